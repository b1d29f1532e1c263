cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_warp; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider -k "warp or posed or mesh_near" > $O/pytest.log 2>&1; echo "pytest rc $?"
grep -n "passed\|failed" $O/pytest.log | tail -2; grep -n "^E  " $O/pytest.log | cut -c1-400 | head -30
for fl in 1 0; do
AC_WARP_FLIST=$fl timeout 600 python bench.py --sds-steps 0 --no-occupancy --no-cpu-baseline --repeat 1 --steps 8 --posed-frames 6 > $O/bench_$fl.json 2> $O/bench_$fl.err; echo "bench flist=$fl rc $?"
python - <<PY
import json
r=json.loads(open("gpurun_out/r04_warp/bench_$fl.json").read().strip().splitlines()[-1])
p=r.get("posed_frame",{}); print("flist=$fl posed ms", p.get("ms_per_frame"), "8192-batches", p.get("ms_per_frame_8192_ray_batches"), "identical", p.get("pixels_identical_across_batch_sizes"), p.get("error"))
PY
done
