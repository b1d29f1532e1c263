# usage (through gpurun): bash tools/gpu_r05_occ.sh <tag>   -- the occupancy-path tests + the occupancy leg of bench.py
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_${1:-occ}; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider -rs -k "run_cuda or raymarch or occupancy or march or density or pins" > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
grep -n "passed\|failed\|SKIPPED" $O/pytest.log | tail -5; grep -n "^E  " $O/pytest.log | cut -c1-300 | head -40
timeout 600 python bench.py --steps 8 --warmup 2 --repeat 1 --sds-steps 0 --posed-frames 0 --no-fine-view --no-viewdirs --no-geometry --no-cpu-baseline --sd-arch-steps 0 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<PY
import json
r=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
o=r.get("occupancy_render",{})
for k,v in o.items(): print(k, json.dumps(v)[:400])
PY
