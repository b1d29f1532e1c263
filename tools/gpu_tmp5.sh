cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_${1:-t5}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider -rs -k "warp or posed or inference_drivers" > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
grep -n "passed\|failed\|SKIPPED" $O/pytest.log | tail -5; grep -n "^E  " $O/pytest.log | cut -c1-300 | head -30
timeout 600 python bench.py --no-cpu-baseline --posed-frames 8 --sds-steps 0 --no-occupancy --sd-arch-steps 0 --no-geometry --no-fine-view --no-viewdirs > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<PY
import json
r=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
p=r["posed_frame"]; print("posed", p.get("ms_per_frame"), p.get("phase_ms")); print(json.dumps(p.get("search_roofline"))[:900])
PY
