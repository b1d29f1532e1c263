cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_${1:-t6}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider -rs -k "viewdirs or variants or render_bitwise or repeat_launch or soak" > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
grep -n "passed\|failed\|SKIPPED" $O/pytest.log | tail -5; grep -n "^E  " $O/pytest.log | cut -c1-300 | head -30
timeout 600 python bench.py --no-cpu-baseline --posed-frames 0 --no-occupancy --sd-arch-steps 0 --no-geometry --no-fine-view > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<PY
import json
r=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value",r["value"],"kern",r["roofline"]["kernel_ms"]); print("sds",r["sds_step"]["ms_per_step"]); print("vd",{k:v for k,v in r.get("viewdirs").items() if k!="note"})
PY
