cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_quick; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider -k "$1" > $O/pytest.log 2>&1; echo "pytest rc $?"
grep -n "passed\|failed" $O/pytest.log | tail -2; grep -n "^E  " $O/pytest.log | cut -c1-400 | head -30
if [ -n "$2" ]; then timeout 600 python bench.py $2 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; python - <<'PY'
import json
r=json.loads(open("gpurun_out/r04_quick/bench.json").read().strip().splitlines()[-1])
print("value",r["value"],"kern",r["roofline"]["kernel_ms"])
for k in ("sds_step","posed_frame","occupancy_render"):
    if k in r: print(k, json.dumps(r[k])[:1800])
PY
fi
cat gpurun_out/render_beside_gemm.json 2>/dev/null
