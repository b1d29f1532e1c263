cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_check; mkdir -p $O
SEL="${1:-}"
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider $SEL > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
grep -n "passed\|failed" $O/pytest.log | tail -2; grep -n "^E  " $O/pytest.log | cut -c1-300 | head -30
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -c 300 $O/bench.err
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r04_check/bench.json").read().strip().splitlines()[-1])
print("value",r["value"],"ms/step",r["ms_per_step"],"frac",r["roofline"]["frac"],"kern",r["roofline"]["kernel_ms"])
s=r.get("sds_step",{}); print("sds",s.get("ms_per_step"),s.get("phase_ms"),s.get("error"))
p=r.get("posed_frame",{}); print("posed",p.get("ms_per_frame"),p.get("error"))
print("occ",json.dumps(r.get("occupancy_render"))[:1500]); print("real_sd",r.get("real_sd"))
PY
