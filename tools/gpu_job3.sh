# round 2, GPU call 3: suite (coverage rows f3 / f4 / a18 / fixes), default bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/j3; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
tail -12 $O/pytest.log
grep -n "^E  " $O/pytest.log | head -30
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; python - <<'PY'
import json
d = json.loads(open("gpurun_out/j3/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"])
s = d["sds_step"]; print("sds", s.get("ms_per_step"), s.get("phase_ms"), "frac", s.get("roofline", {}).get("frac"), "cpu", s.get("cpu_baseline", {}).get("value"))
pf = d["posed_frame"]; print("posed", pf.get("ms_per_frame"), pf.get("roofline", {}).get("frac"), pf.get("cpu_baseline"))
print("cpu", d["cpu_baseline"])
PY
