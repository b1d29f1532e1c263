cd $GRAFT_REPO_ROOT; O=gpurun_out/j4; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
tail -5 $O/pytest.log; grep -n "^E  " $O/pytest.log | head -20
bash tools/prof_waits.sh r02a 2>&1 | tail -30
python - <<'PY'
import sys, time, json, importlib.util
sys.argv=['bench.py']
spec=importlib.util.spec_from_file_location("bench","bench.py"); b=importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
from tests.common import load_golden, make_table
p=load_golden("nsr_params.npz")
table=make_table(int(p["offsets"][-1]), seed=int(p["table_seed"]), offsets=p["offsets"], level_amp=p["level_amp"])
for th in (8, 32, 128):
    t=time.time(); r=b.cpu_baseline_sds(p, table, n_side=16, threads=th); print("threads", th, r["value"], r["sample"][:60], round(time.time()-t,1), flush=True)
PY
