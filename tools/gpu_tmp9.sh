cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_occ6; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider -k "run_cuda or raymarch or occupancy or march or density or pins" > $O/pytest.log 2>&1; echo "pytest rc $?"
grep -n "passed\|failed" $O/pytest.log | tail -3; grep -n "^E  " $O/pytest.log | cut -c1-300 | head -20
python tools/occ_train_probe.py 2>/dev/null | tail -1
for v in rm4k rm16k; do AC_LIB_PATH=$PWD/tools/_bin/lib_$v.so python tools/occ_train_probe.py 2>/dev/null | tail -1; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -o p -- python $GRAFT_REPO_ROOT/tools/occ_train_probe.py > /dev/null 2>&1
python - <<PY
import csv,glob
for f in glob.glob("$GRAFT_REPO_ROOT/$O/kt/**/p_kernel_stats.csv", recursive=True):
    rows=sorted(csv.DictReader(open(f)), key=lambda r:-float(r['TotalDurationNs']))
    for r in rows[:6]: print("%-60s calls %5s avg %9.1f us" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
