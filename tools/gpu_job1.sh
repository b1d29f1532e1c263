# round 2, GPU call 1: the whole -m gpu suite (new parity pins, the render-core operator), the default bench line, the rank-1 A/B of the
# SDF-query backward, and a kernel trace of the SDS step.  Everything lands in gpurun_out/j1/.
cd $GRAFT_REPO_ROOT; O=gpurun_out/j1; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --maxfail=12 --durations=8 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
tail -40 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -c 3000 $O/bench.json; tail -5 $O/bench.err
BENCH_ARGS="--steps 32 --warmup 8 --no-cpu-baseline --sds-steps 8 --posed-frames 0" bash tools/run_variants.sh head rank0
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --sds-steps 6 --posed-frames 2 > $GRAFT_REPO_ROOT/$O/kt.log 2>&1 )
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/j1/kt/**/p_kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:22]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us tot {float(r['TotalDurationNs'])/1e6:8.2f} ms {float(r['Percentage']):5.1f}%")
PY
ls gpurun_out/*.json 2>/dev/null
