cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_sds; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sds-steps 2 > $O/kt.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kt/p_kernel_stats.csv")))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]: print("%-70s calls %6s total %9.2f ms  %5.1f%%" % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6, 100*float(r['TotalDurationNs'])/tot))
PY
