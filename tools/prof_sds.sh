# rocprofv3 kernel stats of the SDS step: tools/prof_sds.sh [sds-steps]  (AC_LIB_PATH selects a library variant)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_sds; rm -rf $O; mkdir -p $O
N=${1:-4}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sds-steps $N --posed-frames 0 > $O/kt.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kt/p_kernel_stats.csv")))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]: print("%-70s calls %6s avg %9.4f ms  %5.1f%%" % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6/int(r['Calls']), 100*float(r['TotalDurationNs'])/tot))
PY
