cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_occ; rm -rf $O; mkdir -p $O
AC_OCC_GLOG=${1:-4} rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python $R/bench.py --steps 4 --sds-steps 0 --posed-frames 0 --no-cpu-baseline --sd-arch-steps 0 --repeat 1 > $O/kt.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kt/p_kernel_stats.csv")))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:14]: print("%-64s calls %5s avg %9.4f ms min %8.4f max %8.4f" % (r['Name'][:64], r['Calls'], float(r['AverageNs'])/1e6, float(r['MinNs'])/1e6, float(r['MaxNs'])/1e6))
PY
rm -rf $O/kt
