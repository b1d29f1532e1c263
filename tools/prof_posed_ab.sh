# rocprofv3 kernel stats of one-batch posed frames, face-list search on / off (AC_WARP_FLIST)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for fl in ${1:-1 0}; do
O=$R/gpurun_out/prof_posed_fl$fl; rm -rf $O; mkdir -p $O
AC_WARP_FLIST=$fl rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python $R/tools/posed_kernels.py 6 > $O/kt.log 2>&1
tail -1 $O/kt.log
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kt/p_kernel_stats.csv")))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("flist=$fl total kernel time %.2f ms over 7 frames" % (tot/1e6))
for r in rows[:12]: print("%-60s calls %5s avg %9.4f ms  per frame %8.3f ms" % (r['Name'][:60], r['Calls'], float(r['TotalDurationNs'])/1e6/int(r['Calls']), float(r['TotalDurationNs'])/1e6/7))
PY
done
