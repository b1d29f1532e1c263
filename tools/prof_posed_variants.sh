# rocprofv3 kernel stats of one-batch posed frames for library variants (tools/_bin/lib_<name>.so; "head" = the in-tree build)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in "$@"; do
lib=$R/tools/_bin/lib_$n.so; [ "$n" = "head" ] && lib=$R/avatarcraft_amd/libavatarcraft_hip.so
O=$R/gpurun_out/prof_posed_$n; rm -rf $O; mkdir -p $O
AC_LIB_PATH=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python $R/tools/posed_kernels.py 6 > $O/kt.log 2>&1
tail -1 $O/kt.log
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kt/p_kernel_stats.csv")))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("$n: total kernel time per frame %.3f ms" % (tot/1e6/7))
for r in rows[:6]: print("   %-60s calls %5s avg %9.4f ms  per frame %8.3f ms" % (r['Name'][:60], r['Calls'], float(r['TotalDurationNs'])/1e6/int(r['Calls']), float(r['TotalDurationNs'])/1e6/7))
PY
rm -rf $O/kt
done
