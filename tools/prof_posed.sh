# rocprofv3 kernel stats of posed frames (render_warp configuration)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_posed; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sds-steps 0 --posed-frames 3 > $O/kt.log 2>&1
tail -1 $O/kt.log | python -c "import sys,json; print(json.loads(sys.stdin.read())['posed_frame'])"
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kt/p_kernel_stats.csv")))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel time %.2f ms" % (tot/1e6))
for r in rows[:14]: print("%-70s calls %6s avg %9.4f ms total %8.2f ms %5.1f%%" % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6/int(r['Calls']), float(r['TotalDurationNs'])/1e6, 100*float(r['TotalDurationNs'])/tot))
PY
