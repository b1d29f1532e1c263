# instruction-cache counters of the big kernels (separate --pmc pass, kernel trace off)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_icache; rm -rf $O; mkdir -p $O
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sds-steps 2 --posed-frames 0"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $O/a -o p -- $B > $O/a.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/b -o p -- $B > $O/b.log 2>&1
python - <<PY
import csv, collections, glob, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::|void ", "", r['Kernel_Name'])[:34]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'SQC_ICACHE_REQ' not in d or sum(d['SQC_ICACHE_REQ']) / len(d['SQC_ICACHE_REQ']) < 1e5: continue
    print(k, {c: "%.3g" % (sum(v) / len(v)) for c, v in sorted(d.items())})
PY
