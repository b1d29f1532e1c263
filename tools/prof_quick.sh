# quick PMC profile of the bench (run through gpurun): tools/prof_quick.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$1; rm -rf $O; mkdir -p $O
B="python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --sds-steps 0 --posed-frames 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python $R/bench.py --steps 32 --warmup 4 --no-cpu-baseline --sds-steps 0 --posed-frames 0 > $O/kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/sq -o p -- $B > $O/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM --output-format csv -d $O/sq2 -o p -- $B > $O/sq2.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tc -o p -- $B > $O/tc.log 2>&1
rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $O/ta -o p -- $B > $O/ta.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- $B > $O/fetch.log 2>&1
python - <<PY
import csv, collections, glob, json, os
O="$O"
out={}
for f in glob.glob(O+"/*/p_counter_collection.csv"):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'render_rays' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    out.update({k:sum(v)/len(v) for k,v in agg.items()})
for r in csv.DictReader(open(O+"/kt/p_kernel_stats.csv")):
    if 'render_rays' in r['Name']: out['kernel_avg_ns']=float(r['AverageNs']); out['kernel_calls']=int(r['Calls'])
json.dump(out, open(O+"/summary.json","w"), indent=1)
print(json.dumps(out))
PY
