cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$1; mkdir -p $O
B="python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline"
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS SQ_INSTS_BRANCH SQ_WAVE_CYCLES --output-format csv -d $O/x1 -o p -- $B > $O/x1.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_SCA --output-format csv -d $O/x2 -o p -- $B > $O/x2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC --output-format csv -d $O/x3 -o p -- $B > $O/x3.log 2>&1
python - <<PY
import csv, collections, glob, json
O="$O"; out={}
for f in glob.glob(O+"/x*/p_counter_collection.csv"):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'render_rays' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    out.update({k:sum(v)/len(v) for k,v in agg.items()})
print(json.dumps(out))
PY
