# wait-state / issue counters of the hot kernels (each --pmc group is its own pass): tools/prof_waits.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/waits_$1; rm -rf $O; mkdir -p $O
B="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --sds-steps 2 --posed-frames 1"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_WAVES" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA_WRREQ_sum TCC_EA_RDREQ_sum TCC_EA_ATOMIC_sum TCC_ATOMIC_sum TCC_WRITE_sum TCC_READ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/g$i -o p -- $B > $O/g$i.log 2>&1 || echo "group $i failed: $(tail -2 $O/g$i.log)"
done
python - <<PY
import csv, collections, glob, json
O="$O"
out=collections.defaultdict(dict)
names=("hash_stencil_bwd_binned","bucket_accumulate","sdf_stencil_bwd","color_bwd","render_rays_kernel<0>","render_rays_kernel<3>","warp_samples_accel")
for f in glob.glob(O+"/*/p_counter_collection.csv"):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        for n in names:
            if n in r['Kernel_Name']: agg[(n,r['Counter_Name'])].append(float(r['Counter_Value']))
    for (n,c),v in agg.items(): out[n][c]=sum(v)/len(v)
json.dump(out, open(O+"/summary.json","w"), indent=1)
for n in names:
    if n in out: print(n, {k: ("%.4g" % v) for k,v in sorted(out[n].items())})
PY
