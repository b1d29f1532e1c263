cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_vdprobe; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python $R/tools/vd_probe.py > $O/kt.log 2>&1
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); timeout 240 rocprofv3 --pmc $grp --output-format csv -d $O/g$i -o p -- python $R/tools/vd_probe.py > $O/g$i.log 2>&1 || echo "group $i failed"
done
python - <<PY
import csv, glob, collections, re
O="$O"
for f in glob.glob(O+"/kt/**/p_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_rays" in r["Name"]: print(r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3)
agg=collections.defaultdict(list)
for f in glob.glob(O+"/g*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "render_rays" in r["Kernel_Name"]:
            agg[(re.sub(r".*render_rays_kernel", "rk", r["Kernel_Name"])[:30], r["Counter_Name"])].append(float(r["Counter_Value"]))
names=sorted({k[0] for k in agg}); ctrs=sorted({k[1] for k in agg})
for c in ctrs:
    print(f"{c:32s}", "  ".join(f"{n}: {sum(agg[(n,c)])/max(1,len(agg[(n,c)])):14.0f}" for n in names))
PY
