# round-6 evidence at this HEAD (run through gpurun): kernel trace of the driver's bench command, then every PMC group in its own pass.
#   bash tools/prof_r06.sh <tag>      -> gpurun_out/prof_<tag>/{kt,...}/, summary.json, traffic.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$1; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- python $R/bench.py --steps 20 --warmup 5 --no-occupancy --sd-arch-steps 0 --no-fine-view --no-viewdirs > $O/kt_bench.json 2> $O/kt.log
B="python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --sds-steps 2 --posed-frames 1 --repeat 1 --no-occupancy --sd-arch-steps 0 --no-fine-view --no-viewdirs"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_WR" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SMEM" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_ACTIVE_INST_LDS SQ_INSTS_SALU" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $grp --output-format csv -d $O/g$i -o p -- $B > $O/g$i.log 2>&1 || echo "group $i failed: $(tail -2 $O/g$i.log | cut -c1-200)"
done
python - <<PY
import csv, collections, glob, json, re
O="$O"
per=collections.defaultdict(dict); calls=collections.Counter()
short=lambda n: re.sub(r"\(anonymous namespace\)::|void ", "", n).split("(")[0]
for f in glob.glob(O+"/g*/p_counter_collection.csv"):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(short(r['Kernel_Name']), r['Counter_Name'])].append(float(r['Counter_Value']))
    for (n,c),v in agg.items():
        per[n][c]=sum(v)/len(v); calls[n]=max(calls[n], len(v))
stats={}
for f in glob.glob(O+"/kt/**/p_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        stats[short(r['Name'])]=dict(calls=int(r['Calls']), avg_us=float(r['AverageNs'])/1e3, total_ms=float(r['TotalDurationNs'])/1e6, min_us=float(r['MinNs'])/1e3, max_us=float(r['MaxNs'])/1e3)
keep=[n for n in per if any(k in n for k in ("render_rays","hash_stencil","bucket_acc","sdf_stencil","color_bwd","composite","core_","warp_samples","mesh_near","accel_","field_prepare","eikonal","field_sdf_grid","mc_","density_grid","adam","sh_bias"))]
out={n: dict(per[n], dispatches_in_pmc_run=calls[n], **({"trace": stats[n]} if n in stats else {})) for n in sorted(keep)}
json.dump(out, open(O+"/summary.json","w"), indent=1)
for n in sorted(keep):
    v=out[n]; t=v.get("trace",{})
    print(f"{n[:46]:46s} avg {t.get('avg_us',0):8.1f} us x{t.get('calls',0):4d}  FETCH {v.get('FETCH_SIZE',0)/1e6:8.3f} GB  WRITE {v.get('WRITE_SIZE',0)/1e6:7.3f} GB  VALU {v.get('SQ_INSTS_VALU',0)/1e6:7.1f}M MFMA {v.get('SQ_INSTS_MFMA',0)/1e6:6.1f}M  L2miss {v.get('TCC_MISS_sum',0)/1e6:6.1f}M")
PY
tail -c 2500 $O/kt_bench.json
