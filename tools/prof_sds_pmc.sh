# PMC counters of the training kernels (own passes, kernel-trace separate): tools/prof_sds_pmc.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_sds_pmc; rm -rf $O; mkdir -p $O
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sds-steps 2 --posed-frames 0"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/sq -o p -- $B > $O/sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- $B > $O/write.log 2>&1
python - <<PY
import csv, collections, glob, json
O="$O"
out=collections.defaultdict(dict)
names=("hash_stencil_bwd_binned","bucket_accumulate","sdf_stencil_bwd","sdf_stencil_fwd","color_bwd","color_fwd","composite_bwd","composite_fwd","render_rays_kernel<0>","render_rays_kernel<2>")
for f in glob.glob(O+"/*/p_counter_collection.csv"):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        for n in names:
            if n in r['Kernel_Name']: agg[(n,r['Counter_Name'])].append(float(r['Counter_Value']))
    for (n,c),v in agg.items(): out[n][c]=sum(v)/len(v)
json.dump(out, open(O+"/summary.json","w"), indent=1)
for n in names:
    if n in out: print(n, {k: ("%.3g" % v) for k,v in sorted(out[n].items())})
PY
