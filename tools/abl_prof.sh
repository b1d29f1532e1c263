# per-kernel timing of library variants under rocprofv3: tools/abl_prof.sh "name:-DFLAG" ...  (KERNELS = regex of kernel names to print)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ablp; mkdir -p $O
C=$R/avatarcraft_amd/csrc
SRC="$C/ac_capi.hip $C/hashgrid.hip $C/hash_stencil.hip $C/shencoder.hip $C/raymarching.hip $C/render_fused.hip $C/sdf_train.hip $C/warp.hip"
BENCH_ARGS=${BENCH_ARGS:---steps 2 --warmup 1 --no-cpu-baseline --sds-steps 4 --posed-frames 0}
KERNELS=${KERNELS:-binned|bucket_acc|sdf_stencil|render_rays}
for spec in "$@"; do
  n=${spec%%:*}; fl=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-result $fl -o $O/lib_$n.so $SRC > $O/build_$n.log 2>&1 &
done
wait
for spec in "$@"; do
  n=${spec%%:*}; rm -rf $O/kt_$n
  AC_LIB_PATH=$O/lib_$n.so rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$n -o p -- python $R/bench.py $BENCH_ARGS > $O/kt_$n.log 2>&1
  python - <<PY
import csv, re
rows=list(csv.DictReader(open("$O/kt_$n/p_kernel_stats.csv")))
out=[]
for r in rows:
    if re.search(r"$KERNELS", r['Name']):
        nm=re.sub(r"\(anonymous namespace\)::|void ", "", r['Name'])[:28]
        out.append("%s %.3f" % (nm, float(r['TotalDurationNs'])/1e6/int(r['Calls'])))
print("%-12s" % "$n", " | ".join(sorted(out)))
PY
done
