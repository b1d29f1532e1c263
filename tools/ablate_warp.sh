# where does warp_samples_accel_kernel spend its time?  (ablated builds give wrong results; timing / counting only)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/abl_warp; mkdir -p $O
C=$R/avatarcraft_amd/csrc
SRC="$C/ac_capi.hip $C/hashgrid.hip $C/hash_stencil.hip $C/shencoder.hip $C/raymarching.hip $C/render_fused.hip $C/sdf_train.hip $C/warp.hip"
V="base nocand nobatch count"
for v in $V; do
  fl=""; case $v in nocand) fl="-DAC_ABL_NOCAND";; nobatch) fl="-DAC_ABL_NOBATCH";; count) fl="-DAC_COUNT_CAND";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-result $fl -o $O/lib_$v.so $SRC > /dev/null 2>&1 &
done
wait
for v in $V; do [ $v = count ] && continue; echo "== $v"; AC_LIB_PATH=$O/lib_$v.so python $R/tools/bench_warp.py 2>&1 | grep culled; done
AC_LIB_PATH=$O/lib_count.so COUNT=1 python $R/tools/bench_warp.py 2>&1 | grep -E "candidate"
