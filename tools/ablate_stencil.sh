# timing ablations of hash_stencil_bwd_kernel (NOT correct results): atomics vs run-combining vs the rest
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/abl_stencil; mkdir -p $O
C=$R/avatarcraft_amd/csrc
for v in base noatomic norun noatomic_norun; do
  fl=""; case $v in noatomic) fl="-DAC_ABL_NOATOMIC";; norun) fl="-DAC_ABL_NORUN";; noatomic_norun) fl="-DAC_ABL_NOATOMIC -DAC_ABL_NORUN";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-result $fl -o $O/lib_$v.so $C/ac_capi.hip $C/hashgrid.hip $C/hash_stencil.hip $C/shencoder.hip $C/raymarching.hip $C/render_fused.hip $C/warp.hip > /dev/null 2>&1 &
done
wait
for v in base noatomic norun noatomic_norun; do echo "== $v"; AC_LIB_PATH=$O/lib_$v.so python $R/tools/bench_hash_stencil.py 2>&1 | grep -E "level|sum" | awk '{printf "%s ", $NF=="ms"?$(NF-1):$0} END {print ""}'; done
