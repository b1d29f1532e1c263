# queue fill of the binned scatter: is the run combining worth its shuffles on the fine levels?  (both variants give correct sums)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/abl_fill; mkdir -p $O
C=$R/avatarcraft_amd/csrc
SRC="$C/ac_capi.hip $C/hashgrid.hip $C/hash_stencil.hip $C/shencoder.hip $C/raymarching.hip $C/render_fused.hip $C/sdf_train.hip $C/warp.hip"
for v in base finenorun norun; do
  fl=""; case $v in finenorun) fl="-DAC_FINE_NORUN";; norun) fl="-DAC_ABL_NORUN";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-result $fl -o $O/lib_$v.so $SRC > /dev/null 2>&1 &
done
wait
for v in base finenorun norun; do echo "== $v"; for i in 1 2; do AC_LIB_PATH=$O/lib_$v.so python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --sds-steps 4 --posed-frames 0 2>&1 | tail -1 | grep -o "sds_step.\{40\}"; done; done
