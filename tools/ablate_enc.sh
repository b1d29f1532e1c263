# render_rays_kernel: gather rounds in flight per lane (AC_ENC_ROUND) -- timing only
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/abl_enc; mkdir -p $O
C=$R/avatarcraft_amd/csrc
SRC="$C/ac_capi.hip $C/hashgrid.hip $C/hash_stencil.hip $C/shencoder.hip $C/raymarching.hip $C/render_fused.hip $C/sdf_train.hip $C/warp.hip"
for v in 1 2 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-result -DAC_ENC_ROUND=$v -o $O/lib_$v.so $SRC > /dev/null 2>&1 &
done
wait
for v in 1 2 4; do echo "== AC_ENC_ROUND=$v"; AC_LIB_PATH=$O/lib_$v.so python $R/bench.py --steps 64 --warmup 8 --no-cpu-baseline --sds-steps 0 --posed-frames 0 2>&1 | tail -1 | grep -o "kernel_ms.\{22\}"; done
