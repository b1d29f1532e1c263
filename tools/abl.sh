# generic timing ablation: tools/abl.sh "name:-DFLAG ..." ... ; builds one library per variant (in parallel) and runs the render bench on each
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/abl; mkdir -p $O
C=$R/avatarcraft_amd/csrc
SRC="$C/ac_capi.hip $C/hashgrid.hip $C/hash_stencil.hip $C/shencoder.hip $C/raymarching.hip $C/render_fused.hip $C/sdf_train.hip $C/warp.hip"
BENCH_ARGS=${BENCH_ARGS:---steps 64 --warmup 8 --no-cpu-baseline --sds-steps 0 --posed-frames 0}
for spec in "$@"; do
  n=${spec%%:*}; fl=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-result $fl -o $O/lib_$n.so $SRC > $O/build_$n.log 2>&1 &
done
wait
for spec in "$@"; do
  n=${spec%%:*}
  for i in 1 2; do
    AC_LIB_PATH=$O/lib_$n.so python $R/bench.py $BENCH_ARGS 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); s = d.get('sds_step') or {}; p = d.get('posed_frame') or {}
    print('%-14s render %.4f ms  sds %s  posed %s' % ('$n', d['roofline']['kernel_ms'], s.get('ms_per_step'), p.get('ms_per_frame')))
except Exception as e: print('$n', 'FAILED', l[-300:])
"
  done
done
