# on the GPU box: time bench.py on every tools/_bin/lib_<name>.so given (built beforehand by tools/build_variants.py)
#   BENCH_ARGS="--steps 64 --warmup 8 --no-cpu-baseline --sds-steps 6 --posed-frames 0" bash tools/run_variants.sh base rank0 ...
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/variants; mkdir -p $O
BENCH_ARGS=${BENCH_ARGS:---steps 64 --warmup 8 --no-cpu-baseline --sds-steps 6 --posed-frames 2}
for n in "$@"; do
  lib=$R/tools/_bin/lib_$n.so; [ "$n" = "head" ] && lib=$R/avatarcraft_amd/libavatarcraft_hip.so
  for i in 1 2; do
    AC_LIB_PATH=$lib timeout 300 python $R/bench.py $BENCH_ARGS 2>$O/err_$n.log | tail -1 > $O/line_$n_$i.json
    python - "$n" $O/line_$n_$i.json <<'PY'
import sys, json
n, f = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(f).read().strip()); s = d.get('sds_step') or {}; p = d.get('posed_frame') or {}
    print('%-16s render %.4f ms  sds %s  posed %s  phases %s' % (n, d['roofline']['kernel_ms'], s.get('ms_per_step', s.get('error')), p.get('ms_per_frame', p.get('error')), s.get('phase_ms')))
except Exception as e:
    print(n, 'FAILED', repr(e), open(f).read()[-300:])
PY
  done
done 2>&1 | tee -a $O/summary.txt
