cd $GRAFT_REPO_ROOT; O=gpurun_out/j13; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_render.py tests/test_gpu_model.py -m gpu -q --maxfail=15 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
tail -3 $O/pytest.log; grep -n "^E  " $O/pytest.log | cut -c1-300 | head -20
for i in 1 2; do for fl in "" "--no-prepared-field"; do timeout 200 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --sds-steps 0 --posed-frames 0 $fl 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('prepared' if '$fl' == '' else 'unprepared', 'render %.4f ms' % d['roofline']['kernel_ms'])"; done; done
BENCH_ARGS="--steps 8 --warmup 2 --no-cpu-baseline --sds-steps 8 --posed-frames 2" bash tools/run_variants.sh head tg256 head tg256 2>&1 | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl"
