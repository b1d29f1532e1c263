cd $GRAFT_REPO_ROOT; O=gpurun_out/j30; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_model.py -m gpu -q --maxfail=15 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
tail -4 $O/pytest.log; grep -n "^E  " $O/pytest.log | cut -c1-300 | head -20
cat gpurun_out/fast_vs_exact.json | head -30
BENCH_ARGS="--steps 64 --warmup 8 --no-cpu-baseline --sds-steps 8 --posed-frames 2" bash tools/run_variants.sh head prevcolor head prevcolor 2>&1 | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl"
