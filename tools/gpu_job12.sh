cd $GRAFT_REPO_ROOT; O=gpurun_out/j12; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_render.py -m gpu -q --maxfail=15 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
tail -3 $O/pytest.log; grep -n "^E  " $O/pytest.log | cut -c1-300 | head -20
BENCH_ARGS="--steps 8 --warmup 2 --no-cpu-baseline --sds-steps 8 --posed-frames 0" bash tools/run_variants.sh head prevrender head prevrender 2>&1 | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl"
cat gpurun_out/sds_step_parity.json gpurun_out/render_core_parity_100.json | tr -d '\n ' | cut -c1-1500
