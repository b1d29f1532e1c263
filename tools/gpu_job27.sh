cd $GRAFT_REPO_ROOT; O=gpurun_out/j27; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_model.py -m gpu -q --maxfail=15 -p no:cacheprovider -k "render_core_operator" > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
tail -4 $O/pytest.log; grep -n "^E  " $O/pytest.log | cut -c1-400 | head -10
cat gpurun_out/render_core_parity_4096.json
BENCH_ARGS="--steps 8 --warmup 2 --no-cpu-baseline --sds-steps 0 --posed-frames 4" bash tools/run_variants.sh head c18_17 c18_16 c17_16 c20_17 head 2>&1 | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl"
