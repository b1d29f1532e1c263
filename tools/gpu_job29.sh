cd $GRAFT_REPO_ROOT; O=gpurun_out/j29; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_render.py -m gpu -q --maxfail=15 -p no:cacheprovider -k "warp or posed or mesh or near_far" > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
tail -3 $O/pytest.log; grep -n "^E  " $O/pytest.log | cut -c1-300 | head -20
echo "== count"; AC_LIB_PATH=$GRAFT_REPO_ROOT/tools/_bin/lib_count.so COUNT=1 timeout 300 python tools/bench_warp.py 2>&1 | grep candidate | tee $O/count.txt
echo "== profile"; timeout 300 python tools/warp_profile.py 2>&1 | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl" | tee $O/warp_profile.txt
BENCH_ARGS="--steps 8 --warmup 2 --no-cpu-baseline --sds-steps 0 --posed-frames 4" bash tools/run_variants.sh head c17_16 head c17_16 2>&1 | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl"
