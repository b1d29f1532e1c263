# round 2, GPU call 2: suite again (fixes), occupancy ablations of the render kernel (timing only), sorted-flush A/B of the scatter
cd $GRAFT_REPO_ROOT; O=gpurun_out/j2; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
tail -15 $O/pytest.log
BENCH_ARGS="--steps 64 --warmup 8 --no-cpu-baseline --sds-steps 0 --posed-frames 0" bash tools/run_variants.sh head w8f1 w8f3a w12f1a w12f3a
BENCH_ARGS="--steps 8 --warmup 2 --no-cpu-baseline --sds-steps 8 --posed-frames 0" bash tools/run_variants.sh head unsorted
for v in head unsorted; do lib=tools/_bin/lib_$v.so; [ $v = head ] && lib=avatarcraft_amd/libavatarcraft_hip.so; echo "== per-level scatter, $v"; AC_LIB_PATH=$PWD/$lib timeout 300 python tools/bench_hash_stencil.py 2>&1 | tail -18; done
cat gpurun_out/sds_step_parity.json gpurun_out/render_core_parity_100.json 2>/dev/null
