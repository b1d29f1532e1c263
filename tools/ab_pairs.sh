cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "bitwise or golden or soak or field_sdf or fused_sdf or run_cuda or stencil" 2>&1 | tail -3
BENCH_ARGS="--steps 64 --warmup 8 --no-cpu-baseline --sds-steps 6 --posed-frames 2 --no-occupancy --sd-arch-steps 0" bash tools/run_variants.sh head nopairs head nopairs 2>&1 | grep -v "^$" | cut -c1-260
