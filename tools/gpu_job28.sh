cd $GRAFT_REPO_ROOT
BENCH_ARGS="--steps 8 --warmup 2 --no-cpu-baseline --sds-steps 0 --posed-frames 4" bash tools/run_variants.sh c17_16 c17_15 c16_16 c16_15 c15_15 c17_16 2>&1 | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl"
