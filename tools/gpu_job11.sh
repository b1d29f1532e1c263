cd $GRAFT_REPO_ROOT
BENCH_ARGS="--steps 64 --warmup 8 --no-cpu-baseline --sds-steps 0 --posed-frames 0" bash tools/run_variants.sh head persist1 head persist1 2>&1 | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl"
