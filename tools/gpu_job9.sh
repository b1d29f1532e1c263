cd $GRAFT_REPO_ROOT
BENCH_ARGS="--steps 64 --warmup 8 --no-cpu-baseline --sds-steps 0 --posed-frames 0" bash tools/run_variants.sh head prevrender head prevrender 2>&1 | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl"
BENCH_ARGS="--steps 64 --warmup 8 --no-cpu-baseline --sds-steps 0 --posed-frames 0 --precision exact" bash tools/run_variants.sh head prevrender 2>&1 | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl"
