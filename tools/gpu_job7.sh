cd $GRAFT_REPO_ROOT
BENCH_ARGS="--steps 8 --warmup 2 --no-cpu-baseline --sds-steps 8 --posed-frames 0" bash tools/run_variants.sh gx96 gx64 gx48 gx32 gx16 2>&1 | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl"
