cd $GRAFT_REPO_ROOT
BENCH_ARGS="--steps 16 --warmup 4 --no-cpu-baseline --sds-steps 12 --posed-frames 0" bash tools/run_variants.sh head tld18 tld19 head tld18 tld19 2>&1 | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl"
