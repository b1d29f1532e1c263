cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=10 -k "backward or binned or hash or stencil or sds_step or stylize or soak or training or reconstruct or render_core or run_cuda" 2>&1 | grep -v "^$\|Warn\|warn" | tail -12
cat gpurun_out/hip_vs_oracle_backward_4096.json 2>/dev/null | head -20
BENCH_ARGS="--steps 16 --warmup 4 --repeat 1 --no-cpu-baseline --sds-steps 8 --posed-frames 0 --no-occupancy --sd-arch-steps 0" bash tools/run_variants.sh head rec12 head rec12 2>&1 | grep -v "^$" | cut -c1-250
