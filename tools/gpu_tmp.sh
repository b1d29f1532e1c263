cd /tmp && export TMPDIR=/tmp
SKIPS=1 SIZES=65536 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/posedprof -o s -- python $GRAFT_REPO_ROOT/tools/posed_batch_sweep.py 2>&1 | grep "ms per frame"
