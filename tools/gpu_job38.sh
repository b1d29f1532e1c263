cd $GRAFT_REPO_ROOT; O=gpurun_out/j38; mkdir -p $O
AC_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 8 --warmup 2 --sds-steps 3 2>$O/err.log | tail -1 > $O/bench2.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/j38/bench2.json'))
print('n_gpus', d['n_gpus'], 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'sds', d['sds_step'].get('ms_per_step'), d['sds_step'].get('grad_allreduce_ms'), d['sds_step'].get('grad_allreduce_mb'), d['sds_step'].get('error'))
PY
tail -3 $O/err.log | cut -c1-300
