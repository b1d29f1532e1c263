cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -q -m gpu -x -k "warp or posed or mesh or accel" 2>&1 | tail -3
python - <<'PY'
import torch, time
from avatarcraft_amd import _lib as L, ray_utils
import numpy as np
torch.manual_seed(0)
from tools.bench_warp import *  # noqa
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/sortprof -o s -- python $GRAFT_REPO_ROOT/tools/bench_warp.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/sortprof/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'accel' in r['Name'] or 'mesh_near' in r['Name']: print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
