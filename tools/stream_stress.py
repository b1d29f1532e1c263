"""stress of concurrent renders on several streams (per-stream hand-out scratch, cross-XCD hand-offs when two persistent kernels share the device):
every launch must equal its single-stream result bit for bit"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from avatarcraft_amd import nsr_ops
from tests.common import make_rays, load_golden
from tests.gpu_common import device_field
dev = "cuda:0"
p = load_golden("nsr_params.npz")
f, _ = device_field(p, device=torch.device(dev)); f.prepare()
inv_s = float(p["inv_s"])
ro, rd = make_rays(256, 256, dist=1.7, f=200.0, yaw=0.0, pitch=0.0)
NS, REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 60
sizes = [4096, 1000, 4096, 2500, 8192, 4096, 300, 4096]
batches = []
for k in range(NS):
    n = sizes[k % len(sizes)]
    batches.append((torch.from_numpy(ro[k * 4096:k * 4096 + n].copy()).to(dev), torch.from_numpy(rd[k * 4096:k * 4096 + n].copy()).to(dev)))
keys = ("image", "weights_sum", "depth", "normal_map", "gradient_error")
ref = []
for o, d in batches:
    r = nsr_ops.render_rays(f, o, d, 64, 64, 1.6, inv_s)
    ref.append({k: r[k].clone() for k in keys})
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in batches]
bad = 0
for rep in range(REPS):
    outs = []
    for s, (o, d) in zip(streams, batches):
        with torch.cuda.stream(s):
            outs.append(nsr_ops.render_rays(f, o, d, 64, 64, 1.6, inv_s))
    torch.cuda.synchronize()
    for i, (r, out) in enumerate(zip(ref, outs)):
        for k in keys:
            if not torch.equal(out[k], r[k]):
                bad += 1
                print("rep %d stream %d (%d rays): %s differs in %d values, nan %d" % (rep, i, batches[i][0].shape[0], k, int((out[k] != r[k]).sum()), int(torch.isnan(out[k]).sum())))
print("%d streams x %d rounds: %d mismatching outputs" % (NS, REPS, bad))
