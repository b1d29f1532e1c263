"""the render launch with and without view directions, 20 launches each (for rocprofv3 --pmc / --kernel-trace: which instructions the SH = true instantiation adds)"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from avatarcraft_amd import nsr_ops
from avatarcraft_amd.instant_nsr import NeRFNetwork
dev = torch.device("cuda", 0)
p, field, table, ro, rd = bench.make_inputs(dev, 0)
field.prepare()
ro_t, rd_t = torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev)
torch.manual_seed(0)
net = NeRFNetwork(use_viewdirs=True)
sd = {k: torch.from_numpy(np.asarray(p[k])) for k in p if k.startswith(("sdf_net", "color_net.1", "color_net.2", "deviation_net"))}
sd["encoder.embeddings"] = torch.from_numpy(table); sd["encoder.offsets"] = torch.from_numpy(np.asarray(p["offsets"]))
net.load_state_dict(sd, strict=False)
net = net.to(dev).eval()
with torch.no_grad():
    fv = net._field()
    for f, inv in ((field, float(p["inv_s"])), (fv, float(p["inv_s"]))):
        out = {}
        for k in range(20):
            b = k % 16
            sl = slice(b * 4096, (b + 1) * 4096)
            nsr_ops.render_rays(f, ro_t[sl], rd_t[sl], 64, 64, 1.6, inv, out=out)
        torch.cuda.synchronize()
print("done")
