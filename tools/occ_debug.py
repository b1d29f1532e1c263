import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from avatarcraft_amd.synthetic import load_field_params, field_table, make_rays
from avatarcraft_amd import nsr_ops, raymarching
dev = torch.device("cuda:0")
p = load_field_params(); table = field_table(p)
net = bench.make_net(p, table, dev, False, cuda_ray=True)
with torch.no_grad():
    net.deviation_net.variance.fill_(float(np.log(512.0) / 10.0))
net.update_extra_state(1.6)
ro, rd = make_rays(48, 48, dist=1.8, f=36.0)
ro, rd = torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev)
kw = dict(num_steps=64, bound=1.6, upsample_steps=64, bg_color=None, cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0)
with torch.no_grad():
    net.occupancy_rounds = True; a = net.render(ro[None], rd[None], **kw)
    net.occupancy_rounds = False; b = net.render(ro[None], rd[None], **kw)
for k in ("weight_sum", "rgb", "normal", "depth"):
    d = (a[k] - b[k]).abs().reshape(ro.shape[0], -1).amax(1)
    bad = torch.nonzero(d > 0).reshape(-1)
    print(k, "differing rays", int(bad.numel()), "max", float(d.max()), "first", bad[:8].tolist())
bad = torch.nonzero((a["weight_sum"] - b["weight_sum"]).abs().reshape(-1) > 0).reshape(-1)
if bad.numel():
    i = int(bad[0]); print("ray", i, "loop ws", float(a["weight_sum"][i]), "one", float(b["weight_sum"][i]))
    # single-round chain for that ray
    from avatarcraft_amd.instant_nsr import near_far_from_bound
    near, far = near_far_from_bound(ro, rd, 1.6); near, far = near.reshape(-1).contiguous(), far.reshape(-1).contiguous()
    N = ro.shape[0]
    alive = torch.arange(N, dtype=torch.int32, device=dev); rt = near.clone()
    x, d_, dl = raymarching.march_rays(N, 64, alive, rt, ro, rd, 1.6, net.density_grid, net.mean_density, near, far, 128, False)
    fs = nsr_ops.field_samples(net._field(), x, d_, dl, 1.6, 0.005, net.forward_variance(), 1.0)
    ws = torch.zeros(N, device=dev); dp = torch.zeros(N, device=dev); im = torch.zeros(N, 3, device=dev); nm = torch.zeros(N, 3, device=dev)
    raymarching.composite_rays(N, 64, alive, rt, fs["alpha"], fs["rgb"], fs["normal"], dl, ws, dp, im, nm)
    print("single round ws", float(ws[i]), "n samples", int((dl.view(N, 64, 2)[i, :, 0] > 0).sum()), "alphas", fs["alpha"].view(N, 64)[i, :12].tolist())
    print("equal single-round vs one-launch:", int((ws != b["weight_sum"][:, 0]).sum()), " vs loop:", int((ws != a["weight_sum"][:, 0]).sum()))
