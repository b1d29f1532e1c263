import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
import bench as B
from avatarcraft_amd import nsr_ops
from avatarcraft_amd.render_utils import render_instantnsr_naive
from avatarcraft_amd.synthetic import make_rays, make_body_sequence
dev = torch.device("cuda:0")
p, field, table, ro, rd = B.make_inputs(dev, 0)
net = B.make_net(p, table, dev, False); net.skip_masked_samples = True; net.warp_temporal_seeds = True
ro_h, rd_h = make_rays(256, 256, dist=1.8, f=443.405 / 2, yaw=0.3, pitch=-0.1)
ro, rd = torch.from_numpy(ro_h).to(dev), torch.from_numpy(rd_h).to(dev)
seq_v, faces, seq_T = make_body_sequence(8, 83, 83)
def run(overlap):
    t0 = time.perf_counter()
    for wm in nsr_ops.warp_mesh_sequence(zip(seq_v, seq_T), faces, dev, overlap=overlap):
        render_instantnsr_naive(net, ro, rd, rays_per_batch=65536, requires_grad=False, render_can=False, perturb=False, verts=wm, faces=None, Ts=None, num_steps=32, upsample_steps=32, bound=1.6)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / len(seq_v) * 1e3
run(True); run(True)
print("overlap", run(True), "inline", run(False), "overlap", run(True))
