"""One-batch posed frames only (what drivers.render_animation does), for a per-kernel trace of the posed frame:
    rocprofv3 --kernel-trace --stats ... -- python tools/posed_kernels.py [frames]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from avatarcraft_amd.render_utils import render_instantnsr_naive
from tests.common import make_rays, make_body, load_golden, make_table

dev = torch.device("cuda:0")
p = load_golden("nsr_params.npz")
table = make_table(int(p["offsets"][-1]), seed=int(p["table_seed"]), offsets=p["offsets"], level_amp=p["level_amp"])
net = bench.make_net(p, table, dev, False)
net.skip_masked_samples = True
verts, faces, Ts = make_body(n_lat=83, n_lon=83)
ro_h, rd_h = make_rays(256, 256, dist=1.8, f=443.405 / 2, yaw=0.3, pitch=-0.1)
ro, rd = torch.from_numpy(ro_h).to(dev), torch.from_numpy(rd_h).to(dev)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def frame():
    return render_instantnsr_naive(net, ro, rd, rays_per_batch=65536, requires_grad=False, render_can=False, perturb=False, verts=verts, faces=faces, Ts=Ts,
                                   num_steps=32, upsample_steps=32, bound=1.6)[0]


frame(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(frames):
    frame()
torch.cuda.synchronize()
print("ms per frame %.3f over %d frames (+1 warm-up frame in the trace)" % ((time.perf_counter() - t0) / frames * 1e3, frames))
