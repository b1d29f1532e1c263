"""posed 256x256 frame (bench.py's posed_frame workload) against the size of the ray batches: same pixels, fewer and fuller launches"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from avatarcraft_amd.render_utils import render_instantnsr_naive
from tests.common import make_rays, make_body

dev = torch.device("cuda:0")
p, _, table, _, _ = bench.make_inputs(dev, 0)
net = bench.make_net(p, table, dev, False)
verts, faces, Ts = make_body(n_lat=83, n_lon=83)
ro_h, rd_h = make_rays(256, 256, dist=1.8, f=443.405 / 2, yaw=0.3, pitch=-0.1)
ro, rd = torch.from_numpy(ro_h).to(dev), torch.from_numpy(rd_h).to(dev)
ref = None
SKIPS = [bool(int(x)) for x in os.environ.get('SKIPS', '1,0').split(',')]
SIZES = [int(x) for x in os.environ.get('SIZES', '4096,8192,16384,32768,65536').split(',')]
for skip in SKIPS:
    net.skip_masked_samples = skip
    for rpb in SIZES:
        def frame():
            return render_instantnsr_naive(net, ro, rd, rays_per_batch=rpb, requires_grad=False, render_can=False, perturb=False, verts=verts, faces=faces,
                                           Ts=Ts, num_steps=32, upsample_steps=32, bound=1.6)[0]
        rgb = frame(); torch.cuda.synchronize()
        if ref is None: ref = rgb.clone()
        t0 = time.perf_counter()
        for _ in range(6): frame()
        torch.cuda.synchronize()
        print("skip_masked %d  rays_per_batch %6d: %.2f ms per frame, pixels %s" % (skip, rpb, (time.perf_counter() - t0) / 6 * 1e3,
              "identical" if torch.equal(rgb, ref) else "DIFFER (max %.3g)" % float((rgb - ref).abs().max())))
