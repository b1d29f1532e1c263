"""timing of the SMPL-guided warp kernels at render_warp.py sizes (256x256 rays, 32 coarse / 64 final samples, SMPL-sized mesh)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from avatarcraft_amd import ray_utils as RY
from tests.common import make_body, make_rays
dev = "cuda:0"
verts, faces, Ts = make_body(n_lat=83, n_lon=83)
print("mesh", verts.shape, faces.shape)
n = int(os.environ.get("RES", 128))
ro, rd = make_rays(n, n, dist=1.8, f=0.78125 * n)
def timeit(fn, k=3):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(k): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / k
tro, trd, tv, tf, tT = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (ro, rd, verts, faces.astype(np.int32), Ts))
print("near/far %d rays x %d verts: %.3f ms" % (ro.shape[0], verts.shape[0], timeit(lambda: RY.geometry_guided_near_far(tro, trd, tv, 0.05))))
for S in (32, 64):
    z = torch.linspace(0.8, 2.8, S, device=dev)
    pts = (tro[:, None, :] + trd[:, None, :] * z[None, :, None]).contiguous()
    P = pts.shape[0] * S
    t = timeit(lambda: RY.warp_samples_to_canonical(pts, tv, tf, tT, 0.05, accel=False))
    print("warp brute force %d pts x %d faces: %.2f ms  (%.1f G point-face tests/s)" % (P, faces.shape[0], t, P * faces.shape[0] / t / 1e6))
    t = timeit(lambda: RY.warp_samples_to_canonical(pts, tv, tf, tT, 0.05, accel=True))
    print("warp culled (incl. per-call build) %d pts: %.2f ms  (%.1f M samples/s)" % (P, t, P / t / 1e3))
