"""Ceiling of a better first bound in the culled closest-face search: the same samples searched twice, the second time with every sample's running minimum
started at its TRUE distance^2 (from the first run) -- -DAC_WARP_SEED_DEBUG build (tools/build_variants.py wseed:"-DAC_WARP_SEED_DEBUG")."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from avatarcraft_amd import _lib as L
L.LIB_PATH = os.path.join(ROOT, "tools", "_bin", "lib_wseed.so")
L._SIGS["ac_debug_warp_seed"] = ([C.c_void_p], None)
from avatarcraft_amd import ray_utils as RY
from tests.common import make_body, make_rays
dev = "cuda:0"
verts, faces, Ts = make_body(n_lat=83, n_lon=83)
ro, rd = make_rays(256, 256, dist=1.8, f=443.405 / 2, yaw=0.3, pitch=-0.1)
tro, trd, tv, tf, tT = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (ro, rd, verts, faces.astype(np.int32), Ts))
nr, fr = RY.geometry_guided_near_far(tro, trd, tv, 0.05)
hit = torch.isfinite(nr) & torch.isfinite(fr)
zg = nr[hit][:, None] + (fr[hit] - nr[hit])[:, None] * torch.linspace(0.0, 1.0, 64, device=dev)[None, :]
pts = (tro[hit][:, None, :] + trd[hit][:, None, :] * zg[:, :, None]).contiguous().reshape(-1, 3)
P = pts.shape[0]
nb = int(L.lib().ac_warp_accel_bytes(faces.shape[0]))
acc = torch.zeros(nb, dtype=torch.uint8, device=dev)
st = L.current_stream(torch.device(dev))
L.check(L.lib().ac_warp_accel_build(tv.data_ptr(), tf.data_ptr(), tv.shape[0], tf.shape[0], acc.data_ptr(), nb, st))
can = torch.empty(P, 3, device=dev); mask = torch.empty(P, dtype=torch.uint8, device=dev)
d2 = torch.empty(P, dtype=torch.float64, device=dev); fid = torch.empty(P, dtype=torch.int32, device=dev)


def run():
    L.check(L.lib().ac_warp_samples_accel(pts.data_ptr(), tv.data_ptr(), tf.data_ptr(), tT.data_ptr(), P, tv.shape[0], tf.shape[0], 0.05, acc.data_ptr(), None,
                                          can.data_ptr(), None, d2.data_ptr(), fid.data_ptr(), mask.data_ptr(), st))


def timed(k=5):
    run(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(k): run()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / k


t0 = timed()
seed = d2.clone(); f0 = fid.clone(); c0 = can.clone()
L.lib().ac_debug_warp_seed(seed.data_ptr())
t1 = timed()
print("%d samples in the mesh-guided range of %d rays: %.3f ms with the cell's seed face as first bound, %.3f ms with the true distance as first bound; same faces: %s"
      % (P, int(hit.sum()), t0, t1, bool(torch.equal(f0, fid) and torch.equal(c0, can))))
