"""timing of the SMPL-guided warp kernels at render_warp.py sizes (256x256 rays, 32 coarse / 64 final samples, SMPL-sized mesh)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from avatarcraft_amd import _lib as L
if os.environ.get('AC_LIB_PATH'): L.LIB_PATH = os.environ['AC_LIB_PATH']
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

if os.environ.get("COUNT"):
    # -DAC_COUNT_CAND build: candidate tiles per sample, near the body vs the whole frustum
    nb = int(L.lib().ac_warp_accel_bytes(faces.shape[0]))
    nr, fr = RY.geometry_guided_near_far(tro, trd, tv, 0.05)
    hit = torch.isfinite(nr) & torch.isfinite(fr)
    zg = nr[hit][:, None] + (fr[hit] - nr[hit])[:, None] * torch.linspace(0.0, 1.0, 64, device=dev)[None, :]
    for name, pp in (("whole ray 0.8..2.8", tro[:, None, :] + trd[:, None, :] * torch.linspace(0.8, 2.8, 64, device=dev)[None, :, None]),
                     ("near the body 1.5..2.1", tro[:, None, :] + trd[:, None, :] * torch.linspace(1.5, 2.1, 64, device=dev)[None, :, None]),
                     ("mesh-guided range (%d rays)" % int(hit.sum()), tro[hit][:, None, :] + trd[hit][:, None, :] * zg[:, :, None])):
        pts = pp.contiguous().reshape(-1, 3)
        acc = torch.zeros(nb, dtype=torch.uint8, device=dev)
        st = L.current_stream(torch.device(dev))
        L.check(L.lib().ac_warp_accel_build(tv.data_ptr(), tf.data_ptr(), tv.shape[0], tf.shape[0], acc.data_ptr(), nb, st))
        P = pts.shape[0]
        can = torch.empty(P, 3, device=dev); mask = torch.empty(P, dtype=torch.uint8, device=dev)
        L.check(L.lib().ac_warp_samples_accel(pts.data_ptr(), tv.data_ptr(), tf.data_ptr(), tT.data_ptr(), P, tv.shape[0], tf.shape[0], 0.05, acc.data_ptr(), None,
                                              can.data_ptr(), None, None, None, mask.data_ptr(), st))
        torch.cuda.synchronize()
        cnt = int(acc[16:24].view(torch.int64)[0]); cf = int(acc[24:32].view(torch.int64)[0])
        full = cf >> 40; cf &= (1 << 40) - 1                     # samples that took the full bounding pass (no cell list) are counted in the high bits
        hdr = acc[:256].view(torch.int32)
        print("candidate tiles per sample, %s: %.2f, faces through the disc test: %.1f, full bounding pass for %.1f %% of the samples  (mask fraction %.3f; "
              "grid %s cells of %.2f cm)" % (name, cnt / P, cf / P, 100.0 * full / P, float(mask.float().mean()), hdr[13:16].tolist(),
                                            100.0 / float(hdr[11:12].view(torch.float32)[0])))
