"""Per-phase s_memtime breakdown of warp_samples_accel_kernel (the culled closest-face search) on samples near the body.
    python tools/warp_profile.py   (on the GPU box; builds an instrumented library)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
out = os.path.join(ROOT, "tools", "_bin", "lib_wprof.so")           # python tools/build_variants.py wprof:"-DAC_PROFILE_WARP" (here, before the GPU job)
from avatarcraft_amd import _lib as L
L.LIB_PATH = out
L._SIGS["ac_debug_warp_prof"] = ([ctypes.c_void_p, ctypes.c_int], None)
from tests.common import make_body, make_rays
dev = "cuda:0"
verts, faces, Ts = make_body(n_lat=83, n_lon=83)
ro, rd = make_rays(128, 128, dist=1.8, f=0.78125 * 128)
tro, trd, tv, tf, tT = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (ro, rd, verts, faces.astype(np.int32), Ts))
names = ["cell path: listed boxes + candidates", "full path: bounding pass + seed choice", "full path: seed test (exact, 2 tiles)", "group stage: sub-boxes of the queued tiles",
         "disc tests of the queued groups", "exact batches", "final reduction", "prologue (cell, seed face) + epilogue (blend, inverse)"]
from avatarcraft_amd import ray_utils as RY
nr, fr = RY.geometry_guided_near_far(tro, trd, tv, 0.05)
hit = torch.isfinite(nr) & torch.isfinite(fr)
u = torch.linspace(0.0, 1.0, 64, device=dev)
zg = nr[hit][:, None] + (fr[hit] - nr[hit])[:, None] * u[None, :]
sets = [("whole ray 0.8..2.8", (tro[:, None, :] + trd[:, None, :] * torch.linspace(0.8, 2.8, 64, device=dev)[None, :, None])),
        ("near the body 1.5..2.1", (tro[:, None, :] + trd[:, None, :] * torch.linspace(1.5, 2.1, 64, device=dev)[None, :, None])),
        ("mesh-guided range of the %d rays that have one" % int(hit.sum()), tro[hit][:, None, :] + trd[hit][:, None, :] * zg[:, :, None])]
for label, pp in sets:
    pts = pp.contiguous().reshape(-1, 3)
    P = pts.shape[0]
    nb = int(L.lib().ac_warp_accel_bytes(faces.shape[0]))
    acc = torch.zeros(nb, dtype=torch.uint8, device=dev)
    st = L.current_stream(torch.device(dev))
    L.check(L.lib().ac_warp_accel_build(tv.data_ptr(), tf.data_ptr(), tv.shape[0], tf.shape[0], acc.data_ptr(), nb, st))
    can = torch.empty(P, 3, device=dev); mask = torch.empty(P, dtype=torch.uint8, device=dev)
    buf = (ctypes.c_ulonglong * 8)()
    call = lambda: L.check(L.lib().ac_warp_samples_accel(pts.data_ptr(), tv.data_ptr(), tf.data_ptr(), tT.data_ptr(), P, tv.shape[0], tf.shape[0], 0.05, acc.data_ptr(),
                                                         None, can.data_ptr(), None, None, None, mask.data_ptr(), st))
    call(); L.lib().ac_debug_warp_prof(buf, 1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); call(); e.record(); torch.cuda.synchronize()
    L.lib().ac_debug_warp_prof(buf, 1)
    a = np.array(list(buf), dtype=np.float64)
    print("%s: %d samples, %.3f ms; wave-clocks per sample and share" % (label, P, s.elapsed_time(e)))
    for n, v in zip(names, a):
        print("  %-32s %8.0f  %5.1f%%" % (n, v / P, 100 * v / a.sum()))
