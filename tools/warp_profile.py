"""Per-phase s_memtime breakdown of warp_samples_accel_kernel (the culled closest-face search) on samples near the body.
    python tools/warp_profile.py   (on the GPU box; builds an instrumented library)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
csrc = os.path.join(ROOT, "avatarcraft_amd", "csrc")
out = os.path.join(ROOT, "gpurun_out", "libac_warpprof.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
srcs = [os.path.join(csrc, f) for f in ("ac_capi.hip", "hashgrid.hip", "shencoder.hip", "raymarching.hip", "render_fused.hip", "hash_stencil.hip", "sdf_train.hip", "warp.hip")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-DAC_PROFILE_WARP",
                       "-Wno-unused-result", "-o", out] + srcs + sys.argv[1:])
from avatarcraft_amd import _lib as L
L.LIB_PATH = out
L._SIGS["ac_debug_warp_prof"] = ([ctypes.c_void_p, ctypes.c_int], None)
from tests.common import make_body, make_rays
dev = "cuda:0"
verts, faces, Ts = make_body(n_lat=83, n_lon=83)
ro, rd = make_rays(128, 128, dist=1.8, f=0.78125 * 128)
tro, trd, tv, tf, tT = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (ro, rd, verts, faces.astype(np.int32), Ts))
names = ["bounding pass", "bound reductions + seed choice", "seed test (exact, 2 tiles)", "candidate tile list", "disc tests + ring", "exact batches", "final reduction", "epilogue (blend, inverse)"]
for label, zz in (("whole ray 0.8..2.8", torch.linspace(0.8, 2.8, 64, device=dev)), ("near the body 1.5..2.1", torch.linspace(1.5, 2.1, 64, device=dev))):
    pts = (tro[:, None, :] + trd[:, None, :] * zz[None, :, None]).contiguous().reshape(-1, 3)
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
