"""Per-phase s_memtime breakdown of hash_stencil_bwd_binned_kernel (the queue fill of the binned table-gradient scatter) on the
sample distribution of a real training batch.   python tools/fill_profile.py   (on the GPU box; builds an instrumented library)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
out = os.path.join(ROOT, "tools", "_bin", "lib_fprof.so")           # python tools/build_variants.py fprof:"-DAC_PROFILE_FILL" (here, before the GPU job)
from avatarcraft_amd import _lib as L
L.LIB_PATH = out
L._SIGS["ac_debug_fill_prof"] = ([ctypes.c_void_p, ctypes.c_int], None)
from avatarcraft_amd import nsr_ops
from tests.common import make_rays, load_golden
from tests.gpu_common import device_field
dev = "cuda:0"
p = load_golden("nsr_params.npz"); f, _ = device_field(p)
ro, rd = make_rays(64, 64, dist=1.8, f=50.0)
t = lambda a: torch.from_numpy(a).to(dev)
o = nsr_ops.render_rays(f, t(ro), t(rd), 64, 64, 1.6, float(p["inv_s"]), extras=True)
z = o["z_vals"]; d = z[:, 1:] - z[:, :-1]
zm = torch.cat([z[:, :-1] + 0.5 * d, z[:, -1:]], 1)
x = (t(ro)[:, None, :] + t(rd)[:, None, :] * zm[:, :, None]).clamp(-1.6, 1.6).reshape(-1, 3).contiguous()
B = x.shape[0]
offs = np.ascontiguousarray(p["offsets"], dtype=np.int32)
Lv = len(offs) - 1
S, H = f.S, f.H
gg = torch.zeros(int(offs[-1]), 2, device=dev)
grad = torch.randn(7, Lv, B, 2, device=dev)
nb = int(L.lib().ac_hash_stencil_backward_scratch(offs.ctypes.data, Lv, S, H, 16, B))
scr = torch.empty(nb, dtype=torch.uint8, device=dev)
st = L.current_stream(torch.device(dev))
buf = (ctypes.c_ulonglong * (32 * 6))()
call = lambda: L.check(L.lib().ac_hash_stencil_backward(grad.data_ptr(), x.data_ptr(), offs.ctypes.data, gg.data_ptr(), B, 2, Lv, S, H, 0.005, 1.6,
                                                        scr.data_ptr(), nb, st))
call(); L.lib().ac_debug_fill_prof(buf, 1)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); call(); e.record(); torch.cuda.synchronize()
L.lib().ac_debug_fill_prof(buf, 1)
a = np.array(list(buf), dtype=np.float64).reshape(32, 6)[:Lv]
print("points %d, operator %.3f ms; s_memtime ticks per level, summed over waves (M ticks): combine+scan | records->LDS | flush: reservation | flush: write-out | whole wave | flushes" % (B, s.elapsed_time(e)))
for l in range(Lv):
    print("  level %2d  %8.1f %8.1f %8.1f %8.1f %8.1f  %7d" % ((l,) + tuple(a[l, :5] / 1e6) + (int(a[l, 5]),)))
t_ = a.sum(0)
print("  total     %8.1f %8.1f %8.1f %8.1f %8.1f  %7d   (%.0f%% / %.0f%% / %.0f%% / %.0f%% of the phases)" % (tuple(t_[:5] / 1e6) + (int(t_[5]),) + tuple(100 * t_[:4] / t_[:4].sum())))
