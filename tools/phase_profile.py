"""Per-phase cycle breakdown of render_rays_kernel (s_memtime instrumentation, -DAC_PROFILE build).
    python tools/phase_profile.py        (on the GPU box; builds a separate instrumented library)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
out = os.path.join(ROOT, "tools", "_bin", "lib_rprof.so")           # python tools/build_variants.py rprof:"-DAC_PROFILE" (here, before the GPU job)
from avatarcraft_amd import _lib
_lib.LIB_PATH = out
_lib._SIGS["ac_debug_set_prof"] = ([ctypes.c_void_p], None)
from avatarcraft_amd import nsr_ops
from tests.common import load_golden, make_rays
from tests.gpu_common import device_field
p = load_golden("nsr_params.npz"); f, _ = device_field(p)
ro, rd = make_rays(256, 256, dist=1.7, f=200.0, yaw=0.0, pitch=0.0)
B0 = int(os.environ.get("BATCH", 0)) * 4096
ro, rd = torch.from_numpy(ro[B0:B0 + 4096].copy()).cuda(), torch.from_numpy(rd[B0:B0 + 4096].copy()).cuda()
prof = torch.zeros(4096 * 11, dtype=torch.int64, device="cuda")
_lib.lib().ac_debug_set_prof(prof.data_ptr())
for _ in range(3):
    prof.zero_(); o_ = nsr_ops.render_rays(f, ro, rd, 64, 64, 1.6, float(p["inv_s"]), precision=os.environ.get("PRECISION", "fast")); torch.cuda.synchronize()
pall = prof.cpu().numpy().astype(np.float64)
pa = pall[:40960].reshape(4096, 10)
nw = int((pa[:, 8] > 0).sum())
pa = pa[pa[:, 8] > 0]                      # the waves that ran (rays are handed out dynamically: fewer waves than rays)
ray_us = pall[40960:] / 100.0
pr = pa[:, :8]
names = ["coarse(64 sdf evals)", "upsample math+merge", "upsample sdf eval", "final: stencil gather+interp", "final: 7x sdf mlp", "final: colour mlp",
         "final: alpha+composite", "final: tile setup"]
tot = pr.sum(1).mean()
print("s_memtime ticks per ray: total %.0f" % tot)
print("whole wave: %.0f s_memtime ticks in %.0f s_memrealtime ticks (100 MHz) = %.1f us  ->  s_memtime runs at %.3f GHz"
      % (pa[:, 8].mean(), pa[:, 9].mean(), pa[:, 9].mean() / 100.0, pa[:, 8].mean() / pa[:, 9].mean() * 0.1))
for n, v in zip(names, pr.mean(0)):
    print("  %-32s %9.0f  %5.1f%%" % (n, v, 100 * v / tot))

# how evenly the work of a launch is spread: waves fetch rays dynamically; per-ray wall times and what they correlate with
ws = o_["weights_sum"].cpu().numpy()
hit = ws > 0.5
print("%d waves; per-ray wall time (us): mean %.1f  std %.1f  min %.1f  max %.1f;  rays that hit the body (%d): %.1f, that miss: %.1f"
      % (nw, ray_us.mean(), ray_us.std(), ray_us.min(), ray_us.max(), int(hit.sum()), ray_us[hit].mean() if hit.any() else 0, ray_us[~hit].mean()))
col = np.arange(4096) % 256
rowi = np.arange(4096) // 256
print("by image row (16):", " ".join("%.0f" % ray_us[rowi == b].mean() for b in range(16)))
print("by image column (16 bins):", " ".join("%.0f" % ray_us[(col // 16) == b].mean() for b in range(16)))
print("per-wave busy time (us): mean %.1f  min %.1f  max %.1f  (kernel ends with the slowest wave)" % ((pa[:, 9] / 100).mean(), (pa[:, 9] / 100).min(), (pa[:, 9] / 100).max()))
