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
ro, rd = torch.from_numpy(ro[:4096].copy()).cuda(), torch.from_numpy(rd[:4096].copy()).cuda()
prof = torch.zeros(4096 * 10, dtype=torch.int64, device="cuda")
_lib.lib().ac_debug_set_prof(prof.data_ptr())
for _ in range(3):
    prof.zero_(); nsr_ops.render_rays(f, ro, rd, 64, 64, 1.6, float(p["inv_s"]), precision=os.environ.get("PRECISION", "fast")); torch.cuda.synchronize()
pa = prof.cpu().numpy().reshape(4096, 10).astype(np.float64)
pr = pa[:, :8]
names = ["coarse(64 sdf evals)", "upsample math+merge", "upsample sdf eval", "final: stencil gather+interp", "final: 7x sdf mlp", "final: colour mlp",
         "final: alpha+composite", "final: tile setup"]
tot = pr.sum(1).mean()
print("s_memtime ticks per ray: total %.0f" % tot)
print("whole wave: %.0f s_memtime ticks in %.0f s_memrealtime ticks (100 MHz) = %.1f us  ->  s_memtime runs at %.3f GHz"
      % (pa[:, 8].mean(), pa[:, 9].mean(), pa[:, 9].mean() / 100.0, pa[:, 8].mean() / pa[:, 9].mean() * 0.1))
for n, v in zip(names, pr.mean(0)):
    print("  %-32s %9.0f  %5.1f%%" % (n, v, 100 * v / tot))

# how evenly the work of a launch is spread: one wave = one ray here (512 workgroups of 8 rays, one workgroup per CU at a time, two rounds)
us = pa[:, 9] / 100.0
wg = us.reshape(512, 8)
print("per-ray wall time (us): mean %.1f  std %.1f  min %.1f  max %.1f" % (us.mean(), us.std(), us.min(), us.max()))
print("per-workgroup (8 rays): mean of the rays %.1f, max of the rays: mean %.1f  min %.1f  max %.1f" % (wg.mean(), wg.max(1).mean(), wg.max(1).min(), wg.max(1).max()))
print("=> a workgroup waits for its slowest ray: %.1f %% above the mean ray; two rounds of 256 workgroups: sum of the two slowest-ray times per CU slot ~ %.1f us vs 2 x mean ray %.1f us"
      % (100 * (wg.max(1).mean() / us.mean() - 1), 2 * wg.max(1).mean(), 2 * us.mean()))
