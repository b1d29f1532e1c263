"""per-level timing of the stencil hash backward on the REAL sample distribution of a training batch (4096 rays x 128
importance-sampled mid points from the fused renderer), to see where the atomics serialise"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from avatarcraft_amd import _lib as L
if os.environ.get('AC_LIB_PATH'): L.LIB_PATH = os.environ['AC_LIB_PATH']
from avatarcraft_amd import nsr_ops
from tests.common import make_rays, load_golden
from tests.gpu_common import device_field
dev = "cuda:0"
p = load_golden("nsr_params.npz")
f, _ = device_field(p)
ro, rd = make_rays(64, 64, dist=1.8, f=50.0)
t = lambda a: torch.from_numpy(a).to(dev)
out = nsr_ops.render_rays(f, t(ro), t(rd), 64, 64, 1.6, float(p["inv_s"]), extras=True)
z = out["z_vals"]
d = z[:, 1:] - z[:, :-1]
zm = torch.cat([z[:, :-1] + 0.5 * d, z[:, -1:]], 1)
x = (t(ro)[:, None, :] + t(rd)[:, None, :] * zm[:, :, None]).clamp(-1.6, 1.6).reshape(-1, 3).contiguous()
B = x.shape[0]
print("points", B, "weights_sum mean", float(out["weights_sum"].mean()))
scales = [15, 21.1, 29.6, 41.2, 57.4, 79.6, 110.4, 153.0, 211.8, 293.1, 405.4, 560.6, 775.0, 1071.4, 1481.0, 2047.0]
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
tot = 0
st = L.current_stream(torch.device(dev))
for l, sc in enumerate(scales):
    H = int(round(sc)) + 1
    size = min(2 ** 19, (H + 1) ** 3)
    offs = np.array([0, size], np.int32)
    gg = torch.zeros(size, 2, device=dev)
    grad = torch.randn(7, 1, B, 2, device=dev)
    nb = int(L.lib().ac_hash_stencil_backward_scratch(offs.ctypes.data, 1, 0.0, H, int(os.environ.get('COPIES', 16)), B if os.environ.get('BINNED', '1') == '1' else 0))
    scr = torch.empty(nb, dtype=torch.uint8, device=dev) if nb else None
    tm = timeit(lambda: L.check(L.lib().ac_hash_stencil_backward(grad.data_ptr(), x.data_ptr(), offs.ctypes.data, gg.data_ptr(), B, 2, 1, 0.0, H, 0.005, 1.6, scr.data_ptr() if scr is not None else None, nb, st)))
    tot += tm
    print(f"level {l:2d} scale {sc:7.1f} size {size:7d} {'dense' if (H+1)**3 <= size else 'hash '}  {tm:7.3f} ms")
print("sum", tot)
