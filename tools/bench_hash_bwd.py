"""per-level timing of the stand-alone hash-grid backward (B = 4096 rays x 128 samples), to locate atomic contention"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from avatarcraft_amd.encoder.hashencoder.backend import _backend
from tests.common import make_rays
dev = "cuda:0"
ro, rd = make_rays(64, 64, dist=1.8, f=50.0)
z = np.linspace(0.4, 3.2, 128, dtype=np.float32)
pts = (ro[:, None, :] + rd[:, None, :] * z[None, :, None]).reshape(-1, 3).clip(-1.6, 1.6)
x = torch.from_numpy(((pts + 1.6) / 3.2).astype(np.float32)).to(dev)
B = x.shape[0]
scales = [15, 21.1, 29.6, 41.2, 57.4, 79.6, 110.4, 153.0, 211.8, 293.1, 405.4, 560.6, 775.0, 1071.4, 1481.0, 2047.0]
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
tot = 0
for l, sc in enumerate(scales):
    H = int(round(sc)) + 1
    res = H
    size = min(2 ** 19, (res + 1) ** 3)
    offsets = torch.tensor([0, size], dtype=torch.int32, device=dev)
    emb = torch.zeros(size, 2, device=dev); gg = torch.zeros_like(emb)
    grad = torch.randn(1, B, 2, device=dev)
    dummy = torch.zeros(1, device=dev)
    t = timeit(lambda: _backend.hash_encode_backward(grad, x, emb, offsets, gg, B, 3, 2, 1, 0.0, H, False, dummy, dummy))
    tot += t
    print(f"level {l:2d} scale {sc:7.1f} size {size:7d} {'dense' if (res+1)**3 <= size else 'hash '}  {t:7.3f} ms")
print("sum", tot)
# full 16-level call
from oracle import oracle as O
offs, pls = O.hash_offsets(desired_resolution=2048)
offsets = torch.from_numpy(offs).to(dev); emb = torch.zeros(int(offs[-1]), 2, device=dev); gg = torch.zeros_like(emb)
grad = torch.randn(16, B, 2, device=dev); dummy = torch.zeros(1, device=dev)
print("16-level call", timeit(lambda: _backend.hash_encode_backward(grad, x, emb, offsets, gg, B, 3, 2, 16, float(np.log2(pls)), 16, False, dummy, dummy)), "ms")
out = torch.empty(16, B, 2, device=dev)
print("16-level forward", timeit(lambda: _backend.hash_encode_forward(x, emb, offsets, out, B, 3, 2, 16, float(np.log2(pls)), 16, False, dummy)), "ms")

# the same 16-level call through the binned scatter
from avatarcraft_amd import _lib as L
nb = int(L.lib().ac_hash_encode_backward_scratch(offs.ctypes.data, 3, 2, 16, float(np.float32(np.log2(pls))), 16, B))
sc = torch.empty(nb, dtype=torch.uint8, device=dev)
st = L.current_stream(torch.device(dev))
print("16-level call, binned", timeit(lambda: L.check(L.lib().ac_hash_encode_backward_ws(grad.data_ptr(), x.data_ptr(), emb.data_ptr(), offsets.data_ptr(), offs.ctypes.data,
      gg.data_ptr(), B, 3, 2, 16, float(np.float32(np.log2(pls))), 16, 0, dummy.data_ptr(), dummy.data_ptr(), sc.data_ptr(), nb, st))), "ms")

# and on the real sample distribution of a training batch (importance-sampled mid points of the fused renderer)
from avatarcraft_amd import nsr_ops
from tests.common import load_golden
from tests.gpu_common import device_field
pp = load_golden("nsr_params.npz"); f, _ = device_field(pp)
t = lambda a: torch.from_numpy(a).to(dev)
z = nsr_ops.sample_rays(f, t(ro), t(rd), 64, 64, 1.6)
d = z[:, 1:] - z[:, :-1]; zm = torch.cat([z[:, :-1] + 0.5 * d, z[:, -1:]], 1)
xr = (((t(ro)[:, None, :] + t(rd)[:, None, :] * zm[:, :, None]).clamp(-1.6, 1.6).reshape(-1, 3) + 1.6) / 3.2).contiguous()
print("real distribution: direct", timeit(lambda: L.check(L.lib().ac_hash_encode_backward_ws(grad.data_ptr(), xr.data_ptr(), emb.data_ptr(), offsets.data_ptr(), offs.ctypes.data,
      gg.data_ptr(), B, 3, 2, 16, float(np.float32(np.log2(pls))), 16, 0, dummy.data_ptr(), dummy.data_ptr(), None, 0, st))), "ms;  binned",
      timeit(lambda: L.check(L.lib().ac_hash_encode_backward_ws(grad.data_ptr(), xr.data_ptr(), emb.data_ptr(), offsets.data_ptr(), offs.ctypes.data,
      gg.data_ptr(), B, 3, 2, 16, float(np.float32(np.log2(pls))), 16, 0, dummy.data_ptr(), dummy.data_ptr(), sc.data_ptr(), nb, st))), "ms")
