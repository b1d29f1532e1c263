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
