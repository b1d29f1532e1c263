"""stand-alone hash encoder forward (ac_hash_encode_forward through HashEncoder) on uniform points and on the importance-sampled points of a training batch"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from avatarcraft_amd import _lib as L
if os.environ.get('AC_LIB_PATH'): L.LIB_PATH = os.environ['AC_LIB_PATH']
from avatarcraft_amd.encoder.hashencoder.hashgrid import HashEncoder
dev = "cuda:0"
enc = HashEncoder(3, 16, 2, 1.381912879967776, 16, 19, 2048).to(dev)          # the default NeRFNetwork's grid
with torch.no_grad():
    enc.embeddings.uniform_(-0.1, 0.1)
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for B in (524288, 4 * 524288):
    x = torch.rand(B, 3, device=dev) * 2 - 1
    with torch.no_grad():
        t = timeit(lambda: enc(x, 1.0))
    print("hash forward, %d uniform points x 16 levels: %.3f ms  (%.2f TB/s of 8-byte corner gathers)" % (B, t, B * 16 * 8 * 8 / t / 1e9))
    xg = x.clone().requires_grad_(True)
    t = timeit(lambda: enc(xg, 1.0))
    print("   with d/dx (dy_dx written): %.3f ms" % t)
