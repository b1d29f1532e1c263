"""does the ORDER of the rays of a batch matter?  The kernel hands consecutive rays to the waves of an XCD at about the same time, and the XCD's L2
turns over every ~20 us: rays share table sectors only with rays that march in step with them.  Same 65 536 rays of the bench view, 4096 per launch:
  rows   : a batch = 16 image rows (the bench; XCD k = 2 rows)
  blocks : the same 16-row batch, permuted so that XCD k holds a 16 x 32 block, as two 16 x 16 blocks one after the other
  tiles  : a batch = a 64 x 64 image tile; XCD k holds 8 rows x 64 = two 8 x 32 ... (here: 16 x 16 blocks, four per XCD? no: 512 rays = two 16 x 16 blocks)
"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from avatarcraft_amd import nsr_ops

dev = torch.device("cuda:0")
p, field, table, ro, rd = bench.make_inputs(dev, 0)
field.prepare()
H = W = 256
idx = np.arange(H * W).reshape(H, W)


def blocks_of(region, bh, bw):          # region [h, w] of ray ids -> list of blocks [bh, bw] row-major over the region, each flattened row-major
    h, w = region.shape
    return [region[i:i + bh, j:j + bw].reshape(-1) for i in range(0, h, bh) for j in range(0, w, bw)]


orders = {
    "rows": np.concatenate([idx[16 * b:16 * b + 16].reshape(-1) for b in range(16)]),
    "blocks16x16": np.concatenate([np.concatenate(blocks_of(idx[16 * b:16 * b + 16], 16, 16)) for b in range(16)]),
    "blocks8x32": np.concatenate([np.concatenate(blocks_of(idx[16 * b:16 * b + 16], 8, 32)) for b in range(16)]),
    "tiles64_16x16": np.concatenate([np.concatenate(blocks_of(t.reshape(64, 64), 16, 16)) for t in blocks_of(idx, 64, 64)]),
    "tiles64_rows": np.concatenate(blocks_of(idx, 64, 64)),
}
ref = None
for name, perm in orders.items():
    assert np.array_equal(np.sort(perm), np.arange(H * W))
    pt = torch.from_numpy(perm).to(dev)
    ro_t, rd_t = torch.from_numpy(ro).to(dev)[pt].contiguous(), torch.from_numpy(rd).to(dev)[pt].contiguous()
    outs = [dict() for _ in range(16)]
    def step(k, ev=None):
        b = k % 16
        nsr_ops.render_rays(field, ro_t[b * 4096:(b + 1) * 4096], rd_t[b * 4096:(b + 1) * 4096], 64, 64, 1.6, float(p["inv_s"]), out=outs[b], events=ev, precision="fast")
    for k in range(16): step(k)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
    torch.cuda.synchronize()
    for k in range(64): step(k, evs[k])
    torch.cuda.synchronize()
    ms = np.array([s.elapsed_time(e) for s, e in evs])
    img = torch.empty(H * W, 3, device=dev)
    img[pt] = torch.cat([o["image"] for o in outs])
    if ref is None: ref = img.clone()
    print("%-14s kernel %.4f ms per 4096-ray launch (min %.3f max %.3f), image %s" % (name, ms.mean(), ms.min(), ms.max(), "identical" if torch.equal(img, ref) else "DIFFERS"))
