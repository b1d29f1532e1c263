"""ray order inside the 64 x 64 stride-4 training patch of the SDS step (BASELINE configuration 3): the same 4096 rays, one launch, different orders.
The kernel deals the batch to the XCDs in chunks of 512 consecutive rays (XCD k = 8 rows of the patch in row order)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from avatarcraft_amd import nsr_ops
from tests.common import load_golden
from tests.gpu_common import device_field

dev = torch.device("cuda:0")
p = load_golden("nsr_params.npz")
field, _ = device_field(p, device=dev)
field.prepare()
ro, rd = bench.sds_view(0)
idx = np.arange(4096).reshape(64, 64)


def blocks_of(region, bh, bw):
    h, w = region.shape
    return [region[i:i + bh, j:j + bw].reshape(-1) for i in range(0, h, bh) for j in range(0, w, bw)]


def morton():
    def part(v):
        v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F; v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555
        return v
    y, x = np.divmod(np.arange(4096), 64)
    return np.argsort(part(x) | (part(y) << 1), kind="stable")


orders = {"rows": idx.reshape(-1), "blocks8x8": np.concatenate(blocks_of(idx, 8, 8)), "blocks16x16": np.concatenate(blocks_of(idx, 16, 16)),
          "blocks16x32": np.concatenate(blocks_of(idx, 16, 32)), "blocks32x16": np.concatenate(blocks_of(idx, 32, 16)), "morton": morton(),
          "cols": idx.T.reshape(-1)}
noise = torch.rand((4096, 64), generator=torch.Generator().manual_seed(3)).to(dev)
ref = None
prec = sys.argv[1] if len(sys.argv) > 1 else "exact"
for rep in range(2):
    for name, perm in orders.items():
        pt = torch.from_numpy(perm).to(dev)
        ro_t, rd_t, nz = torch.from_numpy(ro).to(dev)[pt].contiguous(), torch.from_numpy(rd).to(dev)[pt].contiguous(), noise[pt].contiguous()
        out = {}
        for k in range(8):
            nsr_ops.render_rays(field, ro_t, rd_t, 64, 64, 1.6, float(p["inv_s"]), noise=nz, out=out, precision=prec)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(48)]
        torch.cuda.synchronize()
        for k in range(48):
            nsr_ops.render_rays(field, ro_t, rd_t, 64, 64, 1.6, float(p["inv_s"]), noise=nz, out=out, events=evs[k], precision=prec)
        torch.cuda.synchronize()
        ms = np.array([s.elapsed_time(e) for s, e in evs])
        img = torch.empty(4096, 3, device=dev); img[pt] = out["image"]
        if ref is None: ref = img.clone()
        print("%-12s kernel %.4f ms (min %.3f max %.3f) image %s" % (name, ms.mean(), ms.min(), ms.max(), "identical" if torch.equal(img, ref) else "DIFFERS"))
