"""Would the two renders of net_style in one SDS step (render_val and the training forward: same rays, different jitter noise) gain from running in ONE launch
with the two copies of a ray next to each other in the hand-out order, so that they meet in their XCD's L2?  (round-2 verdict, item 3a)
    A + B      two 4096-ray launches (what the step does)
    concat     one 8192-ray launch, [A rays, B rays]                  (control: what a larger launch alone buys)
    pairs      one 8192-ray launch, a0 b0 a1 b1 ...                   (the proposal)
    pairs_same the same with identical noise in both copies           (upper bound of the sharing)
"""
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
prec = sys.argv[1] if len(sys.argv) > 1 else "exact"
g = torch.Generator().manual_seed(3)
n1, n2 = torch.rand((4096, 64), generator=g).to(dev), torch.rand((4096, 64), generator=g).to(dev)


def timed(ro, rd, nz, reps=40):
    out = {}
    for _ in range(6):
        nsr_ops.render_rays(field, ro, rd, 64, 64, 1.6, float(p["inv_s"]), noise=nz, out=out, precision=prec)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    torch.cuda.synchronize()
    for k in range(reps):
        nsr_ops.render_rays(field, ro, rd, 64, 64, 1.6, float(p["inv_s"]), noise=nz, out=out, events=evs[k], precision=prec)
    torch.cuda.synchronize()
    return float(np.mean([s.elapsed_time(e) for s, e in evs])), out["image"].clone()


def inter(a, b):
    return torch.stack([a, b], 1).reshape((-1,) + a.shape[1:]).contiguous()


for view_name, (ro_np, rd_np) in (("sds view (64 x 64, stride 4)", bench.sds_view(0)),):
    ro, rd = torch.from_numpy(ro_np).to(dev), torch.from_numpy(rd_np).to(dev)
    for rep in range(2):
        ta, ia = timed(ro, rd, n1)
        tb, ib = timed(ro, rd, n2)
        tc, ic = timed(torch.cat([ro, ro]), torch.cat([rd, rd]), torch.cat([n1, n2]))
        tp, ip = timed(inter(ro, ro), inter(rd, rd), inter(n1, n2))
        ts, _ = timed(inter(ro, ro), inter(rd, rd), inter(n1, n1))
        ok = torch.equal(ic[:4096], ia) and torch.equal(ic[4096:], ib) and torch.equal(ip[0::2], ia) and torch.equal(ip[1::2], ib)
        print("%s [%s]: A %.4f + B %.4f = %.4f ms | concat %.4f | pairs %.4f | pairs, same noise %.4f | images %s"
              % (view_name, prec, ta, tb, ta + tb, tc, tp, ts, "identical" if ok else "DIFFER"))
