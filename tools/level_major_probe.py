"""How fast is a LEVEL-MAJOR gather of the final pass's stencil (one lane per point, one level per workgroup row: ac_hash_stencil_forward, the stand-alone
operator) on the sample positions of a real 4096-ray batch -- against the ~0.28 ms the fused renderer spends in its stencil phase per launch?"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from avatarcraft_amd import _lib as L, nsr_ops
from tests.common import make_rays, load_golden
from tests.gpu_common import device_field

dev = torch.device("cuda:0")
p = load_golden("nsr_params.npz")
f, _ = device_field(p, device=dev)
for name, (ro, rd) in (("bench batch", tuple(a[:4096] for a in make_rays(256, 256, dist=1.7, f=200.0, yaw=0.0, pitch=0.0))), ("sds view", bench.sds_view(0))):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    out = nsr_ops.render_rays(f, t(ro), t(rd), 64, 64, 1.6, float(p["inv_s"]), extras=True, train_extras=True)
    x = out["pts"].reshape(-1, 3).contiguous()
    B = x.shape[0]
    offs = np.ascontiguousarray(p["offsets"], dtype=np.int32)
    res = torch.empty((7, 16, B, 2), device=dev)
    st = L.current_stream(dev)
    call = lambda: L.check(L.lib().ac_hash_stencil_forward(x.data_ptr(), f.t["table"].data_ptr(), offs.ctypes.data, res.data_ptr(), B, 2, 16, f.S, f.H, 0.005, 1.6, st))
    for _ in range(3): call()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for s, e in evs:
        s.record(); call(); e.record()
    torch.cuda.synchronize()
    ms = np.mean([s.elapsed_time(e) for s, e in evs])
    print("%s: %d points x 7 x 16 levels, level-major stand-alone operator %.3f ms  (writes %.2f GB of features)" % (name, B, ms, res.numel() * 4 / 1e9))
