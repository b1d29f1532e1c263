"""What the model variants cost next to the default model: use_viewdirs=True (inside the fused renderer since round 5) and
curvature_loss=True (round 6: the fused operators + one more stencil query with a position gradient; rounds 3 - 5: torch MLPs over the HIP hash encoder,
5.3 / 93 ms); AC_VARIANT_ONLY=curvature restricts the run to one variant (for a kernel trace).
    python tools/generic_path_timing.py      (on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from avatarcraft_amd.instant_nsr import NeRFNetwork
from avatarcraft_amd.synthetic import make_rays

dev = "cuda:0"
ro, rd = make_rays(64, 64, dist=1.7, f=50.0)
ro, rd = torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev)
kw = dict(num_steps=64, bound=1.6, upsample_steps=64, staged=False, bg_color=None, cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, render_can=True)


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


only = os.environ.get("AC_VARIANT_ONLY", "")
for name, opts in (("default (fused)", {}), ("use_viewdirs=True", dict(use_viewdirs=True)), ("curvature_loss=True", dict(curvature_loss=True))):
    if only and only not in name:
        continue
    torch.manual_seed(1)
    net = NeRFNetwork(**opts).to(dev)
    with torch.no_grad():
        net.encoder.embeddings.uniform_(-0.05, 0.05)
        net.sdf_net[0].weight_v[:, 3:].normal_(0, 0.05)

    def infer():
        with torch.no_grad():
            net.eval().render(ro[None], rd[None], perturb=False, **kw)

    def train():
        net.train().zero_grad()
        out = net.render(ro[None], rd[None], perturb=True, **kw)
        (out["rgb"].sum() + out["gradient_error"] + (out["curvature_error"] if torch.is_tensor(out["curvature_error"]) else 0.0)).backward()
    print(f"{name:22s} 4096 rays x (64+64): inference {timed(infer):8.3f} ms   forward + backward {timed(train):8.3f} ms")
