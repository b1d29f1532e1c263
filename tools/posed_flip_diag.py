"""Which rays of the posed goldens (tests/golden/warp_render.npz) does the GPU render differently from the CPU oracle, and where does the difference start?"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from tests.common import make_body, oracle_field_from_golden, load_golden
from tests.test_gpu_model import golden_net, DEV
from avatarcraft_amd import nsr_ops

g = load_golden("warp_render.npz"); p = load_golden("nsr_params.npz")
verts, faces, Ts = make_body()
field = oracle_field_from_golden(p)
net, _ = golden_net(); net.eval()
ro, rd = torch.from_numpy(g["rays_o"]).to(DEV), torch.from_numpy(g["rays_d"]).to(DEV)
for tag, guide in (("guide", True), ("noguide", False)):
    r = O.render_rays(field, g["rays_o"], g["rays_d"], 32, 32, 1.6, float(p["inv_s"]), bg=g["bg"], warp=dict(verts=verts, faces=faces, Ts=Ts, use_mesh_guide=guide))
    with torch.no_grad():
        out = net.render(ro[None], rd[None], num_steps=32, bound=1.6, upsample_steps=32, staged=False, bg_color=torch.from_numpy(g["bg"]).to(DEV),
                         cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, render_can=False, verts=verts, faces=faces, Ts=Ts, perturb=False, use_mesh_guide=guide)
    zg, zo, zr = out["z_vals"].cpu().numpy(), np.asarray(r["z_vals"]).reshape(256, 64), g[f"{tag}_z_vals"]
    neq = np.nonzero((zg.view(np.uint32) != zo.view(np.uint32)).any(1))[0]
    print(tag, "rays where GPU z != oracle z bitwise:", neq.tolist())
    print(tag, "GPU flips vs reference:", np.nonzero(np.abs(zg - zr).max(1) > 1e-4)[0].tolist())
    print(tag, "oracle flips vs reference:", np.nonzero(np.abs(zo - zr).max(1) > 1e-4)[0].tolist())
    for ray in neq[:4]:
        i = int(np.nonzero(zg[ray].view(np.uint32) != zo[ray].view(np.uint32))[0][0])
        print("  ray", ray, "first differing sample", i, "gpu", zg[ray, i], "oracle", zo[ray, i], "ref", zr[ray, i])
    img = out["rgb"][0].cpu().numpy()
    print(tag, "image max |gpu - oracle|", np.abs(img - np.asarray(r["image"])).max())
