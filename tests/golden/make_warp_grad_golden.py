#!/usr/bin/env python3
"""tests/golden/warp_grad.npz: the reference's posed-space render UNDER AUTOGRAD -- NeRFRenderer.run(render_can=False, verts, faces, Ts)
(models/instant_nsr.py:147-172,198-203,246-249) is differentiable w.r.t. the network (the SMPL inverse warp itself is numpy: the
warped points and the mask are constants).  Run HERE on the CPU by importing the reference (it cannot travel; only the vectors do):

    python tests/golden/make_warp_grad_golden.py

Training mode (jittered coarse samples, the noise is recorded), 32 + 32 samples, mesh-guided near / far, the libigl stand-in of
make_golden.py (closest point from the oracle's fp64 Ericson routine).  Loss = sum(rgb * G) + 0.01 * gradient_error + sum(weight_sum * Gw)
+ sum(normal * Gn): one backward; recorded: inputs, forward outputs, the .grad of every parameter (4096 sampled table entries)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG           # stubs the CUDA back ends, imports the reference's models/instant_nsr.py   # noqa: E402

import numpy as np                 # noqa: E402
import torch                       # noqa: E402


def main():
    from tests.common import make_body, make_rays
    MG.install_igl_standin()
    verts, faces, Ts = make_body()
    net = MG.build_reference_net()
    net.train(True)
    net.zero_grad()
    ro, rd = make_rays(16, 16, dist=1.8, f=14.0, jitter_seed=5)
    N = ro.shape[0]
    bg = np.random.RandomState(2).uniform(0, 1, (N, 3)).astype(np.float32)
    torch.manual_seed(77)
    noise = torch.rand(N, 32).numpy().copy()
    torch.manual_seed(77)
    out = net.render(torch.from_numpy(ro)[None], torch.from_numpy(rd)[None], num_steps=32, bound=1.6, upsample_steps=32, staged=False,
                     bg_color=torch.from_numpy(bg), cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, render_can=False, verts=verts,
                     faces=faces, Ts=Ts, perturb=True, use_mesh_guide=True)
    rs = np.random.RandomState(9)
    G = np.clip(rs.normal(0, 1, (N, 3)), -1, 1).astype(np.float32)
    Gw = rs.normal(0, 1, (N, 1)).astype(np.float32)
    Gn = rs.normal(0, 0.3, (N, 3)).astype(np.float32)
    G[0] = 0.0; Gw[0] = 0.0; Gn[0] = 0.0       # ray 0 runs along the test body's medial axis, where the closest face (hence the warp) flips with the last
                                               # ulp of z (tests/test_oracle_golden.py:MEDIAL_RAY): it carries no upstream gradient
    loss = (out["rgb"][0] * torch.from_numpy(G)).sum() + 0.01 * out["gradient_error"] + (out["weight_sum"] * torch.from_numpy(Gw)).sum() + \
           (out["normal"] * torch.from_numpy(Gn)).sum()
    loss.backward()
    g = dict(rays_o=ro, rays_d=rd, bg=bg, noise=noise, G=G, Gw=Gw[:, 0], Gn=Gn, rgb=out["rgb"][0].detach().numpy(),
             weight_sum=out["weight_sum"][:, 0].detach().numpy(), normal=out["normal"].detach().numpy(), z_vals=out["z_vals"].detach().numpy(),
             alpha=out["pts_alpha"].detach().numpy(), gradient_error=np.float32(out["gradient_error"].item()))
    for k, prm in net.named_parameters():
        if k != "encoder.embeddings":
            g["grad." + k] = prm.grad.numpy().copy()
    ge = net.encoder.embeddings.grad.numpy()
    nz = np.flatnonzero(np.abs(ge).sum(1))
    pick = nz[np.random.RandomState(6).choice(len(nz), min(4096, len(nz)), replace=False)]
    g["emb_idx"] = pick.astype(np.int64); g["emb_grad"] = ge[pick].copy(); g["emb_nnz"] = np.int64(len(nz))
    g["emb_l2"] = np.float64(np.sqrt((ge.astype(np.float64) ** 2).sum()))
    np.savez_compressed(os.path.join(HERE, "warp_grad.npz"), **g)
    print("warp_grad: rays", N, "mean opacity", float(out["weight_sum"].mean()), "masked-in samples", float((out["pts_alpha"] > 0).float().mean()),
          "emb nnz", len(nz), "emb l2", g["emb_l2"], "variance grad", g["grad.deviation_net.variance"])


if __name__ == "__main__":
    main()
