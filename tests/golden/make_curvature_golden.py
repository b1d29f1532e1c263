#!/usr/bin/env python3
"""Golden vectors of the curvature term (models/instant_nsr.py:276-288, NeRFNetwork(curvature_loss=True)): tests/golden/curvature.npz.

    python tests/golden/make_curvature_golden.py            (in the build container: imports the reference's Python through make_golden.py's stubs)

One training render of 256 rays by the REFERENCE's NeRFRenderer.run on the CPU (the hash back end is served by oracle/, including its dy_dx /
grad_inputs path -- the perturbed points are a function of the normal, so the encoder's input requires grad: hashgrid.py calc_grad_inputs), then
`curvature_error.backward()` alone on the reference's own autograd graph.  Recorded: the inputs (rays, the jitter draw of :162, the torch.randn_like draw of :278),
curvature_error, gradient_error, the image, and .grad of every parameter (the table gradient at 4096 sampled entries + its norm and non-zero count).
The network is make_golden.build_reference_net (nsr_params.npz): the same parameters as every other golden."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.argv = sys.argv[:1]
import make_golden as MG  # noqa: E402  (stubs the absent extensions, imports the reference)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests.common import make_rays  # noqa: E402


def main():
    net = MG.build_reference_net(curvature_loss=True)
    assert net.curvature_loss
    ro, rd = make_rays(16, 16, dist=1.8, f=10.0, jitter_seed=11)            # the rays of train_grad.npz
    bg = np.ones((ro.shape[0], 3), np.float32)
    net.train(True)
    net.zero_grad()
    torch.manual_seed(42)
    noise = torch.rand(ro.shape[0], 64).numpy().copy()
    torch.manual_seed(42)
    drawn = []
    orig = torch.randn_like

    def rec(t, *a, **k):
        r = orig(t, *a, **k); drawn.append(r.detach().clone()); return r
    torch.randn_like = rec
    try:
        out = net.render(torch.from_numpy(ro)[None], torch.from_numpy(rd)[None], num_steps=64, bound=1.6, upsample_steps=64, staged=False,
                         bg_color=torch.from_numpy(bg), cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, render_can=True, perturb=True)
    finally:
        torch.randn_like = orig
    assert len(drawn) == 1 and drawn[0].shape == (ro.shape[0] * 128, 3)
    cerr = out["curvature_error"]
    cerr.backward()
    g = dict(rays_o=ro, rays_d=rd, bg=bg, noise=noise, randn=drawn[0].numpy().copy(), rgb=out["rgb"][0].detach().numpy(), z_vals=out["z_vals"].detach().numpy(),
             curvature_error=np.float64(cerr.item()), gradient_error=np.float64(out["gradient_error"].item()))
    for k, prm in net.named_parameters():
        if k != "encoder.embeddings":
            g["grad." + k] = (prm.grad.numpy().copy() if prm.grad is not None else np.zeros(tuple(prm.shape), np.float32))
    ge = net.encoder.embeddings.grad.numpy()
    nz = np.flatnonzero(np.abs(ge).sum(1))
    pick = nz[np.random.RandomState(6).choice(len(nz), 4096, replace=False)]
    g["emb_idx"] = pick.astype(np.int64); g["emb_grad"] = ge[pick].copy()
    g["emb_nnz"] = np.int64(len(nz)); g["emb_l2"] = np.float64(np.sqrt((ge.astype(np.float64) ** 2).sum())); g["emb_max"] = np.float64(np.abs(ge).max())
    np.savez_compressed(os.path.join(HERE, "curvature.npz"), **g)
    print("curvature_error", g["curvature_error"], "gradient_error", g["gradient_error"], "emb nnz", len(nz), "l2", g["emb_l2"],
          {k: float(np.abs(v).max()) for k, v in g.items() if k.startswith("grad.")})


if __name__ == "__main__":
    sys.exit(main())
