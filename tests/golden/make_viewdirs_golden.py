#!/usr/bin/env python3
"""Golden vectors for NeRFNetwork(use_viewdirs=True) (models/instant_nsr.py:565-569, 644-653: the colour network reads
h = cat[x, sh(d), n, geo_feat], sh = the degree-4 spherical-harmonics encoder of the ray direction), by IMPORTING THE REFERENCE'S PYTHON here.

    python tests/golden/make_viewdirs_golden.py [/root/reference]        -> tests/golden/viewdirs.npz

Same arrangement as make_golden.py (whose stubs it re-uses): the reference's run() / NeRFNetwork / SHEncoder python side are the reference's own; the two
JIT-built CUDA back ends it calls (hash grid, spherical harmonics) cannot be built in this image and are served by the CPU oracle's restatements
(oracle/ac_oracle_ops.c), which the known-answer tests pin.  Stored: the network's parameters (raw weight_v / weight_g as a checkpoint holds them, the
effective matrices), an eval render and a training render (64 + 64 samples) with their per-sample outputs, and the gradients of one training render
(rgb.backward(image_grad) + (0.01 * eikonal).backward(), stylize.py:163-169) for every parameter."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG          # noqa: E402  (stubs; `import models.instant_nsr as ref_nsr`; sys.argv[1] = the reference checkout)
import numpy as np                # noqa: E402
import torch                      # noqa: E402

O = MG.O


class _ShBackend:
    """Stands in for the JIT-built `_sh_encoder` extension (encoder/shencoder/backend.py): the oracle's restatement of shencoder.cu"""

    @staticmethod
    def sh_encode_forward(inputs, outputs, B, D, C, calc_grad_inputs, dy_dx):
        out, dd = O.sh_encode_forward(inputs.detach().numpy(), C, calc_grad_inputs)
        outputs.copy_(torch.from_numpy(out))
        if calc_grad_inputs:
            dy_dx.copy_(torch.from_numpy(dd).reshape(dy_dx.shape))

    @staticmethod
    def sh_encode_backward(grad, inputs, B, D, C, dy_dx, grad_inputs):
        grad_inputs.copy_(torch.from_numpy(O.sh_encode_backward(grad.numpy(), inputs.detach().numpy(), C, dy_dx.numpy())))


def build_net():
    import encoder.shencoder.sphere_harmonics as ref_sh
    ref_sh._backend = _ShBackend
    torch.manual_seed(0)
    net = MG.ref_nsr.NeRFNetwork(use_viewdirs=True)
    assert net.color_net[0].weight_v.shape == (64, 37) and net.encoder_dir.degree == 4
    rs = np.random.RandomState(1234)
    with torch.no_grad():
        scale, _ = O.hash_level_table(16, np.float32(np.log2(net.encoder.per_level_scale)), 16)
        net.level_amp = MG.smooth_level_amp(scale)
        net.encoder.embeddings.copy_(torch.from_numpy(MG.make_table(int(net.encoder.offsets[-1]), offsets=net.encoder.offsets.numpy(), level_amp=net.level_amp)))
        v = net.sdf_net[0].weight_v
        v[:, 3:] = torch.from_numpy(rs.normal(0.0, 0.05, size=(64, 32)).astype(np.float32))
        net.sdf_net[0].bias.copy_(torch.from_numpy(rs.normal(0.0, 0.05, size=64).astype(np.float32)))
        net.sdf_net[1].bias.copy_(torch.from_numpy(rs.normal(0.0, 0.02, size=16).astype(np.float32)))
        net.sdf_net[1].bias[0] = -0.45
        net.deviation_net.variance.fill_(0.3)
        # the stock initialisation gives the 16 direction columns the same small uniform range as the others; make the view dependence clearly visible
        net.color_net[0].weight_v[:, 3:19] *= 3.0
    return net


def main():
    net = build_net()
    ew = MG.effective_weights(net)
    assert ew["Wc1"].shape == (64, 37)
    out = dict(table_seed=np.int64(MG.TABLE_SEED), level_amp=net.level_amp, offsets=net.encoder.offsets.numpy(), per_level_scale=np.float64(net.encoder.per_level_scale),
               inv_s=np.float32(net.forward_variance().item()), **ew)
    for i, l in enumerate(net.sdf_net):
        out[f"sdf_net.{i}.weight_g"] = l.weight_g.detach().numpy(); out[f"sdf_net.{i}.weight_v"] = l.weight_v.detach().numpy(); out[f"sdf_net.{i}.bias"] = l.bias.detach().numpy()
    for i, l in enumerate(net.color_net):
        out[f"color_net.{i}.weight_g"] = l.weight_g.detach().numpy(); out[f"color_net.{i}.weight_v"] = l.weight_v.detach().numpy()
    out["deviation_net.variance"] = net.deviation_net.variance.detach().numpy()
    rs = np.random.RandomState(17)
    ro, rd = MG.make_rays(8, 8, dist=1.7, f=6.25)
    bg = rs.uniform(0, 1, size=(ro.shape[0], 3)).astype(np.float32)
    keys = ("image", "weights_sum", "depth", "normal_map", "weights", "alpha", "color", "z_vals", "gradient_error", "noise")
    for tag, train, seed in (("eval", False, 0), ("train", True, 43)):
        N = ro.shape[0]
        net.train(train)
        noise = None
        if train:
            torch.manual_seed(seed); noise = torch.rand(N, 64).numpy().copy(); torch.manual_seed(seed)
        with torch.no_grad():
            r = net.render(torch.from_numpy(ro)[None], torch.from_numpy(rd)[None], num_steps=64, bound=1.6, upsample_steps=64, staged=False,
                           bg_color=torch.from_numpy(bg), cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, render_can=True, perturb=train)
        c = dict(image=r["rgb"][0].numpy(), weights_sum=r["weight_sum"][:, 0].numpy(), depth=r["depth"][0].numpy(), normal_map=r["normal"].numpy(),
                 weights=r["weights"].numpy(), alpha=r["pts_alpha"].numpy(), color=r["pts_color"].numpy(), z_vals=r["z_vals"].numpy(),
                 gradient_error=np.float32(r["gradient_error"].item()))
        if noise is not None:
            c["noise"] = noise
        for k in keys:
            if k in c:
                out[f"{tag}_{k}"] = c[k]
        print(tag, "weights_sum mean", c["weights_sum"].mean(), "image mean", c["image"].mean())
    out["rays_o"], out["rays_d"], out["bg"] = ro, rd, bg
    # rays on which the oracle's sample positions differ from the reference's by more than the interpolation noise (a flipped searchsorted knife-edge of the
    # up-sampling moves every later sample of the ray: tests/golden/make_golden.py `oracle_ss_flips`): recorded, so that the parity tests require
    # exactly this set instead of tolerating a percentage
    table = MG.make_table(int(net.encoder.offsets[-1]), offsets=net.encoder.offsets.numpy(), level_amp=net.level_amp)
    of = O.Field(table, net.encoder.offsets.numpy(), ew["W1"], ew["b1"], ew["W2"], ew["b2"], ew["Wc1"], ew["Wc2"], ew["Wc3"], float(net.encoder.per_level_scale))
    for tag in ("eval", "train"):
        r = O.render_rays(of, ro, rd, 64, 64, 1.6, float(out["inv_s"]), bg=bg, noise=out.get(f"{tag}_noise"))
        dz = np.abs(np.asarray(r["z_vals"]).reshape(out[f"{tag}_z_vals"].shape) - out[f"{tag}_z_vals"]).max(1)
        out[f"{tag}_oracle_z_flips"] = np.nonzero(dz > 2e-3)[0].astype(np.int32)
        print(tag, "rays with a z flip:", out[f"{tag}_oracle_z_flips"].tolist())
    # how much of the picture is view dependence: the same render with the direction columns zeroed
    with torch.no_grad():
        keep = net.color_net[0].weight_v[:, 3:19].clone()
        net.color_net[0].weight_v[:, 3:19] = 0.0
        net.eval()
        r0 = net.render(torch.from_numpy(ro)[None], torch.from_numpy(rd)[None], num_steps=64, bound=1.6, upsample_steps=64, staged=False,
                        bg_color=torch.from_numpy(bg), cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, render_can=True, perturb=False)
        net.color_net[0].weight_v[:, 3:19] = keep
    out["view_dependence_max"] = np.float32(np.abs(r0["rgb"][0].numpy() - out["eval_image"]).max())
    print("max |image - image without direction columns|", out["view_dependence_max"])
    # gradients of one training render (256 jittered rays, white background)
    ro2, rd2 = MG.make_rays(16, 16, dist=1.8, f=10.0, jitter_seed=12)
    bg2 = np.ones((ro2.shape[0], 3), np.float32)
    net.train(True); net.zero_grad()
    torch.manual_seed(44); noise_g = torch.rand(ro2.shape[0], 64).numpy().copy(); torch.manual_seed(44)
    og = net.render(torch.from_numpy(ro2)[None], torch.from_numpy(rd2)[None], num_steps=64, bound=1.6, upsample_steps=64, staged=False,
                    bg_color=torch.from_numpy(bg2), cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, render_can=True, perturb=True)
    img_grad = np.clip(np.random.RandomState(8).normal(0, 1, (ro2.shape[0], 3)), -1, 1).astype(np.float32)
    og["rgb"][0].backward(gradient=torch.from_numpy(img_grad), retain_graph=True)
    (og["gradient_error"] * 0.01).backward()
    out.update(g_rays_o=ro2, g_rays_d=rd2, g_bg=bg2, g_noise=noise_g, g_img_grad=img_grad, g_rgb=og["rgb"][0].detach().numpy(), g_z_vals=og["z_vals"].detach().numpy())
    for k, prm in net.named_parameters():
        if k != "encoder.embeddings":
            out["grad." + k] = prm.grad.numpy().copy()
    ge = net.encoder.embeddings.grad.numpy()
    nz = np.flatnonzero(np.abs(ge).sum(1))
    pick = nz[np.random.RandomState(6).choice(len(nz), 4096, replace=False)]
    out["emb_idx"] = pick.astype(np.int64); out["emb_grad"] = ge[pick].copy(); out["emb_nnz"] = np.int64(len(nz))
    print("grad color_net.0.weight_v: |direction columns| max", np.abs(out["grad.color_net.0.weight_v"][:, 3:19]).max(), "others", np.abs(out["grad.color_net.0.weight_v"]).max())
    np.savez_compressed(os.path.join(HERE, "viewdirs.npz"), **out)


if __name__ == "__main__":
    main()
