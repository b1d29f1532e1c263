"""The fp64 backward of the render core in oracle/ac_oracle_bwd.c (test infrastructure):
  * CPU: pinned against the reference's own autograd -- tests/golden/train_grad.npz, recorded by running the reference's
    NeRFNetwork.render + the three backward passes of stylize.py:163-193 on 256 rays (tests/golden/make_golden.py);
  * -m gpu: the HIP backward (ac_render_core_backward: sdf_train.hip + hash_stencil.hip) against it on the 4096-ray patch of BASELINE
    configuration 3 -- every MLP gradient and 65 536 sampled table entries within 3e-4 of each tensor's largest entry (fp32 accumulation of 524 288 terms against fp64)."""
import numpy as np
import pytest

from tests.common import load_golden, oracle_field_from_golden


def _chain_to_raw(O, p, r):
    """effective-matrix gradients -> gradients of the reference's parameters (weight_v / weight_g / bias / variance), float64"""
    out = {}
    for i, (w, b) in enumerate((("W1", "b1"), ("W2", "b2"))):
        gv, gg = O.weight_norm_backward(p[f"sdf_net.{i}.weight_v"], p[f"sdf_net.{i}.weight_g"], r["g_" + w])
        out[f"sdf_net.{i}.weight_v"], out[f"sdf_net.{i}.weight_g"], out[f"sdf_net.{i}.bias"] = gv, gg, r["g_" + b]
    for i, w in enumerate(("Wc1", "Wc2", "Wc3")):
        gv, gg = O.weight_norm_backward(p[f"color_net.{i}.weight_v"], p[f"color_net.{i}.weight_g"], r["g_" + w])
        out[f"color_net.{i}.weight_v"], out[f"color_net.{i}.weight_g"] = gv, gg
    inv_s = float(p["inv_s"])
    out["deviation_net.variance"] = np.float64(r["g_inv_s"] * 10.0 * inv_s)          # inv_s = exp(10 variance) (instant_nsr.py:725-726; the clip is inactive)
    return out


def test_oracle_backward_matches_reference_autograd(oracle):
    O = oracle
    p, g = load_golden("nsr_params.npz"), load_golden("train_grad.npz")
    f = oracle_field_from_golden(p)
    N = g["rays_o"].shape[0]
    assert N == 256
    inv_s = float(p["inv_s"])
    # (1) rgb.backward(image_grad) + (0.01 * eikonal).backward()   stylize.py:163-169
    r = O.render_core_backward(f, g["rays_o"], g["rays_d"], g["z_vals"], 64, 64, 1.6, inv_s, bg=g["bg"], g_image=g["img_grad"], g_eik=0.01)
    assert np.abs(r["image"] - g["rgb"]).max() <= 2e-5                        # the fp64 forward reproduces the reference's fp32 image
    assert np.abs(r["weights_sum"] - g["opacity_pred"]).max() <= 2e-5
    raw = _chain_to_raw(O, p, r)
    worst = {}
    for k, mine in raw.items():
        ref = g["grad." + k].astype(np.float64)
        worst[k] = float(np.abs(np.asarray(mine).reshape(ref.shape) - ref).max() / np.abs(ref).max())
        assert worst[k] <= 3e-4, (k, worst[k])                               # observed <= 4e-5: the reference's own fp32 round-off
    ge = r["g_table"][g["emb_idx"]]
    assert np.abs(ge - g["emb_grad"]).max() <= 3e-4 * np.abs(g["emb_grad"]).max()
    assert int((np.abs(r["g_table"]).sum(1) > 0).sum()) == int(g["emb_nnz"])
    assert abs(np.sqrt((r["g_table"] ** 2).sum()) - float(g["emb_l2"])) <= 1e-4 * float(g["emb_l2"])
    # (2) + the opacity term: smooth_l1(clamp(weights_sum), clamp(weights_sum of the frozen net_gt)) * 1e5 (mean over the rays)   stylize.py:185-193
    pred, gt = g["opacity_pred"].astype(np.float64), g["opacity_gt"].astype(np.float64)
    d = np.clip(pred, 0, 1) - gt
    g_ws = np.where(np.abs(d) < 1.0, d, np.sign(d)) * (1e5 / N) * ((pred >= 0) & (pred <= 1))
    r3 = O.render_core_backward(f, g["rays_o"], g["rays_d"], g["z_vals"], 64, 64, 1.6, inv_s, bg=g["bg"], g_image=g["img_grad"], g_weights_sum=g_ws, g_eik=0.01)
    raw3 = _chain_to_raw(O, p, r3)
    for k, mine in raw3.items():
        ref = g["grad3." + k].astype(np.float64)
        e = float(np.abs(np.asarray(mine).reshape(ref.shape) - ref).max() / np.abs(ref).max())
        assert e <= 3e-4, (k, e)
    # (the 1e5-weighted opacity term dominates the table gradient; the reference adds its three backward passes in fp32, entries up to 19)
    assert np.abs(r3["g_table"][g["emb_idx"]] - g["emb_grad3"]).max() <= 1e-3 * np.abs(g["emb_grad3"]).max()



def _posed_inputs(O, g):
    """the constants of a posed render's differentiation, recomputed by the oracle from the golden's z values: mesh-guided near / far, the SMPL inverse
    warp of the mid points (instant_nsr.py:190-207) and its mask"""
    from tests.common import make_body
    verts, faces, Ts = make_body()
    ro, rd, z = g["rays_o"], g["rays_d"], g["z_vals"]
    N, T = z.shape
    near, far = O.mesh_near_far(ro, rd, verts, 0.05)
    delta = z[:, 1:] - z[:, :-1]
    zmid = np.concatenate([z[:, :-1] + np.float32(0.5) * delta, z[:, -1:]], 1).astype(np.float32)
    pts = (ro[:, None, :] + rd[:, None, :] * zmid[:, :, None]).astype(np.float32)
    can, _, _, _, mask = O.warp_samples(pts.reshape(-1, 3), verts, faces, Ts, 0.05)
    return (verts, faces, Ts), (near, far), can.astype(np.float32).reshape(N, T, 3), mask.reshape(N, T)


def test_oracle_posed_backward_matches_reference_autograd(oracle):
    """run(render_can=False, verts, faces, Ts) under autograd (tests/golden/warp_grad.npz: the reference's own backward through its posed-space render,
    training mode, 32 + 32 samples, mesh guide): the fp64 oracle at the reference's sample positions, with the warp's outputs as constants."""
    O = oracle
    p, g = load_golden("nsr_params.npz"), load_golden("warp_grad.npz")
    f = oracle_field_from_golden(p)
    _, nf, can, mask = _posed_inputs(O, g)
    assert abs(float((g["alpha"] > 0).mean()) - float(mask.mean())) <= 0.02
    r = O.render_core_backward(f, g["rays_o"], g["rays_d"], g["z_vals"], 32, 32, 1.6, float(p["inv_s"]), bg=g["bg"], g_image=g["G"], g_weights_sum=g["Gw"],
                               g_normal_map=g["Gn"], g_eik=0.01, ext_pts=can, mask=mask, near_far=nf)
    assert np.abs(r["image"] - g["rgb"]).max() <= 1e-4 and np.abs(r["weights_sum"] - g["weight_sum"]).max() <= 1e-4
    assert abs(r["gradient_error"] - float(g["gradient_error"])) <= 1e-4 * float(g["gradient_error"])
    raw = _chain_to_raw(O, p, r)
    for k, mine in raw.items():
        ref = g["grad." + k].astype(np.float64)
        e = float(np.abs(np.asarray(mine).reshape(ref.shape) - ref).max() / np.abs(ref).max())
        # observed <= 8e-5 everywhere except ONE row of the first colour layer (hidden unit 20, 3.6e-3, the same 0.7 % in every column): a sample whose
        # pre-activation is ~0 sits on the other side of the ReLU in the reference's fp32 forward.  Such a sample's input is orthogonal to the row
        # (no bias: w . in = 0), so weight_g's gradient -- the component along the row -- agrees to 1e-5 and only weight_v's shows it.
        assert e <= (5e-3 if k == "color_net.0.weight_v" else 3e-4), (k, e)
    assert np.abs(r["g_table"][g["emb_idx"]] - g["emb_grad"]).max() <= 3e-4 * np.abs(g["emb_grad"]).max()
    assert abs(np.sqrt((r["g_table"] ** 2).sum()) - float(g["emb_l2"])) <= 1e-4 * float(g["emb_l2"])


@pytest.mark.gpu
def test_hip_backward_matches_oracle_backward_on_the_4096_ray_patch(oracle):
    """BASELINE configuration 3's patch: 64 x 64 stride-4 rays of a 256 x 256 training camera, 64 + 64 jittered samples.  Forward = the fused
    renderer with its per-sample outputs kept, backward = ac_render_core_backward (compositing -> colour MLP -> normals + eikonal -> fused SDF
    query -> binned table scatter), upstream gradients on every differentiable output; against the oracle's independent fp64 reverse pass."""
    import torch
    import bench
    from avatarcraft_amd import nsr_ops
    from tests.gpu_common import device_field, oracle_field
    O = oracle
    dev = "cuda:0"
    p = load_golden("nsr_params.npz")
    f, table = device_field(p, device=dev)
    of = oracle_field(p, table)
    ro, rd = bench.sds_view(0)
    N = ro.shape[0]
    rs = np.random.RandomState(3)
    noise = rs.uniform(0, 1, (N, 64)).astype(np.float32)
    bg = rs.uniform(0, 1, (N, 3)).astype(np.float32)
    g_img = np.clip(rs.normal(0, 1, (N, 3)), -1, 1).astype(np.float32)
    g_ws, g_dp, g_nm = rs.normal(0, 1, N).astype(np.float32), rs.normal(0, 1, N).astype(np.float32), rs.normal(0, 1, (N, 3)).astype(np.float32)
    g_eik = 7.0
    t = lambda a: torch.from_numpy(a).to(dev)
    inv_s = float(p["inv_s"])
    tro, trd, tbg = t(ro), t(rd), t(bg)
    res = {}
    for precision in ("exact", "fast"):
        out = nsr_ops.render_rays(f, tro, trd, 64, 64, 1.6, inv_s, bg=tbg, noise=t(noise), extras=True, train_extras=True, precision=precision)
        g_table = torch.zeros_like(f.t["table"])
        g_sdf_p, g_col_p, g_invs = nsr_ops.render_core_backward(f, out.opts, out, tro, trd, tbg, t(g_img), t(g_ws), t(g_dp), t(g_nm),
                                                                torch.tensor(g_eik, device=dev), g_table)
        torch.cuda.synchronize()
        gW1b = g_sdf_p[:64 * 36].view(64, 36)
        res[precision] = dict(W1=gW1b[:, :35], b1=gW1b[:, 35], W2=g_sdf_p[64 * 36:64 * 36 + 1024].view(16, 64), b2=g_sdf_p[64 * 36 + 1024:],
                              Wc1=g_col_p[:2048].view(64, 32)[:, :21], Wc2=g_col_p[2048:6144].view(64, 64), Wc3=g_col_p[6144:].view(16, 64)[:3],
                              inv_s=g_invs.sum(), table=g_table, z_vals=out["z_vals"].cpu().numpy(), image=out["image"].cpu().numpy())
    z = res["exact"]["z_vals"]
    assert np.array_equal(z, res["fast"]["z_vals"])
    r = O.render_core_backward(of, ro, rd, z, 64, 64, 1.6, inv_s, bg=bg, g_image=g_img, g_weights_sum=g_ws, g_depth=g_dp, g_normal_map=g_nm, g_eik=g_eik)
    assert np.abs(r["image"] - res["exact"]["image"]).max() <= 2e-5
    touched = np.flatnonzero(np.abs(r["g_table"]).sum(1))
    pick = touched[np.random.RandomState(4).choice(len(touched), 65536, replace=False)]
    worst = {}
    for precision in ("exact", "fast"):
        g = res[precision]
        for k in ("W1", "b1", "W2", "b2", "Wc1", "Wc2", "Wc3"):
            ref = r["g_" + k]
            e = float(np.abs(g[k].cpu().numpy().astype(np.float64) - ref).max() / np.abs(ref).max())
            worst[f"{precision}.{k}"] = e
        worst[f"{precision}.inv_s"] = abs(float(g["inv_s"]) - r["g_inv_s"]) / abs(r["g_inv_s"])
        gt = g["table"][torch.from_numpy(pick).to(dev)].cpu().numpy().astype(np.float64)
        worst[f"{precision}.table"] = float(np.abs(gt - r["g_table"][pick]).max() / np.abs(r["g_table"]).max())
        nz_gpu = int((g["table"].abs().sum(1) > 0).sum())
        worst[f"{precision}.table_nnz_rel"] = abs(nz_gpu - len(touched)) / len(touched)
    import json, os
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(worst, open("gpurun_out/hip_vs_oracle_backward_4096.json", "w"), indent=1)
    # fp32 sums of 524 288 signed terms (MFMA accumulators per wave, per-wave partials joined in double) against fp64: observed <= 1.9e-4 (exact),
    # <= 3.5e-4 (fast: the colour network's weights and activations are split bf16) of each tensor's largest entry
    for k, e in worst.items():
        tol = 3e-4 if k.startswith("exact") else 6e-4
        if k.endswith("table_nnz_rel"):
            tol = 1e-3
        assert e <= tol, (k, e, worst)
