"""-m gpu: the host mirror of the reference's model API (NeRFNetwork.render, render_instantnsr_naive, the SDS step)
against goldens recorded from the reference's Python."""
import numpy as np
import pytest
import torch

from tests.common import load_golden, make_table, make_rays

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def golden_net(train=False):
    from avatarcraft_amd.instant_nsr import NeRFNetwork
    p = load_golden("nsr_params.npz")
    torch.manual_seed(0)
    net = NeRFNetwork()
    sd = {k: torch.from_numpy(np.asarray(p[k])) for k in p if k.startswith(("sdf_net", "color_net", "deviation_net"))}
    sd["encoder.embeddings"] = torch.from_numpy(make_table(int(p["offsets"][-1]), seed=int(p["table_seed"]), offsets=p["offsets"],
                                                           level_amp=p["level_amp"]))
    sd["encoder.offsets"] = torch.from_numpy(p["offsets"])
    net.load_state_dict(sd, strict=True)           # same keys as the reference's checkpoints
    return net.to(DEV).train(train), p


@pytest.mark.parametrize("name", ["eval_64_64", "eval_32_32", "eval_64_0"])
def test_render_matches_reference_render(name):
    net, p = golden_net()
    g = load_golden(f"run_{name}.npz")
    with torch.no_grad():
        out = net.render(torch.from_numpy(g["rays_o"]).to(DEV)[None], torch.from_numpy(g["rays_d"]).to(DEV)[None],
                         num_steps=int(g["num_steps"]), bound=1.6, upsample_steps=int(g["upsample_steps"]), staged=False,
                         bg_color=torch.from_numpy(g["bg"]).to(DEV), cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, render_can=True, perturb=False)
    N, T = g["rays_o"].shape[0], int(g["num_steps"]) + int(g["upsample_steps"])
    assert set(out) == {"depth", "weights", "weight_sum", "rgb", "normal", "gradient_error", "curvature_error", "pts_color", "pts_alpha", "z_vals"}
    assert out["rgb"].shape == (1, N, 3) and out["depth"].shape == (1, N) and out["weight_sum"].shape == (N, 1)
    assert out["weights"].shape == (N, T) and out["pts_color"].shape == (N, T, 3) and out["normal"].shape == (N, 3)
    c = lambda t: t.detach().cpu().numpy()
    assert np.abs(c(out["rgb"])[0] - g["image"]).max() <= 1e-3
    assert np.abs(c(out["weight_sum"])[:, 0] - g["weights_sum"]).max() <= 1e-3
    assert np.abs(c(out["depth"])[0] - g["depth"]).max() <= 1e-3
    assert abs(float(out["gradient_error"]) - float(g["gradient_error"])) <= 1e-4


def test_render_instantnsr_naive_batches_and_shapes():
    from avatarcraft_amd.render_utils import render_instantnsr_naive, WHITE_BKG, BLACK_BKG
    net, p = golden_net()
    ro, rd = make_rays(16, 16, dist=1.7, f=12.5)
    ro_t, rd_t = torch.from_numpy(ro).to(DEV), torch.from_numpy(rd).to(DEV)
    rgb, eik, extra = render_instantnsr_naive(net, ro_t, rd_t, rays_per_batch=100, bkg_key=WHITE_BKG, render_can=True, perturb=False, return_raw=True)
    assert rgb.shape == (256, 3) and extra["depth"].shape == (256, 1) and extra["weight_sum"].shape == (256, 1) and extra["normal"].shape == (256, 3)
    # ragged batching (100,100,56) == one batch, bit for bit (rays are independent)
    rgb1, eik1, _ = render_instantnsr_naive(net, ro_t, rd_t, rays_per_batch=6400, bkg_key=WHITE_BKG, render_can=True, perturb=False, return_raw=True)
    assert torch.equal(rgb, rgb1)
    rgbk, _ = render_instantnsr_naive(net, ro_t, rd_t, rays_per_batch=256, bkg_key=BLACK_BKG, render_can=True, perturb=False)
    ws = extra["weight_sum"]
    assert torch.allclose(rgb - rgbk, (1 - ws).expand(-1, 3), atol=1e-6)          # image = colour + (1 - w) * bg
    with pytest.raises(RuntimeError, match="needs verts"):
        render_instantnsr_naive(net, ro_t, rd_t, render_can=False)       # the reference's default: posed space, needs the frame's mesh


def _oracle_raw_grads(O, p, table, ro, rd, z_vals, bg, g_image=None, g_ws=None, g_eik=0.0):
    """the oracle's fp64 backward (oracle/ac_oracle_bwd.c, pinned to the reference's autograd by tests/test_oracle_backward.py) chained to the
    reference's parameters: {name: float64 gradient}, + "encoder.embeddings" """
    from tests.gpu_common import oracle_field
    from tests.test_oracle_backward import _chain_to_raw
    r = O.render_core_backward(oracle_field(p, table), ro, rd, z_vals, 64, 64, 1.6, float(p["inv_s"]), bg=bg, g_image=g_image, g_weights_sum=g_ws, g_eik=g_eik)
    raw = _chain_to_raw(O, p, r)
    raw["encoder.embeddings"] = r["g_table"]
    return raw, r


def test_training_gradients_match_reference_autograd(oracle):
    """stylize.py:163-169: rgb.backward(image_grad) then (0.01*eikonal).backward() on 256 rays.  Two comparisons:
    (a) against the oracle's fp64 backward evaluated at the sample positions THIS forward produced (the oracle itself is pinned to the reference's
        autograd at the reference's sample positions, <= 4e-5 of max: tests/test_oracle_backward.py) -- the tight one;
    (b) against the reference's own .grad (tests/golden/train_grad.npz) directly.  The two forwards agree within the north-star tolerance, not bit
        for bit: on 11 of the 256 rays an up-sampled position differs by up to 1e-3 between the reference's torch-CPU arithmetic and this one
        (a near-tie in the inverse-CDF lerp), so the gradient of THAT ray is taken at a different point -- the bound of (b) reflects it."""
    net, p = golden_net(train=True)
    g = load_golden("train_grad.npz")
    ro, rd = torch.from_numpy(g["rays_o"]).to(DEV), torch.from_numpy(g["rays_d"]).to(DEV)
    # feed the recorded jitter so that the sample positions are the reference's (torch.rand streams differ across devices)
    orig_rand = torch.rand
    torch.rand = lambda *a, **k: torch.from_numpy(g["noise"]).to(DEV)
    try:
        out = net.render(ro[None], rd[None], num_steps=64, bound=1.6, upsample_steps=64, staged=False, bg_color=torch.from_numpy(g["bg"]).to(DEV),
                         cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, render_can=True, perturb=True)
    finally:
        torch.rand = orig_rand
    assert np.abs(out["rgb"][0].detach().cpu().numpy() - g["rgb"]).max() <= 1e-3
    out["rgb"][0].backward(gradient=torch.from_numpy(g["img_grad"]).to(DEV), retain_graph=True)
    (out["gradient_error"] * 0.01).backward()
    table = make_table(int(p["offsets"][-1]), seed=int(p["table_seed"]), offsets=p["offsets"], level_amp=p["level_amp"])
    raw, _ = _oracle_raw_grads(oracle, p, table, g["rays_o"], g["rays_d"], out["z_vals"].detach().cpu().numpy(), g["bg"], g_image=g["img_grad"], g_eik=0.01)
    worst = {}
    for k, prm in net.named_parameters():
        got = prm.grad.detach().cpu().numpy().astype(np.float64)
        orc = np.asarray(raw[k]).reshape(got.shape)
        e_orc = float(np.abs(got - orc).max() / np.abs(orc).max())
        if k == "encoder.embeddings":
            ref, gsub = g["emb_grad"], got[g["emb_idx"]]
        else:
            ref, gsub = g["grad." + k], got
        e_ref = float(np.abs(gsub - ref).max() / (np.abs(ref).max() + 1e-12))
        worst[k] = (e_orc, e_ref)
    import json, os
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(worst, open("gpurun_out/train_grad_parity.json", "w"), indent=1)
    for k, (e_orc, e_ref) in worst.items():
        assert e_orc <= 3e-4, (k, e_orc, worst)            # (a): fp32 kernels vs fp64 witness, through the weight-norm projection (observed <= 9.5e-5)
        assert e_ref <= 1.5e-2, (k, e_ref, worst)          # (b): see the docstring
    ge = net.encoder.embeddings.grad
    l2 = float(torch.sqrt((ge.double() ** 2).sum()))
    assert abs(l2 - float(g["emb_l2"])) <= 2e-2 * float(g["emb_l2"])
    nnz = int((ge.abs().sum(1) > 0).sum())
    assert abs(nnz - int(g["emb_nnz"])) <= 0.01 * int(g["emb_nnz"])


def test_posed_render_under_grad_matches_oracle_and_reference(oracle):
    """run(render_can=False, verts, faces, Ts) under autograd (instant_nsr.py:166-217 are differentiable w.r.t. the network; the SMPL inverse warp is
    numpy in the reference: warped points, mask and mesh-guided range are constants).  Forward = the posed launch sequence with per-sample outputs kept,
    backward = ac_render_core_backward on the warped points with the alpha mask.  (a) against the oracle's fp64 backward at THIS forward's sample
    positions (the oracle is pinned to the reference's autograd of the posed render by tests/test_oracle_backward.py); (b) against the reference's .grad
    (tests/golden/warp_grad.npz) directly, loose for the reason given at test_training_gradients_match_reference_autograd."""
    from tests.common import make_body
    from tests.gpu_common import oracle_field
    from tests.test_oracle_backward import _chain_to_raw, _posed_inputs
    net, p = golden_net(train=True)
    g = load_golden("warp_grad.npz")
    verts, faces, Ts = make_body()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    ro, rd = t(g["rays_o"]), t(g["rays_d"])
    orig_rand = torch.rand
    torch.rand = lambda *a, **k: t(g["noise"])
    try:
        out = net.render(ro[None], rd[None], num_steps=32, bound=1.6, upsample_steps=32, staged=False, bg_color=t(g["bg"]), cos_anneal_ratio=1.0,
                         normal_epsilon_ratio=0.0, render_can=False, verts=verts, faces=faces, Ts=Ts, perturb=True, use_mesh_guide=True)
    finally:
        torch.rand = orig_rand
    c = lambda x: x.detach().cpu().numpy()
    from tests.test_oracle_golden import MEDIAL_RAY              # (the ray on the body's medial axis: the warp is discontinuous in the last ulp of z there)
    keep = np.ones(ro.shape[0], bool); keep[MEDIAL_RAY] = False
    assert np.abs(c(out["rgb"])[0] - g["rgb"])[keep].max() <= 1e-3 and np.abs(c(out["weight_sum"])[:, 0] - g["weight_sum"])[keep].max() <= 1e-3
    assert abs(float(out["gradient_error"]) - float(g["gradient_error"])) <= 1e-3 * float(g["gradient_error"])
    loss = (out["rgb"][0] * t(g["G"])).sum() + 0.01 * out["gradient_error"] + (out["weight_sum"][:, 0] * t(g["Gw"])).sum() + (out["normal"] * t(g["Gn"])).sum()
    loss.backward()
    # (a) the oracle at this forward's z values (its warp is the same fp64 routine: tests/test_gpu_render.py::test_warped_render_bitwise_vs_oracle)
    table = make_table(int(p["offsets"][-1]), seed=int(p["table_seed"]), offsets=p["offsets"], level_amp=p["level_amp"])
    gz = dict(g); gz["z_vals"] = c(out["z_vals"])
    _, nf, can, mask = _posed_inputs(oracle, gz)
    assert np.array_equal(mask, c(out["pts_alpha"]) > 0) or abs(float(mask.mean()) - float((c(out["pts_alpha"]) > 0).mean())) <= 5e-3
    r = oracle.render_core_backward(oracle_field(p, table), g["rays_o"], g["rays_d"], gz["z_vals"], 32, 32, 1.6, float(p["inv_s"]), bg=g["bg"], g_image=g["G"],
                                    g_weights_sum=g["Gw"], g_normal_map=g["Gn"], g_eik=0.01, ext_pts=can, mask=mask, near_far=nf)
    raw = _chain_to_raw(oracle, p, r)
    raw["encoder.embeddings"] = r["g_table"]
    worst = {}
    for k, prm in net.named_parameters():
        got = prm.grad.detach().cpu().numpy().astype(np.float64)
        orc = np.asarray(raw[k]).reshape(got.shape)
        e_orc = float(np.abs(got - orc).max() / np.abs(orc).max())
        ref, gsub = (g["emb_grad"], got[g["emb_idx"]]) if k == "encoder.embeddings" else (g["grad." + k], got)
        worst[k] = (e_orc, float(np.abs(gsub - ref).max() / (np.abs(ref).max() + 1e-12)))
    import json, os
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(worst, open("gpurun_out/warp_grad_parity.json", "w"), indent=1)
    for k, (e_orc, e_ref) in worst.items():
        assert e_orc <= (5e-3 if k == "color_net.0.weight_v" else 3e-4), (k, e_orc, worst)      # (one ReLU-kink row: see test_oracle_posed_backward_...)
        assert e_ref <= 1.5e-2, (k, e_ref, worst)
    # posed space under autograd refuses configurations it would silently get wrong
    net.fused_training = "ops"
    with pytest.raises(NotImplementedError):
        net.render(ro[None], rd[None], num_steps=32, bound=1.6, upsample_steps=32, staged=False, render_can=False, verts=verts, faces=faces, Ts=Ts)


def test_sds_step_matches_reference_step(oracle):
    """One whole optimisation step of stylize.py:143-199 through stylize.sds_step (256 rays; train_grad.npz: grad3.* = .grad after
    rgb.backward(image_grad), (0.01 eikonal).backward() and (1e5 smooth_l1(clamp(opacity), clamp(opacity_gt))).backward() with a frozen net_gt that
    differs from net_style; adam_delta.* = parameter change of Adam(lr 5e-3)'s first step).
    (a) the accumulated gradients against the oracle's fp64 backward fed with THIS step's own forward (sample positions, opacities) -- tight;
    (b) against the reference's .grad directly -- the opacity term is 1e5 x a DIFFERENCE of two opacities whose median is 1.5e-3, so the <= 2.8e-4
        by which the two forwards differ (inside the 1e-3 north-star tolerance) moves that term's gradient by percents;
    (c) Adam's first step against the reference's."""
    from avatarcraft_amd.stylize import sds_step, flat_grad_view
    g = load_golden("train_grad.npz")
    net, p = golden_net(train=True)
    net_gt, _ = golden_net(train=False)
    with torch.no_grad():
        net_gt.sdf_net[1].bias[0] = float(g["gt_sdf_bias"])
    ro, rd = torch.from_numpy(g["rays_o"]).to(DEV), torch.from_numpy(g["rays_d"]).to(DEV)
    n = ro.shape[0]
    img_grad = torch.from_numpy(g["img_grad"]).to(DEV)
    guidance = lambda img: img_grad.reshape(1, n, 1, 3).permute(0, 3, 1, 2).contiguous()        # d loss / d image, [1,3,h,w] with (h, w) = (n, 1)
    opt = torch.optim.Adam([{"params": net.parameters(), "lr": 5e-3}])
    flat = flat_grad_view(net.parameters())
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    orig_rand = torch.rand
    torch.rand = lambda *a, **k: torch.from_numpy(g["noise"]).to(DEV)         # the reference's jitter (torch.rand streams differ across devices)
    try:
        with torch.no_grad():          # the step's own forward, once more (bit-identical launches): sample positions and the two opacities
            bgw = torch.ones((n, 3), device=DEV)
            fw = net.render(ro[None], rd[None], num_steps=64, bound=1.6, upsample_steps=64, staged=False, bg_color=bgw, cos_anneal_ratio=1.0,
                            normal_epsilon_ratio=0.0, render_can=True, perturb=True)
            fw_gt = net_gt.render(ro[None], rd[None], num_steps=64, bound=1.6, upsample_steps=64, staged=False, bg_color=bgw, cos_anneal_ratio=1.0,
                                  normal_epsilon_ratio=0.0, render_can=True, perturb=True)
        stats = sds_step(net, net_gt, ro, rd, (n, 1), opt, guidance, batch_size=4096, w_eikonal=0.01, use_opacity=True, flat_grad=flat)
    finally:
        torch.rand = orig_rand
    assert abs(float(stats["opacity"]) - float(g["opacity_loss"])) <= 2e-2 * float(g["opacity_loss"])
    pred, gt = fw["weight_sum"].reshape(-1).cpu().numpy().astype(np.float64), fw_gt["weight_sum"].reshape(-1).cpu().numpy().astype(np.float64)
    assert np.abs(pred - g["opacity_pred"]).max() <= 1e-3 and np.abs(gt - g["opacity_gt"]).max() <= 1e-3
    d = np.clip(pred, 0, 1) - np.clip(gt, 0, 1)
    g_ws = np.where(np.abs(d) < 1.0, d, np.sign(d)) * (1e5 / n) * ((pred >= 0) & (pred <= 1))
    table = make_table(int(p["offsets"][-1]), seed=int(p["table_seed"]), offsets=p["offsets"], level_amp=p["level_amp"])
    raw, _ = _oracle_raw_grads(oracle, p, table, g["rays_o"], g["rays_d"], fw["z_vals"].cpu().numpy(), np.ones((n, 3), np.float32), g_image=g["img_grad"], g_ws=g_ws, g_eik=0.01)
    worst = {}
    for k, prm in net.named_parameters():
        got = prm.grad.detach().cpu().numpy().astype(np.float64)
        orc = np.asarray(raw[k]).reshape(got.shape)
        e_orc = float(np.abs(got - orc).max() / np.abs(orc).max())
        if k == "encoder.embeddings":
            ref, gsub = g["emb_grad3"], got[g["emb_idx"]]
        else:
            ref, gsub = g["grad3." + k], got
        scale = np.abs(ref).max() + 1e-12
        worst[k] = (e_orc, float(np.abs(gsub - ref).max() / scale))
        # Adam's first step is -lr * g / (|g| + 1e-8): every entry with a clear gradient moves by exactly lr against its sign
        dlt = (prm.detach() - before[k]).cpu().numpy()
        dref = g["adam_delta.emb"] if k == "encoder.embeddings" else g["adam_delta." + k]
        if k == "encoder.embeddings":
            dlt = dlt[g["emb_idx"]]
        clear = np.abs(ref) > 0.2 * scale
        assert clear.any() and np.abs(dlt[clear] - dref[clear]).max() <= 1e-6, k      # (c)
        assert np.abs(dlt - dref).max() <= 2 * 5e-3 + 1e-6
    import json, os
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(worst, open("gpurun_out/sds_step_parity.json", "w"), indent=1)
    for k, (e_orc, e_ref) in worst.items():
        assert e_orc <= 3e-4, (k, e_orc, worst)            # (a) observed <= 1.2e-4
        # (b) per key: the table gradient inherits the reference's fp32 atomicAdd order and the opacity term's 1e5 scale (observed 3.5e-2; the trainer-step
        # test splits it by loss term); every MLP / variance gradient is a plain sum and sits at <= 3.4e-3 -- a 20x regression there must not pass
        assert e_ref <= (6e-2 if k == "encoder.embeddings" else 5e-3), (k, e_ref, worst)
    changed = int(((net.encoder.embeddings.detach() - before["encoder.embeddings"]).abs().sum(1) > 0).sum())
    assert abs(changed - int(g["adam_changed"])) <= 0.01 * int(g["adam_changed"])


def test_sds_step_updates_parameters_and_is_finite():
    from avatarcraft_amd.stylize import sds_step, SyntheticGuidance, flat_grad_view
    net, p = golden_net(train=True)
    net_gt, _ = golden_net(train=False)          # frozen copy of the initial avatar (stylize.py:328-338 loads the ckpt twice)
    ro, rd = make_rays(32, 32, dist=1.8, f=25.0)
    ro_t, rd_t = torch.from_numpy(ro).to(DEV), torch.from_numpy(rd).to(DEV)
    opt = torch.optim.Adam(net.parameters(), lr=5e-3)
    flat = flat_grad_view(net.parameters())
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    stats = sds_step(net, net_gt, ro_t, rd_t, (32, 32), opt, SyntheticGuidance(42), batch_size=512, flat_grad=flat)
    assert torch.isfinite(flat).all() and float(flat.abs().sum()) > 0
    assert torch.isfinite(stats["eikonal"]) and torch.isfinite(stats["opacity"])
    changed = [k for k, v in net.named_parameters() if not torch.equal(v.detach(), before[k])]
    assert "encoder.embeddings" in changed and "sdf_net.0.weight_v" in changed and "color_net.2.weight_v" in changed
    assert flat.numel() == 12248902


def test_posed_render_matches_reference_render():
    """NeRFNetwork.render(render_can=False, verts, faces, Ts) == the reference's (render_warp.py:96-106 call shape)"""
    from tests.common import make_body
    from tests.test_oracle_golden import check_warp_render_vs_golden
    from avatarcraft_amd.render_utils import render_instantnsr_naive
    net, p = golden_net()
    net.eval()
    g = load_golden("warp_render.npz")
    verts, faces, Ts = make_body()
    ro, rd = torch.from_numpy(g["rays_o"]).to(DEV), torch.from_numpy(g["rays_d"]).to(DEV)
    with torch.no_grad():
        out = net.render(ro[None], rd[None], num_steps=32, bound=1.6, upsample_steps=32, staged=False, bg_color=torch.from_numpy(g["bg"]).to(DEV),
                         cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, render_can=False, verts=verts, faces=faces, Ts=Ts, perturb=False)
    m = {"image": out["rgb"][0], "weights_sum": out["weight_sum"][:, 0], "depth": out["depth"][0], "normal_map": out["normal"], "weights": out["weights"],
         "alpha": out["pts_alpha"], "z_vals": out["z_vals"], "gradient_error": out["gradient_error"]}
    check_warp_render_vs_golden(lambda k: m[k].detach().cpu().numpy(), g, "guide", flips="model")      # (model path: see tests/golden/make_z_flips.py)
    # through the harness the animate driver uses
    rgb, _ = render_instantnsr_naive(net, ro, rd, rays_per_batch=100, requires_grad=False, render_can=False, perturb=False, verts=verts, faces=faces,
                                     Ts=Ts, num_steps=32, upsample_steps=32, bound=1.6)
    assert rgb.shape == (256, 3) and np.abs(rgb.cpu().numpy()[1:] - g["guide_image"][1:]).max() <= 1e-3
    # (grad mode: test_posed_render_under_grad_matches_oracle_and_reference; the default normal_epsilon_ratio = 1 gives fd_eps = 0, refused loudly)
    with pytest.raises(RuntimeError):
        net.render(ro[None], rd[None], num_steps=32, bound=1.6, upsample_steps=32, render_can=False, verts=verts, faces=faces, Ts=Ts)


def test_fused_sdf_query_matches_autograd_formulation():
    """sdf_stencil (csrc/sdf_train.hip) against the torch formulation (stencil encoder + nn.Linear/Softplus autograd): values and the
    gradients w.r.t. the hash table and every sdf_net parameter, for a loss that uses all 16 outputs and the finite-difference normals"""
    net, _ = golden_net(train=True)
    rs = np.random.RandomState(4)
    pts = rs.uniform(-1.3, 1.3, size=(5000, 3)).astype(np.float32)
    pts[:50] = np.sign(pts[:50]) * 1.6                                   # on the bound: clamped offsets
    x = torch.from_numpy(pts).to(DEV)
    w16 = torch.from_numpy(rs.normal(size=(1, 16)).astype(np.float32)).to(DEV)
    w3 = torch.from_numpy(rs.normal(size=(5000, 3)).astype(np.float32)).to(DEV)

    def run(fused):
        net.fused_training = fused
        net.zero_grad()
        s, g = net.forward_sdf_stencil(x, 1.6, 0.005)
        loss = (s * w16).sum() + (g * w3).sum() * 0.01 + ((g.norm(dim=-1) - 1) ** 2).mean()
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
        return s.detach(), g.detach(), grads
    s1, g1, G1 = run(True)
    s0, g0, G0 = run(False)
    assert torch.allclose(s1, s0, atol=3e-6, rtol=1e-5) and torch.allclose(g1, g0, atol=3e-3, rtol=1e-3)
    assert set(G1) == set(G0) and "encoder.embeddings" in G1 and "sdf_net.0.weight_v" in G1 and "sdf_net.1.bias" in G1
    for k in G0:
        scale = G0[k].abs().max().item()
        assert scale > 0, k
        err = (G1[k] - G0[k]).abs().max().item()
        assert err <= 2e-3 * scale, (k, err, scale)
    # bit-identical to the stand-alone field query (== oracle) for the value part
    from avatarcraft_amd import nsr_ops
    f = net._field()
    assert torch.equal(s1, nsr_ops.field_sdf(f, x, 1.6))


@pytest.mark.parametrize("B", [0, 1, 17, 1000])
def test_fused_sdf_query_ragged_sizes(B):
    """tile tails (B not a multiple of 16), a single sample and the empty batch"""
    from avatarcraft_amd import nsr_ops
    net, _ = golden_net(train=True)
    rs = np.random.RandomState(B)
    x = torch.from_numpy(rs.uniform(-1.5, 1.5, size=(B, 3)).astype(np.float32)).to(DEV)
    net.fused_training = True
    net.zero_grad()
    s, g = net.forward_sdf_stencil(x, 1.6, 0.005)
    assert s.shape == (B, 16) and g.shape == (B, 3)
    (s.sum() + g.sum()).backward()
    gt = net.encoder.embeddings.grad
    assert gt is not None and torch.isfinite(gt).all()
    if B == 0:
        assert float(gt.abs().max()) == 0.0 and float(net.sdf_net[0].weight_v.grad.abs().max()) == 0.0
        return
    f = net._field()
    assert torch.equal(s.detach(), nsr_ops.field_sdf(f, x, 1.6))
    net.fused_training = False
    g_fused = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    net.zero_grad()
    s0, g0 = net.forward_sdf_stencil(x, 1.6, 0.005)
    (s0.sum() + g0.sum()).backward()
    for k, p in net.named_parameters():
        if p.grad is not None:
            assert (g_fused[k] - p.grad).abs().max() <= 2e-3 * max(float(p.grad.abs().max()), 1e-12), k
    net.fused_training = True


@pytest.mark.parametrize("B", [0, 5, 4099])
def test_fused_color_mlp_matches_autograd(B):
    """color_mlp (csrc/sdf_train.hip) against forward_color with torch autograd: rgb, d normal, d feat and the three weight gradients"""
    from avatarcraft_amd import nsr_ops
    net, _ = golden_net(train=True)
    rs = np.random.RandomState(B + 1)
    t = lambda a: torch.from_numpy(a.astype(np.float32)).to(DEV)
    x = t(rs.uniform(-1.5, 1.5, size=(B, 3)))
    nrm = t(rs.normal(size=(B, 3))).requires_grad_(True)
    so = t(rs.normal(size=(B, 16)) * 0.3).requires_grad_(True)
    up = t(rs.normal(size=(B, 3)))

    def run(fused):
        net.zero_grad(); nrm.grad = None; so.grad = None
        rgb = net.forward_color_fused(x, nrm, so) if fused else net.forward_color(x, None, nrm, so[:, 1:], 1.6)
        (rgb * up).sum().backward()
        return rgb.detach(), nrm.grad.clone(), so.grad.clone(), {k: p.grad.clone() for k, p in net.color_net.named_parameters()}
    r1, n1, s1, G1 = run(True)
    r0, n0, s0, G0 = run(False)
    assert r1.shape == (B, 3)
    if B == 0:
        assert all(float(v.abs().max()) == 0.0 for v in G1.values())
        return
    assert torch.allclose(r1, r0, atol=2e-6) and torch.allclose(n1, n0, atol=1e-5, rtol=1e-4) and torch.allclose(s1, s0, atol=1e-5, rtol=1e-4)
    assert float(s1[:, 0].abs().max()) == 0.0
    for k in G0:
        assert (G1[k] - G0[k]).abs().max() <= 1e-3 * G0[k].abs().max(), k
    f = net._field()
    assert torch.equal(r1, nsr_ops.field_color(f, x, nrm.detach(), so.detach()))            # == the renderer's / the oracle's colour


def test_fused_composite_matches_autograd_formulation():
    """composite (csrc/sdf_train.hip) against the torch formulation of alpha + compositing: all outputs, and the gradients of a loss
    that uses image, weights_sum, depth and normal_map w.r.t. every parameter (incl. the variance); then the whole training render
    with everything fused against everything in torch"""
    net, _ = golden_net(train=True)
    ro, rd = make_rays(24, 24, dist=1.7, f=16.0, jitter_seed=2)
    ro_t, rd_t = torch.from_numpy(ro).to(DEV), torch.from_numpy(rd).to(DEV)
    N = ro.shape[0]
    rs = np.random.RandomState(0)
    wi = torch.from_numpy(rs.normal(size=(1, N, 3)).astype(np.float32)).to(DEV)
    wn = torch.from_numpy(rs.normal(size=(N, 3)).astype(np.float32)).to(DEV)
    bg = torch.from_numpy(rs.uniform(size=(N, 3)).astype(np.float32)).to(DEV)

    def run(fused):
        net.fused_training = fused
        net.zero_grad()
        torch.manual_seed(5)
        out = net.render(ro_t[None], rd_t[None], num_steps=32, bound=1.6, upsample_steps=32, staged=False, bg_color=bg, cos_anneal_ratio=0.7,
                         normal_epsilon_ratio=0.0, render_can=True, perturb=True)
        loss = (out["rgb"] * wi).sum() + 3.0 * out["weight_sum"].clamp(0, 1).sum() + out["depth"].sum() + (out["normal"] * wn).sum() + 0.01 * out["gradient_error"]
        loss.backward()
        return {k: v.detach().clone() for k, v in out.items() if torch.is_tensor(v)}, {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    o1, G1 = run(True)
    o0, G0 = run(False)
    for k in ("rgb", "weight_sum", "depth", "normal", "weights", "pts_alpha", "z_vals"):
        assert torch.allclose(o1[k], o0[k], atol=2e-4, rtol=1e-3), k
    assert "deviation_net.variance" in G1 and set(G1) == set(G0)
    for k in G0:
        scale = float(G0[k].abs().max())
        # the two colour-MLP evaluations differ in the last ulp (MFMA chain vs hipBLASLt), which flips a few ReLU gates: discrete
        # differences of up to ~1 % in single entries of the colour weights; everything upstream of smooth functions agrees to ~1e-4
        tol = 2e-2 if k.startswith("color_net") else 5e-3
        assert float((G1[k] - G0[k]).abs().max()) <= tol * scale + 1e-9, (k, float((G1[k] - G0[k]).abs().max()), scale)
    net.fused_training = True


def test_cap2rays_on_device_equals_host():
    """ray generation on the GPU (the per-view host cost of the stylize loop otherwise) against the numpy path, which the goldens pin"""
    from avatarcraft_amd import render_utils as RU
    poses, _ = RU.style_360_path(np.array([0.0, 0.05, 0.0]), np.array([0.0, 1.0, 0.0]), 1.8, 8)
    for pose in (poses[0], poses[5]):
        cap = RU.pose2cap([96, 128], pose)
        o_h, d_h = RU.cap2rays(cap, device="cpu")
        o_g, d_g = RU.cap2rays(cap, device=DEV)
        assert o_g.shape == (96 * 128, 3) and o_g.is_cuda
        assert torch.equal(o_g.cpu(), o_h) and float((d_g.cpu() - d_h).abs().max()) <= 1.2e-7


def test_inference_drivers():
    """render_canonical.py / render_warp.py main loops as functions: shapes, value ranges, and the posed frame == a direct posed render"""
    from avatarcraft_amd import drivers as DR, smpl as SM
    from avatarcraft_amd.render_utils import render_instantnsr_naive
    from tests.common import make_body
    net, _ = golden_net()
    net.eval()
    views = list(DR.render_canonical_360(net, n_views=2, render_hw=(32, 32), with_head=True))
    assert [(n, i) for n, i, _, _ in views] == [("body", 0), ("body", 1), ("head", 0), ("head", 1)]
    for _, _, rgb, depth in views:
        assert rgb.shape == (32, 32, 3) and depth.shape == (32, 32) and float(rgb.min()) >= 0 and float(rgb.max()) <= 1.0 + 1e-5
    assert float((views[0][2] < 0.99).float().mean()) > 0.02            # the field is visible
    verts, faces, _ = make_body(n_lat=10, n_lon=12)
    bm = SM.BodyModel.synthetic(seed=2, n_verts=verts.shape[0], faces=faces, v_template=verts)
    cam = np.eye(4, dtype=np.float32); cam[:3, 3] = [0.0, 0.0, 2.2]
    rs = np.random.RandomState(1)
    poses = (rs.normal(size=(2, 72)) * 0.2).astype(np.float32)
    frames = list(DR.render_animation(net, bm, cam, poses=poses, resolution=32, max_frames=2))
    assert len(frames) == 2 and frames[0][1].shape == (32, 32, 3) and torch.isfinite(frames[1][1]).all()
    assert float((frames[0][1] - frames[1][1]).abs().max()) > 1e-3        # the pose matters


def test_smpl_on_device_matches_reference_golden():
    """lbs / calc_local_trans with the body model's buffers on the GPU against the reference goldens
    (tests/golden/smpl.npz from models.smpl.lbs, local_trans.npz from render_warp.calc_local_trans), then straight into the posed renderer"""
    from avatarcraft_amd import smpl as SM
    from avatarcraft_amd.render_utils import render_instantnsr_naive
    g = load_golden("smpl.npz")
    bm = SM.BodyModel.synthetic(seed=3, n_verts=600).to(DEV)
    T, v, dv = SM.lbs(torch.from_numpy(g["betas"]).to(DEV), torch.from_numpy(g["pose"]).to(DEV), *bm._args(), return_T=True, concat_joints=True)
    assert T.is_cuda and np.abs(T.cpu().numpy() - g["T"]).max() < 5e-6 and np.abs(v.cpu().numpy() - g["v"]).max() < 2e-6
    lt = load_golden("local_trans.npz")
    bm2 = SM.BodyModel.synthetic(seed=5).to(DEV)
    wv, Ts, n = SM.calc_local_trans(bm2, render_type="animate", poses=lt["poses"])
    assert n == 3 and np.abs(np.stack(Ts)[:, lt["keep"]] - lt["anim_Ts"]).max() < 2e-5
    assert np.abs(np.stack(wv)[:, ::53] - lt["anim_world_verts"]).max() < 2e-5
    # Ts / world_verts of a frame feed the posed renderer unchanged (render_warp.py:88-106)
    net, _ = golden_net()
    net.eval()
    from tests.common import make_rays
    ro, rd = make_rays(8, 8, dist=1.8, f=6.0)
    rgb, _, ex = render_instantnsr_naive(net, torch.from_numpy(ro).to(DEV), torch.from_numpy(rd).to(DEV), 64, requires_grad=False, render_can=False, perturb=False,
                                         return_raw=True, verts=wv[1], faces=np.asarray(bm2.faces), Ts=Ts[1], num_steps=32, upsample_steps=32, bound=1.6)
    assert rgb.shape == (64, 3) and torch.isfinite(rgb).all()


@pytest.mark.parametrize("n_side,T0,up,perturb", [(10, 64, 64, True), (7, 32, 32, False), (3, 64, 0, True), (64, 64, 64, True)])   # the last: BASELINE configuration 3's 4096-ray patch
def test_render_core_operator(n_side, T0, up, perturb):
    """nsr_ops.render_core ("core": forward = the fused renderer itself, backward = ac_render_core_backward) against
    (a) the no-grad render: every forward output bit for bit (the training render IS the inference launch);
    (b) the operator-by-operator training path of round 1 ("ops": sampling launch + fused SDF query / colour / compositing operators with
        torch autograd between them) and (c) plain torch autograd over the stencil hash encoder (False): all parameter gradients, with a loss
        that uses every differentiable output (image, weights_sum, depth, normal_map, gradient_error)."""
    net, p = golden_net(train=True)
    ro, rd = make_rays(n_side, n_side, dist=1.8, f=0.6 * n_side, jitter_seed=n_side)
    ro, rd = torch.from_numpy(ro).to(DEV), torch.from_numpy(rd).to(DEV)
    N = ro.shape[0]
    rs = np.random.RandomState(N)
    t = lambda a: torch.from_numpy(np.asarray(a, np.float32)).to(DEV)
    noise = t(rs.uniform(0, 1, (N, T0)))
    bg = t(rs.uniform(0, 1, (N, 3)))
    w_img, w_ws, w_dp, w_nm = t(rs.normal(size=(N, 3))), t(rs.normal(size=(N, 1))), t(rs.normal(size=(1, N))), t(rs.normal(size=(N, 3)))

    def run(mode):
        net.fused_training = mode
        net.zero_grad()
        orig = torch.rand
        torch.rand = lambda *a, **k: noise
        try:
            out = net.render(ro[None], rd[None], num_steps=T0, bound=1.6, upsample_steps=up, staged=False, bg_color=bg, cos_anneal_ratio=1.0,
                             normal_epsilon_ratio=0.0, render_can=True, perturb=perturb)
        finally:
            torch.rand = orig
        loss = ((out["rgb"][0] * w_img).sum() + (out["weight_sum"] * w_ws).sum() + (out["depth"] * w_dp).sum() + (out["normal"] * w_nm).sum()
                + 3.0 * out["gradient_error"])
        loss.backward()
        return {k: v.detach().clone() for k, v in out.items() if isinstance(v, torch.Tensor)}, {k: q.grad.detach().clone() for k, q in net.named_parameters()}
    o_core, g_core = run("core")
    o_ops, g_ops = run("ops")
    o_ag, g_ag = run(False)
    with torch.no_grad():
        orig = torch.rand
        torch.rand = lambda *a, **k: noise
        try:
            o_ng = net.render(ro[None], rd[None], num_steps=T0, bound=1.6, upsample_steps=up, staged=False, bg_color=bg, cos_anneal_ratio=1.0,
                              normal_epsilon_ratio=0.0, render_can=True, perturb=perturb)
        finally:
            torch.rand = orig
    for k in ("rgb", "weight_sum", "depth", "normal", "weights", "pts_color", "pts_alpha", "z_vals", "gradient_error"):
        assert torch.equal(o_core[k], o_ng[k]), k                       # (a)
        if N < 1024:
            assert torch.allclose(o_core[k], o_ops[k], atol=2e-5, rtol=1e-4), k
        else:       # half a million samples: alpha sits on a steep sigmoid for a few of them -- bound the outliers instead of every element
            dlt = (o_core[k].float() - o_ops[k].float()).abs()
            bad = dlt > (2e-5 + 1e-4 * o_ops[k].float().abs())
            assert float(bad.float().mean()) <= 1e-4 and float(dlt.max()) <= 2e-3 and float(dlt.mean()) <= 2e-6, (k, float(bad.float().mean()), float(dlt.max()), float(dlt.mean()))
    assert set(g_core) == set(g_ops) == set(g_ag) and len(g_core) == 14
    worst = {}
    for k in g_core:
        scale = float(g_ag[k].abs().max())
        assert scale > 0, k
        worst[k] = (float((g_core[k] - g_ops[k]).abs().max()) / scale, float((g_core[k] - g_ag[k]).abs().max()) / scale)
        assert worst[k][0] <= 1e-3 and worst[k][1] <= 3e-3, (k, worst[k])    # (b), (c)
    # gradients accumulate into an existing .grad (stylize.py back-propagates three terms one after the other) ...
    net.fused_training = "core"
    net.zero_grad()
    from avatarcraft_amd.stylize import flat_grad_view
    flat = flat_grad_view(net.parameters())
    orig = torch.rand
    torch.rand = lambda *a, **k: noise
    try:
        out = net.render(ro[None], rd[None], num_steps=T0, bound=1.6, upsample_steps=up, staged=False, bg_color=bg, cos_anneal_ratio=1.0,
                         normal_epsilon_ratio=0.0, render_can=True, perturb=perturb)
    finally:
        torch.rand = orig
    (out["rgb"][0] * w_img).sum().backward(retain_graph=True)
    ((out["weight_sum"] * w_ws).sum() + (out["depth"] * w_dp).sum()).backward(retain_graph=True)
    ((out["normal"] * w_nm).sum() + 3.0 * out["gradient_error"]).backward()
    for k, q in net.named_parameters():                                     # ... and the three partial backward passes add up to the joint one
        scale = float(g_core[k].abs().max())
        assert float((q.grad - g_core[k]).abs().max()) <= 2e-4 * scale + 1e-12, k
        assert q.grad.data_ptr() >= flat.data_ptr() and q.grad.data_ptr() < flat.data_ptr() + flat.numel() * 4        # still views of the flat buffer
    import json, os
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(worst, open(f"gpurun_out/render_core_parity_{N}.json", "w"), indent=1)


def test_render_variants_generic_path():
    """model variants under autograd through net.render(): use_viewdirs=True (SH-encoded directions into the colour net, models/instant_nsr.py:564-569,652-653)
    and curvature_loss=True (:276-288; since round 6 on the fused operators, tests/test_gpu_curvature.py pins its values), with and without gradients; a canonical render inside the mesh-guided range (verts given, :147-153); normal_epsilon_ratio >= 1 is refused"""
    from avatarcraft_amd.instant_nsr import NeRFNetwork
    from tests.common import make_body
    torch.manual_seed(1)
    ro, rd = make_rays(6, 6, dist=1.8, f=4.0)
    ro, rd = torch.from_numpy(ro).to(DEV), torch.from_numpy(rd).to(DEV)
    kw = dict(num_steps=32, bound=1.6, upsample_steps=32, staged=False, bg_color=None, cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, render_can=True)
    for opts in (dict(use_viewdirs=True), dict(curvature_loss=True)):
        net = NeRFNetwork(**opts).to(DEV).train()
        with torch.no_grad():
            net.encoder.embeddings.uniform_(-0.05, 0.05)
            net.sdf_net[0].weight_v[:, 3:].normal_(0, 0.05)
        out = net.render(ro[None], rd[None], perturb=True, **kw)
        assert out["rgb"].shape == (1, 36, 3) and torch.isfinite(out["rgb"]).all()
        (out["rgb"].sum() + out["gradient_error"] + out["curvature_error"]).backward()
        assert net.color_net[0].weight_v.grad is not None and float(net.encoder.embeddings.grad.abs().sum()) > 0
        if opts.get("use_viewdirs"):
            assert net.color_net[0].weight_v.shape == (64, 37)
        else:
            assert float(out["curvature_error"]) > 0
        with torch.no_grad():
            out2 = net.eval().render(ro[None], rd[None], perturb=False, **kw)
        assert torch.isfinite(out2["rgb"]).all() and out2["weights"].shape == (36, 64)
    net, _ = golden_net()
    verts, faces, Ts = make_body(n_lat=12, n_lon=16)
    with torch.no_grad():
        a = net.render(ro[None], rd[None], verts=verts, faces=faces, Ts=Ts, use_mesh_guide=True, **kw)
        b = net.render(ro[None], rd[None], verts=verts, faces=faces, Ts=Ts, use_mesh_guide=False, **kw)
        c = net.render(ro[None], rd[None], **kw)
    assert torch.equal(b["rgb"], c["rgb"]) and not torch.equal(a["z_vals"], c["z_vals"])
    from avatarcraft_amd.ray_utils import geometry_guided_near_far
    nm, fm = geometry_guided_near_far(ro, rd, verts, 0.05)
    hit = torch.isfinite(nm)
    assert hit.any() and (~hit).any()
    assert torch.allclose(a["z_vals"][hit].min(1).values, nm[hit], atol=1e-6) and torch.equal(a["z_vals"][~hit], c["z_vals"][~hit])
    with pytest.raises(RuntimeError, match="fd_eps"):
        with torch.no_grad():
            net.render(ro[None], rd[None], num_steps=32, bound=1.6, upsample_steps=32)      # the reference's default normal_epsilon_ratio = 1: eps = 0


def test_sds_step_through_rccl_world_size_1():
    """BASELINE config 5 readiness without an 8-GPU node: the REAL NeRFNetwork's flat 49 MB gradient goes through an RCCL all-reduce
    (torch.distributed backend "nccl", world size 1 on this box) inside sds_step, and the step equals the step without a process group
    bit for bit (sum over one rank, no division)."""
    import os
    import torch.distributed as dist
    from avatarcraft_amd.stylize import sds_step, SyntheticGuidance, flat_grad_view
    ro, rd = make_rays(16, 16, dist=1.8, f=12.0)
    ro_t, rd_t = torch.from_numpy(ro).to(DEV), torch.from_numpy(rd).to(DEV)

    def one_step(overlap=False, batch=256):
        net, _ = golden_net(train=True)
        net_gt, _ = golden_net(train=False)
        opt = torch.optim.Adam(net.parameters(), lr=5e-3)
        flat = flat_grad_view(net.parameters())
        torch.manual_seed(7)
        marks = []
        sds_step(net, net_gt, ro_t, rd_t, (16, 16), opt, SyntheticGuidance(3), batch_size=batch, flat_grad=flat, timers=marks, overlap_allreduce=overlap)
        torch.cuda.synchronize()
        return {k: v.detach().clone() for k, v in net.named_parameters()}, flat.clone(), [n for n, _ in marks]
    p0, f0, m0 = one_step()
    p0b, f0b, _ = one_step(batch=128)                       # two patches: the table gradient accumulates across them
    assert "grad_allreduce" not in m0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        p1, f1, m1 = one_step()
        # the level-group split: levels 8 - 15 of the table gradient (33.5 MB) are all-reduced from a side stream that waits for exactly the part of
        # the scatter that completes them, the rest after the backward -- three collectives instead of one, the same bits
        p2, f2, m2 = one_step(overlap=True)
        p3, f3, _ = one_step(overlap=True, batch=128)      # (only the LAST patch's backward may release the early slice)
        probe = torch.ones(4, device=DEV); dist.all_reduce(probe); assert float(probe.sum()) == 4.0
        # round 5: on RCCL the data-parallel step averages INSIDE the collective (stylize.reduce_gradients: ReduceOp.AVG = ncclAvg).  One rank cannot show the
        # division, but it shows that this torch + RCCL build accepts the operator on the flat gradient's dtype -- the first multi-GPU lease must not be the first call
        from avatarcraft_amd.stylize import _avg_in_collective
        assert _avg_in_collective(None)
        probe = torch.full((5,), 3.0, device=DEV); dist.all_reduce(probe, op=dist.ReduceOp.AVG); assert float(probe.sum()) == 15.0
    finally:
        dist.destroy_process_group()
    assert "grad_allreduce" in m1 and "grad_allreduce" in m2 and f1.numel() == 12248902
    assert torch.equal(f0, f1) and torch.equal(f0, f2) and torch.equal(f0b, f3)
    for k in p0:
        assert torch.equal(p0[k], p1[k]) and torch.equal(p0[k], p2[k]) and torch.equal(p0b[k], p3[k]), k


def test_sds_step_without_autograd_equals_autograd_step():
    """sds_step's default path for NeRFNetwork writes the upstream gradients down and calls ac_render_core_backward + ac_param_grads itself
    (weight-norm backward, biases, d variance fused, no torch autograd); the autograd formulation of the same step (the operator
    nsr_ops.render_core under torch's weight norm / losses) must give the same gradients -- two patches, multi-patch accumulation included --
    and the same parameters after Adam.  Also: ac_weight_norm_forward == torch._weight_norm."""
    from avatarcraft_amd.stylize import sds_step, SyntheticGuidance, flat_grad_view
    ro, rd = make_rays(32, 16, dist=1.8, f=14.0)
    ro_t, rd_t = torch.from_numpy(ro).to(DEV), torch.from_numpy(rd).to(DEV)

    def one_step(manual):
        net, _ = golden_net(train=True)
        net_gt, _ = golden_net(train=False)
        if not manual:
            net.manual_backward_supported = lambda: False
        opt = torch.optim.Adam(net.parameters(), lr=5e-3)
        flat = flat_grad_view(net.parameters())
        torch.manual_seed(11)
        stats = sds_step(net, net_gt, ro_t, rd_t, (32, 16), opt, SyntheticGuidance(5), batch_size=256, flat_grad=flat)
        torch.cuda.synchronize()
        return net, {k: v.grad.detach().clone() for k, v in net.named_parameters()}, {k: v.detach().clone() for k, v in net.named_parameters()}, stats
    n1, g1, p1, s1 = one_step(True)
    n0, g0, p0, s0 = one_step(False)
    for k in g0:
        scale = float(g0[k].abs().max())
        assert scale > 0 or k.endswith("bias"), k
        err = float((g1[k] - g0[k]).abs().max())
        assert err <= 2e-5 * scale + 1e-12, (k, err, scale)
        assert float((p1[k] - p0[k]).abs().max()) <= 1e-6, k
    assert abs(float(s1["opacity"]) - float(s0["opacity"])) <= 1e-5 * abs(float(s0["opacity"])) + 1e-6
    assert abs(float(s1["eikonal"]) - float(s0["eikonal"])) <= 1e-6 * abs(float(s0["eikonal"])) + 1e-9
    W = n1._effective_weights()
    for l, w in zip(list(n1.sdf_net) + list(n1.color_net), W):
        ref = torch._weight_norm(l.weight_v.detach(), l.weight_g.detach(), 0)
        assert float((w - ref).abs().max()) <= 2e-7 * float(ref.abs().max())


def test_reconstruct_step_matches_reference_step(tmp_path):
    """one step of reconstruct.py:92-112 (smooth_l1(rgb, gt) + 0.1 eikonal, Adam(5e-4, (0.9, 0.99), eps 1e-15)) against the reference's own
    autograd + optimizer (tests/golden/reconstruct_grad.npz), through avatarcraft_amd.reconstruct; then the epoch loop and the dataset reader"""
    from avatarcraft_amd import reconstruct as RC
    g = load_golden("reconstruct_grad.npz")
    net, _ = golden_net(train=True)
    opt, sched = RC.make_optimizer(net, epochs=10)
    assert opt.defaults["lr"] == 5e-4 and opt.defaults["betas"] == (0.9, 0.99) and opt.defaults["eps"] == 1e-15 and sched.eta_min == 0.0
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    ro, rd, gt = (torch.from_numpy(g[k]).to(DEV) for k in ("rays_o", "rays_d", "gt"))
    orig = torch.rand
    torch.rand = lambda *a, **k: torch.from_numpy(g["noise"]).to(DEV)
    try:
        loss = RC.reconstruct_step(net, opt, ro, rd, gt, white_bkg=True)
    finally:
        torch.rand = orig
    assert abs(float(loss) - float(g["loss"])) <= 1e-4 * float(g["loss"])
    for k, prm in net.named_parameters():
        got = prm.grad.detach().cpu().numpy()
        ref = g["emb_grad"] if k == "encoder.embeddings" else g["grad." + k]
        if k == "encoder.embeddings":
            got = got[g["emb_idx"]]
        scale = np.abs(ref).max() + 1e-20
        assert np.abs(got - ref).max() <= 5e-3 * scale, (k, float(np.abs(got - ref).max() / scale))
        d = (prm.detach() - before[k]).cpu().numpy()
        dref = g["adam_delta.emb"] if k == "encoder.embeddings" else g["adam_delta." + k]
        if k == "encoder.embeddings":
            d = d[g["emb_idx"]]
        clear = np.abs(ref) > 1e-3 * scale
        assert clear.any() and np.abs(d[clear] - dref[clear]).max() <= 1e-7, k           # -lr sign(g) where the gradient is clear
    # the dataset reader + two short epochs on a synthetic 3-view set in the reference's layout
    import json
    from PIL import Image
    os = __import__("os")
    root = tmp_path / "set"; (root / "img").mkdir(parents=True)
    rs = np.random.RandomState(0)
    frames = []
    for i in range(3):
        img = (rs.uniform(0, 1, (16, 16, 3)) * 255).astype(np.uint8)
        Image.fromarray(img).save(root / "img" / f"{i:04d}.png")
        c2w = np.eye(4); c2w[:3, 3] = [0.3 * i, 0.0, 2.0]
        frames.append(dict(file_path=f"img/{i:04d}", transform_matrix=c2w.tolist()))
    json.dump(dict(camera_angle_x=float(np.pi / 3), frames=frames), open(root / "transforms_train.json", "w"))
    ds = RC.NeusDataset(str(root), device=DEV)
    assert ds.n_images == 3 and (ds.H, ds.W) == (16, 16) and abs(ds.focal - 8 / np.tan(np.pi / 6)) < 1e-9
    first = np.asarray(Image.open(root / "img" / "0000.png")).astype(np.float32) / 255.0
    assert np.allclose(ds.images[0].numpy(), first[:, ::-1])                              # the reference's flip of the width axis
    aro, ard, argb = ds.all_rays()
    assert aro.shape == (3 * 256, 3) and argb.shape == (3 * 256, 3) and torch.allclose(ard.norm(dim=-1), torch.ones(768, device=DEV), atol=1e-5)
    losses = []
    n = RC.reconstruct_epochs(net, opt, sched, aro, ard, argb, epochs=2, batch_size=256, on_step=lambda s, e, l: losses.append(float(l)))
    assert n == 6 and len(losses) == 6 and np.isfinite(losses).all() and abs(sched.get_last_lr()[0] - 5e-4 * 0.5 * (1 + np.cos(np.pi * 2 / 10))) < 1e-9


def test_density_grid_mesh_export_and_marcher(oracle):
    """update_extra_state (models/instant_nsr.py:303-356) on the fused SDF kernel against the reference's grid and the
    oracle's restatement, two updates (decay / maximum merge); the grid drives raymarching.march_rays_train exactly as the oracle's marcher;
    extract_fields against the reference's SDF volume; extract_geometry returns a closed, outward-oriented surface of the field"""
    from avatarcraft_amd.instant_nsr import NeRFNetwork
    from avatarcraft_amd import raymarching
    from tests.gpu_common import oracle_field as make_of
    g = load_golden("density_grid.npz")
    src, p = golden_net()
    torch.manual_seed(0)
    net = NeRFNetwork(cuda_ray=True)
    net.load_state_dict(src.state_dict(), strict=False)
    net = net.to(DEV).eval()
    assert tuple(net.density_grid.shape) == (129, 129, 129) and tuple(net.step_counter.shape) == (64, 2)
    net.update_extra_state(1.6)
    grid1 = net.density_grid.cpu().numpy()
    assert np.abs(grid1[::4, ::4, ::4] - g["grid1"]).max() <= 5e-3 * float(g["max1"]) and abs(net.mean_density - float(g["mean1"])) <= 1e-3 * float(g["mean1"])
    table = src.encoder.embeddings.detach().cpu().numpy()
    of = make_of(p, table)
    og, om = oracle.update_density_grid(of, np.zeros((129,) * 3, np.float32), 1.6)
    # GPU sdf == oracle sdf bit for bit; the density differs only by torch.exp vs numpy exp (both fp32)
    assert np.abs(grid1 - og).max() <= 2e-4 * float(og.max())
    with torch.no_grad():
        net.sdf_net[1].bias[0] += 0.1
    net.update_extra_state(1.6)
    assert net.iter_density == int(g["iter_density"]) == 2
    assert np.abs(net.density_grid.cpu().numpy()[::4, ::4, ::4] - g["grid2"]).max() <= 5e-3 * float(g["max1"]) and abs(net.mean_density - float(g["mean2"])) <= 1e-3 * float(g["mean2"])
    # the marcher on this grid: GPU == oracle on the same grid values, bit for bit
    ro, rd = make_rays(8, 8, dist=1.8, f=6.0)
    gd = net.density_grid
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(torch.from_numpy(ro).to(DEV), torch.from_numpy(rd).to(DEV), 1.6, gd, net.mean_density, net.iter_density,
                                                           force_all_rays=True)
    x_o, d_o, dl_o, r_o, c_o = oracle.march_rays_train(ro, rd, gd.cpu().numpy(), net.mean_density, 1.6)
    assert np.array_equal(rays.cpu().numpy(), r_o) and int(c_o[0]) > 0
    M = int(c_o[0])
    assert np.array_equal(xyzs[:M].cpu().numpy().view(np.uint32), x_o[:M].view(np.uint32)) and np.array_equal(deltas[:M].cpu().numpy().view(np.uint32), dl_o[:M].view(np.uint32))
    # SDF volume and mesh export
    u = src.extract_fields(1.6, 33)
    assert u.shape == (33, 33, 33) and u.dtype == np.float32 and np.abs(u - g["sdf33"]).max() < 1e-5
    verts, tris = src.extract_geometry(1.6, 48)
    assert verts.shape[1] == 3 and tris.shape[1] == 3 and len(tris) > 500 and np.abs(verts).max() <= 1.6
    with torch.no_grad():
        sd = src.density(torch.from_numpy(verts.astype(np.float32)).to(DEV), 1.6).cpu().numpy()
    assert np.abs(sd).max() < 2e-2                                       # vertices lie on the zero level set (linear interpolation on a 48^3 grid)
    e = np.concatenate([tris[:, [0, 1]], tris[:, [1, 2]], tris[:, [2, 0]]])
    key = e[:, 0].astype(np.int64) * len(verts) + e[:, 1]; rkey = e[:, 1].astype(np.int64) * len(verts) + e[:, 0]
    assert len(np.unique(key)) == len(key) and np.isin(rkey, key).all()          # closed and consistently oriented
    a, b, c = (verts[tris[:, k]] for k in range(3))
    nrm = np.cross(b - a, c - a)
    cen = torch.from_numpy(((a + b + c) / 3).astype(np.float32)).to(DEV)
    with torch.no_grad():
        gsd = src.gradient(cen, 1.6, 0.005).cpu().numpy()
    assert ((nrm * gsd).sum(1) > 0).mean() > 0.97                         # normals point along the SDF gradient: out of the body


def test_nan_guard_raises_after_a_poisoned_training_render():
    """the reference asserts `(gradient == gradient).all()` in every run() (instant_nsr.py:274); here the flag travels to the host asynchronously and
    surfaces at the next poll: a NaN in a parameter makes check_finite() raise after a training render, and a healthy net passes."""
    net, _ = golden_net(train=True)
    ro, rd = make_rays(8, 8, dist=1.8, f=6.0)
    ro_t, rd_t = torch.from_numpy(ro).to(DEV), torch.from_numpy(rd).to(DEV)
    kw = dict(num_steps=32, bound=1.6, upsample_steps=32, staged=False, cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, render_can=True, perturb=True)
    out = net.render(ro_t[None], rd_t[None], **kw)
    out["rgb"].sum().backward()
    net.check_finite()
    with torch.no_grad():
        net.sdf_net[1].bias[0] = float("nan")
    net.render(ro_t[None], rd_t[None], **kw)
    with pytest.raises(FloatingPointError):
        net.check_finite()


def test_variance_forward_kernel_equals_torch_bit_for_bit():
    """ac_variance_forward (forward_variance() without a graph in one launch) against the five torch launches it replaces -- the bits decide every sample
    position of a render, so equality is exact: ordinary values, both clip limits, NaN"""
    from avatarcraft_amd import nsr_ops
    from avatarcraft_amd.instant_nsr import SingleVarianceNetwork
    vals = [0.3, 0.6239, 0.05, -0.2, 1.0, 1.3815, 1.3816, 1.5, -1.3815, -1.3816, -2.0, 0.0, 8.9, -9.0, float("nan")] + list(np.random.RandomState(0).uniform(-1.5, 1.5, 200))
    for val in vals:
        net = SingleVarianceNetwork(float(val)).to(DEV)
        with torch.no_grad():
            want = net(torch.zeros([1, 3]))[:, :1].clip(1e-6, 1e6)
        got = nsr_ops.variance_forward(net.variance)
        assert got.shape == want.shape and got.dtype == want.dtype
        assert np.array_equal(got.cpu().numpy().view(np.uint32), want.cpu().numpy().view(np.uint32)), (val, float(got), float(want))
