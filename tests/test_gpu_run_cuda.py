"""-m gpu: the occupancy-grid render (NeRFRenderer.run_cuda behind render(cuda_ray=True)) on the MI355X
against the oracle's chain (oracle.run_cuda_train / run_cuda_eval, pinned piecewise on the CPU: tests/test_oracle_run_cuda.py):
  * ac_field_samples == orc_field_samples bit for bit on the marcher's packed samples (both delta layouts);
  * training form through net.render: rays[N,3] exactly, pixels / opacity / normal map against the oracle chain;
  * inference loop through net.render: the same number of rounds and alive rays per round, pixels against the oracle chain;
  * gradients of the training form against an independent torch-autograd formulation (torch MLPs over the HIP hash encoder, six forward_sdf
    calls for the normal, torch NeuS alpha) through the same packed compositor."""
import numpy as np
import pytest
import torch

from tests.common import make_rays
from tests.gpu_common import oracle_field as make_of, assert_bitwise
from tests.test_gpu_model import golden_net, DEV

pytestmark = pytest.mark.gpu
INV_S_VARIANCE = float(np.log(512.0) / 10.0)        # forward_variance() = exp(10 v) = 512: the sharpness the grid is built for (instant_nsr.py:325)


@pytest.fixture(scope="module")
def env(oracle):
    from avatarcraft_amd.instant_nsr import NeRFNetwork
    src, p = golden_net()
    torch.manual_seed(0)
    net = NeRFNetwork(cuda_ray=True)
    net.load_state_dict(src.state_dict(), strict=False)
    net = net.to(DEV)
    with torch.no_grad():
        net.deviation_net.variance.fill_(INV_S_VARIANCE)
    # the oracle's field from THIS net's effective matrices (ac_weight_norm_forward on the device): the goldens' effective weights were formed by
    # torch's CPU weight norm and differ from them in the last ulp, which inv_s = 512 would amplify to 1e-5 in alpha -- the comparison below is bitwise
    W = [w.detach().cpu().numpy() for w in net._effective_weights()]
    c = lambda v: v.detach().cpu().numpy()
    of = oracle.Field(c(net.encoder.embeddings), p["offsets"], W[0], c(net.sdf_net[0].bias), W[1], c(net.sdf_net[1].bias), W[2], W[3], W[4],
                      float(p["per_level_scale"]))
    grid, mean = oracle.update_density_grid(of, np.zeros((129,) * 3, np.float32), 1.6)
    # the ORACLE's grid on the device (the GPU's own update_extra_state agrees to 2e-4 of max -- tests/test_gpu_model.py -- which is not bit for bit,
    # and the chain comparison below is)
    net.density_grid.copy_(torch.from_numpy(grid)); net.mean_density = float(mean); net.iter_density = 1
    inv_s = float(net.forward_variance().item())
    return dict(net=net, of=of, grid=grid, mean=float(mean), inv_s=inv_s, O=oracle, p=p)


def test_field_samples_bitwise_vs_oracle(env):
    from avatarcraft_amd import nsr_ops, raymarching
    O, net = env["O"], env["net"]
    ro, rd = make_rays(24, 24, dist=1.8, f=18.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(t(ro), t(rd), 1.6, net.density_grid, net.mean_density, 1, align=128, force_all_rays=True)
    x_o, d_o, dl_o, r_o, c_o = O.march_rays_train(ro, rd, env["grid"], env["mean"], 1.6)
    assert np.array_equal(rays.cpu().numpy(), r_o)                                          # rays[N,3] = (id, offset, count): exactly equal
    M = int(c_o[0])
    assert M > 2000 and xyzs.shape[0] % 128 == 0
    assert_bitwise(xyzs[:M], x_o[:M], "xyzs"); assert_bitwise(dirs[:M], d_o[:M], "dirs"); assert_bitwise(deltas[:M], dl_o[:M], "deltas")
    field = net.eval()._field()
    for car in (1.0, 0.3):
        g = nsr_ops.field_samples(field, xyzs, dirs, deltas, 1.6, 0.005, env["inv_s"], car, want_sdf=True, want_gradient=True)
        r = O.field_samples(env["of"], xyzs.cpu().numpy(), dirs.cpu().numpy(), deltas.cpu().numpy(), 1.6, 0.005, env["inv_s"], car)
        for k in ("sdf", "gradient", "normal", "rgb", "alpha"):
            assert_bitwise(g[k], r[k], f"{k} (car {car})")
    # the [M,2] layout of march_rays, inv_s read from the device, a ragged count
    d2 = torch.stack([deltas, torch.full_like(deltas, 3.0)], 1).contiguous()[:M - 5]
    g2 = nsr_ops.field_samples(field, xyzs[:M - 5], dirs[:M - 5], d2, 1.6, 0.005, net.forward_variance(), 1.0)
    r2 = O.field_samples(env["of"], x_o[:M - 5], d_o[:M - 5], d2.cpu().numpy(), 1.6, 0.005, env["inv_s"], 1.0)
    for k in ("alpha", "rgb", "normal"):
        assert_bitwise(g2[k], r2[k], k + " (stride 2)")
    assert float(g["alpha"][:M].max()) > 0.05
    with pytest.raises(RuntimeError):
        nsr_ops.field_samples(field, xyzs.cpu(), dirs, deltas, 1.6, 0.005, 1.0)


def test_run_cuda_training_form_vs_oracle_chain(env):
    O, net = env["O"], env["net"].train()
    ro, rd = make_rays(32, 32, dist=1.8, f=24.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    bg = np.random.RandomState(3).uniform(0, 1, (1024, 3)).astype(np.float32)
    net.mean_count, net.local_step = 0, 0
    with torch.no_grad():
        out = net.render(t(ro)[None], t(rd)[None], num_steps=64, bound=1.6, upsample_steps=64, bg_color=t(bg), cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0,
                         perturb=False)
    r = O.run_cuda_train(env["of"], ro, rd, env["grid"], env["mean"], 1.6, 0.005, env["inv_s"], bg=bg)
    assert set(out) == {"depth", "weights", "weight_sum", "rgb", "normal", "gradient_error", "curvature_error", "pts_color", "pts_alpha", "z_vals"}
    assert out["rgb"].shape == (1, 1024, 3) and out["weight_sum"].shape == (1024, 1) and out["normal"].shape == (1024, 3)
    c = lambda v: v.detach().cpu().numpy()
    assert tuple(c(net.step_counter[0])) == (int(r["counter"][0]), 1024) and net.local_step == 1
    assert_bitwise(out["weight_sum"][:, 0], r["weights_sum"], "weights_sum")               # same samples, same per-sample bits, same compositor arithmetic
    assert np.abs(c(out["rgb"])[0] - r["image"]).max() <= 1e-6                              # (+ the background blend: torch vs numpy, one rounding)
    assert_bitwise(out["normal"], r["normal_map"], "normal_map")
    assert abs(float(out["gradient_error"]) - r["gradient_error"]) <= 1e-5 * max(1.0, r["gradient_error"])
    # against run() on the same field: the same integral by another quadrature
    net_run = env["net"]
    net_run.cuda_ray = False
    try:
        with torch.no_grad():
            ref = net_run.eval().render(t(ro)[None], t(rd)[None], num_steps=64, bound=1.6, upsample_steps=64, bg_color=t(bg), cos_anneal_ratio=1.0,
                                        normal_epsilon_ratio=0.0)
    finally:
        net_run.cuda_ray = True
    # (grazing rays leave part of their opacity outside the occupied shell of the 129^3 grid: a few silhouette pixels differ by up to 0.1 -- the
    #  oracle's chain shows the same rays, tests/test_oracle_run_cuda.py; the bulk agrees to 1e-3)
    drgb = (ref["rgb"] - out["rgb"]).abs()
    assert float(drgb.max()) <= 0.12 and float(drgb.mean()) <= 2e-3 and float((drgb.amax(-1) > 1e-2).float().mean()) <= 0.03
    # perturbed march with the per-epoch sample budget: no host synchronisation, same pixels for the rays that fit
    net.train(); net.mean_count = int(r["counter"][0]); net.local_step = 5
    with torch.no_grad():
        out2 = net.render(t(ro)[None], t(rd)[None], num_steps=64, bound=1.6, upsample_steps=64, bg_color=t(bg), cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0,
                          perturb=True)
    r2 = O.run_cuda_train(env["of"], ro, rd, env["grid"], env["mean"], 1.6, 0.005, env["inv_s"], bg=bg, perturb=1, mean_count=int(r["counter"][0]))
    assert_bitwise(out2["weight_sum"][:, 0], r2["weights_sum"], "weights_sum (perturbed, budgeted)")
    assert tuple(c(net.step_counter[5])) == (int(r2["counter"][0]), 1024)
    net.mean_count = 0


@pytest.mark.parametrize("n_rays,perturb,budget,bg_kind", [(1024, False, "loose", "rays"), (1024, True, "loose", "scalar"), (1024, True, "tight", "triple"),
                                                         (1000, True, "tight", "none"), (37, False, "loose", "scalar"), (1, True, "loose", "triple"),
                                                         (4096, True, "loose", "rays"), (300, True, None, "rays")])
def test_training_form_in_one_launch_equals_the_chain_of_operators(env, n_rays, perturb, budget, bg_kind):
    """round 5: run_cuda's train() branch under no_grad as ONE launch (ac_render_rays_occupancy_train: count, grid barrier, march + field + the packed
    compositor twice + eikonal term + background) against the chain it replaces (march_rays_train / ac_field_samples / composite_rays_train x 2 / torch):
    pixels, opacity and normal map bit for bit, the step counter exactly, the eikonal term to the rounding of a differently ordered sum -- with and without
    the marcher's jitter, with a budget that fits and with one that leaves rays out (raymarching.cu:133, 249), ragged ray counts; an un-budgeted call (no
    layout size before the count) stays with the operators in both settings"""
    net = env["net"].train()
    side = int(np.ceil(np.sqrt(n_rays)))
    ro, rd = make_rays(side, side, dist=1.8, f=0.75 * side)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a[:n_rays])).to(DEV)
    rs = np.random.RandomState(n_rays)
    bg = {"rays": torch.from_numpy(rs.uniform(0, 1, (n_rays, 3)).astype(np.float32)).to(DEV), "scalar": None,
          "triple": torch.tensor([[0.2, 0.7, 0.4]], device=DEV), "none": torch.zeros(1, 3, device=DEV)}[bg_kind]
    kw = dict(num_steps=64, bound=1.6, upsample_steps=64, bg_color=bg, cos_anneal_ratio=0.7, normal_epsilon_ratio=0.0, perturb=perturb)
    outs, counters = {}, {}
    total = None
    for one in (False, True):
        net.occupancy_train_one_launch = one
        if budget is None:
            net.mean_count = 0
        else:
            if total is None:                                   # the batch's sample count, from an un-budgeted pass
                net.mean_count, net.local_step = 0, 0
                with torch.no_grad():
                    net.render(t(ro)[None], t(rd)[None], **kw)
                total = int(net.step_counter[0, 0].item())
                assert total > 0
            net.mean_count = total if budget == "loose" else (2 * total) // 3
        net.local_step = 7
        try:
            with torch.no_grad():
                outs[one] = net.render(t(ro)[None], t(rd)[None], **kw)
        finally:
            net.occupancy_train_one_launch = True
        counters[one] = tuple(net.step_counter[7].cpu().numpy().tolist())
    net.mean_count = 0
    a, b = outs[False], outs[True]
    assert counters[True] == counters[False] and counters[True][1] == n_rays
    for k in ("weight_sum", "rgb", "normal"):
        assert_bitwise(b[k], a[k].cpu().numpy(), k)
    ga, gb = float(a["gradient_error"]), float(b["gradient_error"])
    assert np.isfinite(gb) and abs(ga - gb) <= 2e-5 * max(1.0, abs(ga)), (ga, gb)
    hit = float((a["weight_sum"] > 0).float().mean())
    if budget == "tight":
        assert 0.05 < hit < 0.95                                # the budget left the later rays out: zeros in both forms
        assert float(a["weight_sum"][-max(1, n_rays // 8):].abs().max()) == 0.0
    elif n_rays >= 37:
        assert hit > 0.05


def test_run_cuda_inference_loop_vs_oracle_chain(env):
    O, net = env["O"], env["net"].eval()
    ro, rd = make_rays(48, 48, dist=1.8, f=36.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    net.occupancy_rounds = True                     # the reference-shaped loop of rounds (the one-launch form is compared with it below)
    try:
        with torch.no_grad():
            out = net.render(t(ro)[None], t(rd)[None], num_steps=64, bound=1.6, upsample_steps=64, bg_color=None, cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0)
    finally:
        net.occupancy_rounds = False
    r = O.run_cuda_eval(env["of"], ro, rd, env["grid"], env["mean"], 1.6, 0.005, env["inv_s"])
    assert net._last_cuda_rounds == r["rounds"]
    assert_bitwise(out["weight_sum"][:, 0], r["weights_sum"], "weights_sum")
    assert_bitwise(out["normal"], r["normal_map"], "normal_map")
    c = lambda v: v.detach().cpu().numpy()
    assert np.abs(c(out["rgb"])[0] - r["image"]).max() <= 1e-6
    hit = r["weights_sum"] > 0.5
    assert hit.any() and np.abs(c(out["depth"])[0][hit] - r["depth"][hit]).max() <= 1e-6
    assert float(out["gradient_error"]) == 0.0 and out["weights"] is None and out["z_vals"] is None
    # the same render as ONE launch (ac_render_rays_occupancy, the default of run_cuda's eval()).  Bit for bit what the three operators give when run as a
    # SINGLE round (n_step = 1024): the marcher continues from its own t.  The loop of rounds restarts each round from the compositor's t = near + sum of
    # (t_after - t_before) differences, which is t up to one rounding when a ray jumps across a long empty stretch (front and back of the body): a handful of
    # rays then march from a position one ulp away -- so against the rounds the comparison is 1e-5, not bitwise.
    from avatarcraft_amd import nsr_ops, raymarching
    from avatarcraft_amd.instant_nsr import near_far_from_bound
    bg = torch.from_numpy(np.random.RandomState(9).uniform(0, 1, (ro.shape[0], 3)).astype(np.float32)).to(DEV)
    for rays in ((ro, rd), make_rays(40, 25, dist=1.7, f=30.0, yaw=1.1, pitch=0.3), make_rays(200, 200, dist=1.8, f=150.0)):   # 2304 | 1000 (not a multiple of 16) | 40 000 rays
        o_, d_ = t(rays[0]), t(rays[1])
        n_ = o_.shape[0]
        b_ = bg[:n_] if n_ <= bg.shape[0] else None
        with torch.no_grad():
            net.occupancy_rounds = True
            loop = net.render(o_[None], d_[None], num_steps=64, bound=1.6, upsample_steps=64, bg_color=b_, cos_anneal_ratio=0.7, normal_epsilon_ratio=0.0)
            net.occupancy_rounds = False
            one = net.render(o_[None], d_[None], num_steps=64, bound=1.6, upsample_steps=64, bg_color=b_, cos_anneal_ratio=0.7, normal_epsilon_ratio=0.0)
        assert net._last_cuda_rounds == 0
        for k in ("weight_sum", "rgb", "normal", "depth"):
            dmax = float((loop[k] - one[k]).abs().max())
            assert dmax <= 2e-5, (k, dmax)
            assert float((loop[k] != one[k]).float().mean()) <= 0.01, k                  # (and all but a handful of rays are equal bit for bit)
        assert float(one["weight_sum"].max()) > 0.9
        if n_ > 3000:
            continue
        # single round through the stand-alone operators
        near, far = near_far_from_bound(o_, d_, 1.6, type='cube')
        near, far = near.reshape(-1).contiguous(), far.reshape(-1).contiguous()
        alive = torch.arange(n_, dtype=torch.int32, device=DEV); rt = near.clone()
        xyzs, dirs, deltas = raymarching.march_rays(n_, 1024, alive, rt, o_, d_, 1.6, net.density_grid, net.mean_density, near, far, -1, False)
        fs = nsr_ops.field_samples(net._field(), xyzs, dirs, deltas, 1.6, 0.005, net.forward_variance(), 0.7)
        ws, dp = torch.zeros(n_, device=DEV), torch.zeros(n_, device=DEV)
        im, nm = torch.zeros(n_, 3, device=DEV), torch.zeros(n_, 3, device=DEV)
        raymarching.composite_rays(n_, 1024, alive, rt, fs["alpha"], fs["rgb"], fs["normal"], deltas, ws, dp, im, nm)
        raw = nsr_ops.render_rays_occupancy(net._field(), o_, d_, net.density_grid, net.mean_density, 1.6, 0.005, net.forward_variance(), 0.7)
        assert torch.equal(raw["weights_sum"], ws) and torch.equal(raw["depth"], dp) and torch.equal(raw["image"], im) and torch.equal(raw["normal_map"], nm)
        # max_steps (ABI 6): the one launch stops a ray after exactly that many samples == one round of the operators with n_step = max_steps
        S = 6
        alive = torch.arange(n_, dtype=torch.int32, device=DEV); rt = near.clone()
        xyzs, dirs, deltas = raymarching.march_rays(n_, S, alive, rt, o_, d_, 1.6, net.density_grid, net.mean_density, near, far, -1, False)
        fs = nsr_ops.field_samples(net._field(), xyzs, dirs, deltas, 1.6, 0.005, net.forward_variance(), 0.7)
        ws, dp = torch.zeros(n_, device=DEV), torch.zeros(n_, device=DEV)
        im, nm = torch.zeros(n_, 3, device=DEV), torch.zeros(n_, 3, device=DEV)
        raymarching.composite_rays(n_, S, alive, rt, fs["alpha"], fs["rgb"], fs["normal"], deltas, ws, dp, im, nm)
        cap = nsr_ops.render_rays_occupancy(net._field(), o_, d_, net.density_grid, net.mean_density, 1.6, 0.005, net.forward_variance(), 0.7, max_steps=S,
                                            count_samples=True)
        assert torch.equal(cap["weights_sum"], ws) and torch.equal(cap["depth"], dp) and torch.equal(cap["image"], im) and torch.equal(cap["normal_map"], nm)
        assert 0 < int(cap["n_samples"].item()) <= n_ * S and not torch.equal(cap["weights_sum"], raw["weights_sum"])
    cnt = nsr_ops.render_rays_occupancy(net._field(), t(ro), t(rd), net.density_grid, net.mean_density, 1.6, 0.005, env["inv_s"], 1.0, count_samples=True)["n_samples"]
    assert 0 < int(cnt.item()) <= ro.shape[0] * 1024


@pytest.mark.parametrize("side,n_rays,max_steps", [(7, 37, 0), (32, 1000, 0), (64, 4096, 0), (64, 4096, 5), (256, 65536, 0), (256, 65536, 23)])
def test_inference_in_phases_equals_the_one_wave_per_group_launch(env, side, n_rays, max_steps):
    """round 5: ac_render_rays_occupancy_phased (rounds of march | field | composite inside one launch, grid barriers, tiles dealt to all waves) gives the
    bits of ac_render_rays_occupancy -- any ray count, with and without the step cap, the sample count too"""
    from avatarcraft_amd import nsr_ops
    net = env["net"].eval()
    ro, rd = make_rays(side, side, dist=1.8, f=0.75 * side)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a[:n_rays])).to(DEV)
    args = (net._field(), t(ro), t(rd), net.density_grid, net.mean_density, 1.6, 0.005, env["inv_s"], 0.7)
    a = nsr_ops.render_rays_occupancy(*args, count_samples=True, max_steps=max_steps, phased=False)
    for rep in range(2):                                     # (twice: the second call runs on the scratch the first one re-armed)
        b = nsr_ops.render_rays_occupancy(*args, count_samples=True, max_steps=max_steps, phased=True)
        for k in ("weights_sum", "depth", "image", "normal_map"):
            assert torch.equal(a[k], b[k]), (k, rep)
        assert int(b["n_samples"]) > 0 and int(a["n_samples"]) > 0        # (samples EVALUATED: each kernel evaluates some past a ray's last composited one)
    assert float(a["weights_sum"].max()) > (0.5 if max_steps else 0.9)
    assert nsr_ops.occupancy_launch_failures() == 0
    if n_rays >= 2048:                                       # what run_cuda picks by itself from that many rays on
        c = nsr_ops.render_rays_occupancy(*args, max_steps=max_steps)
        assert torch.equal(c["image"], a["image"])


def test_run_cuda_gradients_vs_torch_formulation(env):
    """the training form under autograd: fused SDF-query / colour operators + packed compositor, against torch MLPs over the HIP hash encoder with the
    normal from six more forward_sdf calls (the reference's formulation of the same per-sample arithmetic), same samples, same compositor"""
    from avatarcraft_amd import raymarching
    import torch.nn as nn
    net = env["net"].train()
    ro, rd = make_rays(16, 16, dist=1.8, f=12.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    gi = torch.from_numpy(np.random.RandomState(5).normal(0, 1, (256, 3)).astype(np.float32)).to(DEV)
    net.mean_count, net.local_step = 0, 0
    net.zero_grad()
    out = net.render(t(ro)[None], t(rd)[None], num_steps=64, bound=1.6, upsample_steps=64, bg_color=None, cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, perturb=False)
    loss = (out["rgb"][0] * gi).sum() + 0.1 * out["gradient_error"] + 3.0 * out["weight_sum"].sum()
    loss.backward()
    got = {k: v.grad.detach().clone() for k, v in net.named_parameters() if v.grad is not None}
    assert set(got) >= {"encoder.embeddings", "sdf_net.0.weight_v", "color_net.2.weight_v", "deviation_net.variance"}
    # the independent formulation
    net.zero_grad()
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(t(ro), t(rd), 1.6, net.density_grid, net.mean_density, 1, align=128, force_all_rays=True)
    M = int((rays[:, 2]).sum())
    sdf_out = net.forward_sdf(xyzs, 1.6)
    gradient = net.gradient(xyzs, 1.6, 0.005).squeeze()
    normal = gradient / (1e-5 + torch.linalg.norm(gradient, ord=2, dim=-1, keepdim=True))
    rgb = net.forward_color(xyzs, dirs, normal, sdf_out[:, 1:], 1.6)
    inv_s = net.forward_variance()
    tc = (dirs * normal).sum(-1, keepdim=True)
    act = nn.Softplus(beta=100)
    half = -(act(-tc * 0.5 + 0.5) * 0.0 + act(-tc) * 1.0) * deltas.reshape(-1, 1) * 0.5
    pc, nc = torch.sigmoid((sdf_out[:, :1] - half) * inv_s), torch.sigmoid((sdf_out[:, :1] + half) * inv_s)
    alpha = ((pc - nc + 1e-5) / (pc + 1e-5)).reshape(-1).clip(0.0, 1.0)
    ws, img = raymarching.composite_rays_train(alpha, rgb, deltas, rays, 1.6)
    img = img + (1 - ws).unsqueeze(-1)
    valid = (torch.arange(xyzs.shape[0], device=DEV) < M).float()
    relax = (torch.linalg.norm(xyzs, dim=-1) < 1.2).float() * valid
    gerr = ((torch.linalg.norm(gradient, dim=-1) - 1.0) ** 2 * relax).sum() / (relax.sum() + 1e-5)
    assert float((img - out["rgb"][0]).abs().max()) <= 2e-5 and abs(float(gerr) - float(out["gradient_error"])) <= 1e-5
    ((img * gi).sum() + 0.1 * gerr + 3.0 * ws.sum()).backward()
    worst = {}
    for k, v in net.named_parameters():
        if v.grad is None:
            continue
        a, b = got[k].double(), v.grad.double()
        worst[k] = float((a - b).abs().max() / (b.abs().max() + 1e-30))
    import json, os
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(worst, open("gpurun_out/run_cuda_grad_parity.json", "w"), indent=1)
    assert all(e <= 2e-3 for e in worst.values()), worst
    net.zero_grad()


def test_harness_hands_an_eval_occupancy_net_the_whole_view(env):
    """render_instantnsr_naive batches by 4096 rays (the reference bounds ITS memory that way); an eval() occupancy-grid net renders a
    launch of any size, and a launch costs its longest ray's latency however few rays it holds -- so the harness gives it the view in one piece.  Same pixels
    as the batches, bit for bit (rays are independent); a training net, the loop of rounds or a random background keep the batches."""
    from avatarcraft_amd import render_utils as RU
    net = env["net"].eval()
    ro, rd = make_rays(96, 96, dist=1.8, f=72.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    calls = []
    orig = net.run_cuda

    def spy(rays_o, *a, **k):
        calls.append(rays_o.shape[1]); return orig(rays_o, *a, **k)
    net.run_cuda = spy
    try:
        kw = dict(requires_grad=False, bkg_key=RU.WHITE_BKG, render_can=True, perturb=False, return_raw=True, num_steps=64, upsample_steps=64, bound=1.6)
        rgb1, _, ex1 = RU.render_instantnsr_naive(net, t(ro), t(rd), rays_per_batch=1024, **kw)
        assert calls == [96 * 96]
        calls.clear()
        net.occupancy_rounds = True
        rgb2, _, ex2 = RU.render_instantnsr_naive(net, t(ro), t(rd), rays_per_batch=1024, **kw)
        net.occupancy_rounds = False
        assert calls == [1024] * 9
        calls.clear()
        rgb3, _, ex3 = RU.render_instantnsr_naive(net, t(ro), t(rd), rays_per_batch=96 * 96, **kw)          # the same single launch, asked for explicitly
        with torch.no_grad():
            parts = [net.render(t(ro)[None, i:i + 1024], t(rd)[None, i:i + 1024], num_steps=64, bound=1.6, upsample_steps=64,
                                bg_color=torch.ones(3, device=DEV), cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0) for i in range(0, 96 * 96, 1024)]
    finally:
        net.run_cuda = orig
        net.occupancy_rounds = False
    assert torch.equal(rgb1, rgb3) and torch.equal(ex1["weight_sum"], ex3["weight_sum"])
    one = torch.cat([p["rgb"][0] for p in parts])
    assert torch.equal(rgb1, one) and torch.equal(ex1["weight_sum"], torch.cat([p["weight_sum"] for p in parts]))
    assert float((rgb1 - rgb2).abs().max()) <= 2e-5                  # (the loop of rounds: equal up to the one-ulp restarts documented above)
    assert float(ex1["weight_sum"].max()) > 0.9


# ---- residency of the phased launches: never a partial result ------------------------------------------------
def _hold_half_the_device(ms):
    """a foreign workload on a stream of its own that HOLDS half of the compute units for `ms` milliseconds: one 1024-thread workgroup with 100 KB of LDS per
    unit -- no phased workgroup (145 KB of LDS) fits beside it"""
    from avatarcraft_amd import nsr_ops
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    side = torch.cuda.Stream()
    nsr_ops.debug_hold_cus(cus // 2, 100 * 1024, ms, side)
    return side


def test_phased_inference_beside_a_kernel_that_holds_half_the_device(env):
    """The phased inference launch next to a foreign kernel that holds half of the compute units for 1.5 s, with the barriers' spin bounded to 0.2 s: the
    launch's workgroups cannot all be resident, a barrier times out -- and render_rays_occupancy must return the pixels of the barrier-free kernel bit for bit
    (the library queues that kernel behind the phased one, conditional on the launch's verdict word), never the partial ones (NaN in weights_sum[0], stale rays elsewhere).  Then, with the default
    bound (2 s > the hold): the launch simply waits for the units and succeeds.  The reference's loop (raymarching.py:136-188) cannot return partial results."""
    from avatarcraft_amd import nsr_ops, _lib as L
    net = env["net"].eval()
    ro, rd = make_rays(64, 64, dist=1.8, f=48.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    args = (net._field(), t(ro), t(rd), net.density_grid, net.mean_density, 1.6, 0.005, env["inv_s"], 0.7)
    ref = nsr_ops.render_rays_occupancy(*args, phased=False)
    nsr_ops.render_rays_occupancy(*args, phased=True)                 # (scratch allocated, library warm)
    torch.cuda.synchronize()
    f0, b0 = nsr_ops.occupancy_launch_failures(), nsr_ops.occupancy_fallbacks()
    main = torch.cuda.Stream()
    prev = L.lib().ac_set_occupancy_barrier_ms(200)
    try:
        side = _hold_half_the_device(1500)
        with torch.cuda.stream(main):
            out = nsr_ops.render_rays_occupancy(*args, phased=True)
        torch.cuda.synchronize()
    finally:
        L.lib().ac_set_occupancy_barrier_ms(0 if prev == 2000 else prev)
    for k in ("weights_sum", "depth", "image", "normal_map"):
        assert bool(torch.isfinite(out[k]).all()), k
        assert torch.equal(out[k], ref[k]), k
    timed_out = nsr_ops.occupancy_launch_failures() - f0                    # (answered on the device: the conditional barrier-free launch queued behind it)
    assert nsr_ops.occupancy_fallbacks() == b0                              # the inference form needs no host-side answer
    # (whether the barrier timed out depends on the dispatcher: if the foreign kernel's workgroups had not started yet the launch may have won the units;
    #  both outcomes are correct -- what must never happen is a partial result)
    # default bound: the launch outlasts the hold
    side = _hold_half_the_device(300)
    with torch.cuda.stream(main):
        out2 = nsr_ops.render_rays_occupancy(*args, phased=True)
    torch.cuda.synchronize()
    for k in ("weights_sum", "depth", "image", "normal_map"):
        assert torch.equal(out2[k], ref[k]), k
    print(f"phased inference beside a half-device hold: barriers timed out in {timed_out} launch(es), answered by the barrier-free kernel")


def test_training_form_beside_a_kernel_that_holds_half_the_device(env):
    """the same for run_cuda's one-launch training form under no_grad: on a timed-out barrier the wrapper raises OccupancyBarrierTimeout, run_cuda renders
    through the chain of operators -- the same pixels bit for bit as an undisturbed one-launch render, the step counter too"""
    from avatarcraft_amd import nsr_ops, _lib as L
    net = env["net"].train()
    ro, rd = make_rays(64, 64, dist=1.8, f=48.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    kw = dict(num_steps=64, bound=1.6, upsample_steps=64, bg_color=torch.tensor([[0.2, 0.7, 0.4]], device=DEV), cos_anneal_ratio=0.7, normal_epsilon_ratio=0.0, perturb=True)
    net.mean_count, net.local_step = 0, 0
    with torch.no_grad():
        net.render(t(ro)[None], t(rd)[None], **kw)
    net.mean_count = int(net.step_counter[0, 0].item())
    assert net.mean_count > 0
    try:
        net.local_step = 3
        with torch.no_grad():
            ref = net.render(t(ro)[None], t(rd)[None], **kw)
        ref_counter = net.step_counter[3].cpu().numpy().tolist()
        torch.cuda.synchronize()
        b0 = nsr_ops.occupancy_fallbacks()
        main = torch.cuda.Stream()
        prev = L.lib().ac_set_occupancy_barrier_ms(200)
        try:
            side = _hold_half_the_device(1500)
            net.local_step = 5
            with torch.cuda.stream(main), torch.no_grad():
                out = net.render(t(ro)[None], t(rd)[None], **kw)
            torch.cuda.synchronize()
        finally:
            L.lib().ac_set_occupancy_barrier_ms(0 if prev == 2000 else prev)
        for k in ("weight_sum", "rgb", "normal"):
            assert bool(torch.isfinite(out[k]).all()), k
            assert torch.equal(out[k], ref[k]), k
        assert np.isfinite(float(out["gradient_error"]))
        assert net.step_counter[5].cpu().numpy().tolist() == ref_counter
        print(f"training form beside a half-device hold: {nsr_ops.occupancy_fallbacks() - b0} launch(es) answered by the chain of operators")
    finally:
        net.mean_count = 0


@pytest.mark.parametrize("car,budget", [(1.0, False), (0.4, True)])
def test_fused_packed_shading_equals_the_torch_glue(env, car, budget):
    """round 6: run_cuda's train() branch under autograd with normal / NeuS alpha / eikonal terms of the packed samples as one launch
    each way (nsr_ops.packed_shading) against the torch formulation it replaces (occupancy_fused_shading = False): pixels, opacity and the eikonal term to
    float rounding (torch's softplus / sigmoid against the kernels' table / polynomial: <= 2e-6), the gradient of every parameter to <= 1e-4 of its largest entry --
    with and without cosine annealing, with and without a sample budget (rows past the marched samples are alignment padding: no eikonal share)"""
    net = env["net"].train()
    ro, rd = make_rays(24, 24, dist=1.8, f=18.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    gi = torch.from_numpy(np.random.RandomState(7).normal(0, 1, (576, 3)).astype(np.float32)).to(DEV)
    kw = dict(num_steps=64, bound=1.6, upsample_steps=64, bg_color=torch.tensor([[0.3, 0.6, 0.1]], device=DEV), cos_anneal_ratio=car, normal_epsilon_ratio=0.0, perturb=True)
    net.mean_count, net.local_step = 0, 0
    if budget:
        with torch.no_grad():
            net.render(t(ro)[None], t(rd)[None], **kw)
        net.mean_count = int(net.step_counter[0, 0].item())
    res = {}
    try:
        for fused in (True, False):
            net.occupancy_fused_shading = fused
            net.zero_grad()
            net.local_step = 9
            out = net.render(t(ro)[None], t(rd)[None], **kw)
            ((out["rgb"][0] * gi).sum() + 0.1 * out["gradient_error"] + 3.0 * out["weight_sum"].sum()).backward()
            res[fused] = (out["rgb"].detach().clone(), out["weight_sum"].detach().clone(), float(out["gradient_error"]),
                          {k: v.grad.detach().clone() for k, v in net.named_parameters() if v.grad is not None})
    finally:
        net.occupancy_fused_shading = True
        net.mean_count = 0
        net.zero_grad()
    (ia, wa, ea, ga), (ib, wb, eb, gb) = res[True], res[False]
    assert float((ia - ib).abs().max()) <= 2e-6 and float((wa - wb).abs().max()) <= 2e-6 and abs(ea - eb) <= 2e-6 * max(1.0, abs(eb))
    assert set(ga) == set(gb) and "deviation_net.variance" in ga and "encoder.embeddings" in ga
    worst = {k: float((ga[k] - gb[k]).abs().max() / (gb[k].abs().max() + 1e-30)) for k in ga}
    assert all(e <= 1e-4 for e in worst.values()), worst
