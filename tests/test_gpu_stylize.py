"""-m gpu: SURVEY 8(f) rank 1 on the device -- the stylisation outer loop (avatarcraft_amd.stylize.stylize_epochs) with the SD adapter
(guidance.SDSGuidance over guidance.StableDiffusion) driving the REAL NeRFNetwork through the fused training path.

* step 0 against the reference: tests/golden/trainer_step.npz was recorded by running the reference's own Trainer.train (stylize.py:47-217)
  in the build container (tests/golden/make_trainer_golden.py) over the tiny seeded SD stand-ins of tests/common_sd.py: camera jitter, a head
  close-up, noise background, view-dependent prompt, render_val -> mannual_backward -> patch loop with its three backward passes.  The test
  replays the recorded random draws (device RNG streams differ from the CPU's) and compares the image handed to the guidance, its gradient,
  the loss values and the accumulated gradient of every parameter at the first optimizer.step().
* the loop itself: one coarse + one fine epoch (head views, background / prompt augmentation) and one fine step of a 256 x 256 view
  (16 patches of 4096 rays through the shared backward scratch), parameters finite and moving."""
import os
import random

import numpy as np
import pytest
import torch

from tests.common import load_golden, make_rays
from tests.test_gpu_model import golden_net, DEV

pytestmark = pytest.mark.gpu


def _sd(dev):
    from avatarcraft_amd.guidance import StableDiffusion
    from tests import common_sd as SD
    return StableDiffusion(torch.device(dev), "1.5", components=SD.components())


class _Replay:
    """hands the recorded draws to the code under test, in the order the reference consumed them"""

    def __init__(self, g):
        self.g = g
        self.rand = [g["noise_val"], g["noise_grad"]]
        self.bkg = [g["bkg_val"], g["bkg"], g["bkg"]]            # render_val | grad render | frozen net (its own draw in the reference: unused by the loss)
        self.randn = list(g["sd_randn_like"])
        self.randint = [g["sd_t"]]

    def __enter__(self):
        import avatarcraft_amd.render_utils as RU
        self.o = (torch.rand, torch.randn_like, torch.randint, RU.select_background)
        t = lambda a, like=None: torch.from_numpy(np.ascontiguousarray(a))
        torch.rand = lambda *a, **k: t(self.rand.pop(0)).to(k.get("device", DEV))
        torch.randn_like = lambda x, **k: t(self.randn.pop(0)).to(x.device).reshape(x.shape)
        torch.randint = lambda *a, **k: t(self.randint.pop(0)).to(k.get("device", "cpu"))
        RU.select_background = lambda shape, key: t(self.bkg.pop(0))
        return self

    def __exit__(self, *a):
        import avatarcraft_amd.render_utils as RU
        torch.rand, torch.randn_like, torch.randint, RU.select_background = self.o


def test_stylize_step0_matches_reference_trainer():
    from avatarcraft_amd import render_utils as RU
    from avatarcraft_amd.guidance import SDSGuidance
    from avatarcraft_amd.stylize import sds_step, flat_grad_view
    g = load_golden("trainer_step.npz")
    # the view: the reference's jittered head close-up -> my pose2cap / cap2rays / sparse_ray_sampling on the recorded pose and offsets
    pose = RU.CameraPose(g["pose_c2w"])
    offs = g["py_randint"]
    assert tuple(offs[0][:2]) == (0, 2) and int(offs[0][2]) == int(g["bkg_key"])          # random.randint(WHITE_BKG, NOISE_BKG)
    ro, rd = torch.from_numpy(g["rays_o"]).to(DEV), torch.from_numpy(g["rays_d"]).to(DEV)
    o_full, d_full = RU.cap2rays(RU.pose2cap([64, 64], pose), device=DEV)
    o_sub = o_full.reshape(64, 64, 3)[int(offs[1][2])::4, int(offs[2][2])::4].reshape(-1, 3).float()
    d_sub = d_full.reshape(64, 64, 3)[int(offs[1][2])::4, int(offs[2][2])::4].reshape(-1, 3).float()
    assert float((o_sub - ro).abs().max()) <= 1e-6 and float((d_sub - rd).abs().max()) <= 2e-6
    net, _ = golden_net(train=True)
    net_gt, _ = golden_net(train=False)
    with torch.no_grad():
        net_gt.sdf_net[1].bias[0] = float(g["gt_sdf_bias"])
    prompt = str(g["prompt"])
    sd = _sd(DEV)
    guide = SDSGuidance(sd, "Hulk, photorealistic style", 100.0)
    seen = {}

    def guidance(rgb):
        seen["image"] = rgb.detach().clone()
        seen["grad"] = guide(rgb, text=prompt)
        return seen["grad"]
    opt = torch.optim.SGD(net.parameters(), lr=0.0)            # the gradients at the first optimizer.step() are what the golden holds
    flat = flat_grad_view(net.parameters())
    # The three loss terms' table gradients SEPARATELY (the golden holds .grad after each of the reference's three backward() calls): the step's one
    # combined backward runs first (the product path, judged below on the total), then the same saved forward is back-propagated once per term into a
    # zeroed buffer, and the combined gradient is put back.  Gradients are linear in the upstream gradient: the terms must add up to the total.
    emb_idx = torch.from_numpy(g["emb_idx"]).to(DEV)
    terms = {}
    combined_backward = net.backward_last

    def backward_per_term(g_image=None, g_weights_sum=None, g_eik=None, split=None):
        saved = net._last_train
        combined_backward(g_image=g_image, g_weights_sum=g_weights_sum, g_eik=g_eik, split=split)
        total = flat.clone()
        for name, kw in (("rgb", dict(g_image=g_image)), ("eikonal", dict(g_eik=g_eik)), ("opacity", dict(g_weights_sum=g_weights_sum))):
            flat.zero_()
            net._last_train = saved
            combined_backward(**kw)
            terms[name] = net.encoder.embeddings.grad.detach().clone()
        # ... and the opacity term once more with the REFERENCE's upstream gradient (from its recorded opacities: d = clip(ws) - clip(ws_gt), SmoothL1'(d) 1e5 / n,
        # stylize.py:187-190) through OUR saved forward: what is left of that term's error when the upstream is taken out of it (the attribution below)
        pred, gt = g["weight_sum"].reshape(-1).astype(np.float64), g["weight_sum_gt"].reshape(-1).astype(np.float64)
        dref = np.clip(pred, 0, 1) - np.clip(gt, 0, 1)
        gws_ref = np.where(np.abs(dref) < 1.0, dref, np.sign(dref)) * (1e5 / pred.shape[0]) * ((pred >= 0) & (pred <= 1))
        terms["_upstream_mine"] = g_weights_sum.detach().reshape(-1).double().cpu().numpy()
        terms["_upstream_ref"] = gws_ref
        flat.zero_()
        net._last_train = saved
        combined_backward(g_weights_sum=torch.from_numpy(gws_ref).to(g_weights_sum).reshape(g_weights_sum.shape))
        terms["opacity_ref_upstream"] = net.encoder.embeddings.grad.detach().clone()
        # ... and the share of the rays whose opacity the two forwards render most differently (a sample placed on the other side of a near-tie of the
        # inverse-CDF sampler moves a ray's fine-level cells): their own opacity-term gradient, alone
        ws_mine = saved[0]["weights_sum"].detach().reshape(-1).double().cpu().numpy()
        dws = np.abs(ws_mine - pred) + np.abs(saved[0]["image"].detach().double().cpu().numpy() - g["rgb_grad_render"]).max(axis=1)
        terms["_dws"] = dws
        suspects = np.argsort(-dws)[:3]
        terms["_suspects"] = suspects
        only = torch.zeros_like(g_weights_sum).reshape(-1)
        only[torch.from_numpy(suspects).to(only.device)] = g_weights_sum.reshape(-1)[torch.from_numpy(suspects).to(only.device)]
        flat.zero_()
        net._last_train = saved
        combined_backward(g_weights_sum=only.reshape(g_weights_sum.shape))
        terms["opacity_suspects_only"] = net.encoder.embeddings.grad.detach().clone()
        if __import__("os").environ.get("AC_DIAG_OPACITY_ENTRIES"):           # diagnosis (profiles/r06_experiments.txt section 9): which rays feed the entries above 3e-3
            ref_o_ = g["emb_grad_terms"][2]
            e_ = np.abs(terms["opacity"][emb_idx].cpu().numpy() - ref_o_).max(axis=1) / np.abs(ref_o_).max()
            ab_ = np.nonzero(e_ > 3e-3)[0]
            rows = emb_idx[torch.from_numpy(ab_).to(emb_idx.device)]
            per_ray = np.zeros((g_weights_sum.numel(), ab_.size, 2))
            for r in range(g_weights_sum.numel()):
                one = torch.zeros_like(g_weights_sum).reshape(-1); one[r] = g_weights_sum.reshape(-1)[r]
                flat.zero_(); net._last_train = saved
                combined_backward(g_weights_sum=one.reshape(g_weights_sum.shape))
                per_ray[r] = net.encoder.embeddings.grad.detach()[rows].cpu().numpy()
            terms["_per_ray"] = (ab_, per_ray, terms["opacity"][rows].cpu().numpy(), ref_o_[ab_], rows.cpu().numpy())
        flat.copy_(total)
    net.backward_last = backward_per_term
    with _Replay(g) as rp:
        stats = sds_step(net, net_gt, ro, rd, (16, 16), opt, guidance, batch_size=4096, w_eikonal=0.01, use_opacity=True, bkg_key=int(g["bkg_key"]), flat_grad=flat)
        assert not rp.rand and not rp.randn and not rp.randint and len(rp.bkg) == 0, "the step consumed a different number of random draws than the reference"
    net.check_finite()
    # render_val against the reference's: within the 1e-3 north-star tolerance except where the two inverse-CDF samplers place a jittered up-sample
    # differently (exp rounding at a near-tie: the reference's own torch-CPU and torch-GPU builds differ there too) -- 1 ray of this 256-ray
    # close-up, by 1.3e-3 (the CPU oracle shows the same ray: tests/test_oracle_golden.py::test_trainer_view_forward_vs_reference)
    dimg = np.abs(seen["image"].cpu().numpy() - g["guidance_image"])
    assert dimg.max() <= 3e-3 and float((dimg > 1e-3).mean()) <= 0.01, (dimg.max(), float((dimg > 1e-3).mean()))
    gs = np.abs(g["guidance_grad"]).max()
    assert np.abs(seen["grad"].cpu().numpy() - g["guidance_grad"]).max() <= 2e-3 * gs                  # SD stand-ins: MIOpen vs oneDNN convolutions
    assert abs(float(stats["opacity"]) - float(g["opacity_loss"])) <= 3e-3 * float(g["opacity_loss"])
    assert abs(float(stats["eikonal"]) - 0.01 * float(g["eikonal"])) <= 2e-3 * 0.01 * float(g["eikonal"])
    worst = {}
    for k, prm in net.named_parameters():
        got = prm.grad.detach().cpu().numpy()
        if k == "encoder.embeddings":
            ref, got = g["emb_grad"], got[g["emb_idx"]]
        else:
            ref = g["grad." + k]
        worst[k] = float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))
    import json, os
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(worst, open("gpurun_out/trainer_step0_parity.json", "w"), indent=1)
    # (the MLP gradients agree with the reference's to 1e-4; the sampled table entries to 1e-2: single entries collect the 1e5-weighted opacity
    #  term of few rays, and that term is a difference of two opacities that the two forwards produce to ~1e-5 each -- the tight statement about the
    #  backward itself is the chain through the oracle, tests/test_gpu_model.py::test_sds_step_matches_reference_step (a))
    # per term: rgb and eikonal are smooth in the forward and agree like the MLP gradients; the opacity term is 1e5 x a difference of two opacities
    # (|d| ~ 1e-3) that the two forwards reproduce to ~1e-5 each, and a sample placed differently on one ray moves single table entries -- it alone
    # carries the loose bound, and the bound is on ITS scale, not on the sum's
    net.backward_last = combined_backward
    assert {"rgb", "eikonal", "opacity", "opacity_ref_upstream"} <= set(terms)
    # ---- the loosest margin of the repository, diagnosed (What's weak 1c; profiles/r06_experiments.txt section 9).  4 of the 4096 sampled
    # entries of the opacity term exceed 3e-3 of the term's max (worst 1.0e-2).  What they are NOT: the upstream d = clip(weight_sum) - clip(weight_sum_gt)
    # (fed with the REFERENCE's upstream through our forward and backward they keep their error: 9.97e-3), the 8-byte queue records (the full-fp32 library
    # gives the same four numbers), the rays whose opacity the two forwards render most differently (they do not touch these entries).  What they ARE
    # (per-ray decomposition, AC_DIAG_OPACITY_ENTRIES=1): entries of the two finest levels (13, 14: cells of 2.2 / 1.6 mm) that receive ONE sample of ONE
    # ray (at most three), at a corner FAR from that sample -- a trilinear weight of ~1e-2 .. 1e-3 -- times the 1e5-weighted upstream: e.g. entry 4733259,
    # mine (-0.2977, -1.7123), reference (-0.3440, -1.9778): both channels scaled by 0.866, i.e. the corner's WEIGHT differs by 13 %, which is a 1.5e-6
    # difference in the sample's position (the two fp32 forwards' inverse-CDF lerps: the same roundings that cause the recorded index flip of the goldens).
    # They are small entries (<= 5 % of the term's max).  Recorded here: their number, level, size; every OTHER sampled entry is held to 3e-3.
    ref_o = g["emb_grad_terms"][2]
    mx_o = float(np.abs(ref_o).max())
    err_mine = np.abs(terms["opacity"][emb_idx].cpu().numpy() - ref_o).max(axis=1) / mx_o
    err_refup = np.abs(terms["opacity_ref_upstream"][emb_idx].cpu().numpy() - ref_o).max(axis=1) / mx_o
    above = np.nonzero(err_mine > 3e-3)[0]
    up_m, up_r = terms["_upstream_mine"], terms["_upstream_ref"]
    rel_up = np.abs(up_m - up_r) / np.maximum(np.abs(up_r), 1e-30)
    opacity_attribution = {"entries_sampled": int(err_mine.shape[0]), "entries_above_3e-3_with_own_upstream": int(above.size),
                           "worst_with_own_upstream": float(err_mine.max()), "worst_with_reference_upstream": float(err_refup.max()),
                           "worst_of_those_entries_with_reference_upstream": float(err_refup[above].max()) if above.size else 0.0,
                           "upstream_rel_diff_median_and_max_over_rays": [float(np.median(rel_up)), float(rel_up.max())],
                           "upstream_abs_diff_max_over_rays_in_units_of_1e5_over_n": float(np.abs(up_m - up_r).max() / (1e5 / up_r.shape[0]))}
    sus = np.abs(terms["opacity_suspects_only"][emb_idx].cpu().numpy()).max(axis=1) / mx_o
    opacity_attribution.update({"suspect_rays": [int(r) for r in terms["_suspects"]], "suspect_rays_abs_opacity_diff": [float(terms["_dws"][r]) for r in terms["_suspects"]],
                                "median_abs_opacity_diff": float(np.median(terms["_dws"])),
                                "entries_above": [int(i) for i in above[:16]], "their_error": [float(err_mine[i]) for i in above[:16]],
                                "their_error_with_reference_upstream": [float(err_refup[i]) for i in above[:16]],
                                "suspect_rays_own_gradient_at_those_entries_of_max": [float(sus[i]) for i in above[:16]],
                                "entries_touched_by_suspect_rays": int((sus > 0).sum()),
                                "worst_error_outside_suspect_rays_entries": float(err_mine[sus == 0].max())})
    print("opacity attribution:", json.dumps(opacity_attribution))
    if "_per_ray" in terms:
        ab_, per_ray, mine_, ref_, rows_ = terms["_per_ray"]
        offs = np.asarray(net.encoder.offsets.cpu().numpy())
        for j in range(ab_.size):
            c = np.abs(per_ray[:, j]).max(axis=1)
            top = np.argsort(-c)[:5]
            lvl = int(np.searchsorted(offs, rows_[j], side="right") - 1)
            print(f"  entry {int(rows_[j])} (level {lvl}): mine {mine_[j]}, reference {ref_[j]}, sum over rays {per_ray[:, j].sum(0)}; contributing rays {int((c > 0).sum())}; "
                  f"top rays {[(int(r), [float(x) for x in per_ray[r, j]]) for r in top]}; |dws| of those {[float(terms['_dws'][r]) for r in top]}")
    offs_h = np.asarray(net.encoder.offsets.cpu().numpy())
    lv_above = [int(np.searchsorted(offs_h, int(emb_idx[i]), side="right") - 1) for i in above]
    size_above = [float(np.abs(ref_o[i]).max() / mx_o) for i in above]
    opacity_attribution.update({"levels_of_entries_above": lv_above, "their_size_of_max": size_above})
    worst["opacity_term_attribution"] = opacity_attribution
    json.dump(worst, open("gpurun_out/trainer_step0_parity.json", "w"), indent=1)
    assert above.size <= 8, opacity_attribution                              # observed 4 of 4096
    assert all(l >= 12 for l in lv_above) and all(z <= 0.1 for z in size_above), opacity_attribution      # fine levels, small entries (see above)
    assert float(err_mine[np.setdiff1d(np.arange(err_mine.shape[0]), above)].max(initial=0.0)) <= 3e-3   # the tightened bound on everything else
    tsum = terms["rgb"].double() + terms["eikonal"].double() + terms["opacity"].double()
    tot = net.encoder.embeddings.grad.detach().double()
    lin = float((tsum - tot).abs().max() / tot.abs().max())
    term_err = {}
    for j, name in enumerate(("rgb", "eikonal", "opacity")):
        ref = g["emb_grad_terms"][j]
        got = terms[name][emb_idx].cpu().numpy()
        term_err[name] = float(np.abs(got - ref).max() / np.abs(ref).max())
        l2 = float(torch.sqrt((terms[name].double() ** 2).sum()))
        term_err[name + "_l2_rel"] = abs(l2 - float(g["emb_terms_l2"][j])) / float(g["emb_terms_l2"][j])
    worst.update({"table_term." + k: v for k, v in term_err.items()}); worst["table_terms_sum_vs_combined"] = lin
    worst["opacity_term_attribution"] = opacity_attribution
    json.dump(worst, open("gpurun_out/trainer_step0_parity.json", "w"), indent=1)
    # (the binned scatter sums in a fixed-point scale chosen from the level's largest |v| of THAT launch, so a term alone and the sum are rounded on
    #  different grids: 1.5e-5 of max observed)
    assert lin <= 5e-5, lin
    # observed on MI355X: rgb 1.8e-3, eikonal 5.1e-3 (a term of absolute size 2e-5), opacity 1.0e-2 of each term's own max; L2 norms 3e-4 / 1e-6 / 8e-4
    assert term_err["rgb"] <= 3e-3 and term_err["eikonal"] <= 1e-2 and term_err["opacity"] <= 2e-2, term_err
    assert term_err["rgb_l2_rel"] <= 1e-3 and term_err["eikonal_l2_rel"] <= 1e-3 and term_err["opacity_l2_rel"] <= 3e-3, term_err
    for k, e in worst.items():
        if not k.startswith("table_term") and not isinstance(e, dict):
            assert e <= (2e-2 if k == "encoder.embeddings" else 1e-3), (k, e, worst)           # total: the opacity term dominates it (1.0e-2 observed)
    nnz = int((net.encoder.embeddings.grad.abs().sum(1) > 0).sum())
    assert abs(nnz - int(g["emb_nnz"])) <= 0.01 * int(g["emb_nnz"])


def test_stylize_epochs_coarse_and_fine_on_the_device():
    from avatarcraft_amd.guidance import SDSGuidance
    from avatarcraft_amd.stylize import stylize_epochs, flat_grad_view
    net, _ = golden_net(train=True)
    net_gt, _ = golden_net(train=False)
    opt = torch.optim.Adam([{"params": net.parameters(), "lr": 5e-3}], fused=True)
    flat = flat_grad_view(net.parameters())
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    guide = SDSGuidance(_sd(DEV), "Hulk, photorealistic style", 100.0)
    log = []
    torch.manual_seed(0); random.seed(0)
    steps = stylize_epochs(net, net_gt, opt, guide, hw=(64, 64), n_cap=4, coarse_epochs=1, fine_epochs=1, subsample_scale=4, augment_cam=True, stylize_head=True,
                           coarse_head=0.5, fine_head=0.5, augment_bkg=True, augment_text=True, tgt_text="Hulk, photorealistic style", batch_size=4096, device=DEV,
                           flat_grad=flat, on_step=lambda s, e, st: log.append((s, e, float(st["eikonal"]), float(st["opacity"]))))
    net.check_finite()
    assert steps == len(log) and steps >= 8 and {e for _, e, _, _ in log} == {0, 1}                   # 4 + head views per epoch, two epochs
    assert all(np.isfinite(v) for _, _, a, b in log for v in (a, b))
    for k, v in net.named_parameters():
        assert torch.isfinite(v).all(), k
        assert float((v.detach() - before[k]).abs().max()) > 0, k
    # one fine step of a full 256 x 256 view: 65 536 rays = a 16-batch render_val + 16 patches of 4096 rays through the shared 4.5 GB scratch
    from avatarcraft_amd import render_utils as RU
    from avatarcraft_amd.stylize import sds_step
    poses, _ = RU.default_360_path(np.zeros(3), np.array([0.0, 1.0, 0.0]), 1.8, 4, add_noise=False)
    ro, rd = RU.cap2rays(RU.pose2cap([256, 256], poses[1]), device=DEV)
    marks = []
    st = sds_step(net, net_gt, ro.reshape(-1, 3).float().contiguous(), rd.reshape(-1, 3).float().contiguous(), (256, 256), opt, guide, batch_size=4096,
                  flat_grad=flat, timers=marks)
    net.check_finite()
    assert sum(1 for n, _ in marks if n == "backward") == 16
    assert np.isfinite(float(st["eikonal"])) and np.isfinite(float(st["opacity"]))
    for k, v in net.named_parameters():
        assert torch.isfinite(v).all(), k


def test_real_sd_guidance_runs_or_says_why_not():
    """the real Stable-Diffusion guidance (models/diffusion.py:28-69,92-149) either runs one SDS step on this box or the
    probe names what is missing -- recorded in gpurun_out/real_sd.json either way (bench.py carries the same field)."""
    import json, os
    from avatarcraft_amd.guidance import real_sd_probe
    ok, why = real_sd_probe("1.5")
    rec = {"available": ok, "detail": why}
    if ok:
        from avatarcraft_amd.guidance import StableDiffusion, SDSGuidance
        from avatarcraft_amd.stylize import sds_step, flat_grad_view
        net, _ = golden_net(train=True)
        net_gt, _ = golden_net(train=False)
        opt = torch.optim.Adam(net.parameters(), lr=5e-3)
        flat = flat_grad_view(net.parameters())
        from avatarcraft_amd.synthetic import make_rays
        ro, rd = make_rays(64, 64, dist=1.8, f=50.0)
        guide = SDSGuidance(StableDiffusion(torch.device(DEV), "1.5"), "Hulk, photorealistic style", 100.0)
        before = net.encoder.embeddings.detach().clone()
        sds_step(net, net_gt, torch.from_numpy(ro).to(DEV), torch.from_numpy(rd).to(DEV), (64, 64), opt, guide, batch_size=4096, flat_grad=flat)
        net.check_finite()
        assert torch.isfinite(flat).all() and float((net.encoder.embeddings.detach() - before).abs().max()) > 0
        rec["ran"] = True
    else:
        assert isinstance(why, str) and len(why) > 10
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rec, open("gpurun_out/real_sd.json", "w"))


def test_sd_architecture_standin_has_the_checkpoints_shapes_and_drives_a_step():
    """avatarcraft_amd.sd_arch: the UNet / VAE-encoder of Stable-Diffusion 1.5 restated from the published configuration (random weights; a clock for
    bench.py's sds_step_sd_arch_standin).  Parameter counts equal the checkpoint's (UNet2DConditionModel 859 520 964, AutoencoderKL encoder + quant_conv
    34 163 664); one SDS step through guidance.StableDiffusion over it moves the parameters and stays finite."""
    from avatarcraft_amd import sd_arch
    from avatarcraft_amd.guidance import StableDiffusion, SDSGuidance
    from avatarcraft_amd.stylize import sds_step, flat_grad_view
    from avatarcraft_amd.synthetic import make_rays
    assert sd_arch.parameter_counts() == (859520964, 34163664)
    net, _ = golden_net(train=True)
    net_gt, _ = golden_net(train=False)
    opt = torch.optim.Adam(net.parameters(), lr=5e-3)
    flat = flat_grad_view(net.parameters())
    guide = SDSGuidance(StableDiffusion(torch.device(DEV), "1.5", components=sd_arch.components(device=DEV)), "Hulk, photorealistic style", 100.0)
    ro, rd = make_rays(64, 64, dist=1.8, f=50.0)
    seen = {}

    def g(rgb):
        seen["grad"] = guide(rgb)
        return seen["grad"]
    before = net.encoder.embeddings.detach().clone()
    sds_step(net, net_gt, torch.from_numpy(ro).to(DEV), torch.from_numpy(rd).to(DEV), (64, 64), opt, g, batch_size=4096, flat_grad=flat)
    net.check_finite()
    assert seen["grad"].shape == (1, 3, 64, 64) and torch.isfinite(seen["grad"]).all() and float(seen["grad"].abs().max()) > 0
    assert torch.isfinite(flat).all() and float((net.encoder.embeddings.detach() - before).abs().max()) > 0
    del guide
    torch.cuda.empty_cache()


def test_pair_launch_only_with_a_guidance_that_keeps_off_the_global_rng():
    """the pair launch draws the training render's jitter before the guidance runs; a guidance that draws from the GLOBAL generator
    (the real SD guidance: timestep, noise) must therefore get the reference's order render_val -> guidance -> training render.  With such a guidance
    the step is the same whether pairing is allowed or not; forcing the pair on it changes the draws (which is why it is gated)."""
    from avatarcraft_amd import stylize as S
    from avatarcraft_amd.synthetic import make_rays
    ro, rd = make_rays(16, 16, dist=1.8, f=12.0)
    ro, rd = torch.from_numpy(ro).to(DEV), torch.from_numpy(rd).to(DEV)

    class GlobalRngGuidance:
        def __call__(self, rgb, text=None):
            return torch.randn(rgb.shape, device=rgb.device).clamp_(-1, 1)      # the global CUDA generator, like StableDiffusion.mannual_backward

    def step(pair_allowed, private=None):
        net, _ = golden_net(train=True)
        net_gt, _ = golden_net(train=False)
        opt = torch.optim.SGD(net.parameters(), lr=0.0)
        flat = S.flat_grad_view(net.parameters())
        g = GlobalRngGuidance()
        if private is not None:
            g.private_rng = private
        old = S.PAIR_STEP_RENDERS
        S.PAIR_STEP_RENDERS = pair_allowed
        try:
            torch.manual_seed(1234)
            marks = []
            S.sds_step(net, net_gt, ro, rd, (16, 16), opt, g, batch_size=4096, flat_grad=flat, timers=marks)
        finally:
            S.PAIR_STEP_RENDERS = old
        return flat.clone(), [n for n, _ in marks]
    a, ma = step(True)
    b, mb = step(False)
    assert "render_val" in ma and "render_val_and_grad_forward" not in ma and ma == mb       # not paired: the reference's order
    assert torch.equal(a, b)
    c, mc = step(True, private=True)
    assert "render_val_and_grad_forward" in mc
    assert not torch.equal(a, c)                                                              # the draws really are consumed in another order


def test_adam_step_kernel_equals_torch_adam():
    """stylize.Adam (ac_adam_step: every tensor in one launch) against torch.optim.Adam on the same gradients: ragged, unaligned and multi-block tensors,
    several steps (bias corrections), state_dict interchange both ways, and the in-step clearing of the gradients"""
    from avatarcraft_amd.stylize import Adam
    torch.manual_seed(3)
    shapes = [(300001, 2), (64, 35), (64,), (1,), (4099,), (16, 64), (5000, 3)]
    flat = torch.randn(sum(int(np.prod(s)) for s in shapes) + 3, device=DEV)
    ps_a, ps_b, off = [], [], 1                                  # offset 1: every tensor but the first is not 16-byte aligned
    for sh in shapes:
        n = int(np.prod(sh))
        ps_a.append(torch.nn.Parameter(flat[off:off + n].clone().view(sh))); ps_b.append(torch.nn.Parameter(flat[off:off + n].clone().view(sh)))
        off += n
    gflat = torch.zeros(off + 3, device=DEV)
    o = 1
    for p in ps_a:
        p.grad = gflat[o:o + p.numel()].view_as(p); o += p.numel()
    ours, ref = Adam(ps_a, lr=5e-3, zero_grad_in_step=True), torch.optim.Adam(ps_b, lr=5e-3)
    for step in range(5):
        for pa, pb in zip(ps_a, ps_b):
            g = torch.randn_like(pb) * (10.0 ** (step - 2))
            pa.grad.copy_(g); pb.grad = g.clone()
        ours.step(); ref.step()
        assert ours.grads_cleared and float(gflat.abs().max()) == 0.0
        for pa, pb in zip(ps_a, ps_b):
            d = float((pa - pb).abs().max())
            assert d <= 2e-6 * max(1.0, float(pb.abs().max())), (step, tuple(pa.shape), d)          # a few ulps: fma contraction and operation order differ
            assert float((ours.state[pa]["exp_avg_sq"] - ref.state[pb]["exp_avg_sq"]).abs().max()) <= 1e-6 * float(ref.state[pb]["exp_avg_sq"].abs().max())
    assert int(ours.state[ps_a[0]]["step"].item()) == 5
    # state dicts are interchangeable: torch's state into ours and back
    ours2 = Adam(ps_a, lr=5e-3); ours2.load_state_dict(ref.state_dict())
    ref2 = torch.optim.Adam(ps_b, lr=5e-3); ref2.load_state_dict(ours.state_dict())
    for pa, pb in zip(ps_a, ps_b):
        g = torch.randn_like(pb)
        pa.grad.copy_(g); pb.grad = g.clone()
    with torch.no_grad():
        for pa, pb in zip(ps_a, ps_b):
            pa.copy_(pb)
    ours2.step(); ref2.step()
    for pa, pb in zip(ps_a, ps_b):
        assert float((pa - pb).abs().max()) <= 2e-6 * max(1.0, float(pb.abs().max()))
    assert not ours2.grads_cleared and float(gflat.abs().max()) > 0.0             # zero_grad_in_step off: gradients left alone
    with pytest.raises(NotImplementedError):
        bad = Adam(ps_a, lr=1e-3); bad.param_groups[0]["weight_decay"] = 0.1; bad.step()


def test_sds_steps_with_the_one_launch_adam():
    """stylisation steps from identical state.  (a) ONE step with stylize.Adam against torch.optim.Adam: the parameters agree to rounding (later steps are
    no test of an optimizer: with lr = 5e-3 on table entries of 1e-4 the trajectory amplifies a last-bit difference to 1e-5 within two steps).  (b) three
    steps with the gradients cleared INSIDE the optimizer's launch (sds_step then skips its own clearing) against three steps that clear them explicitly:
    bit for bit -- a gradient left over, or cleared too early, would show."""
    from avatarcraft_amd.stylize import sds_step, flat_grad_view, SyntheticGuidance, Adam
    from avatarcraft_amd.synthetic import make_rays
    ro, rd = make_rays(64, 64, dist=1.8, f=50.0)
    ro, rd = torch.from_numpy(ro).to(DEV), torch.from_numpy(rd).to(DEV)

    def run(kind, steps, stray=False):
        net, _ = golden_net(train=True)
        net_gt, _ = golden_net(train=False)
        opt = torch.optim.Adam(net.parameters(), lr=5e-3) if kind == "torch" else Adam(net.parameters(), lr=5e-3, zero_grad_in_step=kind == "in_step")
        flat = flat_grad_view(net.parameters())
        guide = SyntheticGuidance(11)
        torch.manual_seed(5)
        for _ in range(steps):
            sds_step(net, net_gt, ro, rd, (64, 64), opt, guide, batch_size=4096, flat_grad=flat)
            if stray:
                # something accumulates into the gradients between two steps (a manual backward, a wrapped optimizer, ...): `grads_cleared` is a checked
                # claim (version counters), so the next sds_step clears the buffer itself instead of adding this to its gradient
                assert opt.grads_cleared
                net.sdf_net[0].bias.grad.add_(1.0)
                assert not opt.grads_cleared
        net.check_finite()
        if kind == "in_step" and not stray:
            assert opt.grads_cleared and float(flat.abs().max()) == 0.0
        return {k: v.detach().clone() for k, v in net.named_parameters()}

    a, b = run("in_step", 1), run("torch", 1)
    for k in a:
        assert float((a[k] - b[k]).abs().max()) <= 2e-8 + 1e-6 * float(b[k].abs().max()), (k, float((a[k] - b[k]).abs().max()))
    a, b = run("in_step", 3), run("explicit", 3)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    c = run("in_step", 3, stray=True)
    for k in a:
        assert torch.equal(a[k], c[k]), k


@pytest.mark.parametrize("bkg", ["white", "noise"])
def test_fine_view_whole_view_renders_equal_patch_by_patch(bkg):
    """a fine-stage view (several patches) renders render_val and the frozen avatar ONCE per view (NeRFRenderer.render_view_nograd)
    instead of once per patch.  Rays are independent and the random draws are made in the harness's own order, so: render_val and the frozen weight_sum
    bit for bit, hence the guidance input, the flat gradient and the parameters after the step bit for bit -- against the patch-by-patch step (round 4) from
    the same state and streams.  A random background is drawn per patch from the host generator in the reference's order; the frozen avatar's render is
    hoisted all the same (round 6: its background never reaches weight_sum, so the launch gets a constant one and the draw is made -- and dropped -- where the
    reference's render would make it): the host generator ends the step in the same state as the patch-by-patch step."""
    import avatarcraft_amd.stylize as ST
    from avatarcraft_amd.render_utils import WHITE_BKG, NOISE_BKG, render_instantnsr_naive, _background_on, NSR_BOUND
    ro, rd = make_rays(48, 40, dist=1.8, f=36.0)                      # 1920 rays: 4 patches of 512 (the last one shorter: 384)
    ro_t, rd_t = torch.from_numpy(ro).to(DEV), torch.from_numpy(rd).to(DEV)
    key = WHITE_BKG if bkg == "white" else NOISE_BKG

    class Rec(ST.SyntheticGuidance):
        def __call__(self, rgb, text=None):
            self.seen = rgb.detach().clone()
            return super().__call__(rgb, text)

    def one(whole):
        net, _ = golden_net(train=True)
        net_gt, _ = golden_net(train=False)
        opt = ST.Adam(net.parameters(), lr=5e-3, zero_grad_in_step=False)
        flat = ST.flat_grad_view(net.parameters())
        guide = Rec(3)
        torch.manual_seed(21); random.seed(21)
        marks = []
        prev = ST.WHOLE_VIEW_RENDERS
        ST.WHOLE_VIEW_RENDERS = whole
        try:
            st = ST.sds_step(net, net_gt, ro_t, rd_t, (48, 40), opt, guide, batch_size=512, flat_grad=flat, bkg_key=key, timers=marks)
        finally:
            ST.WHOLE_VIEW_RENDERS = prev
        net.check_finite()
        return guide.seen, flat.clone(), {k: v.detach().clone() for k, v in net.named_parameters()}, [n for n, _ in marks], (st, torch.get_rng_state().clone())
    img_a, g_a, p_a, m_a, (s_a, rng_a) = one(True)
    img_b, g_b, p_b, m_b, (s_b, rng_b) = one(False)
    assert "render_gt_view" in m_a and "render_gt_view" not in m_b
    assert torch.equal(rng_a, rng_b)                                  # the host generator made the same draws
    assert m_a.count("backward") == m_b.count("backward") == 4
    assert torch.equal(img_a, img_b)                                  # render_val of the view, bit for bit
    assert torch.equal(g_a, g_b) and float(g_a.abs().max()) > 0       # the accumulated gradient of the four patches
    for k in p_a:
        assert torch.equal(p_a[k], p_b[k]), k
    assert float(s_a["opacity"]) == float(s_b["opacity"]) and float(s_a["eikonal"]) == float(s_b["eikonal"])
    # the one-launch view render against the harness on an eval net as well (no noise at all)
    net, _ = golden_net(train=False)
    with torch.no_grad():
        rgb_v, ws_v = net.render_view_nograd(ro_t, rd_t, 64, 64, NSR_BOUND, lambda n: _background_on(DEV, (n, 3), WHITE_BKG), 512)
        rgb_h, _, ex = render_instantnsr_naive(net, ro_t, rd_t, rays_per_batch=512, requires_grad=False, bkg_key=WHITE_BKG, render_can=True, perturb=True,
                                               return_raw=True, num_steps=64, upsample_steps=64, bound=NSR_BOUND)
    assert torch.equal(rgb_v, rgb_h) and torch.equal(ws_v, ex["weight_sum"])


def test_a_cuda_ray_style_net_trains_through_its_own_render():
    """a cuda_ray style net's render() is run_cuda's occupancy march; the fixed-step fused launches (pair, whole-view render_val,
    backward_last) would render DIFFERENT samples for it.  manual_backward_supported() is False for such a net, so sds_step takes the harness path
    (render_instantnsr_naive -> net.render -> run_cuda, autograd): render_val in one step == what render_instantnsr_naive gives for the same net and
    draws, and a step moves the parameters and keeps them finite."""
    from avatarcraft_amd.instant_nsr import NeRFNetwork
    from avatarcraft_amd.render_utils import render_instantnsr_naive, WHITE_BKG
    from avatarcraft_amd.stylize import sds_step, flat_grad_view, SyntheticGuidance
    from avatarcraft_amd.synthetic import make_rays
    src, _ = golden_net(train=True)
    net = NeRFNetwork(cuda_ray=True)
    net.load_state_dict(src.state_dict(), strict=False)
    net = net.to(DEV).train()
    assert not net.manual_backward_supported() and src.manual_backward_supported()
    net.update_extra_state(1.6)
    net_gt, _ = golden_net(train=False)
    ro, rd = make_rays(32, 32, dist=1.8, f=25.0)
    ro, rd = torch.from_numpy(ro).to(DEV), torch.from_numpy(rd).to(DEV)
    seen = {}

    class Guide(SyntheticGuidance):
        def __call__(self, rgb, text=None):
            seen["img"] = rgb.detach().clone()
            return super().__call__(rgb, text)

    opt = torch.optim.Adam(net.parameters(), lr=5e-3)
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    ls = net.local_step
    torch.manual_seed(3)
    sds_step(net, net_gt, ro, rd, (32, 32), opt, Guide(11), batch_size=4096, flat_grad=None)
    net.check_finite()
    after = dict(net.named_parameters())
    assert all(bool(torch.isfinite(v).all()) for v in after.values())
    assert any(not torch.equal(before[k], after[k].detach()) for k in before)
    # render_val of that step == the harness on the same net state and draws (parameters restored, RNG re-seeded, step counter rewound)
    with torch.no_grad():
        for k, v in net.named_parameters():
            v.copy_(before[k])
    net.invalidate_caches()
    net.local_step = ls
    torch.manual_seed(3)
    rgb_val, _ = render_instantnsr_naive(net, ro, rd, rays_per_batch=4096, requires_grad=False, bkg_key=WHITE_BKG, render_can=True, perturb=True,
                                         num_steps=64, upsample_steps=64, bound=1.6)
    img = rgb_val.reshape(32, 32, 3).permute(2, 0, 1).unsqueeze(0)
    assert torch.equal(img, seen["img"])


@pytest.mark.parametrize("bkg", ["white", "noise"])
def test_fine_view_whole_view_backward_equals_patch_by_patch(bkg):
    """round 6: the training render of a view of several patches as ONE launch and its backward as ONE ac_render_core_backward over all
    patches (NeRFNetwork.render_view_train, stylize.WHOLE_VIEW_BACKWARD) against the patch-by-patch step from the same state and random streams.
    Forward: the image handed to the guidance and both loss values bit for bit / to a last-bit mean (same draws in the same order, the eikonal term a ratio per
    patch).  Backward: the accumulated gradient of every parameter within 2e-6 of its largest entry -- NOT bit for bit, by construction: the table gradient is
    summed per bucket in a fixed-point scale chosen from the launch's record count and largest value and rounded to fp32 once instead of once per patch; the MLP
    gradients join one set of per-wave partial sums instead of four."""
    import avatarcraft_amd.stylize as ST
    from avatarcraft_amd.render_utils import WHITE_BKG, NOISE_BKG
    ro, rd = make_rays(64, 32, dist=1.8, f=40.0)                      # 2048 rays: 4 patches of 512
    ro_t, rd_t = torch.from_numpy(ro).to(DEV), torch.from_numpy(rd).to(DEV)
    key = WHITE_BKG if bkg == "white" else NOISE_BKG

    class Rec(ST.SyntheticGuidance):
        def __call__(self, rgb, text=None):
            self.seen = rgb.detach().clone()
            return super().__call__(rgb, text)

    def one(whole):
        net, _ = golden_net(train=True)
        net_gt, _ = golden_net(train=False)
        opt = torch.optim.SGD(net.parameters(), lr=0.0)
        flat = ST.flat_grad_view(net.parameters())
        guide = Rec(3)
        torch.manual_seed(21); random.seed(21)
        marks = []
        prev = ST.WHOLE_VIEW_BACKWARD
        ST.WHOLE_VIEW_BACKWARD = whole
        try:
            st = ST.sds_step(net, net_gt, ro_t, rd_t, (64, 32), opt, guide, batch_size=512, flat_grad=flat, bkg_key=key, timers=marks)
        finally:
            ST.WHOLE_VIEW_BACKWARD = prev
        net.check_finite()
        return guide.seen, {k: v.grad.detach().clone() for k, v in net.named_parameters()}, [n for n, _ in marks], st
    img_a, g_a, m_a, s_a = one(True)
    img_b, g_b, m_b, s_b = one(False)
    assert m_a.count("backward") == 1 and m_b.count("backward") == 4 and m_a.count("render_grad_forward") == 1
    assert torch.equal(img_a, img_b)
    assert abs(float(s_a["opacity"]) - float(s_b["opacity"])) <= 1e-6 * abs(float(s_b["opacity"]))
    assert abs(float(s_a["eikonal"]) - float(s_b["eikonal"])) <= 1e-6 * abs(float(s_b["eikonal"]))
    worst = {}
    for k in g_a:
        scale = float(g_b[k].abs().max())
        assert scale > 0, k
        worst[k] = float((g_a[k] - g_b[k]).abs().max()) / scale
        assert worst[k] <= 5e-6, (k, worst)                          # observed <= 2.4e-6
    print("whole-view backward vs patch by patch, worst |d grad| / max:", max(worst.values()))
