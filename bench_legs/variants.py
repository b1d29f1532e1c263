"""bench_legs.variants -- use_viewdirs=True, the geometry of the learned surface (mesh export, density grid), the occupancy-grid render (cuda_ray=True)."""
import time

import numpy as np
import torch

from bench_legs.common import NUM_STEPS, RAYS_PER_BATCH, UPSAMPLE_STEPS, _grid_roofline, make_net, sds_view

def time_viewdirs(dev, p, table, ro_t, rd_t, steps=8):
    """NeRFNetwork(use_viewdirs=True) (models/instant_nsr.py:565-569, 644-653: colour layer 1 reads cat[x, sh(d), n, geo_feat]) through the same fused paths as
    the default model: the 16 spherical harmonics of the ray direction are folded into a per-ray bias of colour layer 1 in the renderer's prologue, so the
    headline launch and the SDS step should cost what they cost without view directions (round 4: 2.1x / 2.2x through the generic path)."""
    from avatarcraft_amd import nsr_ops
    from avatarcraft_amd.instant_nsr import NeRFNetwork
    from avatarcraft_amd.stylize import sds_step, SyntheticGuidance, flat_grad_view, Adam

    def make(train):
        torch.manual_seed(0)
        net = NeRFNetwork(use_viewdirs=True)
        sd = {k: torch.from_numpy(np.asarray(p[k])) for k in p if k.startswith(("sdf_net", "color_net.1", "color_net.2", "deviation_net"))}
        sd["encoder.embeddings"] = torch.from_numpy(table); sd["encoder.offsets"] = torch.from_numpy(np.asarray(p["offsets"]))
        net.load_state_dict(sd, strict=False)                    # (color_net.0 keeps its own [64,37] initialisation)
        return net.to(dev).train(train)
    net = make(False)
    with torch.no_grad():
        f, inv_s = net._field(), net.forward_variance()
        out = {}
        # (two whole passes over the view's 16 batches after one of warm-up: a batch's launch takes 0.68 - 0.81 ms depending on what its rays see, so a ratio
        # against the headline -- an average over whole views -- needs whole views on this side as well)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(32)]
        for k in range(48):
            b = k % 16
            sl = slice(b * RAYS_PER_BATCH, (b + 1) * RAYS_PER_BATCH)
            nsr_ops.render_rays(f, ro_t[sl], rd_t[sl], NUM_STEPS, UPSAMPLE_STEPS, 1.6, inv_s, out=out, events=evs[k - 16] if k >= 16 else None)
        torch.cuda.synchronize()
        k_ms = float(np.mean([a_.elapsed_time(b_) for a_, b_ in evs]))
    net, net_gt = make(True), make(False)
    opt = Adam(net.parameters(), lr=5e-3, zero_grad_in_step=True)
    flat = flat_grad_view(net.parameters())
    guide = SyntheticGuidance(42)
    so, sd_ = sds_view(0)
    so, sd_ = torch.from_numpy(so).to(dev), torch.from_numpy(sd_).to(dev)
    for _ in range(2):
        sds_step(net, net_gt, so, sd_, (64, 64), opt, guide, batch_size=4096, flat_grad=flat)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sds_step(net, net_gt, so, sd_, (64, 64), opt, guide, batch_size=4096, flat_grad=flat)
    torch.cuda.synchronize()
    return {"render_kernel_ms_per_4096_rays": k_ms, "rays_per_s": RAYS_PER_BATCH / (k_ms * 1e-3), "sds_step_ms": (time.perf_counter() - t0) / steps * 1e3,
            "note": "use_viewdirs=True: sh(d) of degree 4 folded into a per-ray bias of colour layer 1 (ac_field.Wc1_sh); compare with the headline's kernel_ms and sds_step.ms_per_step"}


def time_geometry(dev, p, table, reps=3):
    """SURVEY 8(f) rank 3 at the reference's own sizes: the mesh export -- extract_geometry(NSR_BOUND, 512) (stylize.py:267: 512^3 = 134 M forward_sdf
    queries + marching cubes) -- and the density-grid update of update_extra_state (129^3 queries -> density -> max pool -> merge -> mean), both on the
    device (csrc/geometry.hip).  Per-launch times by HIP events on the launch stream; gather-request roofline like the headline's (1024 B per query)."""
    from avatarcraft_amd import nsr_ops
    from avatarcraft_amd.render_utils import NSR_BOUND
    net = make_net(p, table, dev, False, cuda_ray=True)
    res_ = 512
    ev = lambda: torch.cuda.Event(enable_timing=True)
    out = {}
    with torch.no_grad():
        f = net._field()
        ax = net._grid_axis(NSR_BOUND, res_)
        vol = torch.empty((res_,) * 3, dtype=torch.float32, device=dev)
        nsr_ops.field_sdf_grid(f, ax, ax, ax, NSR_BOUND, negate=True, out=vol)                   # warm-up
        t_sdf, t_mc = [], []
        for _ in range(reps):
            e0, e1, e2 = ev(), ev(), ev()
            e0.record()
            nsr_ops.field_sdf_grid(f, ax, ax, ax, NSR_BOUND, negate=True, out=vol)
            e1.record()
            v, t = nsr_ops.marching_cubes(vol, 0.0, den=res_ - 1.0, span=[3.2] * 3, lo=[-1.6] * 3)
            e2.record(); torch.cuda.synchronize()
            t_sdf.append(e0.elapsed_time(e1)); t_mc.append(e1.elapsed_time(e2))
        t0 = time.perf_counter(); vh, th = v.cpu().numpy(), t.cpu().numpy(); t_copy = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter(); verts, tris = net.extract_geometry(NSR_BOUND, res_); t_e2e = (time.perf_counter() - t0) * 1e3
        sdf_ms, mc_ms = float(np.median(t_sdf)), float(np.median(t_mc))
        evals = res_ ** 3
        gb = evals * 1024 / 1e9
        out["mesh_export_512"] = {
            "ms": sdf_ms + mc_ms, "sdf_grid_ms": sdf_ms, "marching_cubes_ms": mc_ms, "mesh_to_host_ms": t_copy, "extract_geometry_call_ms": t_e2e,
            "field_evaluations": evals, "vertices": int(v.shape[0]), "triangles": int(t.shape[0]),
            "roofline": _grid_roofline("field_sdf_grid_kernel", evals, sdf_ms),
            "note": "reference: extract_geometry(NSR_BOUND, 512) of stylize.py:267 (512^3 forward_sdf queries in 256^3 blocks assembled on the host + PyMCubes "
                    "on the CPU); here one ac_field_sdf_grid launch + ac_marching_cubes_count / _emit (classify, scan, emit; one 8-byte read-back between "
                    "them), the volume never leaves the device; marching_cubes_ms includes that read-back and the allocation of the scratch"}
        del vol, v, t
        # density grid: the reference's call, update_extra_state(bound) once per epoch
        ts = []
        for _ in range(reps + 1):
            e0, e1 = ev(), ev()
            e0.record(); mean = nsr_ops.density_grid_update(f, net._grid_axis(NSR_BOUND, 129), net.density_grid, NSR_BOUND, 512.0, 0.95); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t0 = time.perf_counter(); net.update_extra_state(NSR_BOUND); t_call = (time.perf_counter() - t0) * 1e3
        net.fused_density_grid = False
        net.update_extra_state(NSR_BOUND); torch.cuda.synchronize()
        t0 = time.perf_counter(); net.update_extra_state(NSR_BOUND); torch.cuda.synchronize(); t_torch = (time.perf_counter() - t0) * 1e3
        net.fused_density_grid = True
        k_ms = float(np.median(ts[1:]))
        halo_evals = 137 * 145 * 145               # 129^3 grid points + the +1 halo of every 16 x 8 x 8 brick that lies inside the grid
        out["density_grid_update"] = {
            "ms": k_ms, "update_extra_state_call_ms": t_call, "torch_chain_call_ms": t_torch, "grid": [129] * 3, "field_evaluations": halo_evals,
            "roofline": _grid_roofline("density_grid_kernel", 129 ** 3, k_ms),
            "note": "one launch: SDF -> logistic density -> 2^3 max pool -> max(grid * decay, new) in place -> mean; update_extra_state_call_ms = the whole "
                    "method with its one read-back of (mean, step counts); torch_chain_call_ms = the reference-shaped torch formulation on the same fused SDF "
                    "query (round 4's path)"}
    return out


def time_occupancy_render(dev, p, table, ro, rd, reps=3):
    """A SEPARATE figure, not the headline and not run()'s result: the occupancy-grid render (render(cuda_ray=True) -> NeRFRenderer.run_cuda, the path
    models/instant_nsr.py:358-363 dispatches to and the reference never defines): density grid (update_extra_state) -> march -> fused per-sample field
    (ac_field_samples) -> packed compositor, on the same 256 x 256 view.  The grid is built for a sharp variance (inv_s = 512 hard-coded at :325), so
    this leg sets forward_variance() = 512 and reports how far its pixels are from run()'s at that variance (two quadratures of one integral)."""
    from avatarcraft_amd.render_utils import NSR_BOUND
    net = make_net(p, table, dev, False, cuda_ray=True)
    with torch.no_grad():
        net.deviation_net.variance.fill_(float(np.log(512.0) / 10.0))
    t0 = time.perf_counter(); net.update_extra_state(NSR_BOUND); torch.cuda.synchronize(); t_grid = time.perf_counter() - t0
    kw = dict(num_steps=64, bound=NSR_BOUND, upsample_steps=64, bg_color=None, cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0)
    n = ro.shape[0]

    def view(rpb):
        rounds = 0
        outs = []
        for i in range(0, n, rpb):
            outs.append(net.render(ro[None, i:i + rpb], rd[None, i:i + rpb], **kw)["rgb"][0])
            rounds += net._last_cuda_rounds
        return torch.cat(outs), rounds

    res = {}
    with torch.no_grad():
        # the one-launch form (ac_render_rays_occupancy: march + field + composite per ray; run_cuda's default in eval()) and the reference-shaped loop of
        # compact / march / field / composite rounds with one host read-back each (the same pixels bit for bit)
        for mode, rounds_on in (("one_launch", False), ("rounds", True)):
            net.occupancy_rounds = rounds_on
            for rpb in (RAYS_PER_BATCH, n):
                img, rounds = view(rpb); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps * (4 if not rounds_on else 1)):
                    img, rounds = view(rpb)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / (reps * (4 if not rounds_on else 1))
                res[f"eval_{mode}_{rpb}_ray_batches"] = {"ms_per_view": dt * 1e3, "rays_per_s": n / dt, "march_rounds_per_view": rounds}
        net.occupancy_rounds = False
        # the two one-launch kernels side by side on the whole view (run_cuda picks the phased one from 2048 rays on: same pixels)
        from avatarcraft_amd import nsr_ops as _o
        fa = (net._field(), ro, rd, net.density_grid, net.mean_density, NSR_BOUND, 0.005, net.forward_variance(), 1.0)
        for nm, ph in (("phases_rounds_inside_the_launch", True), ("one_wave_per_ray_group", False)):
            a_img = _o.render_rays_occupancy(*fa, phased=ph)["image"]; torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(12):
                _o.render_rays_occupancy(*fa, phased=ph)
            torch.cuda.synchronize()
            res.setdefault("eval_kernels_65536_rays", {})[nm] = {"ms_per_view": (time.perf_counter() - t0) / 12 * 1e3}
            res["eval_kernels_65536_rays"].setdefault("_img", []).append(a_img)
        _imgs = res["eval_kernels_65536_rays"].pop("_img")
        res["eval_kernels_65536_rays"]["pixels_identical"] = bool(torch.equal(_imgs[0], _imgs[1]))
        # what a driver gets: the harness (render_instantnsr_naive, rays_per_batch = 4096 like render_canonical.py) hands an eval() occupancy net the whole view
        from avatarcraft_amd.render_utils import render_instantnsr_naive as _harness, WHITE_BKG as _W
        hk = dict(rays_per_batch=RAYS_PER_BATCH, requires_grad=False, bkg_key=_W, render_can=True, perturb=False, return_raw=True, num_steps=64, upsample_steps=64,
                  bound=NSR_BOUND)
        himg = _harness(net, ro, rd, **hk)[0]; torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps * 4):
            himg = _harness(net, ro, rd, **hk)[0]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (reps * 4)
        res["eval_through_the_harness_4096_ray_batches"] = {"ms_per_view": dt * 1e3, "rays_per_s": n / dt,
                                                            "note": "render_instantnsr_naive(rays_per_batch=4096): one launch per view for an eval() occupancy net (same pixels)"}
        from avatarcraft_amd import nsr_ops as _ops
        res["samples_evaluated_per_view"] = int(_ops.render_rays_occupancy(net._field(), ro, rd, net.density_grid, net.mean_density, NSR_BOUND, 0.005,
                                                                           net.forward_variance(), 1.0, count_samples=True)["n_samples"].item())
        net.cuda_ray = False
        ref = torch.cat([net.render(ro[None, i:i + RAYS_PER_BATCH], rd[None, i:i + RAYS_PER_BATCH], **kw)["rgb"][0] for i in range(0, n, RAYS_PER_BATCH)])
        net.cuda_ray = True
        res["max_abs_rgb_diff_vs_run_at_inv_s_512"] = float((img - ref).abs().max())
        res["mean_abs_rgb_diff_vs_run_at_inv_s_512"] = float((img - ref).abs().mean())
        # training form (march_rays_train with the per-epoch sample budget: no host synchronisation), no-grad: march + field + two composites per batch
        net.train()
        so, sd_ = sds_view(0)                                    # the 4096-ray training view of the SDS step (every ray aimed at the body)
        ro, rd = torch.from_numpy(so).to(dev), torch.from_numpy(sd_).to(dev)
        b0 = slice(0, RAYS_PER_BATCH)
        net.render(ro[None, b0], rd[None, b0], perturb=True, **kw)
        samples = int(net.step_counter[0, 0].item())
        net.mean_count = samples + 4096
        def train_form(one_launch):
            net.occupancy_train_one_launch = one_launch
            try:
                for _ in range(3):
                    o = net.render(ro[None, b0], rd[None, b0], perturb=True, **kw)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(40):
                    net.render(ro[None, b0], rd[None, b0], perturb=True, **kw)
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / 40, o["rgb"]
            finally:
                net.occupancy_train_one_launch = True
        dt_chain, img_chain = train_form(False)
        dt, img_one = train_form(True)
    # ... and UNDER AUTOGRAD (forward + backward of the same batch): the fused SDF-query / colour operators + the packed compositor,
    # with the shading glue between them as one launch each way (nsr_ops.packed_shading, round 6) and as the torch formulation it replaced
    def train_autograd(fused):
        net.occupancy_fused_shading = fused
        try:
            def step():
                net.zero_grad(set_to_none=True)
                o = net.render(ro[None, b0], rd[None, b0], perturb=True, **kw)
                (o["rgb"].sum() + o["weight_sum"].sum() + 0.1 * o["gradient_error"]).backward()
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                step()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / 20
        finally:
            net.occupancy_fused_shading = True
            net.zero_grad(set_to_none=True)
    dt_ag_torch, dt_ag = train_autograd(False), train_autograd(True)
    with torch.no_grad():                                    # (bookkeeping only)
        res["train_form_4096_ray_batch"] = {"ms_per_batch": dt * 1e3, "rays_per_s": RAYS_PER_BATCH / dt, "samples_per_ray": samples / RAYS_PER_BATCH,
                                            "bytes_per_sample_gathered": 7 * 1024, "gather_gbs": samples * 7 * 1024 / dt / 1e9,
                                            "chain_of_operators_ms_per_batch": dt_chain * 1e3, "pixels_identical_to_the_chain": bool(torch.equal(img_one, img_chain)),
                                            "under_autograd_forward_plus_backward_ms": dt_ag * 1e3, "under_autograd_with_torch_shading_glue_ms": dt_ag_torch * 1e3,
                                            "note": "net.train() under no_grad (stylize.py's render_val of a cuda_ray net): ONE launch (ac_render_rays_occupancy_train: "
                                                    "count, grid barrier, march + field + both composites + eikonal term + background; grid look-ups 8 at a time) "
                                                    "against the chain it replaces (march_rays_train, ac_field_samples, composite_rays_train x 2, torch)"}
    res["density_grid_update_ms"] = t_grid * 1e3
    res["note"] = ("occupancy-grid path (cuda_ray=True): a separate renderer from the headline's run(); the reference ships its operators but no caller "
                   "(run_cuda is undefined there), so there is no reference number for it")
    return res
