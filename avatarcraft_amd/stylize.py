"""One SDS stylisation step (and its data-parallel form): counterpart of Trainer.train's inner loop,
stylize.py:95-199, for one view:

    (A) render_val : no-grad full-view render of net_style (train mode => jittered samples)      stylize.py:115 -> :298
    (B) guidance   : image -> d(loss)/d(image); Stable-Diffusion SDS in the reference              :128-130
    (C) per 4096-ray patch: render net_style with grad; backward(image_grad); backward(w_eik * eikonal);
        render frozen net_gt; backward(1e5 * smooth_l1(clamp(opacity_pred), clamp(opacity_gt)))  :143-196
    (D) optimizer.step()  (Adam lr 5e-3 on all parameters of net_style)                            :199, :355-363

The SD UNet is an opaque `guidance(rgb[1,3,h,w]) -> grad[1,3,h,w]` callable (it stays PyTorch-ROCm; `diffusers` is not
available offline, so `SyntheticGuidance` -- clamp(N(0,1), -1, 1), the statistics of the clamped SDS gradient
(models/diffusion.py:139-146) -- stands in for measurement and tests).

Data parallel (BASELINE config 5): one view per rank, parameters replicated, ONE all-reduce (sum,
averaged inside the collective on RCCL) of the flat fp32 gradient (12 248 902 elements = 49 MB, plus one guard word) before the optimizer step.
"""
import os

import torch
from . import nsr_ops
import torch.nn.functional as F

from .render_utils import render_instantnsr_naive, NSR_BOUND, WHITE_BKG, BLACK_BKG, NOISE_BKG, _background_on, select_background


class SyntheticGuidance:
    """Stand-in for StableDiffusion.mannual_backward: a clamped Gaussian image gradient from a per-call seeded stream."""

    private_rng = True          # draws from its own generator: sds_step may render the training forward before calling it (see sds_step)

    def __init__(self, seed=42):
        self.gen = None
        self.seed = seed

    def __call__(self, rgb, text=None):
        if self.gen is None:
            self.gen = torch.Generator(device=rgb.device); self.gen.manual_seed(self.seed)
        return torch.randn(rgb.shape, generator=self.gen, device=rgb.device, dtype=rgb.dtype).clamp_(-1.0, 1.0)


def flat_grad_view(params):
    """Allocate ONE contiguous fp32 buffer and point every p.grad into it (so that the all-reduce is a single
    collective over 49 MB instead of 15 small ones).  Returns the flat buffer [n_params].
    The allocation is one word longer than what is returned: the GUARD WORD behind the gradients (`flat.ac_guard`, a [n_params + 1] view of the same
    memory) rides in the same collective and carries the step's "a render produced NaN / Inf" flag to every rank (see _reduce_gradients)."""
    params = [p for p in params if p.requires_grad]
    n = sum(p.numel() for p in params)
    full = torch.zeros(n + 1, dtype=torch.float32, device=params[0].device)
    flat = full[:n]
    flat.ac_guard = full
    off = 0
    for p in params:
        p.grad = flat[off:off + p.numel()].view_as(p)
        off += p.numel()
    return flat


class Adam(torch.optim.Adam):
    """torch.optim.Adam -- the reference's optimizer (stylize.py:355-363) -- whose step() is ONE HIP launch over every parameter tensor (ac_adam_step,
    csrc/step_glue.hip) instead of torch's kernel (0.135 ms for the 12.2 M parameters: 2.5 TB/s).  Same hyper-parameters, same update arithmetic
    (torch/optim/adam.py, _single_tensor_adam: no amsgrad / weight decay / maximize -- the reference uses none), same state layout ('step', 'exp_avg',
    'exp_avg_sq' per parameter), so state_dict()s are interchangeable with torch.optim.Adam's.

    zero_grad_in_step: the launch also clears the gradients it has consumed -- the next step's optimizer.zero_grad() (stylize.py:143) for free while the
    values are in registers; `grads_cleared` then tells sds_step that its own clearing of the flat gradient is redundant.

    `grads_cleared` is a CHECKED claim, not a sticky flag: step() / zero_grad() record the version counters of the gradient tensors they have just cleared
    and the property is true only while every counter still has that value.  Anything that writes a gradient in between -- loss.backward(), an in-place
    op on .grad or on the flat buffer it is a view of, NeRFRenderer.backward_last (which bumps the counters of what its kernels wrote) -- makes it false,
    and sds_step then clears the buffer itself (when in doubt, zero)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, zero_grad_in_step=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps)
        self.zero_grad_in_step = bool(zero_grad_in_step)
        self._cleared_token = None
        self._entries = {}

    def _grad_token(self):
        return tuple((id(p.grad), p.grad.data_ptr(), p.grad._version) for g in self.param_groups for p in g["params"] if p.grad is not None)

    @property
    def grads_cleared(self):
        return self._cleared_token is not None and self._cleared_token == self._grad_token()

    @grads_cleared.setter
    def grads_cleared(self, value):
        self._cleared_token = self._grad_token() if value else None

    def zero_grad(self, set_to_none=False):
        # the gradients live in one flat buffer other code holds views of (flat_grad_view): they are cleared in place, never dropped
        super().zero_grad(set_to_none=False)
        self.grads_cleared = True

    def __getstate__(self):
        d = super().__getstate__()                       # (torch's keeps defaults / state / param_groups only)
        d["zero_grad_in_step"] = self.zero_grad_in_step
        return d

    def __setstate__(self, state):
        super().__setstate__(state)
        self.__dict__.setdefault("zero_grad_in_step", False)
        self.__dict__.pop("grads_cleared", None)             # (a pickle of the round-4 class kept a plain attribute of that name)
        self._cleared_token = None
        self._entries = {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # a state saved by torch.optim.Adam(fused=True / capturable=True) keeps its step counts on the device: reading them would synchronise every step
        for st in self.state.values():
            if isinstance(st.get("step"), torch.Tensor) and st["step"].is_cuda:
                st["step"] = st["step"].detach().to("cpu", torch.float32)

    @torch.no_grad()
    def step(self, closure=None):
        """one launch per 16 tensors of a parameter group.  The group's step count is read from its first parameter (torch keeps one per parameter; they
        only differ when parameters are added to a running optimizer, which the reference never does)."""
        from . import _lib as L
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            if group.get("amsgrad") or group.get("maximize") or group.get("weight_decay", 0) != 0:
                raise NotImplementedError("avatarcraft_amd.stylize.Adam: amsgrad / maximize / weight_decay are not implemented (the reference uses none)")
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("avatarcraft_amd.stylize.Adam: contiguous float32 CUDA parameters and gradients only")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            t = int(self.state[ps[0]]["step"].item()) + 1
            for p in ps:
                self.state[p]["step"] += 1
            beta1, beta2 = group["betas"]
            step_size = float(group["lr"]) / (1.0 - beta1 ** t)
            bc2_sqrt = (1.0 - beta2 ** t) ** 0.5
            key = (gi,) + tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr()) for p in ps)
            arrs = self._entries.get(key)
            if arrs is None:
                self._entries = {k: v for k, v in self._entries.items() if k[0] != gi}          # (pointers changed: drop this group's old tables)
                arrs = []
                for i in range(0, len(ps), L.AC_ADAM_MAX_TENSORS):
                    chunk = ps[i:i + L.AC_ADAM_MAX_TENSORS]
                    arr = (L.ac_adam_entry * len(chunk))()
                    for e, p in zip(arr, chunk):
                        e.param, e.grad, e.n = p.data_ptr(), p.grad.data_ptr(), p.numel()
                        e.exp_avg, e.exp_avg_sq = self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr()
                    arrs.append(arr)
                self._entries[key] = arrs
            for arr in arrs:
                L.check(L.lib().ac_adam_step(arr, len(arr), step_size, beta1, 1.0 - beta1, beta2, 1.0 - beta2, float(group["eps"]), bc2_sqrt,
                                             int(self.zero_grad_in_step), L.current_stream(ps[0].device)), "adam_step")
            for p in ps:
                # the kernel wrote through raw pointers: tell autograd / every cache keyed on tensor versions (the network's effective weights and
                # prepared field, instant_nsr.py) that the parameters changed in place
                torch.autograd.graph.increment_version(p)
        self.grads_cleared = self.zero_grad_in_step
        return loss


_CONSTS = {}
# sds_step renders render_val and the training forward of a one-patch view in one launch (ac_render_rays_pair); False = two launches (same values)
PAIR_STEP_RENDERS = True
# a view of several patches (fine stage): render_val and the frozen avatar's render as one launch per VIEW instead of one per patch (same values)
WHOLE_VIEW_RENDERS = os.environ.get("AC_WHOLE_VIEW_RENDERS", "1") != "0"
# Round 6, built, measured, OFF by default: the TRAINING render and its backward of a view of several patches (the fine stage) as
# ONE forward launch + ONE ac_render_core_backward over all patches instead of one pair per patch (NeRFNetwork.render_view_train): the patches' gradients
# add up before the single optimizer.step() (stylize.py:199), the random draws are made patch by patch in the reference's order, the eikonal term stays a
# ratio per patch.  256 x 256 view, same box (bench.py sds_view_fine): training forward 14.65 -> 13.67 ms, backward 34.8 -> 36.5 ms, view 73.7 -> 74.4 ms.
# The 16 x smaller per-patch intermediates (470 MB of feature gradients, ~1 GB of scatter queues, 117 MB of saved stencil features) are written and read
# back within a patch's backward and largely stay in the 256 MB Infinity Cache; the whole view's 7.5 GB + 16 GB stream through HBM -- that costs more than
# the 15 saved fixed parts of bucket_accumulate and ~450 small launches return.  Scratch: ~75 GB.  AC_WHOLE_VIEW_BACKWARD=1 switches it on.
WHOLE_VIEW_BACKWARD = os.environ.get("AC_WHOLE_VIEW_BACKWARD", "0") == "1"
# data-parallel steps: all-reduce the table gradient of levels >= ALLREDUCE_SPLIT_LEVEL (33.5 of the 49 MB) while the scatter still accumulates the
# coarser levels and the MLP gradients are formed (ac_core_grads.side_stream / split_level).  Off by default: no multi-GPU node was available to measure it
# (at most the ~0.15 ms the second accumulation launch + ac_param_grads take can be hidden); AC_OVERLAP_ALLREDUCE=1 or sds_step(overlap_allreduce=True).
OVERLAP_GRAD_ALLREDUCE = os.environ.get("AC_OVERLAP_ALLREDUCE", "0") == "1"
ALLREDUCE_SPLIT_LEVEL = 8
_SIDE_STREAMS = {}


def _side_stream(device):
    k = str(device)
    if k not in _SIDE_STREAMS:
        _SIDE_STREAMS[k] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[k]


def _const_scalar(v, device):
    k = (float(v), str(device))
    if k not in _CONSTS:
        _CONSTS[k] = torch.tensor(float(v), dtype=torch.float32, device=device)
    return _CONSTS[k]


def sds_step(net_style, net_gt, rays_o, rays_d, hw, optimizer, guidance, batch_size=4096, w_eikonal=0.01, use_opacity=True,
             bkg_key=WHITE_BKG, flat_grad=None, process_group=None, num_steps=64, upsample_steps=64, timers=None, overlap_allreduce=None,
             grad_divisor=None):
    """rays_o, rays_d: [h*w, 3] of the (sub-sampled) training view; hw = (h, w).  Returns a dict of scalars.
    timers: a list that receives (phase name, torch.cuda.Event) marks on the current stream (bench.py's per-phase times).
    grad_divisor: the number of ranks that contribute a view to this step's all-reduce (default: the world size; smaller in the last round of an epoch
    whose view count is not a multiple of the world size, see shard_views / sds_idle_step)."""
    dist_on = process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized())
    # under a process group no render of this step judges its NaN flag on this rank alone: they all ride in the guard word of the gradient collective
    with _collective_verdict(dist_on and flat_grad is not None, net_style, net_gt) as verdict:
        return _sds_step(net_style, net_gt, rays_o, rays_d, hw, optimizer, guidance, batch_size, w_eikonal, use_opacity, bkg_key, flat_grad, process_group,
                         num_steps, upsample_steps, timers, overlap_allreduce, grad_divisor, verdict)


def _sds_step(net_style, net_gt, rays_o, rays_d, hw, optimizer, guidance, batch_size, w_eikonal, use_opacity, bkg_key, flat_grad, process_group, num_steps,
              upsample_steps, timers, overlap_allreduce, grad_divisor, verdict):
    h, w = hw
    n_rays = h * w

    def mark(name):
        if timers is not None:
            ev = torch.cuda.Event(enable_timing=True); ev.record(); timers.append((name, ev))
    mark("start")
    manual = rays_o.is_cuda and getattr(net_style, "manual_backward_supported", lambda: False)()
    # One patch covers the view (the coarse stage: 64 x 64 rays): render_val and the training render of the patch are the same rays with two
    # noise draws -- one launch renders both (NeRFNetwork.render_step_pair); the training render does not depend on the guidance, only its
    # backward does.  Larger views keep the reference's order (render_val of the whole view first, then patch by patch).
    # RNG order: the pair launch draws the training render's background and jitter BEFORE the guidance runs; the reference draws them after it
    # (stylize.py:128-152), and its Stable-Diffusion guidance takes its timestep and noise from the GLOBAL generator in between.  Pairing is therefore
    # used only with a guidance that declares `private_rng = True` (its draws do not touch the global streams: SyntheticGuidance); any other guidance --
    # SDSGuidance over the real networks -- gets the reference's order: render_val, guidance, training render (two launches, same values per launch).
    paired = None
    if (manual and PAIR_STEP_RENDERS and n_rays <= batch_size and net_style.training and hasattr(net_style, "render_step_pair")
            and getattr(guidance, "private_rng", False)):
        rgb_val, rgb_p, eik_p, ws_p = net_style.render_step_pair(rays_o, rays_d, num_steps, upsample_steps, NSR_BOUND,
                                                                 lambda: _background_on(rays_o.device, rays_o.shape, bkg_key))
        paired = (rgb_p, eik_p, {"weight_sum": ws_p})
        mark("render_val_and_grad_forward")
    elif manual and WHOLE_VIEW_RENDERS and n_rays > batch_size and hasattr(net_style, "render_view_nograd"):
        # (A) a view of several patches (the fine stage: 256 x 256 = 16 patches): render_val as ONE launch instead of one per patch -- the same
        # random draws in the same order (per patch: background, jitter noise), the same pixels bit for bit
        rgb_val, _ = net_style.render_view_nograd(rays_o, rays_d, num_steps, upsample_steps, NSR_BOUND,
                                                  lambda n: _background_on(rays_o.device, (n, 3), bkg_key), batch_size)
        mark("render_val")
    else:
        # (A) render_val: net_style stays in train mode (stylize.py never calls eval()), no grad
        rgb_val, _ = render_instantnsr_naive(net_style, rays_o, rays_d, rays_per_batch=batch_size, requires_grad=False, bkg_key=bkg_key,
                                             render_can=True, perturb=True, num_steps=num_steps, upsample_steps=upsample_steps, bound=NSR_BOUND)
        mark("render_val")
    img = rgb_val.reshape(h, w, 3).permute(2, 0, 1).unsqueeze(0)             # "(h w) c -> 1 c h w"
    # (B) gradient of the guidance loss w.r.t. the whole image
    grad_img = guidance(img.detach())
    grad_rays = grad_img.squeeze(0).permute(1, 2, 0).reshape(n_rays, 3).detach()
    mark("guidance")
    # (C) patch-wise backward
    if flat_grad is not None:
        if not getattr(optimizer, "grads_cleared", False):       # (stylize.Adam(zero_grad_in_step=True) has cleared them in its last step)
            flat_grad.zero_()
    else:
        optimizer.zero_grad()
    if hasattr(optimizer, "grads_cleared"):
        optimizer.grads_cleared = False
    bs = min(batch_size, n_rays)
    eik_vals, opa_vals, nan_flags = [], [], []
    dist_on = process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized())
    overlap = (OVERLAP_GRAD_ALLREDUCE if overlap_allreduce is None else bool(overlap_allreduce)) and dist_on and flat_grad is not None and manual
    if overlap and _avg_in_collective(process_group) and torch.distributed.get_world_size(process_group) > 1:
        early_op = torch.distributed.ReduceOp.AVG          # (every slice of one step uses the same operator)
    else:
        early_op = torch.distributed.ReduceOp.SUM
    work_hi = hi_range = None
    # the frozen avatar of the opacity loss (stylize.py:176-190), rendered for its weight_sum alone: for a view of several patches ONE launch before the
    # patch loop instead of one per patch.  net_gt is in eval mode (no jitter draw) and its background never reaches weight_sum: the launch gets a constant
    # one.  A RANDOM background (NOISE_BKG: a third of the views under augment_bkg) is still DRAWN per patch from the host generator where the reference's
    # render of net_gt would draw it (gt_bkg_draw below) and dropped -- the generator's stream, and with it every later draw of the run, stays the reference's.
    ws_gt_view = None
    gt_bkg_draw = (bkg_key % 4) == NOISE_BKG
    if (manual and WHOLE_VIEW_RENDERS and n_rays > bs and hasattr(net_gt, "render_view_nograd") and not net_gt.training
            and getattr(net_gt, "_fused_supported", lambda: False)() and not getattr(net_gt, "cuda_ray", False)):
        _, ws_gt_view = net_gt.render_view_nograd(rays_o, rays_d, num_steps, upsample_steps, NSR_BOUND,
                                                  lambda n: _background_on(rays_o.device, (n, 3), WHITE_BKG), bs, opacity_only=True)
        mark("render_gt_view")
    whole_bwd = (manual and WHOLE_VIEW_BACKWARD and paired is None and n_rays > bs and n_rays % bs == 0 and hasattr(net_style, "render_view_train")
                 and hasattr(net_gt, "render_view_nograd") and not net_gt.training and getattr(net_gt, "_fused_supported", lambda: False)()
                 and not getattr(net_gt, "cuda_ray", False))
    if whole_bwd:
        # one training launch + one backward for the whole view (see WHOLE_VIEW_BACKWARD).  Draw order per patch as in the loop below: the training render's
        # background, its jitter noise (device generator), the frozen avatar's background (host generator, after the training render's) -- kept by stashing
        # the frozen avatar's backgrounds when they are random (constant ones were rendered above already: ws_gt_view)
        P = n_rays // bs
        gt_bgs = []

        def draw(k, n):
            b = _background_on(rays_o.device, (n, 3), bkg_key)
            if ws_gt_view is None:
                gt_bgs.append(_background_on(rays_o.device, (n, 3), bkg_key))
            elif gt_bkg_draw:
                select_background((n, 3), bkg_key)                    # (drawn and dropped: see gt_bkg_draw)
            return b
        rgb, eik_p, ws_all = net_style.render_view_train(rays_o, rays_d, num_steps, upsample_steps, NSR_BOUND, draw, bs)
        mark("render_grad_forward")
        with torch.no_grad():
            if ws_gt_view is None:
                _, ws_gt_view = net_gt.render_view_nograd(rays_o, rays_d, num_steps, upsample_steps, NSR_BOUND, lambda n: gt_bgs.pop(0), bs, opacity_only=True)
            g_ws, opa = nsr_ops.sds_upstream(ws_all, ws_gt_view, 1e5 / bs, want_grad=use_opacity)          # (each patch's mean: 1e5 / its ray count)
            opa_vals.append(opa[0] / P)
            nan_flags.append(eik_p.sum())
            g_eik = None
            if w_eikonal > 0.0:
                g_eik = torch.full((P,), float(w_eikonal), dtype=torch.float32, device=rays_o.device)
                eik_vals.append((eik_p * w_eikonal).mean())
            mark("render_gt_and_losses")
            if overlap:
                emb = net_style.encoder.embeddings
                offs = net_style._offsets_host()
                base = (emb.grad.data_ptr() - flat_grad.data_ptr()) // 4
                assert 0 <= base and base + emb.grad.numel() <= flat_grad.numel(), "encoder.embeddings.grad must be a view of flat_grad"
                hi_range = (base + 2 * int(offs[ALLREDUCE_SPLIT_LEVEL]), base + 2 * int(offs[-1]))
                side = _side_stream(rays_o.device)
                net_style.backward_last(g_image=grad_rays, g_weights_sum=g_ws, g_eik=g_eik, split=(ALLREDUCE_SPLIT_LEVEL, side))
                with torch.cuda.stream(side):
                    work_hi = torch.distributed.all_reduce(flat_grad[hi_range[0]:hi_range[1]], op=early_op, group=process_group, async_op=True)
            else:
                net_style.backward_last(g_image=grad_rays, g_weights_sum=g_ws, g_eik=g_eik)
        mark("backward")
    for i in range(0, n_rays, bs) if (manual and not whole_bwd) else ():
        # The same three terms WITHOUT autograd (avatarcraft_amd.NeRFNetwork): the training render keeps its per-sample outputs, the upstream
        # gradients of (image, weights_sum, gradient_error) are written down directly (d sum(rgb * g) = g; d (eik * w) = w; the opacity term through
        # ac_sds_upstream) and go through ac_render_core_backward + ac_param_grads into .grad -- ~15 launches instead of ~80 per patch.
        ro, rd = rays_o[i:i + bs], rays_d[i:i + bs]
        if paired is not None:
            rgb, eik, extra = paired
        else:
            net_style._manual_backward = True
            try:
                rgb, eik, extra = render_instantnsr_naive(net_style, ro, rd, requires_grad=True, bkg_key=bkg_key, rays_per_batch=bs, perturb=1.0,
                                                          return_raw=True, render_can=True, bound=NSR_BOUND, num_steps=num_steps,
                                                          upsample_steps=upsample_steps)
            finally:
                net_style._manual_backward = False
            mark("render_grad_forward")
        with torch.no_grad():
            if ws_gt_view is not None:
                ws_gt = ws_gt_view[i:i + bs]
                if gt_bkg_draw:
                    select_background((ro.shape[0], 3), bkg_key)          # (the draw the reference's net_gt render makes here; host generator only)
            else:
                _, _, extra_gt = render_instantnsr_naive(net_gt, ro, rd, requires_grad=False, bkg_key=bkg_key, rays_per_batch=bs, perturb=True,
                                                         return_raw=True, render_can=True, num_steps=num_steps, upsample_steps=upsample_steps,
                                                         opacity_only=True)           # (only its weight_sum is read: no colour network)
                ws_gt = extra_gt["weight_sum"]
            g_ws, opa = nsr_ops.sds_upstream(extra["weight_sum"], ws_gt, 1e5 / ro.shape[0], want_grad=use_opacity)
            opa_vals.append(opa[0])
            if isinstance(eik, torch.Tensor):
                nan_flags.append(eik)
            g_eik = None
            if w_eikonal > 0.0:
                g_eik = _const_scalar(w_eikonal, ro.device)
                eik_vals.append(eik * g_eik)
            mark("render_gt_and_losses")
            if overlap and i + bs >= n_rays:
                # the last patch of the step: the finer levels' share of the table gradient is final before the backward has ended -- its all-reduce
                # starts on the side stream (ordered behind exactly that part of the scatter) while the main stream finishes the coarser levels
                emb = net_style.encoder.embeddings
                offs = net_style._offsets_host()
                base = (emb.grad.data_ptr() - flat_grad.data_ptr()) // 4
                assert 0 <= base and base + emb.grad.numel() <= flat_grad.numel(), "encoder.embeddings.grad must be a view of flat_grad"
                hi_range = (base + 2 * int(offs[ALLREDUCE_SPLIT_LEVEL]), base + 2 * int(offs[-1]))
                side = _side_stream(ro.device)
                net_style.backward_last(g_image=grad_rays[i:i + bs], g_weights_sum=g_ws, g_eik=g_eik, split=(ALLREDUCE_SPLIT_LEVEL, side))
                with torch.cuda.stream(side):
                    work_hi = torch.distributed.all_reduce(flat_grad[hi_range[0]:hi_range[1]], op=early_op, group=process_group, async_op=True)
            else:
                net_style.backward_last(g_image=grad_rays[i:i + bs], g_weights_sum=g_ws, g_eik=g_eik)
        mark("backward")
    for i in range(0, n_rays, bs) if not manual else ():
        ro, rd = rays_o[i:i + bs], rays_d[i:i + bs]
        rgb, eik, extra = render_instantnsr_naive(net_style, ro, rd, requires_grad=True, bkg_key=bkg_key, rays_per_batch=bs, perturb=1.0,
                                                  return_raw=True, render_can=True, bound=NSR_BOUND, num_steps=num_steps,
                                                  upsample_steps=upsample_steps)
        opacity_pred = extra["weight_sum"]
        if isinstance(eik, torch.Tensor):
            nan_flags.append(eik.detach())
        mark("render_grad_forward")
        # The reference back-propagates the three terms one after the other through the same retained graph
        # (stylize.py:163,169,193).  Gradients are linear in the loss, so ONE backward pass of their sum gives the same
        # parameter gradients with a third of the hash-table scatter traffic.
        total = (rgb * grad_rays[i:i + bs]).sum()
        if w_eikonal > 0.0:
            eik_loss = eik * w_eikonal
            eik_vals.append(eik_loss.detach())
            total = total + eik_loss
        with torch.no_grad():       # frozen reference avatar: its graph is never used (the reference detaches it, :187)
            _, _, extra_gt = render_instantnsr_naive(net_gt, ro, rd, requires_grad=False, bkg_key=bkg_key, rays_per_batch=bs, perturb=True,
                                                     return_raw=True, render_can=True, num_steps=num_steps, upsample_steps=upsample_steps)
        opacity_loss = F.smooth_l1_loss(opacity_pred.clamp(0.0, 1.0), extra_gt["weight_sum"].clamp(0.0, 1.0).detach()) * 1e5
        opa_vals.append(opacity_loss.detach())
        if use_opacity:
            total = total + opacity_loss
        mark("render_gt_and_losses")
        total.backward()
        mark("backward")
    # data parallel: one collective over the flat gradient (issued whenever a process group exists, world size 1 included: the call
    # path RCCL sees on an 8-GPU node is then the one every single-GPU run exercises)
    guard = None
    if dist_on:
        if flat_grad is None:
            if torch.distributed.get_world_size(process_group) > 1:
                raise RuntimeError("data-parallel sds_step needs flat_grad = flat_grad_view(net_style.parameters())")
        else:
            flag = verdict.flag(extra=nan_flags)                # every NaN flag of this step's renders (both nets, grad and no-grad) + the patches' eikonal terms
            guard = reduce_gradients(flat_grad, process_group, grad_divisor, nan_flag=flag, early=(work_hi, hi_range) if work_hi is not None else None)
            mark("grad_allreduce")
    # (D) -- a NaN recorded by this step's renders ON ANY RANK raises here on EVERY rank, before Adam's state and the weights are touched
    _check_finite(net_style, guard)
    optimizer.step()
    mark("optimizer")
    mean = lambda v: v[0].reshape(()) if len(v) == 1 else torch.stack([t.reshape(()) for t in v]).mean()      # (one patch: no reduction kernels)
    return {"eikonal": mean(eik_vals) if eik_vals else torch.zeros(()), "opacity": mean(opa_vals)}


def stylize_epochs(net_style, net_gt, optimizer, guidance, hw=(256, 256), n_cap=100, coarse_epochs=1, fine_epochs=0, subsample_scale=4,
                   center=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), camera_dist=2.0 * 0.9, augment_cam=False, stylize_head=False, coarse_head=0.0,
                   fine_head=0.0, head_offset=0.47 * 0.9, head_dist=0.5 * 0.9, augment_bkg=False, white_bkg=True, augment_text=False, tgt_text="",
                   batch_size=4096, w_eikonal=0.01, use_opacity=True, seed=42, device="cuda", flat_grad=None, on_step=None):
    """The outer loop of Trainer.train (stylize.py:46-215): per epoch a ring of training views (default_360_path, or style_360_path with
    camera jitter / head close-ups), visited in random order; per view the stride-sub-sampled rays of a 256x256 pinhole camera, a random
    background when augment_bkg, the view description prepended to the prompt when augment_text, and one sds_step.  The fine stage halves
    the stride (subsample_scale // 2, at least 1).  Under torch.distributed every rank draws the same poses and permutation (same seed)
    and takes the views rank, rank + world, ...; the gradients meet in sds_step's single all-reduce.
    guidance(rgb[1,3,h,w], text=...) -> d loss / d rgb.  on_step(global_step, epoch, stats) is called after every optimizer step.
    Returns the number of optimizer steps taken by this rank."""
    import random
    import numpy as np
    from .render_utils import (default_360_path, style_360_path, describe_view, pose2cap, cap2rays, sparse_ray_sampling, BLACK_BKG, NOISE_BKG)
    H, W = hw
    rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
    world = torch.distributed.get_world_size() if torch.distributed.is_available() and torch.distributed.is_initialized() else 1
    np.random.seed(seed); random.seed(seed)
    gen = torch.Generator(); gen.manual_seed(seed)
    center, up = np.asarray(center, dtype=np.float64), np.asarray(up, dtype=np.float64)
    step = 0
    for epoch in range(coarse_epochs + fine_epochs):
        coarse = epoch < coarse_epochs
        head_rate = coarse_head if coarse else fine_head
        if augment_cam or stylize_head:
            poses, desc = style_360_path(center, up, camera_dist, n_cap, add_noise=augment_cam, noise_scale=2.0 if augment_cam else 1.0,
                                         style_head=stylize_head, head_offset=head_offset, head_rate=head_rate, head_dist=head_dist)
        else:
            poses, angles = default_360_path(center, up, camera_dist, n_cap, add_noise=False)
            desc = describe_view(angles)
        perm = torch.randperm(len(poses), generator=gen).tolist()
        stride = subsample_scale if coarse else max(1, subsample_scale // 2)
        assert stride in (1, 2, 4, 8, 16), 'subsample scale must be 1, 2, 4, 8, or 16'
        for rnd, k in enumerate(shard_views(len(perm), rank, world)):
            # (the random draws below are made by every rank for every round, idle or not, so that the host-side streams stay in step across ranks)
            n_active = views_in_round(len(perm), world, rnd)
            bkg_key = random.randint(WHITE_BKG, NOISE_BKG) if augment_bkg else (WHITE_BKG if white_bkg else BLACK_BKG)
            if k is None:
                # the epoch's view count is not a multiple of the world size and this rank has no view in the last round (the reference's epoch is every
                # view once, stylize.py:76-78: nothing is dropped, nothing is visited twice): zero gradient into the same collective
                stats = sds_idle_step(net_style, optimizer, flat_grad, n_active)
            else:
                i = perm[k]
                text = f"{desc[i]} {tgt_text}" if augment_text else tgt_text
                ro, rd = cap2rays(pose2cap([H, W], poses[i]), device=device)
                ro, rd = sparse_ray_sampling(ro.reshape(H, W, 3), rd.reshape(H, W, 3), stride)
                h, w = ro.shape[0], ro.shape[1]
                g = guidance
                if _accepts_text(guidance):
                    g = lambda rgb, _t=text: guidance(rgb, text=_t)                       # noqa: E731
                    g.private_rng = getattr(guidance, "private_rng", False)
                short = n_active < world
                stats = sds_step(net_style, net_gt, ro.reshape(-1, 3).float().contiguous(), rd.reshape(-1, 3).float().contiguous(), (h, w), optimizer, g,
                                 batch_size=batch_size, w_eikonal=w_eikonal, use_opacity=use_opacity, bkg_key=bkg_key, flat_grad=flat_grad,
                                 grad_divisor=n_active if short else None, overlap_allreduce=False if short else None)
            if on_step is not None:
                on_step(step, epoch, stats)
            step += 1
    return step


def _accepts_text(guidance):
    import inspect
    try:
        return "text" in inspect.signature(guidance).parameters
    except (TypeError, ValueError):
        return False


def shard_views(n_views, rank, world):
    """Positions of one epoch's view permutation handled by `rank`: round-robin over ceil(n_views / world) ROUNDS, every rank the same number of
    rounds (the all-reduce of a step is a collective: all ranks take part in every round).  The reference's epoch visits every view exactly once
    (stylize.py:76-78); when n_views is not a multiple of world the LAST round has fewer views than ranks and the ranks without one get None --
    they join that round's collective with a zero gradient (sds_idle_step) and the average is taken over the views that exist."""
    rounds = -(-n_views // world)
    return [(rank + world * k if rank + world * k < n_views else None) for k in range(rounds)]


def views_in_round(n_views, world, k):
    """number of ranks that hold a view in round k of shard_views (the divisor of that round's gradient average)"""
    return max(0, min(world, n_views - world * k))


def sds_idle_step(net_style, optimizer, flat_grad, n_active, process_group=None):
    """The step of a rank that holds no view in the last round of an epoch: a zero gradient into the same collective(s) the working ranks issue,
    the same divisor, the same optimizer step -- parameters stay replicated bit for bit."""
    if flat_grad is None:
        raise RuntimeError("data-parallel sds_idle_step needs flat_grad = flat_grad_view(net_style.parameters())")
    flat_grad.zero_()
    guard = reduce_gradients(flat_grad, process_group, max(1, int(n_active)))
    _check_finite(net_style, guard)
    optimizer.step()
    return {"eikonal": torch.zeros(()), "opacity": torch.zeros(()), "idle": True}


def _avg_in_collective(process_group):
    """RCCL averages inside the collective (ncclAvg: no separate pass over the 49 MB); gloo -- the CPU tests' backend -- has no AVG.
    AC_ALLREDUCE_AVG=0 forces sum + one scaling pass on every backend (an RCCL build without ncclAvg; A/B on a multi-GPU node)."""
    if os.environ.get("AC_ALLREDUCE_AVG", "1") == "0":
        return False
    try:
        return str(torch.distributed.get_backend(process_group)).lower() == "nccl"
    except Exception:
        return False


class _Guard:
    """the guard word of one step's gradient collective, on its way to the host: finite() waits for it (4 bytes behind an event)"""

    def __init__(self, word):
        self.host = torch.zeros(1, dtype=torch.float32)
        if word.is_cuda:
            self.host = self.host.pin_memory()
            self.host.copy_(word.detach().reshape(1), non_blocking=True)
            self.event = torch.cuda.Event(); self.event.record()
        else:
            self.host.copy_(word.detach().reshape(1)); self.event = None

    def finite(self):
        if self.event is not None:
            self.event.synchronize()
        import math
        return math.isfinite(float(self.host[0]))


def reduce_gradients(flat_grad, process_group=None, divisor=None, nan_flag=None, early=None):
    """The ONE collective of a data-parallel step: all-reduce of the flat fp32 gradient, averaged over `divisor` contributing ranks
    (default: the world size; fewer in the last round of an epoch whose view count is not a multiple of the world size).

      * the average is taken INSIDE the collective on RCCL (ReduceOp.AVG; a short round rescales by world / divisor afterwards, once per epoch at most);
        backends without AVG (gloo) sum and scale in one pass afterwards.  World size 1: no scaling at all.
      * a buffer made by flat_grad_view carries a GUARD WORD behind the gradients: this rank's "a training render produced NaN / Inf" value (nan_flag: a
        device scalar that is non-finite in that case -- the renders' gradient_error --, or None = 0) rides in the same collective, so every rank
        sees the same verdict and either all of them step or all of them raise (a per-rank flag leaves the healthy ranks stepping on a
        poisoned gradient and then hanging in the next collective).  Returns a _Guard (or None without a guard word).
      * early = (work, (lo, hi)): the slice [lo, hi) is already in flight on a side stream (sds_step's overlap option); the rest goes now."""
    dist = torch.distributed
    world = dist.get_world_size(process_group)
    d = world if divisor is None else int(divisor)
    n = flat_grad.numel()
    buf = getattr(flat_grad, "ac_guard", None)
    if buf is not None and (buf.data_ptr() != flat_grad.data_ptr() or buf.numel() != n + 1):
        buf = None
    if buf is not None:
        if nan_flag is None:
            buf[n:].zero_()
        else:
            buf[n:].copy_(nan_flag.detach().reshape(1).to(device=buf.device, dtype=buf.dtype))
    else:
        buf = flat_grad
    avg = _avg_in_collective(process_group) and world > 1
    op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
    if early is None:
        dist.all_reduce(buf, op=op, group=process_group)
    else:
        work, (lo, hi) = early
        for a, b in ((0, lo), (hi, buf.numel())):                                   # the coarser levels (and whatever precedes the table) | the MLP gradients + guard
            if b > a:
                dist.all_reduce(buf[a:b], op=op, group=process_group)
        work.wait()                                                                  # the current stream waits for the early collective
    if world > 1:
        if avg and d != world:
            flat_grad.mul_(float(world) / float(d))
        elif not avg and d != 1:
            flat_grad.div_(d)
    return _Guard(buf[n:]) if buf.numel() == n + 1 else None


class _collective_verdict:
    """While a data-parallel step runs, the NaN flags of EVERY render of the given nets (training renders, the no-grad render_val of a cuda_ray net, ...)
    are collected instead of judged per rank (NeRFRenderer._guard_finite): `flag()` folds them into one device scalar for the guard word of the step's
    gradient collective, so that all ranks raise or all ranks step.  Outside a process group the nets keep their own event-based guard."""

    def __init__(self, active, *nets):
        self.nets = [n for n in nets if active and hasattr(n, "_guard_finite")]

    def __enter__(self):
        for n in self.nets:
            n.__dict__["_nan_deferred"] = []
        return self

    def __exit__(self, *exc):
        for n in self.nets:
            n.__dict__.pop("_nan_deferred", None)
        return False

    def flag(self, extra=()):
        """sum of every collected flag (+ extra device scalars): non-finite iff one of them is; None if there is none"""
        parts = [t.reshape(-1)[:1].float() for n in self.nets for t in n.__dict__.get("_nan_deferred", ())]
        dev = parts[0].device if parts else None
        for t in extra:
            if isinstance(t, torch.Tensor):
                t = t.detach().reshape(-1)[:1].float()
                if dev is None:
                    dev = t.device
                parts.append(t.to(dev))
        if not parts:
            return None
        return parts[0] if len(parts) == 1 else torch.cat(parts).sum()


def _check_finite(net, guard=None):
    """the reference asserts on a NaN gradient_error before its backward (instant_nsr.py:274); the fused path records a device flag instead, and it is
    resolved HERE -- before the optimizer step consumes the gradients -- at the cost of one 4-byte event wait.  Under a process group the verdict is the
    guard word of the gradient collective (reduce_gradients): the same on every rank."""
    if guard is not None:
        # the collective's verdict is the ONLY one under a process group: a rank-local check behind a finite guard could still raise on one rank alone
        # (a flag that never reached the guard word) and leave the others in the next collective
        pend = getattr(net, "__dict__", {}).get("_nan_pending")
        if pend:
            pend.clear()                                     # (this step's local flags are resolved by the collective verdict)
        if not guard.finite():
            raise FloatingPointError("NaN / Inf in the finite-difference normals of a training render on at least one rank (reference: instant_nsr.py:274); "
                                     "no rank has stepped")
        return
    chk = getattr(net, "check_finite", None)
    if chk is not None:
        chk()
