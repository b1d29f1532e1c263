"""CPU, world_size 2, gloo: the data-parallel SDS step (sharding + one flat all-reduce + replicated optimizer step).
The renderer itself needs the GPU, so a tiny torch module with the same .render() contract stands in for the field;
what is tested is the N>1 plumbing of avatarcraft_amd.stylize."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class TinyField(torch.nn.Module):
    """duck-types NeRFNetwork.render for the harness: rgb/weight_sum/depth/normal/gradient_error from a 2-layer MLP"""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.l1 = torch.nn.Linear(6, 16); self.l2 = torch.nn.Linear(16, 5)

    def render(self, rays_o, rays_d, bg_color=None, perturb=False, **kw):
        x = torch.cat([rays_o[0], rays_d[0]], -1)
        if self.training and perturb:
            x = x + 0.0 * torch.rand_like(x)
        h = self.l2(torch.tanh(self.l1(x)))
        rgb = torch.sigmoid(h[:, :3]); ws = torch.sigmoid(h[:, 3:4])
        return {"rgb": (rgb + (1 - ws) * bg_color)[None], "weight_sum": ws, "depth": h[:, 4][None], "normal": h[:, :3],
                "gradient_error": (h[:, 4] ** 2).mean()}


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from avatarcraft_amd.stylize import sds_step, flat_grad_view, shard_views
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net, net_gt = TinyField().train(), TinyField().eval()
    opt = torch.optim.Adam(net.parameters(), lr=5e-3)
    flat = flat_grad_view(net.parameters())
    views = shard_views(8, rank, world)
    gen = torch.Generator().manual_seed(100 + rank)
    def guidance(img):                                   # per-rank stream, like per-rank SDS noise
        return torch.randn(img.shape, generator=gen).clamp(-1, 1)
    grads = []
    for v in views:
        g = torch.Generator().manual_seed(v)
        ro = torch.randn(64, 3, generator=g); rd = torch.nn.functional.normalize(torch.randn(64, 3, generator=g), dim=-1)
        sds_step(net, net_gt, ro, rd, (8, 8), opt, guidance, batch_size=32, flat_grad=flat)
        grads.append(flat.clone())
    q.put((rank, views, [g.numpy() for g in grads], [p.detach().numpy().copy() for p in net.parameters()]))
    dist.destroy_process_group()


def test_two_rank_sds_step_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, v0, g0, p0), (r1, v1, g1, p1) = res
    assert v0 == [0, 2, 4, 6] and v1 == [1, 3, 5, 7]            # disjoint round-robin shards
    import numpy as np
    for a, b in zip(g0, g1):                                     # the averaged flat gradient is identical on both ranks
        assert np.array_equal(a, b) and np.abs(a).sum() > 0
    for a, b in zip(p0, p1):                                     # hence parameters stay replicated bit for bit
        assert np.array_equal(a, b)


def test_single_rank_equals_manual_average():
    """world_size 1 path and flat-gradient bookkeeping: p.grad are views into one buffer of the right size/order"""
    sys.path.insert(0, ROOT)
    from avatarcraft_amd.stylize import flat_grad_view, shard_views
    net = TinyField()
    flat = flat_grad_view(net.parameters())
    assert flat.numel() == sum(p.numel() for p in net.parameters())
    off = 0
    for p in net.parameters():
        assert p.grad.data_ptr() == flat.data_ptr() + 4 * off
        off += p.numel()
    (net.l2(torch.tanh(net.l1(torch.ones(3, 6)))).sum()).backward()
    assert float(flat.abs().sum()) > 0 and net.l1.weight.grad.data_ptr() == flat.data_ptr()
    # every view of an epoch exactly once (the reference's epoch, stylize.py:76-78); ranks without a view in the last round get None
    assert shard_views(10, 1, 4) == [1, 5, 9] and shard_views(10, 2, 4) == [2, 6, None]
    assert shard_views(100, 7, 8) == [7 + 8 * k for k in range(12)] + [None] and shard_views(100, 3, 8) == [3 + 8 * k for k in range(13)]
    assert shard_views(8, 0, 1) == list(range(8))
    from avatarcraft_amd.stylize import views_in_round
    for n, w in ((100, 8), (150, 8), (10, 4), (8, 2), (7, 1)):
        got = sorted(k for r in range(w) for k in shard_views(n, r, w) if k is not None)
        assert got == list(range(n))
        rounds = len(shard_views(n, 0, w))
        assert all(len(shard_views(n, r, w)) == rounds for r in range(w))
        assert [views_in_round(n, w, k) for k in range(rounds)] == [sum(shard_views(n, r, w)[k] is not None for r in range(w)) for k in range(rounds)]


def _loop_worker(rank, world, port, q, n_cap=8):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from avatarcraft_amd.stylize import stylize_epochs, flat_grad_view, SyntheticGuidance
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    net, net_gt = TinyField().train(), TinyField().eval()
    opt = torch.optim.Adam(net.parameters(), lr=5e-3)
    flat = flat_grad_view(net.parameters())
    seen = []

    class G(SyntheticGuidance):
        def __call__(self, rgb, text=None):
            seen.append((tuple(rgb.shape), text))
            return super().__call__(rgb, text)
    steps = stylize_epochs(net, net_gt, opt, G(1 + rank), hw=(16, 16), n_cap=n_cap, coarse_epochs=1, fine_epochs=1, subsample_scale=4, augment_cam=True,
                           stylize_head=True, coarse_head=0.5, fine_head=0.5, augment_bkg=True, augment_text=True, tgt_text="Hulk", batch_size=8,
                           device="cpu", flat_grad=flat)
    q.put((rank, steps, seen, [p.detach().numpy().copy() for p in net.parameters()]))
    if world > 1:
        dist.destroy_process_group()


def test_stylize_outer_loop_two_ranks_gloo():
    """coarse + fine epoch over jittered body and head views, background / prompt augmentation, views sharded over 2 ranks"""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loop_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, seen0, p0), (r1, s1, seen1, p1) = res
    assert s0 == s1 == 2 * ((8 + 4) // 2)                       # (8 body + 4 head views) / 2 ranks, two epochs
    for a, b in zip(p0, p1):
        assert np.array_equal(a, b)                             # parameters stay replicated
    shapes = [s for s, _ in seen0]
    assert shapes[:6] == [(1, 3, 4, 4)] * 6 and shapes[6:] == [(1, 3, 8, 8)] * 6          # fine stage: half the stride
    texts = {t for _, t in seen0 + seen1}
    assert all(t.endswith(" Hulk") for t in texts) and any("face" in t for t in texts) and any("body" in t for t in texts)


def test_stylize_outer_loop_uneven_views_two_ranks_gloo():
    """an epoch whose view count is not a multiple of the world size: 7 views (body + head close-ups of style_360_path(6)) on 2 ranks = 4 rounds, the last with one view; the rank
    without a view joins the collective with a zero gradient, nothing is dropped, parameters stay replicated"""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loop_worker, args=(r, 2, port, q, 6)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, seen0, p0), (r1, s1, seen1, p1) = res
    assert s0 == s1 == 2 * 4                                   # optimizer steps: every rank takes part in every round
    assert len(seen0) == 2 * 4 and len(seen1) == 2 * 3         # guidance calls = views actually rendered: 7 per epoch in total
    for a, b in zip(p0, p1):
        assert np.array_equal(a, b)


def _recon_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from avatarcraft_amd.reconstruct import reconstruct_epochs, make_optimizer
    from avatarcraft_amd.stylize import flat_grad_view
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = TinyField().train()
    opt, sched = make_optimizer(net, epochs=2)
    flat = flat_grad_view(net.parameters())
    g = torch.Generator().manual_seed(5)
    n = 5 * 8                                                  # five batches of 8 rays on two ranks: three rounds, the last with one batch
    ro = torch.randn(n, 3, generator=g); rd = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1); rgb = torch.rand(n, 3, generator=g)
    seen = []
    steps = reconstruct_epochs(net, opt, sched, ro, rd, rgb, epochs=2, batch_size=8, flat_grad=flat, on_step=lambda s, e, l: seen.append((s, e, float(l))))
    q.put((rank, steps, seen, [p.detach().numpy().copy() for p in net.parameters()]))
    dist.destroy_process_group()


def test_reconstruct_epochs_uneven_batches_two_ranks_gloo():
    """reconstruct_epochs under a process group visits every batch of an epoch once (reconstruct.py:86-92), also when batches % world != 0: the rank
    without a batch in the last round contributes a zero gradient to the same collective; parameters stay replicated"""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_recon_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, seen0, p0), (r1, s1, seen1, p1) = res
    assert s0 == s1 == 2 * 3
    assert [l for _, _, l in seen1][2] == 0.0 and [l for _, _, l in seen0][2] > 0.0          # round 3: rank 1 idle, rank 0 holds the fifth batch
    for a, b in zip(p0, p1):
        assert np.array_equal(a, b)


class _NaNField(TinyField):
    """a field whose training render reports a non-finite gradient_error (the reference's `assert (gradient == gradient).all()`, instant_nsr.py:274)"""
    poison = False

    def render(self, *a, **kw):
        out = super().render(*a, **kw)
        if self.poison and torch.is_grad_enabled():
            out["gradient_error"] = out["gradient_error"] * float("nan")
        return out


def _nan_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from avatarcraft_amd.stylize import sds_step, flat_grad_view
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net, net_gt = _NaNField().train(), TinyField().eval()
    opt = torch.optim.Adam(net.parameters(), lr=5e-3)
    flat = flat_grad_view(net.parameters())
    assert flat.ac_guard.numel() == flat.numel() + 1 and flat.ac_guard.data_ptr() == flat.data_ptr()
    g = torch.Generator().manual_seed(rank)
    ro = torch.randn(64, 3, generator=g); rd = torch.nn.functional.normalize(torch.randn(64, 3, generator=g), dim=-1)
    guidance = lambda img: torch.zeros_like(img)
    sds_step(net, net_gt, ro, rd, (8, 8), opt, guidance, batch_size=64, flat_grad=flat)          # a healthy step: both ranks step
    before = [p.detach().clone() for p in net.parameters()]
    net.poison = rank == 1                                                                       # the NaN happens on rank 1 ONLY
    raised = False
    try:
        sds_step(net, net_gt, ro, rd, (8, 8), opt, guidance, batch_size=64, flat_grad=flat)
    except FloatingPointError:
        raised = True
    stepped = any(not torch.equal(a, b.detach()) for a, b in zip(before, net.parameters()))
    # ... and the group is still usable: no rank is left behind in a collective
    net.poison = False
    sds_step(net, net_gt, ro, rd, (8, 8), opt, guidance, batch_size=64, flat_grad=flat)
    q.put((rank, raised, stepped, [p.detach().numpy().copy() for p in net.parameters()]))
    dist.destroy_process_group()


def test_nan_on_one_rank_stops_every_rank_gloo():
    """the NaN flag of a training render is per rank; the verdict must be collective (the guard word of the gradient all-reduce) --
    every rank raises before its optimizer step, none steps, and the replicas stay identical afterwards"""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nan_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, raised0, stepped0, p0), (_, raised1, stepped1, p1) = res
    assert raised0 and raised1 and not stepped0 and not stepped1
    for a, b in zip(p0, p1):
        assert np.array_equal(a, b) and np.isfinite(a).all()


class _GuardedNaNField(TinyField):
    """a field that guards its renders the way NeRFRenderer does (instant_nsr.py: _guard_finite behind every render, the no-grad ones included) and
    poisons the FIRST patch of a view only: without the collective verdict the guard of that patch raises on this rank alone -- on a CPU tensor at once, on the
    device in the next patch's non-waiting poll -- before the gradient collective, and the healthy ranks hang in the all-reduce"""
    from avatarcraft_amd.instant_nsr import NeRFRenderer as _R
    nan_guard = True
    _guard_finite = _R._guard_finite
    _poll_finite = _R._poll_finite
    check_finite = _R.check_finite
    poison = False
    poison_nograd = False
    calls = 0

    def render(self, *a, **kw):
        out = super().render(*a, **kw)
        grad = torch.is_grad_enabled()
        if grad:
            self.calls += 1
        if (self.poison and grad and self.calls == 1) or (self.poison_nograd and not grad):
            out["gradient_error"] = out["gradient_error"] * float("nan")
        self._guard_finite(out["gradient_error"])
        return out


def _guarded_nan_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from avatarcraft_amd.stylize import sds_step, flat_grad_view
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net, net_gt = _GuardedNaNField().train(), TinyField().eval()
    opt = torch.optim.Adam(net.parameters(), lr=5e-3)
    flat = flat_grad_view(net.parameters())
    g = torch.Generator().manual_seed(rank)
    ro = torch.randn(128, 3, generator=g); rd = torch.nn.functional.normalize(torch.randn(128, 3, generator=g), dim=-1)
    guidance = lambda img: torch.zeros_like(img)
    step = lambda: sds_step(net, net_gt, ro, rd, (8, 16), opt, guidance, batch_size=32, flat_grad=flat)      # a view of FOUR patches
    step()
    out = []
    for mode in ("patch0", "nograd"):
        before = [p.detach().clone() for p in net.parameters()]
        net.calls = 0
        net.poison, net.poison_nograd = (rank == 1 and mode == "patch0"), (rank == 1 and mode == "nograd")
        raised = False
        try:
            step()
        except FloatingPointError:
            raised = True
        stepped = any(not torch.equal(a, b.detach()) for a, b in zip(before, net.parameters()))
        out.append((raised, stepped))
        net.poison = net.poison_nograd = False
        net.calls = 0
        step()                                             # the group is still usable, no rank was left behind in a collective
    # outside a step the net's own guard is back in force (nothing stays deferred)
    assert "_nan_deferred" not in net.__dict__
    q.put((rank, out, [p.detach().numpy().copy() for p in net.parameters()]))
    dist.destroy_process_group()


def test_nan_in_one_patch_or_in_a_nograd_render_on_one_rank_is_a_collective_verdict_gloo():
    """(a) NaN in patch 0 of a four-patch view on rank 1 only; (b) NaN in the no-grad render_val on rank 1 only.  With per-rank guards rank 1
    raises before the gradient collective and rank 0 hangs in it (this test then times out); with the collective verdict both ranks raise after the
    collective, none steps, and the replicas stay bit-identical through the healthy steps that follow"""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_guarded_nan_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    (_, out0, p0), (_, out1, p1) = res
    assert out0 == out1 == [(True, False), (True, False)], (out0, out1)
    for a, b in zip(p0, p1):
        assert np.array_equal(a, b) and np.isfinite(a).all()
