"""bench_legs.cpu -- the cpu_baseline legs: the CPU oracle (oracle/, test infrastructure) timed on the host on a bounded sample.  The ONLY place next to
tests/ and smoke() that touches oracle/."""
import os
import time

import numpy as np
import torch

from bench_legs.common import NUM_STEPS, UPSAMPLE_STEPS, oracle_field, sds_view

def cpu_baseline(p, table, ro, rd, budget_s=12.0):
    """time the CPU oracle on a bounded, strided sample of the same rays"""
    from oracle import oracle as O
    of = oracle_field(p, table)
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    idx = np.arange(0, ro.shape[0], ro.shape[0] // 64)[:64]
    t0 = time.time(); O.render_rays(of, ro[idx], rd[idx], NUM_STEPS, UPSAMPLE_STEPS, 1.6, float(p["inv_s"]), extras=False); dt = time.time() - t0
    n = int(min(ro.shape[0], max(64, (budget_s / max(dt, 1e-3)) * 64)))
    n = (n // 64) * 64
    idx = np.arange(0, ro.shape[0], max(1, ro.shape[0] // n))[:n]
    t0 = time.time(); O.render_rays(of, ro[idx], rd[idx], NUM_STEPS, UPSAMPLE_STEPS, 1.6, float(p["inv_s"]), extras=False); dt = time.time() - t0
    return dict(value=n / dt, unit="rays/s", cores=cores, threads=int(os.environ.get("OMP_NUM_THREADS", cores)), kind="port",
                sample=f"{n} rays (every {max(1, ro.shape[0] // n)}-th ray of the 256x256 view), 64+64 samples, {dt:.1f} s wall, OpenMP over rays",
                note="the C restatement of the reference's algorithm (oracle/), OpenMP over rays on every host core: faster than the reference's own "
                     "torch-CPU path would be; a reported baseline, not the target")


def cpu_baseline_sds(p, table, n_side=16, threads=None):
    """CPU leg of the SDS step on a bounded sample (n_side^2 rays of the same training view): the no-grad renders through the C oracle
    (OpenMP), the differentiable render core as torch-CPU autograd (MKL threads) over a hash encoder served by the oracle's forward /
    backward -- the structure of the reference's own CPU path (pure PyTorch + its hash kernel), with the reference's three backward
    passes folded into one like the GPU path.  kind = "port"."""
    import torch.nn as nn
    from oracle import oracle as O
    from avatarcraft_amd.instant_nsr import NeRFNetwork
    # threads: this leg is many medium-sized torch ops and a hash backward that is parallel over its 16 levels only; on a 256-core host the
    # full thread count is SLOWER than 32 (51 s per 256-ray step against a few seconds), so the leg runs on min(cores, 32) threads and says so
    cores = min(os.cpu_count() or 1, 32) if threads is None else int(threads)
    torch.set_num_threads(cores)
    try:
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)
    except OSError:
        pass

    class _Enc(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x01, emb, offsets, S):
            out, _, _ = O.hash_encode_forward(x01.detach().numpy(), emb.detach().numpy(), offsets, S, 16)
            ctx.save_for_backward(x01, emb); ctx.o = (offsets, S)
            return torch.from_numpy(np.ascontiguousarray(out.transpose(1, 0, 2).reshape(x01.shape[0], -1)))

        @staticmethod
        def backward(ctx, g):
            x01, emb = ctx.saved_tensors
            gl = np.ascontiguousarray(g.numpy().reshape(x01.shape[0], 16, 2).transpose(1, 0, 2))
            gg, _ = O.hash_encode_backward(gl, x01.numpy(), emb.detach().numpy(), ctx.o[0], ctx.o[1], 16, None)
            return None, torch.from_numpy(gg), None, None

    class OracleEncoder(nn.Module):               # stands where HashEncoder stands (no forward_stencil: 7 encoder calls per sample, like the reference)
        def __init__(self, emb, offsets, pls):
            super().__init__()
            self.embeddings = nn.Parameter(emb); self.offsets_np = offsets; self.S = np.float32(np.log2(pls))
            self.num_levels, self.level_dim, self.input_dim, self.per_level_scale, self.base_resolution = 16, 2, 3, pls, 16

        def forward(self, x, size=1):
            return _Enc.apply((x + size) / (2 * size), self.embeddings, self.offsets_np, self.S)

    def cpu_net(train):
        torch.manual_seed(0)
        net = NeRFNetwork()
        sd = {k: torch.from_numpy(np.asarray(p[k])) for k in p if k.startswith(("sdf_net", "color_net", "deviation_net"))}
        net.load_state_dict(sd, strict=False)
        net.encoder = OracleEncoder(torch.from_numpy(table.copy()), np.asarray(p["offsets"], np.int32), float(p["per_level_scale"]))
        net.fused_training = False
        return net.train(train)
    net = cpu_net(True)
    of = oracle_field(p, table)
    ro, rd = sds_view(0)
    idx = np.arange(0, 4096, 4096 // (n_side * n_side))[:n_side * n_side]
    ro, rd = ro[idx], rd[idx]
    n = ro.shape[0]
    opt = torch.optim.Adam(net.parameters(), lr=5e-3)
    rs = np.random.RandomState(0)
    inv_s = float(p["inv_s"])

    def step():
        noise = rs.uniform(0, 1, (n, NUM_STEPS)).astype(np.float32)
        O.render_rays(of, ro, rd, NUM_STEPS, UPSAMPLE_STEPS, 1.6, inv_s, noise=noise, extras=False)                       # (A) render_val
        g_img = torch.from_numpy(np.clip(rs.normal(0, 1, (n, 3)), -1, 1).astype(np.float32))                              # (B) synthetic guidance
        opt.zero_grad()
        noise = rs.uniform(0, 1, (n, NUM_STEPS)).astype(np.float32)
        z = O.render_rays(of, ro, rd, NUM_STEPS, UPSAMPLE_STEPS, 1.6, inv_s, noise=noise)["z_vals"]                       # (C) sampling stage (no grad)
        tro, trd = torch.from_numpy(ro), torch.from_numpy(rd)
        out = net._render_core_autograd(tro, trd, torch.from_numpy(z), NUM_STEPS, UPSAMPLE_STEPS, 1.6, None, 1.0, 0.0, 1, n)
        wgt = O.render_rays(of, ro, rd, NUM_STEPS, UPSAMPLE_STEPS, 1.6, inv_s, extras=False)["weights_sum"]             # frozen net_gt
        opa = torch.nn.functional.smooth_l1_loss(out[2].clamp(0, 1), torch.from_numpy(wgt).reshape(-1, 1).clamp(0, 1)) * 1e5
        ((out[3][0] * g_img).sum() + 0.01 * out[5] + opa).backward()
        opt.step()                                                                                                        # (D)
    step()
    t0 = time.time(); reps = 0
    while reps < 1 or (time.time() - t0 < 8.0 and reps < 8):
        step(); reps += 1
    dt = (time.time() - t0) / reps
    return dict(value=n / dt, unit="rays/s (SDS steps)", ms_per_4096_ray_step_equivalent=dt * 1e3 * 4096 / n, cores=os.cpu_count() or 1, threads=cores,
                threads_note="min(host cores, 32): this leg is many medium-sized torch ops and a hash backward parallel over its 16 levels only; on a 256-core "
                             "host the full thread count is slower (51 s per 256-ray step)", kind="port",
                sample=f"{reps} step(s) of {n} rays (every {4096 // n}-th ray of the 4096-ray training view), 64+64 samples, {dt:.2f} s each: C oracle (OpenMP) "
                       f"for the two no-grad renders and the sampling stage, torch-CPU autograd ({torch.get_num_threads()} threads) over the oracle's hash "
                       f"forward/backward for the render core, torch Adam on 12.2 M parameters")
