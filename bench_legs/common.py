"""bench_legs.common -- constants of the workload (BASELINE.json configs[1]), the synthetic inputs, and the roofline helpers every leg of bench.py shares."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))          # the repository root (bench.py put it on sys.path)

import numpy as np
import torch

RAYS_PER_BATCH = 4096
H = W = 256
NUM_STEPS, UPSAMPLE_STEPS = 64, 64
BYTES_PER_RAY = 1008 * 1024          # 1008 hash-grid evaluations x (16 levels x 8 corners x 2 ch x 4 B)
FLOP_PER_RAY = 1008 * 6528 + 128 * 11264
HBM_PEAK_GBS = 8000.0                # MI355X HBM3E spec (MI355X_MICROARCH.md)


def oracle_field(p, table):
    """the CPU oracle's view of the same field (cpu_baseline legs only)"""
    from oracle import oracle as O
    return O.Field(table, p["offsets"], p["W1"], p["b1"], p["W2"], p["b2"], p["Wc1"], p["Wc2"], p["Wc3"], float(p["per_level_scale"]))


def make_inputs(device, rank):
    from avatarcraft_amd.synthetic import load_field_params, make_rays, device_field
    p = load_field_params()
    field, table = device_field(p, device=device)
    # camera on the 360-degree path of render_canonical.py (dist 1.7, f = 0.78125*256 = 200), one view per rank
    yaw = 2 * np.pi * ((rank * 12) % 100) / 100.0
    ro, rd = make_rays(H, W, dist=1.7, f=200.0, yaw=yaw, pitch=0.0)
    return p, field, table, ro, rd


SAMPLES = NUM_STEPS + UPSAMPLE_STEPS
# ALGORITHMIC bytes of one 4096-ray SDS step (SURVEY 8d per-unit figures: 1024 B gathered per hash evaluation forward, 2048 B
# read-modify-write per evaluation backward), for the work this implementation actually launches ...
SDS_BYTES_LAUNCHED = {
    "render_val (no-grad render of net_style)": RAYS_PER_BATCH * BYTES_PER_RAY,
    "grad render forward (the same fused launch, per-sample outputs kept)": RAYS_PER_BATCH * BYTES_PER_RAY,
    "net_gt render (frozen avatar, opacity target)": RAYS_PER_BATCH * BYTES_PER_RAY,
    "stencil features of the grad render: written once by the forward, streamed back by sdf_stencil_bwd (7 points x 32 floats per sample, each way) "
    "-- the re-gather they replace would be 3.758 GB": 2 * RAYS_PER_BATCH * SAMPLES * 7 * 32 * 4,
    "table-gradient scatter (hash_stencil_bwd_binned + bucket_accumulate)": RAYS_PER_BATCH * SAMPLES * 7 * 2048,
}
# ... and SURVEY 8(d)'s contract figure for the reference's schedule (3 forward renders + 3 backward passes of 7, 6 and 7 evaluations per sample)
SDS_BYTES_SURVEY = 3 * RAYS_PER_BATCH * BYTES_PER_RAY + (7 + 6 + 7) * RAYS_PER_BATCH * SAMPLES * 2048


def _binding(kernel):
    """busy fractions of a kernel from the committed PMC pass (profiles/traffic.json `binding`, tools/collect_profiles.py) or None"""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        b = (tj.get("binding") or {}).get(kernel)
        return (b, tj.get("binding_source") or tj.get("profile")) if b else (None, None)
    except Exception:
        return None, None


def _grid_roofline(kernel, evals, ms):
    """roofline object of the regular-grid SDF kernels (mesh export, density grid).  Their x-tiles make several lanes of a gather share one 64-byte
    sector (the spatial hash is linear in x), so the REQUEST bytes of SURVEY 8(d) -- 1024 B per query -- are not what moves: `frac` is taken against the
    sector bytes the L1s actually asked of L2 (TCP_TCC_READ_REQ x 64 B per query, committed PMC pass), which cannot exceed the peak; the request-byte rate
    stays in the object as `request_gbs` / `request_rate_vs_hbm_peak` (it can exceed 1 and did: 1.09), and `issue` names what binds the kernel."""
    b, src = _binding(kernel)
    req = evals * 1024 / 1e9
    out = {"bound": "hbm", "kernel": kernel, "peak": HBM_PEAK_GBS, "unit": "GB/s", "request_bytes": evals * 1024, "request_gbs": req / (ms * 1e-3),
           "request_rate_vs_hbm_peak": req / (ms * 1e-3) / HBM_PEAK_GBS, "request_floor_ms_at_peak": req / HBM_PEAK_GBS * 1e3}
    if b and b.get("l2_sector_bytes_per_launch"):
        sec = b["l2_sector_bytes_per_launch"] / 1e9
        out.update(algorithmic_bytes=int(b["l2_sector_bytes_per_launch"]), achieved=sec / (ms * 1e-3), frac=sec / (ms * 1e-3) / HBM_PEAK_GBS,
                   bytes_basis="64-byte sectors requested of L2 per launch (TCP_TCC_READ_REQ x 64 B, " + str(src) + "); timed live")
    else:
        out.update(algorithmic_bytes=evals * 1024, achieved=req / (ms * 1e-3), frac=min(1.0, req / (ms * 1e-3) / HBM_PEAK_GBS),
                   bytes_basis="request bytes (no committed sector counter for this kernel): capped at 1")
    if b and "issue" in b:
        out["issue"] = dict(b["issue"], source=src)
    if b and "gather" in b:
        out["gather"] = dict(b["gather"], source=src)
    return out


def make_net(p, table, dev, train, cuda_ray=False):
    from avatarcraft_amd.instant_nsr import NeRFNetwork
    torch.manual_seed(0)
    net = NeRFNetwork(cuda_ray=cuda_ray)
    sd = {k: torch.from_numpy(np.asarray(p[k])) for k in p if k.startswith(("sdf_net", "color_net", "deviation_net"))}
    sd["encoder.embeddings"] = torch.from_numpy(table); sd["encoder.offsets"] = torch.from_numpy(np.asarray(p["offsets"]))
    net.load_state_dict(sd, strict=not cuda_ray)       # (cuda_ray adds the density grid / step counter buffers)
    return net.to(dev).train(train)


def sds_view(rank):
    """the 64x64 stride-4 sub-sampled rays of a 256x256 training camera (stylize.py:98-107), one view per rank"""
    from avatarcraft_amd.synthetic import make_rays
    yaw = 2 * np.pi * ((rank * 12) % 100) / 100.0
    ro, rd = make_rays(256, 256, dist=1.8, f=200.0, yaw=yaw, pitch=0.0)
    return ro.reshape(256, 256, 3)[1::4, 2::4].reshape(-1, 3).copy(), rd.reshape(256, 256, 3)[1::4, 2::4].reshape(-1, 3).copy()


class _NoStep:
    """optimizer stand-in whose step() leaves the gradients and the weights alone (bench.py inspects the gradient of one more step)"""

    def __init__(self, opt):
        self.param_groups = opt.param_groups

    def zero_grad(self, set_to_none=False):
        pass

    def step(self):
        pass


XGMI_LINKS, XGMI_LINK_GBS = 7, 153.0          # per GPU: 7 point-to-point xGMI links x ~153 GB/s (the task's figure for this node type)
