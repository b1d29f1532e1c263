#!/usr/bin/env python3
"""Generate the golden vectors of tests/golden/ by IMPORTING THE REFERENCE'S PYTHON in this
container (it cannot travel to the GPU box; only the vectors do).

    python tests/golden/make_golden.py [/root/reference]

What is imported from the reference: models/instant_nsr.py (NeRFNetwork, NeRFRenderer.run,
up_sample, sample_pdf, cat_z_vals, near_far_from_bound), encoder/ (HashEncoder python side,
get_encoder, freq_encoder), utils/ray_utils.py.  What is NOT the reference: the three JIT-built
CUDA extension back ends (`encoder.hashencoder.backend`, `encoder.shencoder.backend`,
`raymarching.backend`) cannot be built here (no CUDA toolkit), so they are pre-seeded in
sys.modules; the hash back end is served by oracle/ (the CPU restatement).  These goldens therefore
pin everything in NeRFRenderer.run EXCEPT the hash-grid kernel itself, which is pinned by the
known-answer values in tests/golden/kat.json.

Large inputs are not stored: the hash table is regenerated from a numpy RandomState seed
(np.random.RandomState is a frozen stream), see tests/common.py:make_table().
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import numpy as np
import torch

from oracle import oracle as O
from tests.common import make_table, make_rays, smooth_level_amp, TABLE_SEED

torch.set_num_threads(8)


# ---------------------------------------------------------------- stubs for absent third-party modules
def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


for name in ("mcubes", "trimesh", "igl"):
    _stub(name)


class _HashBackend:
    """Stands in for the JIT-built `_hash_encoder` extension (encoder/hashencoder/backend.py)."""

    @staticmethod
    def hash_encode_forward(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, calc_grad_inputs, dy_dx):
        out, dd, _ = O.hash_encode_forward(inputs.detach().numpy(), embeddings.detach().numpy(), offsets.numpy(),
                                           np.float32(S), H, calc_grad_inputs)
        outputs.copy_(torch.from_numpy(out))
        if calc_grad_inputs:
            dy_dx.copy_(torch.from_numpy(dd))

    @staticmethod
    def hash_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, C, L, S, H, calc_grad_inputs,
                             dy_dx, grad_inputs):
        gg, gi = O.hash_encode_backward(grad.numpy(), inputs.detach().numpy(), embeddings.detach().numpy(), offsets.numpy(),
                                        np.float32(S), H, dy_dx.numpy() if calc_grad_inputs else None)
        grad_embeddings.copy_(torch.from_numpy(gg))
        if calc_grad_inputs:
            grad_inputs.copy_(torch.from_numpy(gi))


_stub("encoder.hashencoder.backend", _backend=_HashBackend)
_stub("encoder.shencoder.backend", _backend=object())
_stub("raymarching.backend", _backend=object())

import models.instant_nsr as ref_nsr  # noqa: E402  (the reference)


GT_SDF_BIAS = -0.40            # sdf_net.1.bias[0] of the frozen net_gt in the training golden (net_style: -0.45)


def build_reference_net(**kw):
    """NeRFNetwork(**kw) with seed-0 init, then table/first-layer randomised so that all 16 levels matter
    (with the stock geometric init the hash features get zero weight)."""
    torch.manual_seed(0)
    net = ref_nsr.NeRFNetwork(**kw)
    rs = np.random.RandomState(1234)
    with torch.no_grad():
        scale, _ = O.hash_level_table(16, np.float32(np.log2(net.encoder.per_level_scale)), 16)
        net.level_amp = smooth_level_amp(scale)
        net.encoder.embeddings.copy_(torch.from_numpy(make_table(int(net.encoder.offsets[-1]), offsets=net.encoder.offsets.numpy(),
                                                                  level_amp=net.level_amp)))
        v = net.sdf_net[0].weight_v
        v[:, 3:] = torch.from_numpy(rs.normal(0.0, 0.05, size=(64, 32)).astype(np.float32))
        net.sdf_net[0].bias.copy_(torch.from_numpy(rs.normal(0.0, 0.05, size=64).astype(np.float32)))
        net.sdf_net[1].bias.copy_(torch.from_numpy(rs.normal(0.0, 0.02, size=16).astype(np.float32)))
        net.sdf_net[1].bias[0] = -0.45       # sphere-ish SDF: surface near r ~ 0.45
        net.deviation_net.variance.fill_(0.3)
    return net


def effective_weights(net):
    def wn(layer):
        return torch._weight_norm(layer.weight_v, layer.weight_g, 0).detach().numpy().astype(np.float32)
    return dict(W1=wn(net.sdf_net[0]), b1=net.sdf_net[0].bias.detach().numpy(), W2=wn(net.sdf_net[1]),
                b2=net.sdf_net[1].bias.detach().numpy(), Wc1=wn(net.color_net[0]), Wc2=wn(net.color_net[1]),
                Wc3=wn(net.color_net[2]))


class Recorder:
    """Records the outputs of torch.searchsorted / torch.sort made inside run()."""

    def __init__(self):
        self.ss, self.sort = [], []

    def __enter__(self):
        self._ss, self._sort = torch.searchsorted, torch.sort

        def ss(*a, **k):
            r = self._ss(*a, **k); self.ss.append(r.clone()); return r

        def srt(*a, **k):
            r = self._sort(*a, **k); self.sort.append(r[1].clone()); return r
        torch.searchsorted, torch.sort = ss, srt
        return self

    def __exit__(self, *a):
        torch.searchsorted, torch.sort = self._ss, self._sort


def run_case(net, rays_o, rays_d, num_steps, upsample_steps, train, seed, bg):
    N = rays_o.shape[0]
    net.train(train)
    noise = None
    if train:
        torch.manual_seed(seed)
        noise = torch.rand(N, num_steps).numpy().copy()   # the same draw run() makes first (instant_nsr.py:162)
        torch.manual_seed(seed)
    with Recorder() as rec, torch.no_grad():
        out = net.render(torch.from_numpy(rays_o)[None], torch.from_numpy(rays_d)[None], num_steps=num_steps, bound=1.6,
                         upsample_steps=upsample_steps, staged=False, bg_color=torch.from_numpy(bg),
                         cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, render_can=True, perturb=train)
    nup = upsample_steps // 16
    ss = np.stack([t.numpy() for t in rec.ss], 1).astype(np.int32) if nup else np.zeros((N, 0, 16), np.int32)
    srt = np.full((N, max(nup, 1), 128), -1, np.int32)
    for i, t in enumerate(rec.sort):
        srt[:, i, :t.shape[1]] = t.numpy()
    res = dict(rays_o=rays_o, rays_d=rays_d, bg=bg, image=out["rgb"][0].numpy(), weights_sum=out["weight_sum"][:, 0].numpy(),
               depth=out["depth"][0].numpy(), normal_map=out["normal"].numpy(), weights=out["weights"].numpy(),
               alpha=out["pts_alpha"].numpy(), color=out["pts_color"].numpy(), z_vals=out["z_vals"].numpy(),
               gradient_error=np.float32(out["gradient_error"].item()), ss_inds=ss, sort_index=srt,
               num_steps=np.int32(num_steps), upsample_steps=np.int32(upsample_steps), train=np.int32(train))
    if noise is not None:
        res["noise"] = noise
    # The oracle's own sample indices on these inputs, compared with the reference's: positions (ray, iteration, sample) of the
    # searchsorted indices that differ (knife-edge `cdf[k] <= u` decided by the last ulp of exp: Sleef in torch CPU, ac_math in the
    # oracle and the GPU).  Recorded so that the parity test asserts "exactly these and no others" instead of a percentage.
    if nup:
        from tests.common import oracle_field_from_golden
        global _ORACLE_FIELD
        if "_ORACLE_FIELD" not in globals():
            _ORACLE_FIELD = O.Field(net.encoder.embeddings.detach().numpy(), net.encoder.offsets.numpy(), *[effective_weights(net)[k] for k in
                                    ("W1", "b1", "W2", "b2", "Wc1", "Wc2", "Wc3")], float(net.encoder.per_level_scale))
        r = O.render_rays(_ORACLE_FIELD, rays_o, rays_d, num_steps, upsample_steps, 1.6, float(net.forward_variance().item()), bg=bg, noise=noise)
        res["oracle_ss_flips"] = np.argwhere(r["ss_inds"] != ss).astype(np.int32).reshape(-1, 3)
    else:
        res["oracle_ss_flips"] = np.zeros((0, 3), np.int32)
    return res


def hash_witness_fp64(x01, table, offsets, per_level_scale, H, grad=None):
    """Second, independent witness of the hash-grid kernel's FLOAT output (hashencoder.cu:73-220 forward, :223-308 backward),
    written from the formulas with its own index math in numpy fp64 -- it calls nothing under oracle/.
        scale = exp2f(l*S)*H - 1 (fp32, :122), res = ceil(scale)+1 (:123), pos = x*scale + 0.5 (:131), cell = floor(pos), frac = pos - cell,
        corner weight = prod_d (bit_d ? frac_d : 1 - frac_d) (:141-153),
        index = (stride <= hashmap_size for all dims ? x + y*(res+1) + z*(res+1)^2 : x*1 ^ y*2654435761 ^ z*805459861 (u32)) % hashmap_size (:35-70)
    Only `scale` is rounded to fp32 (it decides the cell); everything else is fp64, so a result agrees with any faithful fp32 evaluation
    to ~1e-7 relative -- and disagrees at the 1e-2 level if a corner weight or index were swapped.
    Returns enc [B, L*C] f64 (HashEncoder.forward's layout, hashgrid.py:41) and, with grad [B, L*C], d(table) [n_entries, C] f64."""
    x = np.asarray(x01, np.float64)
    B, D = x.shape
    L, C = len(offsets) - 1, table.shape[1]
    tab = np.asarray(table, np.float64)
    S = np.float32(np.log2(per_level_scale))                       # hashgrid.py:27 -> `const float S`
    enc = np.zeros((B, L * C), np.float64)
    gtab = np.zeros_like(tab) if grad is not None else None
    inside = ((x >= 0) & (x <= 1)).all(1)                          # :95-119 out-of-range input -> zeros
    primes = (np.uint32(1), np.uint32(2654435761), np.uint32(805459861))
    for l in range(L):
        size = int(offsets[l + 1]) - int(offsets[l])
        scale = np.float32(np.float32(2.0 ** float(np.float32(l) * S)) * np.float32(H)) - np.float32(1.0)
        res = int(np.ceil(scale)) + 1
        pos = x * float(scale) + 0.5
        cell = np.floor(pos)
        frac = pos - cell
        cell = cell.astype(np.int64)
        dense = (res + 1) ** D <= size                             # get_grid_index: hashed as soon as a stride exceeds the level size
        for corner in range(1 << D):
            w = np.ones(B, np.float64)
            cc = np.empty((B, D), np.int64)
            for d in range(D):
                bit = (corner >> d) & 1
                w *= frac[:, d] if bit else 1.0 - frac[:, d]
                cc[:, d] = cell[:, d] + bit
            if dense:
                idx = np.zeros(B, np.int64); st = 1
                for d in range(D):
                    idx += cc[:, d] * st; st *= res + 1
            else:
                h = np.zeros(B, np.uint32)
                for d in range(D):
                    h ^= (cc[:, d].astype(np.uint64) * np.uint64(primes[d]) & np.uint64(0xffffffff)).astype(np.uint32)
                idx = h.astype(np.int64)
            idx = idx % size + int(offsets[l])
            w = np.where(inside, w, 0.0)
            enc[:, l * C:(l + 1) * C] += w[:, None] * tab[idx]
            if gtab is not None:
                np.add.at(gtab, idx, w[:, None] * np.asarray(grad, np.float64)[:, l * C:(l + 1) * C])
    return enc, gtab


def main():
    net = build_reference_net()
    ew = effective_weights(net)
    inv_s = float(net.forward_variance().item())
    common = dict(table_seed=np.int64(TABLE_SEED), level_amp=net.level_amp, offsets=net.encoder.offsets.numpy(),
                  per_level_scale=np.float64(net.encoder.per_level_scale), inv_s=np.float32(inv_s), **ew)
    # raw g/v as the checkpoint stores them (state_dict parity of the host mirror)
    for i, l in enumerate(net.sdf_net):
        common[f"sdf_net.{i}.weight_g"] = l.weight_g.detach().numpy(); common[f"sdf_net.{i}.weight_v"] = l.weight_v.detach().numpy()
        common[f"sdf_net.{i}.bias"] = l.bias.detach().numpy()
    for i, l in enumerate(net.color_net):
        common[f"color_net.{i}.weight_g"] = l.weight_g.detach().numpy(); common[f"color_net.{i}.weight_v"] = l.weight_v.detach().numpy()
    common["deviation_net.variance"] = net.deviation_net.variance.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "nsr_params.npz"), **common)

    rs = np.random.RandomState(7)
    cases = {}
    ro, rd = make_rays(8, 8, dist=1.7, f=6.25)                      # 64 rays through the object
    bg = rs.uniform(0, 1, size=(ro.shape[0], 3)).astype(np.float32)
    cases["eval_64_64"] = run_case(net, ro, rd, 64, 64, False, 0, bg)
    cases["train_64_64"] = run_case(net, ro, rd, 64, 64, True, 42, bg)
    ro2, rd2 = make_rays(6, 6, dist=1.8, f=4.0, jitter_seed=3)
    bg2 = np.ones((ro2.shape[0], 3), np.float32)
    cases["eval_32_32"] = run_case(net, ro2, rd2, 32, 32, False, 0, bg2)
    cases["eval_64_0"] = run_case(net, ro2, rd2, 64, 0, False, 0, bg2)
    for name, c in cases.items():
        np.savez_compressed(os.path.join(HERE, f"run_{name}.npz"), **c)
        print(name, "weights_sum min/mean/max", c["weights_sum"].min(), c["weights_sum"].mean(), c["weights_sum"].max(),
              "eik", c["gradient_error"])

    # gradients of one training render: rgb.backward(image_grad) + (0.01 * eikonal).backward(),
    # stylize.py:163-169, on the reference's own autograd graph (hash backward served by the oracle)
    # (round 3: 256 rays instead of 36 -- a 16 x 16 view with jittered pixel positions; white background, what stylize.py renders by default)
    ro2, rd2 = make_rays(16, 16, dist=1.8, f=10.0, jitter_seed=11)
    bg2 = np.ones((ro2.shape[0], 3), np.float32)
    net.train(True)
    net.zero_grad()
    torch.manual_seed(42)
    noise_g = torch.rand(ro2.shape[0], 64).numpy().copy()
    torch.manual_seed(42)
    outg = net.render(torch.from_numpy(ro2)[None], torch.from_numpy(rd2)[None], num_steps=64, bound=1.6, upsample_steps=64, staged=False,
                      bg_color=torch.from_numpy(bg2), cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, render_can=True, perturb=True)
    img_grad = np.clip(np.random.RandomState(5).normal(0, 1, (ro2.shape[0], 3)), -1, 1).astype(np.float32)
    outg["rgb"][0].backward(gradient=torch.from_numpy(img_grad), retain_graph=True)
    (outg["gradient_error"] * 0.01).backward(retain_graph=True)
    gg = dict(rays_o=ro2, rays_d=rd2, bg=bg2, noise=noise_g, img_grad=img_grad, rgb=outg["rgb"][0].detach().numpy(),
              z_vals=outg["z_vals"].detach().numpy())
    for k, prm in net.named_parameters():
        if k != "encoder.embeddings":
            gg["grad." + k] = prm.grad.numpy().copy()
    ge = net.encoder.embeddings.grad.numpy()
    nz = np.flatnonzero(np.abs(ge).sum(1))
    pick = nz[np.random.RandomState(6).choice(len(nz), 4096, replace=False)]
    gg["emb_idx"] = pick.astype(np.int64); gg["emb_grad"] = ge[pick].copy()
    gg["emb_nnz"] = np.int64(len(nz)); gg["emb_l2"] = np.float64(np.sqrt((ge.astype(np.float64) ** 2).sum())); gg["emb_sum"] = np.float64(ge.astype(np.float64).sum())
    # third term of the step (stylize.py:177-193): the frozen net_gt renders the same rays (eval mode: perturb has no effect,
    # instant_nsr.py:161), opacity loss = smooth_l1(clamp(pred), clamp(gt).detach()) * 1e5, backward into the same .grad
    import torch.nn.functional as F
    net_gt = build_reference_net()
    with torch.no_grad():
        net_gt.sdf_net[1].bias[0] = GT_SDF_BIAS                    # a slightly different body, so that the opacities differ
    net_gt.eval()
    out_gt = net_gt.render(torch.from_numpy(ro2)[None], torch.from_numpy(rd2)[None], num_steps=64, bound=1.6, upsample_steps=64, staged=False,
                           bg_color=torch.from_numpy(bg2), cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, render_can=True, perturb=True)
    opacity_pred = torch.clamp(outg["weight_sum"], 0.0, 1.0)
    opacity_gt = torch.clamp(out_gt["weight_sum"], 0.0, 1.0).detach()
    opacity_loss = F.smooth_l1_loss(opacity_pred, opacity_gt) * 1e5
    opacity_loss.backward(retain_graph=False)
    gg["opacity_loss"] = np.float64(opacity_loss.item()); gg["opacity_gt"] = opacity_gt[:, 0].numpy().copy(); gg["gt_sdf_bias"] = np.float32(GT_SDF_BIAS)
    gg["opacity_pred"] = outg["weight_sum"][:, 0].detach().numpy().copy()
    for k, prm in net.named_parameters():
        if k != "encoder.embeddings":
            gg["grad3." + k] = prm.grad.numpy().copy()
    ge3 = net.encoder.embeddings.grad.numpy()
    gg["emb_grad3"] = ge3[pick].copy(); gg["emb3_l2"] = np.float64(np.sqrt((ge3.astype(np.float64) ** 2).sum()))
    # the optimizer step (stylize.py:199, :355-363): Adam(lr 5e-3) over all parameters, first step from zero moments
    before = {k: prm.detach().clone() for k, prm in net.named_parameters()}
    torch.optim.Adam([{"params": net.parameters(), "lr": 5e-3}]).step()
    for k, prm in net.named_parameters():
        d = (prm.detach() - before[k]).numpy()
        if k == "encoder.embeddings":
            gg["adam_delta.emb"] = d[pick].copy(); gg["adam_changed"] = np.int64((np.abs(d).sum(1) > 0).sum())
        else:
            gg["adam_delta." + k] = d.copy()
    with torch.no_grad():                                          # put the parameters back: the goldens below use the same net
        for k, prm in net.named_parameters():
            prm.copy_(before[k])
    np.savez_compressed(os.path.join(HERE, "train_grad.npz"), **gg)
    print("train_grad: emb nnz", len(nz), "l2", gg["emb_l2"], "variance grad", gg["grad.deviation_net.variance"], "opacity loss", gg["opacity_loss"],
          "emb l2 with the opacity term", gg["emb3_l2"])
    net.zero_grad()

    # forward_sdf / forward_color / gradient point-wise goldens (instant_nsr.py:627-704)
    pts = rs.uniform(-1.6, 1.6, size=(257, 3)).astype(np.float32)
    pts[0] = [1.6, -1.6, 1.6]; pts[1] = [0, 0, 0]
    with torch.no_grad():
        net.eval()
        t = torch.from_numpy(pts)
        sdf = net.forward_sdf(t, 1.6)
        grad = net.gradient(t, 1.6, 0.005)
        nrm = grad / (1e-5 + torch.linalg.norm(grad, ord=2, dim=-1, keepdim=True))
        col = net.forward_color(t, None, nrm, sdf[:, 1:], 1.6)
        enc = net.encoder(t, 1.6)
    # independent fp64 witness of the hash kernel's float output (forward and table gradient), see hash_witness_fp64
    x01 = (pts + np.float32(1.6)) / np.float32(3.2)                 # HashEncoder.forward: (x + size) / (2 size) in fp32, hashgrid.py:130
    wgrad = np.random.RandomState(8).normal(0, 1, (pts.shape[0], 32))
    enc_w, gtab_w = hash_witness_fp64(x01, net.encoder.embeddings.detach().numpy(), net.encoder.offsets.numpy(), net.encoder.per_level_scale, 16,
                                      grad=wgrad)
    nzw = np.flatnonzero(np.abs(gtab_w).sum(1))
    pickw = np.sort(nzw[np.random.RandomState(9).choice(len(nzw), 4096, replace=False)])
    print("hash witness: |enc - witness| max", np.abs(enc.numpy() - enc_w).max(), "table-gradient entries touched", len(nzw))
    np.savez_compressed(os.path.join(HERE, "field_points.npz"), pts=pts, sdf=sdf.numpy(), gradient=grad.numpy(),
                        normal=nrm.numpy(), color=col.numpy(), enc=enc.numpy(), enc_witness=enc_w, witness_grad=wgrad.astype(np.float32),
                        witness_gtab_idx=pickw.astype(np.int64), witness_gtab=gtab_w[pickw], witness_gtab_nnz=np.int64(len(nzw)),
                        witness_gtab_l1=np.float64(np.abs(gtab_w).sum()))

    # HashEncoder python-side facts (hashgrid.py:79-124): offsets, n_params, output_dim for a few configs
    from encoder import get_encoder
    facts = {}
    for tag, cfg in {"default": dict(hash_num_levels=16, hash_level_dim=2, hash_per_level_scale=1.3819, hash_base_resolution=16,
                                     hash_log2_hashmap_size=19, hash_desired_resolution=2048),
                     "small": dict(hash_num_levels=8, hash_level_dim=4, hash_per_level_scale=2.0, hash_base_resolution=4,
                                   hash_log2_hashmap_size=12, hash_desired_resolution=None)}.items():
        enc_m, dim = get_encoder("hashgrid", dict(in_dim=3, **cfg))
        facts[f"{tag}_offsets"] = enc_m.offsets.numpy(); facts[f"{tag}_dim"] = np.int32(dim)
        facts[f"{tag}_pls"] = np.float64(enc_m.per_level_scale); facts[f"{tag}_nparams"] = np.int64(int(enc_m.n_params))
    fe, fdim = get_encoder("frequency", dict(in_dim=3, freq_multires=6))
    x = torch.from_numpy(pts[:16])
    facts["freq_in"] = pts[:16]; facts["freq_out"] = fe(x).numpy(); facts["freq_dim"] = np.int32(fdim)
    np.savez_compressed(os.path.join(HERE, "encoder_facts.npz"), **facts)
    print("n_params", sum(p.numel() for p in net.parameters()))


if __name__ == "__main__":
    main()


# ---------------------------------------------------------------- ray generation goldens
def _prepare_render_utils():
    """stubs the third-party modules utils/render_utils.py imports at module level (none is used by the functions called here)"""
    import numpy
    for name in ("pytorch3d", "pytorch3d.structures", "pytorch3d.renderer", "open3d", "cv2", "torchvision", "torchvision.transforms",
                 "imageio", "lpips", "prompt_toolkit"):
        if name not in sys.modules:
            _stub(name)
    sys.modules["pytorch3d.structures"].Meshes = object
    for n in ("RasterizationSettings", "MeshRenderer", "MeshRasterizer", "HardPhongShader", "PointLights", "TexturesVertex",
              "PerspectiveCameras"):
        setattr(sys.modules["pytorch3d.renderer"], n, object)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    # numpy>=2 rejects np.array(copy=False) used by geometry/transformations.py:1846
    import geometry.transformations as T

    class _NP:
        def __getattr__(self, k):
            return getattr(numpy, k)

        @staticmethod
        def array(*a, **k):
            if k.get("copy", True) is False:
                k.pop("copy"); return numpy.asarray(*a, **k)
            return numpy.array(*a, **k)
    T.numpy = _NP()


def make_ray_goldens():
    """Rays from the reference's own camera code: default_360_path -> pose2cap -> shot_rays
    (utils/render_utils.py:137-154,323-337,363-376; utils/ray_utils.py:25-37), and
    SMPLDataset.gen_rays_pose's formula is exercised in a separate golden."""
    _prepare_render_utils()
    try:
        import utils.render_utils as RU
    except Exception as e:   # pragma: no cover
        print("render_utils import failed:", repr(e)); raise
    import utils.ray_utils as RY
    out = {}
    poses, _ = RU.default_360_path(np.array([0, 0, 0]), np.array([0, 1, 0]), 1.44, 100)
    cap = RU.pose2cap([64, 64], poses[0])
    coords = np.argwhere(np.ones(cap.shape))[:, ::-1]
    o, d = RY.shot_rays(cap, coords)
    out["kat64_o"], out["kat64_d"] = o.astype(np.float32), d.astype(np.float32)
    # config 2: 256x256, dist 1.7 (render_canonical.py:34), poses 0 and 17; keep every 16th pixel of the 256^2 grid
    poses, _ = RU.default_360_path(np.array([0, 0, 0]), np.array([0, 1, 0]), 1.7, 60)
    for pi in (0, 17):
        cap = RU.pose2cap([256, 256], poses[pi])
        coords = np.argwhere(np.ones(cap.shape))[:, ::-1]
        o, d = RY.shot_rays(cap, coords)
        out[f"can256_p{pi}_o"] = o.astype(np.float32)[::16]; out[f"can256_p{pi}_d"] = d.astype(np.float32)[::16]
        out[f"can256_p{pi}_c2w"] = poses[pi].camera_to_world
    np.savez_compressed(os.path.join(HERE, "rays.npz"), **out)
    print("rays: centre d", out["kat64_d"][2080], "corner", out["kat64_d"][0], "o", out["kat64_o"][0])


def make_warp_goldens():
    """SMPL-guided warp rows (a2, a13): geometry_guided_near_far_{torch,np} are pure torch/numpy in the reference;
    warp_samples_to_canonical calls libigl (absent here) for the closest point and the barycentric coordinates: those two
    calls are served by a stand-in (closest point from the oracle, the textbook barycentric formula), everything else in
    the function (mask on the SQUARED distance, blend of the per-vertex 4x4, np.linalg.inv, apply, can_dirs) is the
    reference's own numpy code.  Hence: closest-point parity is pinned against the definition only."""
    import utils.ray_utils as RY
    from tests.common import make_body
    verts, faces, Ts = make_body()
    install_igl_standin()
    ro, rd = make_rays(12, 12, dist=1.8, f=9.0, jitter_seed=11)
    return _make_warp_goldens_body(RY, verts, faces, Ts, ro, rd)


def install_igl_standin():
    igl = sys.modules["igl"]

    def point_mesh_squared_distance(P, V, F):
        can, clo, d2, fid, mask = O.warp_samples(P, V, F, np.tile(np.eye(4)[None], (V.shape[0], 1, 1)))
        return d2, fid, clo

    def barycentric_coordinates_tri(p, a, b, c):
        p, a, b, c = (np.asarray(t, np.float64) for t in (p, a, b, c))       # libigl's double-precision path
        v0, v1, v2 = b - a, c - a, p - a
        d00 = (v0 * v0).sum(1); d01 = (v0 * v1).sum(1); d11 = (v1 * v1).sum(1); d20 = (v2 * v0).sum(1); d21 = (v2 * v1).sum(1)
        den = d00 * d11 - d01 * d01
        v = (d11 * d20 - d01 * d21) / den; w = (d00 * d21 - d01 * d20) / den
        return np.stack([1 - v - w, v, w], 1)
    igl.point_mesh_squared_distance = point_mesh_squared_distance
    igl.barycentric_coordinates_tri = barycentric_coordinates_tri


def _make_warp_goldens_body(RY, verts, faces, Ts, ro, rd):
    near_t, far_t = RY.geometry_guided_near_far_torch(torch.from_numpy(ro), torch.from_numpy(rd), verts, 0.05)
    near_n, far_n = RY.geometry_guided_near_far_np(ro, rd, verts, 0.05)
    z = np.linspace(0.9, 2.7, 24, dtype=np.float32)
    pts = (ro[:, None, :] + rd[:, None, :] * z[None, :, None]).astype(np.float32)
    can, can_dirs, closest, mask = RY.warp_samples_to_canonical(pts, verts, np.concatenate([faces, faces], 1), Ts, 0.05)
    np.savez_compressed(os.path.join(HERE, "warp.npz"), rays_o=ro, rays_d=rd, near_t=near_t.numpy(), far_t=far_t.numpy(), near_n=near_n, far_n=far_n,
                        pts=pts, can_pts=can, can_dirs=can_dirs, closest=closest, mask=mask)
    print("warp: rays hitting the body", int(np.isfinite(near_t.numpy()).sum()), "of", ro.shape[0], "mask frac", float(mask.mean()))


if __name__ == "__main__":
    make_ray_goldens()
    make_warp_goldens()


# ---------------------------------------------------------------- SMPL skinning goldens
def make_smpl_goldens():
    """models.smpl.lbs is a free function (models/smpl.py:351): called with synthetic buffers of the SMPL layout (the licensed
    pickle is not available), for T (return_T, concat_joints) and for the posed vertices / joints."""
    import torch
    import models.smpl as RS
    from avatarcraft_amd.smpl import BodyModel
    bm = BodyModel.synthetic(seed=3, n_verts=600)
    g = np.random.default_rng(11)
    pose = (g.standard_normal((1, 72)) * 0.4).astype(np.float32)
    pose[0, 3:6] = 0.0                                     # one exactly-zero rotation: exercises the 1e-8 epsilon
    betas = g.standard_normal((1, 10)).astype(np.float32)
    args = (bm.v_template, bm.shapedirs, bm.posedirs, bm.J_regressor, bm.parents, bm.lbs_weights)
    T, v, dv = RS.lbs(torch.from_numpy(betas), torch.from_numpy(pose), *args, return_T=True, concat_joints=True)
    verts, joints = RS.lbs(torch.from_numpy(betas), torch.from_numpy(pose), *args)
    R = RS.batch_rodrigues(torch.from_numpy(pose).view(-1, 3))
    np.savez_compressed(os.path.join(HERE, "smpl.npz"), pose=pose, betas=betas, T=T.numpy(), v=v.numpy(), dv=dv.numpy(), verts=verts.numpy(),
                        joints=joints.numpy(), R=R.numpy())
    print("smpl: T", tuple(T.shape), "verts", tuple(verts.shape))


def make_warp_render_golden():
    """run(render_can=False, verts, faces, Ts) (models/instant_nsr.py:147-172,198-203,246-249) at render_warp.py's sampling
    (32 + 32), eval mode, with the same libigl stand-in as make_warp_goldens."""
    from tests.common import make_body
    install_igl_standin()
    verts, faces, Ts = make_body()
    net = build_reference_net()
    net.eval()
    ro, rd = make_rays(16, 16, dist=1.8, f=14.0, jitter_seed=5)
    bg = np.ones((ro.shape[0], 3), np.float32)
    res = {}
    for tag, guide in (("guide", True), ("noguide", False)):
        with torch.no_grad():
            out = net.render(torch.from_numpy(ro)[None], torch.from_numpy(rd)[None], num_steps=32, bound=1.6, upsample_steps=32, staged=False,
                             bg_color=torch.from_numpy(bg), cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, render_can=False, verts=verts,
                             faces=faces, Ts=Ts, perturb=False, use_mesh_guide=guide)
        res.update({f"{tag}_image": out["rgb"][0].numpy(), f"{tag}_weights_sum": out["weight_sum"][:, 0].numpy(), f"{tag}_depth": out["depth"][0].numpy(),
                    f"{tag}_normal_map": out["normal"].numpy(), f"{tag}_weights": out["weights"].numpy(), f"{tag}_alpha": out["pts_alpha"].numpy(),
                    f"{tag}_z_vals": out["z_vals"].numpy(), f"{tag}_gradient_error": np.float32(out["gradient_error"].item())})
        print("warp render", tag, "mean opacity", float(out["weight_sum"].mean()), "rays with opacity > 0.5:", int((out["weight_sum"] > 0.5).sum()))
    np.savez_compressed(os.path.join(HERE, "warp_render.npz"), rays_o=ro, rays_d=rd, bg=bg, **res)


if __name__ == "__main__":
    make_smpl_goldens()
    make_warp_render_golden()


# ---------------------------------------------------------------- config-1 plumbing goldens
def make_vanilla_golden():
    """get_freq_embedder(10 / 4) -> NeRF(8x256, use_viewdirs) -> ray_to_samples(16) -> raw2outputs on the 64x64 pose-0 rays of kat/rays
    (o = (0,0,1.44), f = 50), near 1, far 4, torch.manual_seed(0) default init, CPU."""
    import torch
    import models.nerf as RN
    import utils.ray_utils as RY
    _prepare_render_utils()
    import utils.render_utils as RU
    from encoder.freq_encoder import get_freq_embedder
    pe, pdim = get_freq_embedder(10)
    de, ddim = get_freq_embedder(4)
    torch.manual_seed(0)
    net = RN.NeRF(depth=8, width=256, input_ch=pdim, input_ch_views=ddim, use_viewdirs=True)
    g = np.load(os.path.join(HERE, "rays.npz"))
    ro, rd = torch.from_numpy(g["kat64_o"]), torch.from_numpy(g["kat64_d"])
    R = ro.shape[0]
    batch = dict(origin=ro, direction=rd, near=torch.full((R, 1), 1.0), far=torch.full((R, 1), 4.0))
    with torch.no_grad():
        pts, dirs, z = RY.ray_to_samples(batch, 16)
        raw = net(pe(pts.reshape(-1, 3)), de(dirs.reshape(-1, 3))).reshape(R, 16, 4)
        rgb, disp, acc, w, depth = RU.raw2outputs(raw, z, dirs[:, 0, :], white_bkg=True)
        torch.manual_seed(3)
        _, _, zp = RY.ray_to_samples(batch, 16, perturb=1.0)
        net2 = RN.NeRF(depth=4, width=32, input_ch=pdim, output_ch=4, skips=[1], scale=0.5, scale_type='tanh')
        raw2 = net2(pe(pts.reshape(-1, 3)[:64]))
    sd2 = {k: v.numpy() for k, v in net2.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, "vanilla.npz"), rgb=rgb.numpy(), disp=disp.numpy(), acc=acc.numpy(), weights=w.numpy(), depth=depth.numpy(),
                        z=z.numpy(), z_perturbed=zp.numpy(), raw_sub=raw[::64].numpy(), raw2=raw2.numpy(), n_params=np.int64(sum(p.numel() for p in net.parameters())),
                        **{"net2." + k: v for k, v in sd2.items()})
    print("vanilla: rgb mean", float(rgb.mean()), "acc mean", float(acc.mean()), "params", sum(p.numel() for p in net.parameters()))


if __name__ == "__main__":
    make_vanilla_golden()


# ---------------------------------------------------------------- training-view goldens (the stylize outer loop)
def make_path_goldens():
    """style_360_path / describe_view / the jittered pose_spherical of utils/render_utils.py:57-90,157-208 under a fixed numpy seed"""
    import numpy
    _prepare_render_utils()
    import utils.render_utils as RU
    out = {}
    c, up = np.array([0.0, 0.1, 0.0]), np.array([0.0, 1.0, 0.0])
    poses, desc = RU.style_360_path(c, up, 1.8, 20)
    out["plain_c2w"] = np.stack([p.camera_to_world for p in poses]); out["plain_desc"] = np.array(desc)
    numpy.random.seed(7)
    poses, desc = RU.style_360_path(c, up, 1.8, 20, add_noise=True, noise_scale=2.0, style_head=True, head_offset=0.423, head_rate=0.4, head_dist=0.45)
    out["noisy_c2w"] = np.stack([p.camera_to_world for p in poses]); out["noisy_desc"] = np.array(desc)
    numpy.random.seed(9)
    poses, _ = RU.default_360_path(c, up, 1.8, 8, add_noise=True)
    out["ring_noisy_c2w"] = np.stack([p.camera_to_world for p in poses])
    np.savez_compressed(os.path.join(HERE, "paths.npz"), **out)
    print("paths:", out["plain_c2w"].shape, out["noisy_c2w"].shape, out["noisy_desc"][:2], out["noisy_desc"][-1])


if __name__ == "__main__":
    make_path_goldens()


def make_dataset_ray_golden():
    """SMPLDataset.gen_rays_pose (utils/SMPLDataset.py:86-103) without the dataset files: the method only reads W, H, K, device"""
    for name in ("imageio", "scipy.ndimage"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _stub(name)
    _prepare_render_utils()
    import utils.SMPLDataset as SD
    ds = object.__new__(SD.SMPLDataset)
    ds.W = ds.H = 512; ds.device = torch.device("cpu")
    focal = .5 * 512 / np.tan(.5 * (np.pi / 3))
    ds.K = torch.from_numpy(np.array([[focal, 0, 256.0], [0, focal, 256.0], [0, 0, 1]])).cpu()
    rs = np.random.RandomState(2)
    A = rs.normal(size=(3, 3)); Q, _ = np.linalg.qr(A)
    pose = np.eye(4, dtype=np.float32); pose[:3, :3] = Q; pose[:3, 3] = [0.3, -0.2, 2.1]
    o, v = ds.gen_rays_pose(torch.from_numpy(pose), 8)
    np.savez_compressed(os.path.join(HERE, "dataset_rays.npz"), pose=pose, rays_o=o.numpy(), rays_d=v.numpy())
    print("dataset rays", tuple(v.shape))


if __name__ == "__main__":
    make_dataset_ray_golden()


def make_edge_ray_goldens():
    """NeRFRenderer.run on the rays of tests/common.py:edge_case_rays (axis-parallel, inside the cube, missing it, grazing a face)"""
    from tests.common import edge_case_rays
    net = build_reference_net()
    ro, rd = edge_case_rays()
    bg = np.ones((ro.shape[0], 3), np.float32)
    for name, train in (("eval_edge", False), ("train_edge", True)):
        c = run_case(net, ro, rd, 64, 64, train, 9, bg)
        np.savez_compressed(os.path.join(HERE, f"run_{name}.npz"), **c)
        print(name, "weights_sum", c["weights_sum"])


if __name__ == "__main__":
    make_edge_ray_goldens()


# ---------------------------------------------------------------- render_warp.calc_local_trans + convert_amass
def write_synthetic_smpl_pickle(path, seed=5):
    """a file with the layout of the licensed SMPL_NEUTRAL.pkl (keys f, v_template, shapedirs, posedirs [V,3,207], J_regressor,
    kintree_table [2,24], weights) holding BodyModel.synthetic(seed)'s buffers; V = 6890 and J = 24 because render_warp.py:184 hard-codes them"""
    import pickle
    from avatarcraft_amd.smpl import BodyModel, SMPL_PARENTS
    bm = BodyModel.synthetic(seed=seed)
    V = bm.v_template.shape[0]
    kt = np.stack([np.array(SMPL_PARENTS, np.int64), np.arange(24, dtype=np.int64)])
    kt[0, 0] = 2 ** 32 - 1                                           # the real file stores the root's parent as uint32(-1); SMPL.__init__ overwrites it
    d = dict(f=np.asarray(bm.faces, np.uint32), v_template=bm.v_template.numpy().astype(np.float64), shapedirs=bm.shapedirs.numpy().astype(np.float64),
             posedirs=bm.posedirs.numpy().T.reshape(V, 3, -1).astype(np.float64), J_regressor=bm.J_regressor.numpy().astype(np.float64),
             kintree_table=kt, weights=bm.lbs_weights.numpy().astype(np.float64))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        pickle.dump(d, f, protocol=2)
    return bm


def make_calc_local_trans_golden():
    """render_warp.calc_local_trans (render_warp.py:127-222) run on a synthetic SMPL_NEUTRAL.pkl written into a scratch directory (the
    function loads 'data/smplx/smpl' relative to the working directory), for an animation and a shape interpolation; and
    utils/convert_amass.py (a top-level script with a hard-coded path) run on a synthetic AMASS-layout .npz in the same scratch directory."""
    import runpy
    import tempfile
    _prepare_render_utils()
    for name in ("imageio", "joblib", "scipy.ndimage"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _stub(name)
    import render_warp as RW
    keep = np.concatenate([np.arange(0, 6890, 53), np.arange(6890, 6914)])        # 130 vertices + the 24 joints
    g = np.random.default_rng(21)
    poses = (g.standard_normal((3, 72)) * 0.35).astype(np.float32)
    poses[0] = 0.0                                                   # frame 0: the T pose (not the rest "da" pose: legs move)
    shape_from = np.zeros((1, 10)); shape_from[0, 1] = 2.0           # render_warp.py:36-37,42-43 defaults
    shape_to = np.zeros((1, 10)); shape_to[0, 1] = -2.0
    cwd = os.getcwd()
    out = dict(keep=keep.astype(np.int32), poses=poses, shape_from=shape_from, shape_to=shape_to)
    with tempfile.TemporaryDirectory() as tmp:
        write_synthetic_smpl_pickle(os.path.join(tmp, "data", "smplx", "smpl", "SMPL_NEUTRAL.pkl"))
        os.chdir(tmp)
        try:
            wv, Ts, n = RW.calc_local_trans(render_type="animate", poses=poses, shape_from=shape_from, shape_to=shape_to, max_frames=100)   # as main() calls it
            out["anim_world_verts"] = np.stack(wv)[:, ::53]; out["anim_Ts"] = np.stack(Ts)[:, keep]; out["anim_n"] = np.int32(n)
            wv, Ts, n = RW.calc_local_trans(render_type="interp_shape", shape_from=shape_from, shape_to=shape_to, n_interp=4, max_frames=3)
            out["shape_world_verts"] = np.stack(wv)[:, ::53]; out["shape_Ts"] = np.stack(Ts)[:, keep]; out["shape_n"] = np.int32(n)
            wv, Ts, n = RW.calc_local_trans(scale=1.25, render_type="animate", poses=poses[1:2], shape_from=shape_from, shape_to=shape_to)
            out["scaled_world_verts"] = np.stack(wv)[:, ::53]; out["scaled_Ts"] = np.stack(Ts)[:, keep]
            # AMASS ingestion: poses [F,156] (SMPL-H), betas [16], 10x temporal sub-sampling, hands zeroed
            os.makedirs(os.path.join(tmp, "path", "to", "amass")); os.makedirs(os.path.join(tmp, "data", "amass_processed"))
            amass_poses = g.standard_normal((47, 156)); amass_betas = g.standard_normal(16)
            np.savez(os.path.join(tmp, "path", "to", "amass", "*.npz"), poses=amass_poses, betas=amass_betas, trans=np.zeros((47, 3)),
                     mocap_framerate=np.float64(120.0), gender="neutral")
            runpy.run_path(os.path.join(REF, "utils", "convert_amass.py"), run_name="__main__")
            with open(os.path.join(tmp, "data", "amass_processed", "amass_rope.pkl"), "rb") as f:
                out["amass_out"] = np.load(f)
            out["amass_poses"] = amass_poses; out["amass_betas"] = amass_betas
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, "local_trans.npz"), **out)
    print("calc_local_trans: frames", int(out["anim_n"]), int(out["shape_n"]), "Ts", out["anim_Ts"].shape, out["anim_Ts"].dtype,
          "amass", out["amass_out"].shape, out["amass_out"].dtype)


if __name__ == "__main__":
    make_calc_local_trans_golden()


# ---------------------------------------------------------------- SDS guidance
def make_sds_golden():
    """models/diffusion.py:StableDiffusion imported over stub `diffusers` / `transformers` / `torchvision` modules whose classes are the tiny
    seeded stand-ins of tests/common_sd.py (no library, no weights offline): get_text_embeds and the image gradient of mannual_backward
    (resize to 512^2, VAE encode with grad, add_noise at a random t, UNet x2, CFG 100, (1 - abar_t) weight, clamp, manual backward)."""
    from tests import common_sd as SD
    import torch.nn.functional as F
    saved = {k: sys.modules.get(k) for k in ("transformers", "diffusers", "torchvision", "torchvision.transforms", "torchvision.transforms.functional", "prompt_toolkit")}
    try:
        _stub("prompt_toolkit", prompt=lambda *a, **k: "")
        _stub("transformers", CLIPTextModel=SD.TinyTextEncoder, CLIPTokenizer=SD.TinyTokenizer, logging=types.SimpleNamespace(set_verbosity_error=lambda: None))
        _stub("diffusers", AutoencoderKL=SD.TinyVAE, UNet2DConditionModel=SD.TinyUNet, PNDMScheduler=SD.StubPNDMScheduler)

        def pad(img, padding, fill=0, padding_mode="constant"):            # torchvision.transforms.functional.pad for a 1-tuple: all four sides
            p = padding[0]
            return F.pad(img, (p, p, p, p), mode=padding_mode, value=fill)
        tvf = _stub("torchvision.transforms.functional", pad=pad)
        tvt = _stub("torchvision.transforms", functional=tvf)
        _stub("torchvision", transforms=tvt)
        sys.modules.pop("models.diffusion", None)
        import models.diffusion as RD
        sd = RD.StableDiffusion(torch.device("cpu"), "1.5")
        emb = sd.get_text_embeds(["Hulk, photorealistic style"])
        out = dict(text_embeds=emb.numpy(), alphas_cumprod=sd.alphas.numpy()[::50])
        g = torch.Generator().manual_seed(5)
        for seed, hw in ((11, (64, 64)), (12, (48, 80))):
            pred = torch.rand((1, 3) + hw, generator=g)
            pred.requires_grad_(True)
            torch.manual_seed(seed)
            sd.mannual_backward(emb, pred, 100)
            torch.manual_seed(seed)
            t = torch.randint(sd.min_step, sd.max_step + 1, [1])
            out[f"rgb_{seed}"] = pred.detach().numpy(); out[f"grad_{seed}"] = pred.grad.numpy().copy(); out[f"t_{seed}"] = np.int64(t.item())
            print("sds golden: seed", seed, "t", int(t), "|grad| max", float(pred.grad.abs().max()), "nonzero", float((pred.grad != 0).float().mean()))
        np.savez_compressed(os.path.join(HERE, "sds.npz"), **out)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        sys.modules.pop("models.diffusion", None)


if __name__ == "__main__":
    make_sds_golden()


# ---------------------------------------------------------------- reconstruct.py step
def make_reconstruct_golden():
    """one optimisation step of main_reconstruct (reconstruct.py:92-112): render with perturbation, smooth_l1(rgb, gt) + 0.1 * eikonal,
    backward on the reference's own autograd graph, Adam(lr 5e-4, betas (0.9, 0.99), eps 1e-15): gradients and parameter deltas"""
    import torch.nn.functional as F
    net = build_reference_net()
    net.train(True)
    ro, rd = make_rays(6, 6, dist=1.8, f=4.0, jitter_seed=4)
    n = ro.shape[0]
    bg = np.ones((n, 3), np.float32)
    gt = np.random.RandomState(15).uniform(0, 1, (n, 3)).astype(np.float32)
    torch.manual_seed(21)
    noise = torch.rand(n, 64).numpy().copy()
    torch.manual_seed(21)
    out = net.render(torch.from_numpy(ro)[None], torch.from_numpy(rd)[None], num_steps=64, bound=1.6, upsample_steps=64, staged=False,
                     bg_color=torch.from_numpy(bg), cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, render_can=True, perturb=True)
    loss = F.smooth_l1_loss(out["rgb"][0], torch.from_numpy(gt), reduction='mean') + out["gradient_error"] * 0.1
    net.zero_grad()
    loss.backward()
    gg = dict(rays_o=ro, rays_d=rd, gt=gt, noise=noise, loss=np.float64(loss.item()), rgb=out["rgb"][0].detach().numpy())
    ge = net.encoder.embeddings.grad.numpy()
    nz = np.flatnonzero(np.abs(ge).sum(1))
    pick = np.sort(nz[np.random.RandomState(16).choice(len(nz), 4096, replace=False)])
    gg["emb_idx"] = pick.astype(np.int64); gg["emb_grad"] = ge[pick].copy(); gg["emb_l2"] = np.float64(np.sqrt((ge.astype(np.float64) ** 2).sum()))
    before = {k: p.detach().clone() for k, p in net.named_parameters()}
    for k, p in net.named_parameters():
        if k != "encoder.embeddings":
            gg["grad." + k] = p.grad.numpy().copy()
    torch.optim.Adam(net.parameters(), lr=5e-4, betas=(0.9, 0.99), eps=1e-15).step()
    for k, p in net.named_parameters():
        d = (p.detach() - before[k]).numpy()
        gg["adam_delta." + ("emb" if k == "encoder.embeddings" else k)] = d[pick].copy() if k == "encoder.embeddings" else d.copy()
    np.savez_compressed(os.path.join(HERE, "reconstruct_grad.npz"), **gg)
    print("reconstruct golden: loss", gg["loss"], "emb l2", gg["emb_l2"])


if __name__ == "__main__":
    make_reconstruct_golden()


# ---------------------------------------------------------------- occupancy grid + SDF volume
def make_density_golden():
    """NeRFRenderer.update_extra_state (models/instant_nsr.py:303-356) of a cuda_ray=True reference net (the golden weights) and
    extract_fields (:728-745; extract_geometry's marching cubes needs the absent PyMCubes).  The 129^3 grid is stored at every 4th index."""
    torch.manual_seed(0)
    net = ref_nsr.NeRFNetwork(cuda_ray=True)
    src = build_reference_net()
    net.load_state_dict(src.state_dict(), strict=False)
    net.eval()
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        net.update_extra_state(1.6)
        g1 = net.density_grid.numpy().copy(); m1 = net.mean_density
        with torch.no_grad():
            net.sdf_net[1].bias[0] += 0.1                              # a second update on a changed field: the decay / maximum merge
        net.update_extra_state(1.6)
    g2 = net.density_grid.numpy().copy()
    u = ref_nsr.extract_fields(torch.tensor([-1.6] * 3), torch.tensor([1.6] * 3), 33, lambda pts: src.density(pts, 1.6).detach())
    np.savez_compressed(os.path.join(HERE, "density_grid.npz"), grid1=g1[::4, ::4, ::4], grid2=g2[::4, ::4, ::4], mean1=np.float64(m1), mean2=np.float64(net.mean_density),
                        max1=np.float32(g1.max()), nnz1=np.int64((g1 > 1e-3).sum()), iter_density=np.int64(net.iter_density), sdf33=u)
    print("density grid: mean", m1, net.mean_density, "max", g1.max(), "cells > 1e-3:", int((g1 > 1e-3).sum()), "sdf33 range", u.min(), u.max())


if __name__ == "__main__":
    make_density_golden()
