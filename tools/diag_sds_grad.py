#!/usr/bin/env python3
"""GPU diagnostic: the three-term training gradient of stylize.sds_step (HIP) against the oracle's fp64 backward and the reference golden,
per tensor and, for the hash table, per level.   python tools/diag_sds_grad.py tests/golden/train_grad.npz [more.npz ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tests.common import load_golden, oracle_field_from_golden
from tests.test_gpu_model import golden_net, DEV
from tests.test_oracle_backward import _chain_to_raw
from oracle import oracle as O
from avatarcraft_amd.stylize import sds_step, flat_grad_view

p = load_golden("nsr_params.npz")
offs = np.asarray(p["offsets"], np.int64)
for path in sys.argv[1:]:
    g = dict(np.load(path))
    net, _ = golden_net(train=True)
    net_gt, _ = golden_net(train=False)
    with torch.no_grad():
        net_gt.sdf_net[1].bias[0] = float(g["gt_sdf_bias"])
    ro, rd = torch.from_numpy(g["rays_o"]).to(DEV), torch.from_numpy(g["rays_d"]).to(DEV)
    n = ro.shape[0]
    img_grad = torch.from_numpy(g["img_grad"]).to(DEV)
    guidance = lambda img: img_grad.reshape(1, n, 1, 3).permute(0, 3, 1, 2).contiguous()
    opt = torch.optim.SGD(net.parameters(), lr=0.0)
    flat = flat_grad_view(net.parameters())
    orig = torch.rand
    torch.rand = lambda *a, **k: torch.from_numpy(g["noise"]).to(DEV)
    try:
        sds_step(net, net_gt, ro, rd, (n, 1), opt, guidance, batch_size=4096, w_eikonal=0.01, use_opacity=True, flat_grad=flat)
    finally:
        torch.rand = orig
    hip = {k: q.grad.detach().cpu().numpy().astype(np.float64) for k, q in net.named_parameters()}
    f = oracle_field_from_golden(p)
    pred, gt = g["opacity_pred"].astype(np.float64), g["opacity_gt"].astype(np.float64)
    d = np.clip(pred, 0, 1) - gt
    g_ws = np.where(np.abs(d) < 1.0, d, np.sign(d)) * (1e5 / n) * ((pred >= 0) & (pred <= 1))
    r = O.render_core_backward(f, g["rays_o"], g["rays_d"], g["z_vals"], 64, 64, 1.6, float(p["inv_s"]), bg=g["bg"], g_image=g["img_grad"], g_weights_sum=g_ws, g_eik=0.01)
    raw = _chain_to_raw(O, p, r)
    rep = {"file": path, "rays": int(n)}
    for k, v in raw.items():
        ref = g["grad3." + k].astype(np.float64); sc = np.abs(ref).max()
        rep[k] = dict(hip_vs_oracle=float(np.abs(hip[k].reshape(ref.shape) - np.asarray(v).reshape(ref.shape)).max() / sc),
                      oracle_vs_ref=float(np.abs(np.asarray(v).reshape(ref.shape) - ref).max() / sc), hip_vs_ref=float(np.abs(hip[k].reshape(ref.shape) - ref).max() / sc))
    ht, ot = hip["encoder.embeddings"], r["g_table"]
    sc = np.abs(ot).max()
    rep["table"] = dict(hip_vs_oracle=float(np.abs(ht - ot).max() / sc), max=float(sc),
                        by_level=[dict(level=l, max=float(np.abs(ot[offs[l]:offs[l + 1]]).max()), err=float(np.abs(ht[offs[l]:offs[l + 1]] - ot[offs[l]:offs[l + 1]]).max()))
                                  for l in range(16)],
                        sub_hip_vs_ref=float(np.abs(ht[g["emb_idx"]] - g["emb_grad3"]).max() / np.abs(g["emb_grad3"]).max()),
                        sub_oracle_vs_ref=float(np.abs(ot[g["emb_idx"]] - g["emb_grad3"]).max() / np.abs(g["emb_grad3"]).max()))
    print(json.dumps(rep, indent=1))
