#!/usr/bin/env python3
"""Repeat-launch determinism probe of the fused renderer (GPU box):

    AC_LIB_PATH=tools/_bin/lib_<variant>.so python tools/determinism_probe.py [--reps 20] [--precision fast] [--view bench|sds]

Renders the same rays `reps` times through the dynamic ray hand-out with every optional output kept (per-sample arrays, sample indices,
the 7 x 32 stencil features of every sample) and compares each repeat with the first, bit for bit.  Differences are decoded down to
(ray, sample, stencil point, hash level) so that a timing-dependent value can be traced to the code that produced it."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--precision", default="fast")
    ap.add_argument("--view", default="bench")
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--out", default=None)
    ap.add_argument("--save-ref", default=None, help="write the first launch's stencil features / sdf to this .pt file (run with the trusted build)")
    ap.add_argument("--ref", default=None, help="compare every launch with the outputs saved by --save-ref (another build's)")
    a = ap.parse_args()
    from avatarcraft_amd import nsr_ops
    from tests.common import load_golden, make_rays
    from tests.gpu_common import device_field
    import bench
    dev = torch.device("cuda:0")
    p = load_golden("nsr_params.npz")
    field, _ = device_field(p, device=dev)
    field.prepare()
    if a.view == "bench":
        ro, rd = make_rays(256, 256, dist=1.7, f=200.0, yaw=0.0, pitch=0.0)
        ro, rd = ro[:a.rays], rd[:a.rays]
    else:
        ro, rd = bench.sds_view(0)
    ro, rd = torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev)
    N = ro.shape[0]
    noise = torch.rand((N, 64), generator=torch.Generator().manual_seed(1)).to(dev) if a.view == "sds" else None
    def logical(out):
        """feat7 as the kernels store it, [tile][14][lane = n + 16 g][4] with float k = 8 e + q at [k / 4][lane][k % 4] -> [e][q][sample][g]"""
        f7 = out["feat7"]
        nt = f7.shape[0]
        out["feat7"] = f7.view(nt, 14, 4, 16, 4).permute(1, 4, 0, 3, 2).reshape(7, 8, nt * 16, 4).contiguous()
        return out
    run_raw = lambda: nsr_ops.render_rays(field, ro, rd, 64, 64, 1.6, float(p["inv_s"]), noise=noise, extras=True, train_extras=True, debug_indices=True,
                                      precision=a.precision)
    run = lambda: logical(run_raw())
    first = {k: v.clone() for k, v in run().items() if isinstance(v, torch.Tensor)}
    torch.cuda.synchronize()
    report = {"lib": os.environ.get("AC_LIB_PATH", "head"), "precision": a.precision, "view": a.view, "reps": a.reps, "differing_repeats": 0, "keys": {}}
    T = 128
    if a.save_ref:
        torch.save({k: first[k].cpu() for k in ("feat7", "sdf", "z_vals", "gradient")}, a.save_ref)

    def hist(d4):            # d4: bool [7, 8, N*T, 4] -> counts by stencil point, by level, by lane-in-tile
        by_e = d4.sum(dim=(1, 2, 3)).tolist()
        lv = torch.zeros(16, dtype=torch.long)
        for q in range(8):
            for g in range(4):
                lv[4 * (q >> 1) + g] += int(d4[:, q, :, g].sum())
        by_n = d4.reshape(7, 8, -1, 16, 4).sum(dim=(0, 1, 2, 4)).tolist()
        by_g = d4.sum(dim=(0, 1, 2)).tolist()
        return {"by_point": by_e, "by_level": lv.tolist(), "by_lane_in_tile_n": by_n, "by_lane_group_g": by_g}
    if a.ref:
        ref = torch.load(a.ref)
        rf = ref["feat7"].to(dev)
        report["first_launch_vs_ref_centre_features"] = hist((first["feat7"] != rf) & (torch.arange(7, device=dev) == 0)[:, None, None, None])
        report["first_launch_vs_ref_sdf_differs"] = int((first["sdf"] != ref["sdf"].to(dev)).sum())
        dc = (first["feat7"][0] - rf[0]).abs()
        report["first_launch_vs_ref_centre_maxabs"] = float(dc.max())
        report["first_launch_vs_ref_centre_n_above_1e-6"] = int((dc > 1e-6).sum())
        report["vs_ref_centre_by_repeat"] = []
    for r in range(1, a.reps):
        out = run()
        torch.cuda.synchronize()
        bad = False
        if a.ref:
            report["vs_ref_centre_by_repeat"].append(int((out["feat7"][0] != rf[0]).sum()))
        if r == 1 and not torch.equal(first["feat7"], out["feat7"]):
            report["feat7_repeat1_vs_first"] = hist(first["feat7"] != out["feat7"])
        for k, v in first.items():
            w = out[k]
            if torch.equal(v, w):
                continue
            bad = True
            ne = (v != w) & ~(torch.isnan(v) & torch.isnan(w)) if v.dtype.is_floating_point else (v != w)
            cnt = int(ne.sum())
            if cnt == 0:
                continue
            e = report["keys"].setdefault(k, {"repeats": 0, "max_count": 0, "max_abs": 0.0, "examples": []})
            e["repeats"] += 1; e["max_count"] = max(e["max_count"], cnt)
            if v.dtype.is_floating_point:
                e["max_abs"] = max(e["max_abs"], float((v - w).abs()[ne].max()))
            if len(e["examples"]) < 6:
                idx = ne.nonzero()[:4].tolist()
                for i in idx:
                    ex = {"rep": r, "index": i, "a": float(v[tuple(i)]), "b": float(w[tuple(i)])}
                    if k == "feat7":            # [e][q = 2 j + c][sample][g]: level 4 j + g, channel c
                        ee, q, s, g = i
                        ex.update(point=ee, level=4 * (q >> 1) + g, channel=q & 1, ray=s // T, sample=s % T)
                    e["examples"].append(ex)
        report["differing_repeats"] += int(bad)
    if "feat7" in report["keys"]:               # histogram of the differing features by (stencil point, level)
        out = run(); torch.cuda.synchronize()
    print(json.dumps(report, indent=1)[:6000])
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(report, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
