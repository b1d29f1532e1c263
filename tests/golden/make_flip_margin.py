#!/usr/bin/env python3
"""Show that the ONE recorded searchsorted flip of the run goldens (run_eval_edge.npz: ray 9, up-sampling iteration 3, sample 14) is a true
knife-edge of the reference's own arithmetic, and store the evidence in the fixture.

    python tests/golden/make_flip_margin.py [/root/reference]

The reference's `up_sample` / `sample_pdf` (models/instant_nsr.py:410-459, :21-55) are dtype-agnostic torch code.  The script runs the reference's
render on the edge-case rays (fp32, as recorded), captures the arguments of the iteration-3 `up_sample` call, and calls THE REFERENCE'S OWN
up_sample again on those very inputs cast to float64 (default dtype float64 for that call, so its linspace / zeros / ones are double too),
capturing the `cdf` and `u` that reach torch.searchsorted both times.  `inds = searchsorted(cdf, u, right=True)` counts the entries with
cdf[j] <= u; the reference (fp32, torch CPU / Sleef) and the oracle (fp32, ac_math) disagree on exactly one comparison `cdf[k] <= u[14]`.
Stored next to `oracle_ss_flips`:
    flip_bin          k
    flip_u            u[14] = 0.90625 (exact in fp32)
    flip_cdf_ref_f32  the reference's fp32 cdf[k]
    flip_cdf_f64      the same quantity evaluated by the reference's code in fp64 from the same fp32 inputs
    flip_margin_f64   flip_cdf_f64 - flip_u
    flip_ulp_f32      spacing of fp32 numbers at flip_u (2^-24 * 2 for [0.5, 1))
A flip is legitimate when |flip_margin_f64| is within the rounding error of an fp32 evaluation of the cdf (a cumsum of ~112 terms of magnitude
<= 1: a few ulp), i.e. when exact arithmetic cannot tell which side is right for fp32 inputs; the parity test asserts that."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG          # noqa: E402  (stubs + `import models.instant_nsr as ref_nsr`)
import numpy as np                # noqa: E402
import torch                      # noqa: E402

from tests.common import edge_case_rays    # noqa: E402


def main():
    path = os.path.join(HERE, "run_eval_edge.npz")
    g = dict(np.load(path))
    (ray, it, smp), = g["oracle_ss_flips"].tolist()
    net = MG.build_reference_net().eval()
    ro, rd = edge_case_rays()
    assert np.array_equal(ro, g["rays_o"]) and np.array_equal(rd, g["rays_d"])
    calls, ss_args = [], []
    orig_up = MG.ref_nsr.NeRFRenderer.up_sample
    orig_ss = torch.searchsorted

    def up(self, rays_o, rays_d, z_vals, sdf, n_importance, inv_s):
        calls.append((rays_o.clone(), rays_d.clone(), z_vals.clone(), sdf.clone(), n_importance, inv_s))
        return orig_up(self, rays_o, rays_d, z_vals, sdf, n_importance, inv_s)

    def ss(cdf, u, **k):
        ss_args.append((cdf.clone(), u.clone())); return orig_ss(cdf, u, **k)
    MG.ref_nsr.NeRFRenderer.up_sample = up
    torch.searchsorted = ss
    try:
        with torch.no_grad():
            net.render(torch.from_numpy(ro)[None], torch.from_numpy(rd)[None], num_steps=64, bound=1.6, upsample_steps=64, staged=False,
                       bg_color=torch.from_numpy(g["bg"]), cos_anneal_ratio=1.0, normal_epsilon_ratio=0.0, render_can=True, perturb=False)
        MG.ref_nsr.NeRFRenderer.up_sample = orig_up
        cdf32, u32 = ss_args[it]
        ref_idx = int(orig_ss(cdf32, u32, right=True)[ray, smp])
        assert ref_idx == int(g["ss_inds"][ray, it, smp]), "the fixture's recorded index is the reference's"
        ss_args.clear()
        a = calls[it]
        torch.set_default_dtype(torch.float64)
        with torch.no_grad():
            orig_up(net, a[0].double(), a[1].double(), a[2].double(), a[3].double(), a[4], a[5])
        torch.set_default_dtype(torch.float32)
        cdf64, u64 = ss_args[0]
    finally:
        torch.searchsorted = orig_ss
        MG.ref_nsr.NeRFRenderer.up_sample = orig_up
        torch.set_default_dtype(torch.float32)
    assert cdf64.dtype == torch.float64 and float(u64[ray, smp]) == float(u32[ray, smp])
    idx64 = int(orig_ss(cdf64, u64, right=True)[ray, smp])
    # the oracle's index differs from the reference's by one: the disputed comparison is at the smaller of the two
    from oracle import oracle as O
    r = O.render_rays(MG._ORACLE_FIELD if "_ORACLE_FIELD" in MG.__dict__ else
                      O.Field(net.encoder.embeddings.detach().numpy(), net.encoder.offsets.numpy(),
                              *[MG.effective_weights(net)[k] for k in ("W1", "b1", "W2", "b2", "Wc1", "Wc2", "Wc3")], float(net.encoder.per_level_scale)),
                      ro, rd, 64, 64, 1.6, float(net.forward_variance().item()), bg=g["bg"])
    orc_idx = int(r["ss_inds"][ray, it, smp])
    assert abs(orc_idx - ref_idx) == 1
    # the ORACLE's inputs of the same up_sample call: its 112 z values before the last merge (sort_index of iteration 3 says which of the final 128
    # entries are the 16 new ones) and its own fp32 sdf at them; the reference's up_sample in fp64 on THOSE inputs lands on the oracle's side
    old = r["sort_index"][ray, it, :128] < 112
    z_orc = r["z_vals"][ray][old]
    fld = MG._ORACLE_FIELD if "_ORACLE_FIELD" in MG.__dict__ else None
    if fld is None:
        fld = O.Field(net.encoder.embeddings.detach().numpy(), net.encoder.offsets.numpy(),
                      *[MG.effective_weights(net)[k] for k in ("W1", "b1", "W2", "b2", "Wc1", "Wc2", "Wc3")], float(net.encoder.per_level_scale))
    pts = np.clip((ro[ray][None, :] + rd[ray][None, :] * z_orc[:, None]).astype(np.float32), -1.6, 1.6)
    sdf_orc = fld.sdf(pts, 1.6)[:, 0]
    z_ref, sdf_ref = a[2][ray].numpy(), a[3].reshape(a[2].shape)[ray].numpy()
    ss_args.clear()
    torch.searchsorted = ss
    torch.set_default_dtype(torch.float64)
    try:
        with torch.no_grad():
            orig_up(net, a[0][ray:ray + 1].double(), a[1][ray:ray + 1].double(), torch.from_numpy(z_orc)[None].double(), torch.from_numpy(sdf_orc)[None].double(),
                    a[4], a[5])
    finally:
        torch.set_default_dtype(torch.float32)
        torch.searchsorted = orig_ss
    cdf64_o, u64_o = ss_args[0]
    idx64_o = int(orig_ss(cdf64_o, u64_o, right=True)[0, smp])
    k = min(orc_idx, ref_idx)
    u = float(u32[ray, smp])
    out = dict(flip_bin=np.int32(k), flip_u=np.float64(u), flip_cdf_ref_f32=np.float32(cdf32[ray, k]), flip_cdf_f64=np.float64(cdf64[ray, k]),
               flip_margin_f64=np.float64(float(cdf64[ray, k]) - u), flip_ulp_f32=np.float64(np.spacing(np.float32(u))),
               flip_ref_index=np.int32(ref_idx), flip_oracle_index=np.int32(orc_idx), flip_f64_index=np.int32(idx64),
               flip_neighbours_f64=cdf64[ray, k - 1:k + 2].numpy().astype(np.float64),
               # the same fp64 evaluation from the oracle's own fp32 inputs of that call, and how far the two sets of inputs are apart
               flip_cdf_f64_oracle_inputs=np.float64(cdf64_o[0, k]), flip_f64_index_oracle_inputs=np.int32(idx64_o),
               flip_inputs_max_dz=np.float64(np.abs(z_ref.astype(np.float64) - z_orc).max()),
               flip_inputs_max_dsdf=np.float64(np.abs(sdf_ref.astype(np.float64) - sdf_orc).max()),
               flip_inv_s=np.float64(a[5]))
    for kk, v in out.items():
        print(kk, v)
    g.update(out)
    np.savez(path, **g)


if __name__ == "__main__":
    main()
