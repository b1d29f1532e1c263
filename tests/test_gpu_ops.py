"""-m gpu: the stand-alone operators (hash encoder, SH encoder, raymarching) through the reference-shaped
Python surface (-> C ABI -> HIP), against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from tests.common import load_golden, make_table
from tests.gpu_common import assert_bitwise

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(DEV) if dtype is None else t.to(DEV, dtype)


# ------------------------------------------------------------------ hash encoder
@pytest.mark.parametrize("D,C,L,base,log2T,B", [(3, 2, 16, 16, 19, 5000), (3, 4, 8, 4, 12, 777), (2, 2, 6, 8, 10, 300),
                                                (3, 1, 4, 16, 14, 129), (3, 8, 3, 4, 8, 64), (2, 1, 2, 2, 6, 1)])
def test_hash_forward_backward(oracle, D, C, L, base, log2T, B):
    from avatarcraft_amd.encoder.hashencoder.backend import _backend
    O = oracle
    desired = 2048 if L == 16 else None
    offsets, pls = O.hash_offsets(D, L, C, 1.5, base, log2T, desired)
    S = np.float32(np.log2(pls))
    rs = np.random.RandomState(B)
    grid = rs.uniform(-1, 1, (int(offsets[-1]), C)).astype(np.float32)
    x = rs.uniform(0, 1, (B, D)).astype(np.float32)
    x[0] = 1.0
    if B > 4:
        x[1] = 0.0; x[2, 0] = -0.1; x[3, -1] = 1.5        # out-of-range rows -> zeros
    out_o, dd_o, ci_o = O.hash_encode_forward(x, grid, offsets, S, base, True, True)
    xt, gt, ot = T(x), T(grid), T(offsets)
    out = torch.empty(L, B, C, device=DEV); dd = torch.empty(B, L * D * C, device=DEV)
    _backend.hash_encode_forward(xt, gt, ot, out, B, D, C, L, S, base, True, dd)
    assert_bitwise(out, out_o, "hash outputs")
    assert_bitwise(dd, dd_o, "hash dy_dx")
    # corner indices: bit-exact
    from avatarcraft_amd import _lib as Lb
    ci = torch.empty(L, B, 1 << D, dtype=torch.int32, device=DEV)
    Lb.check(Lb.lib().ac_hash_corner_indices(xt.data_ptr(), offsets.ctypes.data, ci.data_ptr(), B, D, L, float(S), base,
                                             Lb.current_stream()))
    assert np.array_equal(ci.cpu().numpy().view(np.uint32), ci_o)
    # backward: atomics => order-free accumulation, compare with a tolerance
    g = rs.normal(0, 1, (L, B, C)).astype(np.float32)
    gg_o, gi_o = O.hash_encode_backward(g, x, grid, offsets, S, base, dd_o)
    gg = torch.zeros_like(gt); gi = torch.zeros_like(xt)
    _backend.hash_encode_backward(T(g), xt, gt, ot, gg, B, D, C, L, S, base, True, dd, gi)
    np.testing.assert_allclose(gg.cpu().numpy(), gg_o, rtol=2e-5, atol=2e-5)
    assert_bitwise(gi, gi_o, "hash grad_inputs")


def test_hash_float_output_vs_independent_witness(golden_params):
    """HIP hash kernels (stand-alone forward, direct-atomic and binned backward, the stencil operator's centre point) against the numpy-fp64
    witness of tests/golden/field_points.npz, which was computed from hashencoder.cu's formulas without any call into oracle/"""
    from avatarcraft_amd.encoder.hashencoder import backend as BK
    from avatarcraft_amd.encoder.hashencoder.hashgrid import HashEncoder
    from tests.test_oracle_golden import hash_witness_inputs, check_hash_vs_witness
    fp = load_golden("field_points.npz")
    table, x01, g = hash_witness_inputs(fp, golden_params)
    B = x01.shape[0]
    S = np.float32(np.log2(float(golden_params["per_level_scale"])))
    xt, tt, ot = T(x01), T(table), T(golden_params["offsets"].astype(np.int32))
    out = torch.empty(16, B, 2, device=DEV)
    BK._backend.hash_encode_forward(xt, tt, ot, out, B, 3, 2, 16, S, 16, False, torch.empty(1, device=DEV))
    enc = out.permute(1, 0, 2).reshape(B, 32).cpu().numpy()
    for binned in (True, False):
        old = BK.BINNED_SCATTER
        BK.BINNED_SCATTER = binned
        try:
            gg = torch.zeros_like(tt)
            BK._backend.hash_encode_backward(T(g), xt, tt, ot, gg, B, 3, 2, 16, S, 16, False, torch.empty(1, device=DEV), torch.empty(1, device=DEV))
        finally:
            BK.BINNED_SCATTER = old
        check_hash_vs_witness(enc, gg.cpu().numpy(), fp)
    # the module surface (HashEncoder.forward with size = bound) and the centre point of the 7-point stencil operator
    he = HashEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048).to(DEV)
    with torch.no_grad():
        he.embeddings.copy_(tt)
    pts = T(fp["pts"])
    y = he(pts, 1.6)
    (y * T(fp["witness_grad"])).sum().backward()
    check_hash_vs_witness(y.detach().cpu().numpy(), he.embeddings.grad.cpu().numpy(), fp)
    h7 = he.forward_stencil(pts, 1.6, 0.005)
    assert np.abs(h7[0].detach().cpu().numpy().astype(np.float64) - fp["enc_witness"]).max() < 1e-6


def test_hash_module_autograd_and_errors(oracle):
    from avatarcraft_amd.encoder import get_encoder
    enc, dim = get_encoder("hashgrid", dict(in_dim=3, hash_num_levels=8, hash_level_dim=2, hash_per_level_scale=2.0,
                                            hash_base_resolution=4, hash_log2_hashmap_size=12, hash_desired_resolution=None))
    enc = enc.to(DEV)
    with torch.no_grad():
        enc.embeddings.uniform_(-1, 1)
    x = torch.rand(1000, 3, device=DEV) * 2 - 1
    y = enc(x, size=1.0)
    assert y.shape == (1000, 16)
    off = enc.offsets.cpu().numpy()
    yo, _, _ = oracle.hash_encode_forward(((x.cpu().numpy() + np.float32(1.0)) / np.float32(2.0)), enc.embeddings.detach().cpu().numpy(),
                                          off, np.float32(np.log2(2.0)), 4)
    assert_bitwise(y, yo.transpose(1, 0, 2).reshape(1000, 16), "HashEncoder.forward")
    y.sum().backward()
    assert enc.embeddings.grad is not None and float(enc.embeddings.grad.abs().sum()) > 0
    # empty and ragged batches
    assert enc(torch.zeros(0, 3, device=DEV)).shape == (0, 16)
    assert enc(torch.rand(7, 5, 3, device=DEV)).shape == (7, 5, 16)
    from avatarcraft_amd.encoder.hashencoder.backend import _backend
    with pytest.raises(RuntimeError):      # C = 3 is not a supported level_dim ("GridEncoding: C must be 1, 2, 4, or 8.")
        _backend.hash_encode_forward(x, torch.zeros(100, 3, device=DEV), enc.offsets, torch.empty(8, 1000, 3, device=DEV), 1000, 3, 3, 8,
                                     1.0, 4, False, torch.empty(1, device=DEV))
    with pytest.raises(RuntimeError):      # CPU tensor
        _backend.hash_encode_forward(x.cpu(), enc.embeddings, enc.offsets, y, 1000, 3, 2, 8, 1.0, 4, False, torch.empty(1, device=DEV))


# ------------------------------------------------------------------ SH encoder
@pytest.mark.parametrize("degree", [1, 2, 4, 6, 8])
def test_sh_forward_backward(oracle, degree):
    from avatarcraft_amd.encoder.shencoder.backend import _backend
    rs = np.random.RandomState(degree)
    x = rs.normal(0, 1, (513, 3)).astype(np.float32)
    x[:256] /= np.linalg.norm(x[:256], axis=1, keepdims=True)     # unit directions and raw (non-unit) inputs
    out_o, dd_o = oracle.sh_encode_forward(x, degree, True)
    xt = T(x); out = torch.empty(513, degree ** 2, device=DEV); dd = torch.empty(513, 3 * degree ** 2, device=DEV)
    _backend.sh_encode_forward(xt, out, 513, 3, degree, True, dd)
    assert_bitwise(out, out_o, "sh outputs"); assert_bitwise(dd, dd_o, "sh dy_dx")
    g = rs.normal(0, 1, (513, degree ** 2)).astype(np.float32)
    gi_o = oracle.sh_encode_backward(g, x, degree, dd_o)
    gi = torch.zeros_like(xt)
    _backend.sh_encode_backward(T(g), xt, 513, 3, degree, dd, gi)
    assert_bitwise(gi, gi_o, "sh grad_inputs")


def test_sh_kernel_vs_reference_polynomials():
    """the HIP kernel itself (not via the oracle) against the reference's polynomial lines: tests/golden/kat_sh.npz (make_kat_sh.py), all 64 channels and
    their 192 derivatives -- same tolerances as tests/test_oracle_golden.py::test_sh_table_vs_reference_polynomials"""
    from avatarcraft_amd.encoder.shencoder.backend import _backend
    from tests.common import load_golden
    g = load_golden("kat_sh.npz")
    xt = T(g["dirs"]); out = torch.empty(256, 64, device=DEV); dd = torch.empty(256, 3 * 64, device=DEV)
    _backend.sh_encode_forward(xt, out, 256, 3, 8, True, dd)
    assert np.abs(out.cpu().numpy() - g["values"]).max() <= 1e-5
    jac = g["jacobian"]
    assert (np.abs(dd.cpu().numpy().reshape(256, 3, 64) - jac) / np.maximum(1.0, np.abs(jac))).max() <= 2e-5


def test_sh_module(oracle):
    from avatarcraft_amd.encoder import get_encoder
    enc, dim = get_encoder("sphere_harmonics", dict(in_dim=3))
    assert dim == 16
    d = torch.randn(100, 3, device=DEV, requires_grad=True)
    y = enc(d)
    y.square().sum().backward()
    # analytic Jacobian vs central differences of the oracle (fp64-ish check of dy_dx)
    x = d.detach().cpu().numpy()
    _, dd = oracle.sh_encode_forward(x, 4, True)
    eps = 1e-3
    for a in range(3):
        xp, xm = x.copy(), x.copy(); xp[:, a] += eps; xm[:, a] -= eps
        fd = (oracle.sh_encode_forward(xp, 4)[0].astype(np.float64) - oracle.sh_encode_forward(xm, 4)[0]) / (2 * eps)
        assert np.abs(fd - dd.reshape(100, 3, 16)[:, a]).max() < 5e-2
    with pytest.raises(AssertionError):
        from avatarcraft_amd.encoder.shencoder import SHEncoder
        SHEncoder(3, 9)


# ------------------------------------------------------------------ raymarching
def _sphere_grid(H=129, bound=1.6, r=0.5):
    ax = np.linspace(-bound, bound, H, dtype=np.float32)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    return (100.0 * ((X ** 2 + Y ** 2 + Z ** 2) < r * r)).astype(np.float32)


def test_march_rays_train_kat_and_oracle(oracle):
    """SURVEY A.4 known answer + exact equality with the oracle (packed layout, ray triples, samples)."""
    import avatarcraft_amd.raymarching as RM
    import json, os
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))["march_rays_train"]
    r = load_golden("rays.npz"); o, d = r["kat64_o"], r["kat64_d"]
    grid = _sphere_grid()
    for perturb in (0, 1):
        xo, do_, dlo, ro, co = oracle.march_rays_train(o, d, grid, float(grid.mean()), 1.6, perturb=perturb)
        counter = torch.zeros(2, dtype=torch.int32, device=DEV)
        x, dd, dl, rays = RM.march_rays_train(T(o), T(d), 1.6, T(grid), float(grid.mean()), 0, counter, -1, bool(perturb), -1, True)
        torch.cuda.synchronize()
        assert counter.cpu().tolist() == co.tolist()
        assert np.array_equal(rays.cpu().numpy(), ro)
        m = int(co[0])
        assert_bitwise(x[:m], xo[:m], "xyzs"); assert_bitwise(dd[:m], do_[:m], "dirs"); assert_bitwise(dl[:m], dlo[:m], "deltas")
        if not perturb:
            assert co.tolist() == kat["counter"]
            assert int((ro[:, 2] > 0).sum()) == kat["rays_hit"] and int(ro[:, 2].max()) == kat["max_steps"]
            assert ro[2080].tolist() == kat["centre_ray"]


def test_composite_train_forward_backward(oracle):
    import avatarcraft_amd.raymarching as RM
    rs = np.random.RandomState(3)
    N = 300
    steps = rs.randint(0, 60, N).astype(np.int32); steps[5] = 0
    offs = np.concatenate([[0], np.cumsum(steps)[:-1]]).astype(np.int32)
    M = int(steps.sum()) + 1
    perm = rs.permutation(N).astype(np.int32)
    rays = np.stack([perm, offs, steps], 1).astype(np.int32)
    sig = rs.uniform(0, 0.4, M).astype(np.float32); rgb = rs.uniform(0, 1, (M, 3)).astype(np.float32)
    dl = rs.uniform(0.001, 0.01, M).astype(np.float32)
    ws_o, img_o = oracle.composite_rays_train_forward(sig, rgb, dl, rays)
    sg = T(sig).requires_grad_(True); rg = T(rgb).requires_grad_(True)
    ws, img = RM.composite_rays_train(sg, rg, T(dl), T(rays), 1.6)
    assert_bitwise(ws, ws_o, "weights_sum"); assert_bitwise(img, img_o, "image")
    gws = rs.normal(0, 1, N).astype(np.float32); gimg = rs.normal(0, 1, (N, 3)).astype(np.float32)
    (ws * T(gws)).sum().add((img * T(gimg)).sum()).backward()
    gs_o, gc_o = oracle.composite_rays_train_backward(gws, gimg, sig, rgb, dl, rays, ws_o, img_o)
    assert_bitwise(sg.grad, gs_o, "grad_sigmas"); assert_bitwise(rg.grad, gc_o, "grad_rgbs")
    # KAT: alpha = 0.05 constant over 163 steps -> 1 - 0.95^163
    s = np.full(164, 0.05, np.float32); c = np.full((164, 3), 0.5, np.float32)
    w1, i1 = RM.composite_rays_train(T(s), T(c), T(s), T(np.array([[0, 0, 163]], np.int32)), 1.6)
    assert abs(float(w1[0]) - 0.9997662) < 2e-6 and abs(float(i1[0, 0]) - 0.5 * 0.9997662) < 2e-6


def test_inference_march_composite_compact(oracle):
    import avatarcraft_amd.raymarching as RM
    r = load_golden("rays.npz"); o, d = r["kat64_o"], r["kat64_d"]
    grid = _sphere_grid()
    N = o.shape[0]
    rs = np.random.RandomState(9)
    alive = rs.permutation(N)[:1500].astype(np.int32)
    near = np.full(N, 0.05, np.float32); far = np.full(N, 3.0, np.float32)
    t0 = np.full(1500, 0.6, np.float32)
    for perturb in (0, 3):
        xo, do_, dlo = oracle.march_rays(1500, 8, alive, t0, o, d, 1.6, grid, float(grid.mean()), near, far, perturb)
        x, dd, dl = RM.march_rays(1500, 8, T(alive), T(t0), T(o), T(d), 1.6, T(grid), float(grid.mean()), T(near), T(far), -1, perturb)
        assert_bitwise(x, xo, "xyzs"); assert_bitwise(dl, dlo, "deltas"); assert_bitwise(dd, do_, "dirs")
    M = 1500 * 8
    sig = rs.uniform(0, 0.7, M).astype(np.float32); rgb = rs.uniform(0, 1, (M, 3)).astype(np.float32)
    nrm = rs.normal(0, 1, (M, 3)).astype(np.float32)
    wo, dpo, imo, nmo = (np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32), np.zeros((N, 3), np.float32))
    rt_o = t0.copy()
    oracle.composite_rays(1500, 8, alive, rt_o, sig, rgb, nrm, dlo, wo, dpo, imo, nmo)
    w, dp, im, nm, rt = (torch.zeros(N, device=DEV), torch.zeros(N, device=DEV), torch.zeros(N, 3, device=DEV),
                         torch.zeros(N, 3, device=DEV), T(t0))
    RM.composite_rays(1500, 8, T(alive), rt, T(sig), T(rgb), T(nrm), T(dlo), w, dp, im, nm)
    for a, b, n in ((w, wo, "weights"), (dp, dpo, "depth"), (im, imo, "image"), (nm, nmo, "normal"), (rt, rt_o, "rays_t")):
        assert_bitwise(a, b, n)
    ra_o, rtt_o, cnt_o = oracle.compact_rays(1500, alive, rt_o)
    ra = torch.zeros(1500, dtype=torch.int32, device=DEV); rtt = torch.zeros(1500, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    RM.compact_rays(1500, ra, T(alive), rtt, rt, cnt)
    assert int(cnt[0]) == cnt_o and 0 < cnt_o < 1500
    assert np.array_equal(ra.cpu().numpy()[:cnt_o], ra_o[:cnt_o]); assert_bitwise(rtt[:cnt_o], rtt_o[:cnt_o], "rays_t")


# ------------------------------------------------------------------ SMPL-guided warp (rows a2, a13)
def test_mesh_near_far_and_warp(oracle):
    from avatarcraft_amd import ray_utils as RY
    from tests.common import make_body, make_rays
    g = load_golden("warp.npz")
    verts, faces, Ts = make_body()
    # goldens (reference code)
    near, far = RY.geometry_guided_near_far(T(g["rays_o"]), T(g["rays_d"]), verts, 0.05)
    fin = np.isfinite(g["near_t"])
    assert np.array_equal(np.isfinite(near.cpu().numpy()), fin)
    assert np.abs(near.cpu().numpy()[fin] - g["near_t"][fin]).max() < 5e-5
    can, dirs, clo, mask = RY.warp_samples_to_canonical(g["pts"], verts, np.concatenate([faces, faces], 1), Ts, 0.05)   # numpy in -> numpy out
    assert isinstance(can, np.ndarray) and can.dtype == np.float64
    assert np.array_equal(mask, g["mask"]) and np.abs(can - g["can_pts"]).max() < 1e-12
    assert np.abs(dirs - g["can_dirs"]).max() < 1e-9 and np.abs(clo - g["closest"]).max() < 1e-12
    # bitwise vs the oracle on a larger problem, torch in -> torch out
    ro, rd = make_rays(48, 48, dist=1.8, f=40.0, jitter_seed=2)
    n_o, f_o = oracle.mesh_near_far(ro, rd, verts, 0.05)
    n_g, f_g = RY.geometry_guided_near_far(T(ro), T(rd), T(verts), 0.05)
    assert_bitwise(n_g, n_o, "mesh near"); assert_bitwise(f_g, f_o, "mesh far")
    z = np.linspace(0.8, 2.8, 32, dtype=np.float32)
    pts = (ro[:, None, :] + rd[:, None, :] * z[None, :, None]).astype(np.float32)
    can_o, clo_o, d2_o, fid_o, m_o = oracle.warp_samples(pts.reshape(-1, 3), verts, faces, Ts, 0.05)
    can_g, _, clo_g, m_g = RY.warp_samples_to_canonical(T(pts), T(verts), T(faces), T(Ts), 0.05)
    assert can_g.is_cuda and can_g.dtype == torch.float64
    assert np.array_equal(can_g.cpu().numpy().reshape(-1, 3).view(np.uint64), can_o.view(np.uint64))
    assert np.array_equal(clo_g.cpu().numpy().reshape(-1, 3).view(np.uint64), clo_o.view(np.uint64))
    assert np.array_equal(m_g.cpu().numpy(), m_o)


def test_hash_stencil_forward_backward(oracle):
    """the 7-point stencil operator == 7 x hash_encode_forward / backward (oracle) on x, clamp(x +- eps e_k):
    forward bit for bit, backward up to the order of the float atomics; default 16-level grid, points up to the bound"""
    from avatarcraft_amd.encoder.hashencoder.hashgrid import HashEncoder
    O = oracle
    bound, eps = 1.6, 0.005
    enc = HashEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048).to(DEV)
    rs = np.random.RandomState(5)
    with torch.no_grad():
        enc.embeddings.copy_(torch.from_numpy(rs.uniform(-0.5, 0.5, size=tuple(enc.embeddings.shape)).astype(np.float32)))
    B = 3000
    x = rs.uniform(-1.6, 1.6, size=(B, 3)).astype(np.float32)
    x[:40] = np.sign(x[:40]) * 1.6                                   # corners / faces of the cube: the clamp is active
    x[40:80, 0] = 1.6; x[80:120, 1] = -1.6
    x[120:200] = (np.round(x[120:200] * 40) / 40).astype(np.float32)  # points on cell borders of the coarse levels
    xt = torch.from_numpy(x).to(DEV)
    h7 = enc.forward_stencil(xt, bound, eps)
    assert h7.shape == (7, B, 32)
    offsets = enc.offsets.cpu().numpy(); table = enc.embeddings.detach().cpu().numpy()
    S = np.log2(enc.per_level_scale)
    pts = np.repeat(x[None], 7, 0)
    for k in range(3):
        pts[1 + 2 * k, :, k] = np.clip(x[:, k] + np.float32(eps), -np.float32(bound), np.float32(bound))
        pts[2 + 2 * k, :, k] = np.clip(x[:, k] - np.float32(eps), -np.float32(bound), np.float32(bound))
    g = rs.normal(size=(7, B, 32)).astype(np.float32)
    gg_o = np.zeros_like(table, dtype=np.float64)
    for p in range(7):
        u = ((pts[p] + np.float32(bound)) / np.float32(2 * bound)).astype(np.float32)
        out_o, _, _ = O.hash_encode_forward(u, table, offsets, S, 16, False, False)        # [L,B,C]
        assert_bitwise(h7[p], np.ascontiguousarray(out_o.transpose(1, 0, 2).reshape(B, 32)), f"stencil point {p}")
        gp = np.ascontiguousarray(g[p].reshape(B, 16, 2).transpose(1, 0, 2))
        gg_p, _ = O.hash_encode_backward(gp, u, table, offsets, S, 16, None)
        gg_o += gg_p
    (h7 * torch.from_numpy(g).to(DEV)).sum().backward()
    gg = enc.embeddings.grad.cpu().numpy()
    err = np.abs(gg - gg_o)
    assert err.max() <= 1e-4 * max(1.0, np.abs(gg_o).max()), err.max()
    # the model-level wrapper agrees with the 7-call formulation
    from tests.test_gpu_model import golden_net
    net, _ = golden_net(train=True)
    pt = torch.from_numpy(rs.uniform(-1.2, 1.2, size=(2048, 3)).astype(np.float32)).to(DEV)
    s1, g1 = net.forward_sdf_stencil(pt, bound, eps)
    s0, g0 = net.forward_sdf(pt, bound), net.gradient(pt, bound, eps)
    assert torch.allclose(s1, s0, atol=2e-6) and torch.allclose(g1, g0, atol=2e-3, rtol=1e-3)
    (s1.sum() + (g1 ** 2).sum()).backward()
    ga = net.encoder.embeddings.grad.clone(); net.zero_grad()
    (s0.sum() + (g0 ** 2).sum()).backward()
    gb = net.encoder.embeddings.grad
    assert (ga - gb).abs().max() <= 2e-3 * gb.abs().max()


@pytest.mark.parametrize("nlat,nlon", [(40, 80), (83, 83), (3, 5)])
def test_warp_accel_equals_brute_force(oracle, nlat, nlon):
    """the culled closest-face search returns exactly what the exhaustive one returns (all outputs bit for bit), also with
    shuffled face order, on points near, inside and far from the body"""
    from avatarcraft_amd import _lib as Lb
    from tests.common import make_body
    verts, faces, Ts = make_body(n_lat=nlat, n_lon=nlon)
    rs = np.random.RandomState(7)
    faces = faces[rs.permutation(faces.shape[0])]                    # the order along the Morton curve must not matter
    P = 5000
    pts = np.concatenate([rs.uniform(-1.6, 1.6, size=(P // 2, 3)),
                          verts[rs.randint(0, verts.shape[0], P - P // 2)] + rs.normal(0, 0.03, size=(P - P // 2, 3))]).astype(np.float32)
    pts[:7] = verts[:7]                                              # exactly on vertices: many equally close faces
    pts[7] = 0.0                                                     # the centre: medial axis
    tp, tv, tf, tT = T(pts), T(verts), torch.from_numpy(faces).to(DEV), torch.from_numpy(Ts).to(DEV)
    F, V = faces.shape[0], verts.shape[0]
    st = None
    def outs():
        return dict(can=torch.empty(P, 3, dtype=torch.float64, device=DEV), canf=torch.empty(P, 3, device=DEV),
                    clo=torch.empty(P, 3, dtype=torch.float64, device=DEV), d2=torch.empty(P, dtype=torch.float64, device=DEV),
                    fid=torch.empty(P, dtype=torch.int32, device=DEV), mask=torch.empty(P, dtype=torch.uint8, device=DEV))
    a, b = outs(), outs()
    Lb.check(Lb.lib().ac_warp_samples(tp.data_ptr(), tv.data_ptr(), tf.data_ptr(), tT.data_ptr(), P, V, F, 0.05, a["can"].data_ptr(), a["canf"].data_ptr(),
                                      a["clo"].data_ptr(), a["d2"].data_ptr(), a["fid"].data_ptr(), a["mask"].data_ptr(), st))
    nbytes = Lb.lib().ac_warp_accel_bytes(F)
    assert nbytes > 0 and Lb.lib().ac_warp_accel_bytes(16385) == 0 and Lb.lib().ac_warp_accel_bytes(0) == 0
    acc = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    Lb.check(Lb.lib().ac_warp_accel_build(tv.data_ptr(), tf.data_ptr(), V, F, acc.data_ptr(), nbytes, st))
    Lb.check(Lb.lib().ac_warp_samples_accel(tp.data_ptr(), tv.data_ptr(), tf.data_ptr(), tT.data_ptr(), P, V, F, 0.05, acc.data_ptr(), b["can"].data_ptr(),
                                            b["canf"].data_ptr(), b["clo"].data_ptr(), b["d2"].data_ptr(), b["fid"].data_ptr(), b["mask"].data_ptr(), st))
    torch.cuda.synchronize()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # and the oracle agrees with both
    can_o, clo_o, d2_o, fid_o, m_o = oracle.warp_samples(pts[:600], verts, faces, Ts, 0.05)
    assert np.array_equal(b["fid"][:600].cpu().numpy(), fid_o) and np.array_equal(b["d2"][:600].cpu().numpy().view(np.uint64), d2_o.view(np.uint64))
    assert np.array_equal(b["can"][:600].cpu().numpy().view(np.uint64), can_o.view(np.uint64))
    # too small a buffer / too many faces are refused
    assert Lb.lib().ac_warp_accel_build(tv.data_ptr(), tf.data_ptr(), V, F, acc.data_ptr(), 100, st) != 0


def test_warp_accel_degenerate_faces_and_far_points():
    """the conservative fp32 bounds of the culled search (boxes, per-face bounding discs) must not lose the closest face when the
    mesh holds zero-area and sliver triangles, and for query points far outside the mesh: culled == exhaustive, bit for bit"""
    from avatarcraft_amd import _lib as Lb
    from tests.common import make_body
    verts, faces, Ts = make_body(n_lat=20, n_lon=30)
    rs = np.random.RandomState(11)
    V0 = verts.shape[0]
    extra_v, extra_f = [], []
    for k in range(40):
        i, j = rs.randint(0, V0, 2)
        a, b = verts[i], verts[j]
        t = rs.uniform(0.2, 0.8)
        mid = (a + t * (b - a) + rs.normal(0, 1e-7 if k % 2 else 1e-4, 3)).astype(np.float32)      # (nearly) on the segment a-b: a sliver
        extra_v.append(mid)
        extra_f.append([i, j, V0 + k])
    for k in range(10):
        i, j = rs.randint(0, V0, 2)
        extra_f.append([i, i, j])                                    # zero area: two equal corners
        extra_f.append([i, i, i])                                    # a point
    verts2 = np.concatenate([verts, np.stack(extra_v)]).astype(np.float32)
    faces2 = np.concatenate([faces, np.asarray(extra_f, dtype=faces.dtype)])
    faces2 = faces2[rs.permutation(faces2.shape[0])]
    Ts2 = np.concatenate([Ts, Ts[rs.randint(0, Ts.shape[0], len(extra_v))]])
    P = 4096
    pts = np.concatenate([rs.uniform(-1.6, 1.6, size=(P // 4, 3)), rs.uniform(-40.0, 40.0, size=(P // 4, 3)),
                          verts2[rs.randint(0, verts2.shape[0], P // 2)] + rs.normal(0, 0.02, size=(P // 2, 3))]).astype(np.float32)
    tp, tv, tf, tT = T(pts), T(verts2), torch.from_numpy(faces2).to(DEV), torch.from_numpy(Ts2).to(DEV)
    F, V = faces2.shape[0], verts2.shape[0]
    def outs():
        return dict(can=torch.empty(P, 3, dtype=torch.float64, device=DEV), clo=torch.empty(P, 3, dtype=torch.float64, device=DEV),
                    d2=torch.empty(P, dtype=torch.float64, device=DEV), fid=torch.empty(P, dtype=torch.int32, device=DEV),
                    mask=torch.empty(P, dtype=torch.uint8, device=DEV))
    a, b = outs(), outs()
    Lb.check(Lb.lib().ac_warp_samples(tp.data_ptr(), tv.data_ptr(), tf.data_ptr(), tT.data_ptr(), P, V, F, 0.05, a["can"].data_ptr(), None,
                                      a["clo"].data_ptr(), a["d2"].data_ptr(), a["fid"].data_ptr(), a["mask"].data_ptr(), None))
    nbytes = Lb.lib().ac_warp_accel_bytes(F)
    acc = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    Lb.check(Lb.lib().ac_warp_accel_build(tv.data_ptr(), tf.data_ptr(), V, F, acc.data_ptr(), nbytes, None))
    Lb.check(Lb.lib().ac_warp_samples_accel(tp.data_ptr(), tv.data_ptr(), tf.data_ptr(), tT.data_ptr(), P, V, F, 0.05, acc.data_ptr(), b["can"].data_ptr(),
                                            None, b["clo"].data_ptr(), b["d2"].data_ptr(), b["fid"].data_ptr(), b["mask"].data_ptr(), None))
    torch.cuda.synchronize()
    for k in ("fid", "d2", "clo", "mask"):
        assert torch.equal(a[k], b[k]), k
    fin = torch.isfinite(a["can"]).all(1)                            # a degenerate winner can make the blend singular: same non-finite pattern
    assert torch.equal(fin, torch.isfinite(b["can"]).all(1)) and torch.equal(a["can"][fin], b["can"][fin])


def test_warp_accel_nonfinite_points():
    """NaN / infinite query points (a NaN ray) take the culled search's no-cell path; it must report what the exhaustive kernel reports for them
    (no face ever compares closer: face 0, distance +inf) and must not disturb the finite samples of the same wave"""
    from avatarcraft_amd import _lib as Lb
    from tests.common import make_body
    verts, faces, Ts = make_body(n_lat=20, n_lon=30)
    rs = np.random.RandomState(5)
    P = 300
    pts = (verts[rs.randint(0, verts.shape[0], P)] + rs.normal(0, 0.05, size=(P, 3))).astype(np.float32)
    pts[3] = np.nan; pts[70, 1] = np.inf; pts[71] = -np.inf; pts[130, 2] = np.nan; pts[299, 0] = np.inf
    tp, tv, tf, tT = T(pts), T(verts), torch.from_numpy(faces).to(DEV), torch.from_numpy(Ts).to(DEV)
    F, V = faces.shape[0], verts.shape[0]
    def outs():
        return dict(clo=torch.empty(P, 3, dtype=torch.float64, device=DEV), d2=torch.empty(P, dtype=torch.float64, device=DEV),
                    fid=torch.empty(P, dtype=torch.int32, device=DEV), mask=torch.empty(P, dtype=torch.uint8, device=DEV),
                    can=torch.empty(P, 3, dtype=torch.float64, device=DEV))
    a, b = outs(), outs()
    Lb.check(Lb.lib().ac_warp_samples(tp.data_ptr(), tv.data_ptr(), tf.data_ptr(), tT.data_ptr(), P, V, F, 0.05, a["can"].data_ptr(), None,
                                      a["clo"].data_ptr(), a["d2"].data_ptr(), a["fid"].data_ptr(), a["mask"].data_ptr(), None))
    nbytes = Lb.lib().ac_warp_accel_bytes(F)
    acc = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    Lb.check(Lb.lib().ac_warp_accel_build(tv.data_ptr(), tf.data_ptr(), V, F, acc.data_ptr(), nbytes, None))
    Lb.check(Lb.lib().ac_warp_samples_accel(tp.data_ptr(), tv.data_ptr(), tf.data_ptr(), tT.data_ptr(), P, V, F, 0.05, acc.data_ptr(), b["can"].data_ptr(),
                                            None, b["clo"].data_ptr(), b["d2"].data_ptr(), b["fid"].data_ptr(), b["mask"].data_ptr(), None))
    torch.cuda.synchronize()
    for k in ("fid", "d2", "clo", "mask"):
        assert torch.equal(a[k], b[k]), k
    bad = [3, 70, 71, 130, 299]
    assert all(int(b["fid"][i]) == 0 and float(b["d2"][i]) == float("inf") and int(b["mask"][i]) == 0 for i in bad)
    fin = torch.isfinite(a["can"]).all(1)
    assert torch.equal(fin, torch.isfinite(b["can"]).all(1)) and torch.equal(a["can"][fin], b["can"][fin]) and int(fin.sum()) == P - len(bad)


@pytest.mark.parametrize("P", [1, 63, 65])
def test_warp_tiny_point_sets(oracle, P):
    from avatarcraft_amd import ray_utils as RY
    from tests.common import make_body
    verts, faces, Ts = make_body(n_lat=6, n_lon=9)
    rs = np.random.RandomState(P)
    pts = rs.uniform(-1, 1, size=(1, P, 3)).astype(np.float32)
    can_o, clo_o, d2_o, fid_o, m_o = oracle.warp_samples(pts.reshape(-1, 3), verts, faces, Ts, 0.05)
    for accel in (True, False):
        can, dirs, clo, mask = RY.warp_samples_to_canonical(T(pts), T(verts), torch.from_numpy(faces).to(DEV), torch.from_numpy(Ts).to(DEV), 0.05, accel=accel)
        assert np.array_equal(can.cpu().numpy().reshape(-1, 3).view(np.uint64), can_o.view(np.uint64))
        assert np.array_equal(mask.cpu().numpy(), m_o)
    empty = RY.warp_samples_to_canonical(torch.empty(0, 4, 3, device=DEV), T(verts), torch.from_numpy(faces).to(DEV), torch.from_numpy(Ts).to(DEV), 0.05)
    assert empty[0].shape == (0, 4, 3)


def test_hash_stencil_backward_paths_agree():
    """the three scatter paths of ac_hash_stencil_backward give the same table gradient (up to summation order):
    no scratch (hardware float atomics), private copies of the dense levels only, binned two-pass scatter"""
    from avatarcraft_amd import _lib as Lb
    from avatarcraft_amd.encoder.hashencoder.hashgrid import HashEncoder
    enc = HashEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048).to(DEV)
    rs = np.random.RandomState(8)
    B = 70000                                                   # > 64 * RCAP / 56: several flushes per wave, partial last group
    z = np.sort(rs.uniform(0.2, 3.0, size=(B // 100, 100)), axis=1)   # depth-sorted samples along rays: long same-cell runs
    o = rs.uniform(-0.3, 0.3, size=(B // 100, 1, 3)); d = rs.normal(size=(B // 100, 1, 3)); d /= np.linalg.norm(d, axis=2, keepdims=True)
    x = torch.from_numpy(np.clip(o - d * 1.5 + d * z[:, :, None], -1.6, 1.6).reshape(-1, 3).astype(np.float32)).to(DEV)
    g = torch.from_numpy(rs.normal(size=(7, 16, B, 2)).astype(np.float32)).to(DEV)
    g[:, :, ::3] = 0.0                                          # exact zeros are skipped
    oh = enc.offsets.cpu().numpy().astype(np.int32)
    S = float(np.float32(np.log2(enc.per_level_scale)))
    outs = []
    for copies, nb in ((0, 0), (16, 0), (16, B)):
        gt = torch.zeros_like(enc.embeddings)
        nbytes = int(Lb.lib().ac_hash_stencil_backward_scratch(oh.ctypes.data, 16, S, 16, copies, nb)) if copies else 0
        sc = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=DEV)
        Lb.check(Lb.lib().ac_hash_stencil_backward(g.data_ptr(), x.data_ptr(), oh.ctypes.data, gt.data_ptr(), B, 2, 16, S, 16, 0.005, 1.6,
                                                   sc.data_ptr() if nbytes else None, nbytes, None))
        torch.cuda.synchronize()
        outs.append(gt)
    scale = float(outs[0].abs().max())
    assert scale > 0
    for k in (1, 2):
        assert float((outs[k] - outs[0]).abs().max()) <= 2e-5 * scale, k
    # adversarial distribution: two alternating points -> no runs, all records of a level land in a handful of buckets, whose queues
    # overflow into the atomic path; the sums must not change
    x2 = x.clone(); x2[0::2] = torch.tensor([0.31, -0.42, 0.77], device=DEV); x2[1::2] = torch.tensor([-1.01, 0.63, -0.2], device=DEV)
    res = []
    for nb in (0, B):
        gt = torch.zeros_like(enc.embeddings)
        nbytes = int(Lb.lib().ac_hash_stencil_backward_scratch(oh.ctypes.data, 16, S, 16, 16, nb))
        sc = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=DEV)
        Lb.check(Lb.lib().ac_hash_stencil_backward(g.data_ptr(), x2.data_ptr(), oh.ctypes.data, gt.data_ptr(), B, 2, 16, S, 16, 0.005, 1.6, sc.data_ptr(), nbytes, None))
        torch.cuda.synchronize()
        res.append(gt)
    assert float((res[1] - res[0]).abs().max()) <= 1e-4 * float(res[0].abs().max())


@pytest.mark.parametrize("L,base,log2T,pls,binned", [(4, 4, 12, 2.0, True), (8, 16, 15, 1.5, True), (6, 5, 14, 1.7, True), (3, 2, 10, 2.0, False)])
def test_hash_backward_binned_other_grids(oracle, L, base, log2T, pls, binned):
    """binned scatter on grids other than the default one (bucket size = a power of two that covers each level with <= 64 buckets;
    a level of fewer than 64 entries is not binned: the operator falls back to the direct atomics)"""
    from avatarcraft_amd import _lib as Lb
    O = oracle
    offs, _ = O.hash_offsets(num_levels=L, per_level_scale=pls, base_resolution=base, log2_hashmap_size=log2T)
    S = float(np.float32(np.log2(pls)))
    rs = np.random.RandomState(5)
    B = 20011
    x = rs.uniform(0, 1, size=(B, 3)).astype(np.float32)
    x[:64] = rs.uniform(0.4, 0.41, size=(64, 3))                 # a cluster: long runs of lanes in one cell
    g = rs.normal(size=(L, B, 2)).astype(np.float32)
    xt, gt_ = T(x), T(g)
    n = int(offs[-1])
    emb = torch.zeros(n, 2, device=DEV)
    ot = torch.from_numpy(offs).to(DEV)
    dummy = torch.zeros(1, device=DEV)
    nbytes = int(Lb.lib().ac_hash_encode_backward_scratch(offs.ctypes.data, 3, 2, L, S, base, B))
    assert (nbytes > 0) == binned
    sc = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=DEV)
    gg = torch.zeros_like(emb)
    Lb.check(Lb.lib().ac_hash_encode_backward_ws(gt_.data_ptr(), xt.data_ptr(), emb.data_ptr(), ot.data_ptr(), offs.ctypes.data, gg.data_ptr(), B, 3, 2, L,
                                                 S, base, 0, dummy.data_ptr(), dummy.data_ptr(), sc.data_ptr(), nbytes, None))
    torch.cuda.synchronize()
    gg_o, _ = O.hash_encode_backward(g, x, np.zeros((n, 2), np.float32), offs, S, base, None)
    assert np.abs(gg.cpu().numpy() - gg_o).max() <= 2e-5 * np.abs(gg_o).max()


def test_hash_backward_binned_equals_direct(oracle):
    """the reference operator's backward through the binned scatter (ac_hash_encode_backward_ws) against the direct float atomics and
    the oracle, default 16-level grid"""
    from avatarcraft_amd import _lib as Lb
    O = oracle
    offs, pls = O.hash_offsets(desired_resolution=2048)
    S = float(np.float32(np.log2(pls)))
    rs = np.random.RandomState(12)
    B = 50001
    x = rs.uniform(0, 1, size=(B, 3)).astype(np.float32)
    x[:100] = np.round(x[:100] * 15) / 15                       # on coarse cell borders
    g = rs.normal(size=(16, B, 2)).astype(np.float32)
    xt, gt_ = T(x), T(g)
    emb = torch.zeros(int(offs[-1]), 2, device=DEV)
    ot = torch.from_numpy(offs).to(DEV)
    dummy = torch.zeros(1, device=DEV)
    res = []
    for use_ws in (False, True):
        gg = torch.zeros_like(emb)
        nbytes = int(Lb.lib().ac_hash_encode_backward_scratch(offs.ctypes.data, 3, 2, 16, S, 16, B)) if use_ws else 0
        assert (nbytes > 0) == use_ws
        sc = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=DEV)
        Lb.check(Lb.lib().ac_hash_encode_backward_ws(gt_.data_ptr(), xt.data_ptr(), emb.data_ptr(), ot.data_ptr(), offs.ctypes.data, gg.data_ptr(), B, 3, 2, 16,
                                                     S, 16, 0, dummy.data_ptr(), dummy.data_ptr(), sc.data_ptr() if use_ws else None, nbytes, None))
        torch.cuda.synchronize()
        res.append(gg.cpu().numpy())
    gg_o, _ = O.hash_encode_backward(g, x, np.zeros((int(offs[-1]), 2), np.float32), offs, S, 16, None)
    scale = np.abs(gg_o).max()
    assert np.abs(res[0] - gg_o).max() <= 2e-5 * scale and np.abs(res[1] - gg_o).max() <= 2e-5 * scale
    assert Lb.lib().ac_hash_encode_backward_scratch(offs.ctypes.data, 3, 4, 16, S, 16, B) == 0          # C = 4: direct path only


def test_hash_backward_binned_nonfinite_gradients(oracle):
    """An Inf / NaN upstream gradient must reach the table like the reference's atomicAdd delivers it (hashencoder.cu:302-305), not be
    quantised away by the fixed-point sums of the binned scatter: the level that holds it is summed in float, every other level keeps
    its exact fixed-point sums.  Also documents the quantisation floor: contributions below 2^-42 of a level's largest |v| vanish."""
    from avatarcraft_amd import _lib as Lb
    O = oracle
    offs, pls = O.hash_offsets(desired_resolution=2048)
    S = float(np.float32(np.log2(pls)))
    rs = np.random.RandomState(3)
    B = 4099
    x = rs.uniform(0.05, 0.95, size=(B, 3)).astype(np.float32)
    g = rs.normal(size=(16, B, 2)).astype(np.float32)
    g[7, 11, 0] = np.inf; g[12, 500, 1] = np.nan
    g[3, :, :] *= 1e-20; g[3, 77, 0] = 1.0                       # level 3: one record 2^66 times larger than the rest -> the rest falls below the floor
    emb = torch.zeros(int(offs[-1]), 2, device=DEV)
    ot = torch.from_numpy(offs).to(DEV)
    dummy = torch.zeros(1, device=DEV)
    nbytes = int(Lb.lib().ac_hash_encode_backward_scratch(offs.ctypes.data, 3, 2, 16, S, 16, B))
    sc = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    gg = torch.zeros_like(emb)
    gt_, xt = T(g), T(x)                          # (named: a temporary's memory may be handed to the next allocation before the kernel runs)
    Lb.check(Lb.lib().ac_hash_encode_backward_ws(gt_.data_ptr(), xt.data_ptr(), emb.data_ptr(), ot.data_ptr(), offs.ctypes.data, gg.data_ptr(), B, 3, 2, 16,
                                                 S, 16, 0, dummy.data_ptr(), dummy.data_ptr(), sc.data_ptr(), nbytes, None))
    torch.cuda.synchronize()
    got = gg.cpu().numpy()
    with np.errstate(invalid="ignore"):
        ref, _ = O.hash_encode_backward(g, x, np.zeros((int(offs[-1]), 2), np.float32), offs, S, 16, None)
    lv = lambda a, l: a[offs[l]:offs[l + 1]]
    assert np.array_equal(np.isinf(lv(got, 7)), np.isinf(lv(ref, 7))) and np.isinf(lv(got, 7)).sum() == 8          # the 8 corners of sample 11
    assert np.array_equal(np.isnan(lv(got, 12)), np.isnan(lv(ref, 12))) and np.isnan(lv(got, 12)).sum() == 8
    fin7 = np.isfinite(lv(ref, 7))
    assert np.abs(lv(got, 7)[fin7] - lv(ref, 7)[fin7]).max() <= 2e-5 * np.abs(lv(ref, 7)[fin7]).max()               # the rest of that level: float sums
    for l in (0, 5, 9, 15):
        assert np.isfinite(lv(got, l)).all() and np.abs(lv(got, l) - lv(ref, l)).max() <= 2e-5 * np.abs(lv(ref, l)).max()
    big = np.abs(lv(ref, 3)) > 1e-12                                                                                 # level 3: the large record is exact ...
    assert 1 <= big.sum() <= 8 and np.allclose(lv(got, 3)[big], lv(ref, 3)[big], rtol=1e-5, atol=1e-9)
    assert np.abs(lv(got, 3)[~big]).max() <= 1e-19                                                                  # ... and nothing else is invented


def test_unit_div_equals_the_ieee_division_on_the_device_over_the_whole_domain():
    """(x + bound) / (2 bound) without a division for the divisors the library accepts.  tests/test_div_check.py proves the identity on the
    host (all 2^32 dividends, C fmaf); this sweeps it ON THE DEVICE -- v_mul_f32 / v_fma_f32 against the compiler's IEEE division sequence -- over every fp32
    dividend from 1e-30 to 1e30 of both signs (2 x 1.67 G values), +0 and the NaNs: the renderer's dividends are 0, NaN or between 1e-7 and 4."""
    import ctypes
    import re
    import struct
    from avatarcraft_amd import _lib as L
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "avatarcraft_amd", "csrc", "ac_common.hpp")).read()
    divs = [float(t.strip().rstrip("f")) for t in re.search(r"const float ok\[\] = \{([^}]*)\};", src).group(1).split(",")]
    assert 3.2 in [round(d, 6) for d in divs]
    bits = lambda f: struct.unpack("<I", struct.pack("<f", f))[0]
    cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
    st = L.current_stream(torch.device(DEV))
    for d in divs:
        for lo, hi in ((bits(1e-30), bits(1e30)), (bits(-1e-30), bits(-1e30)), (0, 0), (0x7f800001, 0x7fffffff)):
            cnt.zero_()
            L.check(L.lib().ac_debug_unit_div_check(lo, hi, ctypes.c_float(d), cnt.data_ptr(), st), "unit_div_check")
            assert int(cnt.item()) == 0, (d, hex(lo), hex(hi), int(cnt.item()))
    # the identity is not a general one (which is why only verified divisors are accepted): it fails on denormal quotients
    cnt.zero_()
    L.check(L.lib().ac_debug_unit_div_check(1, bits(1e-37), ctypes.c_float(3.2), cnt.data_ptr(), st), "unit_div_check")
    assert int(cnt.item()) > 0
