"""The two encoder extensions on half and double tensors (the reference dispatches them over Float / Half / Double: hashencoder.cu:352,391,
shencoder.cu:337,380).  CPU tier: the oracle's typed restatements against the pinned fp32 routines.  GPU tier: the HIP instantiations
(ac_*_encode_*_typed) against the oracle through the reference-shaped Python surface, and the autocast behaviour of `custom_fwd(cast_inputs=half)`."""
import numpy as np
import pytest
import torch

DEV = "cuda:0"


def _case(O, D=3, C=2, L=8, base=4, log2T=12, B=600, seed=3):
    offsets, pls = O.hash_offsets(D, L, C, 1.5, base, log2T)
    rs = np.random.RandomState(seed)
    grid = rs.uniform(-1, 1, (int(offsets[-1]), C))
    x = rs.uniform(0, 1, (B, D))
    x[0] = 1.0; x[1] = 0.0; x[2, 0] = -0.1; x[3, -1] = 1.5            # the range edges and two out-of-range rows
    g = rs.normal(0, 1, (L, B, C))
    return offsets, np.float32(np.log2(pls)), grid, x, g


# ------------------------------------------------------------------ CPU: the oracle's typed restatements
def test_oracle_hash_double_follows_the_float_routine(oracle):
    O = oracle
    offsets, S, grid, x, g = _case(O)
    # float32-representable inputs: both instantiations see the same cells and weights (fp32 in every instantiation), so they differ by the
    # accumulation only
    x = x.astype(np.float32).astype(np.float64); grid = grid.astype(np.float32).astype(np.float64)
    o64, d64 = O.hash_encode_forward_typed(x, grid, offsets, S, 4, True)
    o32, d32 = O.hash_encode_forward_typed(x.astype(np.float32), grid.astype(np.float32), offsets, S, 4, True)
    assert o64.dtype == np.float64 and d64.dtype == np.float64
    assert np.abs(o64 - o32).max() <= 4e-7 and np.abs(d64 - d32).max() <= 2e-6 * np.abs(d64).max()
    assert np.all(o64[:, 2] == 0) and np.all(o64[:, 3] == 0) and np.all(d64[2] == 0)          # out of range -> zeros
    assert np.any(o64[:, 0] != 0) and np.any(o64[:, 1] != 0)                                  # 0 and 1 are in range
    gg64, gi64 = O.hash_encode_backward_typed(g, x, grid, offsets, S, 4, d64)
    gg32, gi32 = O.hash_encode_backward_typed(g.astype(np.float32), x.astype(np.float32), grid.astype(np.float32), offsets, S, 4, d32)
    assert np.abs(gg64 - gg32).max() <= 1e-5 and np.abs(gi64 - gi32).max() <= 1e-4 * np.abs(gi64).max()
    # a double input that is out of range only in double: (float)x == 1 but x > 1 -> zeros, as `inputs[d] > 1` on the double decides (hashencoder.cu:98)
    xe = x.copy(); xe[5, 1] = 1.0 + 1e-12
    oe, _ = O.hash_encode_forward_typed(xe, grid, offsets, S, 4, False)
    assert np.all(oe[:, 5] == 0)


def test_oracle_hash_half_is_the_float_routine_rounded_once(oracle):
    O = oracle
    offsets, S, grid, x, g = _case(O)
    xh, gh = x.astype(np.float16), grid.astype(np.float16)
    o16, d16 = O.hash_encode_forward_typed(xh, gh, offsets, S, 4, True)
    o32, d32, _ = O.hash_encode_forward(xh.astype(np.float32), gh.astype(np.float32), offsets, S, 4, True)
    assert o16.dtype == np.float16 and np.array_equal(o16, o32.astype(np.float16)) and np.array_equal(d16, d32.astype(np.float16))


@pytest.mark.parametrize("degree", [1, 4, 8])
def test_oracle_sh_double_and_half(oracle, degree):
    O = oracle
    rs = np.random.RandomState(degree)
    d = rs.normal(0, 1, (300, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    s64, j64 = O.sh_encode_forward_typed(d, degree, True)
    s32, j32 = O.sh_encode_forward_typed(d.astype(np.float32), degree, True)
    assert s64.dtype == np.float64 and np.abs(s64 - s32).max() <= 3e-6 and np.abs(j64 - j32).max() <= 6e-5
    # the double basis is orthonormal to double accuracy where the float one is to float accuracy: Y_00 = 1 / (2 sqrt(pi)) exactly rounded
    assert abs(s64[0, 0] - 0.5 / np.sqrt(np.pi)) <= 1e-16
    s16, j16 = O.sh_encode_forward_typed(d.astype(np.float16), degree, True)
    r32, q32 = O.sh_encode_forward(d.astype(np.float16).astype(np.float32), degree, True)
    assert np.array_equal(s16, r32.astype(np.float16)) and np.array_equal(j16, q32.astype(np.float16))
    g = rs.normal(0, 1, s64.shape)
    gi64 = O.sh_encode_backward_typed(g, d, degree, j64)
    assert np.abs(gi64 - np.einsum("bc,bdc->bd", g, j64.reshape(-1, 3, degree * degree))).max() <= 1e-12 * max(1.0, np.abs(gi64).max())


# ------------------------------------------------------------------ GPU: the HIP instantiations
def _t(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV, dtype)


@pytest.mark.gpu
@pytest.mark.parametrize("D,C,L,base,log2T,B", [(3, 2, 8, 4, 12, 600), (3, 4, 5, 4, 10, 333), (2, 2, 6, 8, 10, 300), (3, 1, 4, 16, 14, 129), (3, 8, 3, 4, 8, 64)])
@pytest.mark.parametrize("dtype", ["float16", "float64"])
def test_hash_encoder_half_and_double_vs_oracle(oracle, dtype, D, C, L, base, log2T, B):
    from avatarcraft_amd.encoder.hashencoder.backend import _backend
    O = oracle
    nd, td = getattr(np, dtype), getattr(torch, dtype)
    offsets, S, grid, x, g = _case(O, D, C, L, base, log2T, B, seed=B)
    x, grid, g = x.astype(nd), grid.astype(nd), (0.01 * g).astype(nd)
    out_o, dd_o = O.hash_encode_forward_typed(x, grid, offsets, S, base, True)
    xt, gt, ot = _t(x, td), _t(grid, td), torch.from_numpy(offsets).to(DEV)
    out = torch.empty(L, B, C, device=DEV, dtype=td); dd = torch.empty(B, L * D * C, device=DEV, dtype=td)
    _backend.hash_encode_forward(xt, gt, ot, out, B, D, C, L, S, base, True, dd)
    # same arithmetic on both sides (fp32 cells and weights; fp32 / fp64 fma accumulation; one rounding to half): bit for bit
    assert np.array_equal(out.cpu().numpy().view(np.uint16 if dtype == "float16" else np.uint64), out_o.view(np.uint16 if dtype == "float16" else np.uint64)), "outputs"
    assert np.array_equal(dd.cpu().numpy(), dd_o), "dy_dx"
    gg_o, gi_o = O.hash_encode_backward_typed(g, x, grid, offsets, S, base, dd_o)
    gg = torch.zeros_like(gt); gi = torch.zeros_like(xt)
    _backend.hash_encode_backward(_t(g, td), xt, gt, ot, gg, B, D, C, L, S, base, True, dd, gi)
    assert np.array_equal(gi.cpu().numpy(), gi_o), "grad_inputs"
    if dtype == "float64":
        np.testing.assert_allclose(gg.cpu().numpy(), gg_o, rtol=1e-12, atol=1e-14)
    else:
        # half2 atomics round every contribution and every partial sum to half (as the reference's do, hashencoder.cu:293-299): entries that
        # collect n contributions carry ~sqrt(n) half roundings.  Bound: 2^-10 of the entry's absolute sum per contribution.
        got = gg.float().cpu().numpy()
        ref32, _ = O.hash_encode_backward(g.astype(np.float32), x.astype(np.float32), grid.astype(np.float32), offsets, S, base, None)
        absum, _ = O.hash_encode_backward(np.abs(g).astype(np.float32), x.astype(np.float32), grid.astype(np.float32), offsets, S, base, None)
        cnt, _ = O.hash_encode_backward(np.ones_like(g, np.float32), x.astype(np.float32), grid.astype(np.float32), offsets, S, base, None)
        bound = 2.0 ** -10 * absum * np.maximum(cnt * 8, 1.0) + 1e-7
        assert np.all(np.abs(got - ref32) <= bound), float(np.max(np.abs(got - ref32) / bound))
    # mixed dtypes are an error, like a scalar_t mismatch would be in the reference
    with pytest.raises(RuntimeError):
        _backend.hash_encode_forward(xt, gt.float(), ot, out, B, D, C, L, S, base, True, dd)
    with pytest.raises(RuntimeError):
        _backend.hash_encode_forward(xt.to(torch.bfloat16), gt, ot, out, B, D, C, L, S, base, True, dd)


@pytest.mark.gpu
@pytest.mark.parametrize("degree", [1, 3, 8])
@pytest.mark.parametrize("dtype", ["float16", "float64"])
def test_sh_encoder_half_and_double_vs_oracle(oracle, dtype, degree):
    from avatarcraft_amd.encoder.shencoder.sphere_harmonics import SHEncoder
    O = oracle
    nd, td = getattr(np, dtype), getattr(torch, dtype)
    rs = np.random.RandomState(degree)
    d = rs.normal(0, 1, (257, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(nd)
    s_o, j_o = O.sh_encode_forward_typed(d, degree, True)
    enc = SHEncoder(degree=degree)
    xt = _t(d, td).requires_grad_(True)
    y = enc(xt)
    assert y.dtype == td and np.array_equal(y.detach().cpu().numpy(), s_o)
    g = rs.normal(0, 1, s_o.shape).astype(nd)
    y.backward(_t(g, td))
    gi_o = O.sh_encode_backward_typed(g, d, degree, j_o)
    assert xt.grad.dtype == td and np.array_equal(xt.grad.cpu().numpy(), gi_o)


@pytest.mark.gpu
def test_hash_module_on_half_double_and_under_autocast(oracle):
    from avatarcraft_amd.encoder.hashencoder.hashgrid import HashEncoder
    O = oracle
    torch.manual_seed(0)
    enc = HashEncoder(input_dim=3, num_levels=6, level_dim=2, per_level_scale=1.5, base_resolution=4, log2_hashmap_size=11).to(DEV)
    with torch.no_grad():
        enc.embeddings.uniform_(-1, 1)
    x = torch.rand(500, 3, device=DEV) * 2 - 1
    y32 = enc(x)
    # .double() / .half() modules: the encoder follows the dtype of its tensors
    for td, tol in ((torch.float64, 1e-6), (torch.float16, 2e-2)):
        e2 = HashEncoder(input_dim=3, num_levels=6, level_dim=2, per_level_scale=1.5, base_resolution=4, log2_hashmap_size=11).to(DEV).to(td)
        with torch.no_grad():
            e2.embeddings.copy_(enc.embeddings.to(td))
        xin = x.to(td).requires_grad_(True)
        y = e2(xin)
        assert y.dtype == td and float((y.float() - enc(xin.detach().float())).abs().max()) <= tol
        y.sum().backward()
        assert e2.embeddings.grad.dtype == td and xin.grad.dtype == td and bool(torch.isfinite(e2.embeddings.grad.float()).all())
    # autocast: operands are cast to half on the way in (hashgrid.py:13 of the reference), gradients come back in the parameters' dtype
    enc.zero_grad()
    with torch.autocast("cuda", dtype=torch.float16):
        ya = enc(x)
    assert ya.dtype == torch.float16
    # (x + 1) / 2 is evaluated in fp32 by the module before the cast; the oracle gets the same half inputs
    x01 = ((x + 1) / 2).half().cpu().numpy()
    o16, _ = O.hash_encode_forward_typed(x01, enc.embeddings.detach().half().cpu().numpy(), enc.offsets.cpu().numpy(), np.float32(np.log2(1.5)), 4, False)
    assert np.array_equal(ya.detach().cpu().numpy().reshape(500, 6, 2).transpose(1, 0, 2), o16)
    ya.float().sum().backward()
    assert enc.embeddings.grad.dtype == torch.float32 and float(enc.embeddings.grad.abs().sum()) > 0
    assert float((ya.float() - y32).abs().max()) <= 2e-2
