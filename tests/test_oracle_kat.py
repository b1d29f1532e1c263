"""CPU: pin the oracle against the reference's known answers (tests/golden/kat.json: SURVEY.md Appendix A.4 / B,
values produced by the reference's own kernel bodies) and against public PCG vectors."""
import json
import os

import numpy as np
import pytest

from tests.common import load_golden

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))


def test_pcg32(oracle):
    for seed, (a, b) in KAT["pcg32_next_float"].items():
        g = oracle.Pcg32(int(seed))
        assert abs(g.next_float() - a) < 5e-10 and abs(g.next_float() - b) < 5e-10
    pub = KAT["pcg32_public"]
    g = oracle.Pcg32(pub["seed"], pub["seq"])
    assert [g.next_uint() for _ in pub["uint"]] == pub["uint"]


def test_hash_integer_pieces(oracle):
    for pos, h in KAT["fast_hash3"]:
        assert oracle.fast_hash(pos) == h
    for k in KAT["grid_index"]:
        assert oracle.grid_index(k["D"], k["C"], k["ch"], k["hashmap_size"], k["resolution"], k["pos"]) == k["index"]


def test_level_table_and_offsets(oracle):
    lt = KAT["level_table"]
    offsets, pls = oracle.hash_offsets(3, 16, 2, 1.3819, 16, 19, 2048)
    assert offsets.tolist() == lt["offsets"] and abs(pls - lt["per_level_scale"]) < 1e-15
    scale, res = oracle.hash_level_table(16, np.float32(np.log2(pls)), lt["H"])
    assert res.tolist() == lt["res"]
    np.testing.assert_allclose(scale, np.float32(lt["scale"]), rtol=3e-7)
    assert scale[15] == np.float32(2047.0)                 # the ulp-sensitive level (Appendix B)
    assert int(offsets[-1]) * 2 + 9188 == KAT["n_params"]  # 12 248 902 parameters of NeRFNetwork()


def _sphere_grid(H=129, bound=1.6, r=0.5):
    ax = np.linspace(-bound, bound, H, dtype=np.float32)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    return (100.0 * ((X ** 2 + Y ** 2 + Z ** 2) < r * r)).astype(np.float32)


def test_march_rays_train_kat(oracle):
    k = KAT["march_rays_train"]
    r = load_golden("rays.npz")           # rays made by the reference's own camera code (make_golden.py)
    grid = _sphere_grid()
    xyzs, dirs, deltas, rays, counter = oracle.march_rays_train(r["kat64_o"], r["kat64_d"], grid, float(grid.mean()), k["bound"])
    assert counter.tolist() == k["counter"]
    assert int((rays[:, 2] > 0).sum()) == k["rays_hit"] and int(rays[:, 2].max()) == k["max_steps"]
    assert rays[2080].tolist() == k["centre_ray"]
    # packed layout is contiguous in ray order
    assert np.array_equal(rays[:, 0], np.arange(4096)) and np.array_equal(rays[1:, 1], np.cumsum(rays[:-1, 2]))


def test_composite_kat(oracle):
    k = KAT["composite_rays_train_forward"]
    n = k["steps"]
    s = np.full(n + 1, k["alpha"], np.float32); c = np.full((n + 1, 3), k["rgb_in"], np.float32)
    ws, img = oracle.composite_rays_train_forward(s, c, s, np.array([[0, 0, n]], np.int32))
    assert abs(ws[0] - k["weights_sum"]) < 2e-6 and abs(img[0, 0] - k["rgb_in"] * k["weights_sum"]) < 2e-6
    assert abs(ws[0] - (1 - 0.95 ** n)) < 2e-6


def test_deterministic_math_accuracy(oracle):
    """exp/log1p/sigmoid of ac_math.h against float64 (<= 1.5 ulp); softplus within 2.1e-9 absolute + 1 ulp of the value + half an ulp of x / 2"""
    L = oracle.lib()
    rs = np.random.RandomState(0)
    xs = rs.uniform(-87, 88, 4000).astype(np.float32)
    e = np.array([L.orc_test_expf(float(v)) for v in xs]); ref = np.exp(xs.astype(np.float64))
    assert np.max(np.abs(e - ref) / ref) < 1.5 * 2 ** -24 * 2
    us = np.exp(rs.uniform(-30, 20, 4000)).astype(np.float32)
    l = np.array([L.orc_test_log1pf(float(v)) for v in us]); ref = np.log1p(us.astype(np.float64))
    assert np.max(np.abs(l - ref) / ref) < 1.5 * 2 ** -24 * 2
    ts = rs.uniform(-0.5, 0.5, 4000).astype(np.float32)
    sp = np.array([L.orc_test_softplus100(float(v)) for v in ts])
    t32 = (ts * np.float32(100.0)).astype(np.float64)        # the reference rounds x*beta in fp32 too
    ref = np.where(t32 > 20, ts, np.log1p(np.exp(t32)) / 100)
    # table softplus: 0.5 x + 0.5 |x| + G(|100x|), |error of G| <= 1.9e-9 (tools/gen_softplus_table.py) + the rounding of the result + the rounding
    # of the intermediate 0.5 x + G (<= half an ulp of x / 2: the uncertainty x itself carries as a sum of 35 fp32 products)
    assert np.max(np.abs(sp - ref) - 1.3e-7 * np.abs(ref) - 3.0e-8 * np.abs(ts)) < 2.1e-9
    big = ts[t32 > 20]
    assert len(big) and all(L.orc_test_softplus100(float(v)) == np.float32(v) for v in big)      # torch's linear branch: x, exactly
    assert L.orc_test_softplus100(-5.0) >= 0.0 and L.orc_test_softplus100(-5.0) < 1e-15 and np.isnan(L.orc_test_softplus100(float("nan")))
    sg = np.array([L.orc_test_sigmoid(float(v)) for v in xs]); ref = 1 / (1 + np.exp(-xs.astype(np.float64)))
    assert np.max(np.abs(sg - ref) / ref) < 4e-7
    assert L.orc_test_expf(100.0) == np.inf and L.orc_test_expf(-100.0) == 0.0 and L.orc_test_sigmoid(-200.0) == 0.0
