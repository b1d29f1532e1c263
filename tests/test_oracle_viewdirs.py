"""CPU: NeRFNetwork(use_viewdirs=True) -- models/instant_nsr.py:565-569, 644-653: colour layer 1 reads cat[x, sh(d), n, geo_feat] -- in the oracle (the 16
view-direction columns as a per-ray bias of layer 1: oracle/ac_oracle.c orc_color_mlp_d) against tests/golden/viewdirs.npz, recorded from the
reference's own run() and autograd (tests/golden/make_viewdirs_golden.py)."""
import numpy as np
import pytest

from tests.common import load_golden, make_table
from tests.test_oracle_backward import _chain_to_raw


def viewdirs_field(O, g):
    table = make_table(int(g["offsets"][-1]), seed=int(g["table_seed"]), offsets=g["offsets"], level_amp=g["level_amp"])
    f = O.Field(table, g["offsets"], g["W1"], g["b1"], g["W2"], g["b2"], g["Wc1"], g["Wc2"], g["Wc3"], float(g["per_level_scale"]))
    assert f.has_viewdirs and f.arrs["Wc1"].shape == (64, 21) and f.arrs["Wsh"].shape == (64, 16)
    return f, table


def check_viewdirs_render(get, g, tag):
    """shared by the CPU (oracle) and GPU tests.  Rays with a recorded up-sampling flip (tests/golden/make_viewdirs_golden.py: exactly that set, none
    tolerated beyond it) are compared through their pixels only: one new sample sits in the neighbouring bin, every later per-sample value is shifted."""
    d = lambda k: np.abs(get(k).reshape(g[f"{tag}_{k}"].shape) - g[f"{tag}_{k}"])
    flips = np.nonzero(d("z_vals").max(1) > 2e-3)[0]
    assert np.array_equal(flips, g[f"{tag}_oracle_z_flips"]), (flips.tolist(), g[f"{tag}_oracle_z_flips"].tolist())
    ok = np.ones(g["rays_o"].shape[0], bool); ok[flips] = False
    assert d("image").max() <= 1e-3 and d("weights_sum").max() <= 1e-3 and d("depth").max() <= 1e-3 and d("normal_map").max() <= 2e-3
    assert d("color")[ok].max() <= 1e-3 and d("alpha")[ok].max() <= 2e-3 and d("weights")[ok].max() <= 2e-3


@pytest.mark.parametrize("tag", ["eval", "train"])
def test_render_with_view_directions_vs_reference(oracle, tag):
    g = load_golden("viewdirs.npz")
    f, _ = viewdirs_field(oracle, g)
    noise = g["train_noise"] if tag == "train" else None
    r = oracle.render_rays(f, g["rays_o"], g["rays_d"], 64, 64, 1.6, float(g["inv_s"]), bg=g["bg"], noise=noise)
    check_viewdirs_render(lambda k: np.asarray(r[k]), g, tag)
    assert abs(float(r["gradient_error"]) - float(g[f"{tag}_gradient_error"])) <= 1e-4
    assert float(g["view_dependence_max"]) > 5e-3                # the direction columns matter in this fixture: leaving them out would not pass
    # ... which is what happens when the field is built without them
    f0 = oracle.Field(f.arrs["table"], g["offsets"], g["W1"], g["b1"], g["W2"], g["b2"], f.arrs["Wc1"], g["Wc2"], g["Wc3"], float(g["per_level_scale"]))
    r0 = oracle.render_rays(f0, g["rays_o"], g["rays_d"], 64, 64, 1.6, float(g["inv_s"]), bg=g["bg"], noise=noise)
    assert np.abs(np.asarray(r0["image"]) - g[f"{tag}_image"]).max() > 5e-3


def test_colour_of_a_point_is_the_37_input_network(oracle):
    """the stand-alone colour query with a direction per point against a plain numpy evaluation of sigmoid(Wc3 relu(Wc2 relu(Wc1_37 [x, sh(d), n, feat])))"""
    g = load_golden("viewdirs.npz")
    f, _ = viewdirs_field(oracle, g)
    rs = np.random.RandomState(0)
    B = 257
    x = rs.uniform(-1, 1, (B, 3)).astype(np.float32); n = rs.normal(size=(B, 3)).astype(np.float32); n /= np.linalg.norm(n, axis=1, keepdims=True)
    d = rs.normal(size=(B, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    s16 = rs.normal(0, 0.3, (B, 16)).astype(np.float32)
    rgb = f.color(x, n, s16, dirs=d)
    sh, _ = oracle.sh_encode_forward(d, 4)
    h = np.concatenate([x, sh, n, s16[:, 1:]], 1).astype(np.float64)
    h = np.maximum(h @ g["Wc1"].astype(np.float64).T, 0); h = np.maximum(h @ g["Wc2"].astype(np.float64).T, 0)
    ref = 1.0 / (1.0 + np.exp(-(h @ g["Wc3"].astype(np.float64).T)))
    assert np.abs(rgb - ref).max() <= 2e-6
    with_zero_dir = f.color(x, n, s16, dirs=np.zeros_like(d))
    assert np.abs(with_zero_dir - rgb).max() > 1e-3


def test_backward_with_view_directions_matches_reference_autograd(oracle):
    O = oracle
    g = load_golden("viewdirs.npz")
    f, _ = viewdirs_field(O, g)
    r = O.render_core_backward(f, g["g_rays_o"], g["g_rays_d"], g["g_z_vals"], 64, 64, 1.6, float(g["inv_s"]), bg=g["g_bg"], g_image=g["g_img_grad"], g_eik=0.01)
    assert np.abs(r["image"] - g["g_rgb"]).max() <= 2e-5
    assert r["g_Wc1_37"].shape == (64, 37) and np.abs(r["g_Wsh"]).max() > 0
    r["g_Wc1"] = r["g_Wc1_37"]                                   # the reference's parameter is the [64,37] matrix
    raw = _chain_to_raw(O, g, r)
    for k, mine in raw.items():
        ref = g["grad." + k].astype(np.float64)
        e = float(np.abs(np.asarray(mine).reshape(ref.shape) - ref).max() / np.abs(ref).max())
        assert e <= (5e-3 if k == "color_net.0.weight_v" else 3e-4), (k, e)      # (one ReLU of layer 1 on the other side in fp32: see test_oracle_backward)
    gv = np.asarray(raw["color_net.0.weight_v"]).reshape(64, 37)
    ref = g["grad.color_net.0.weight_v"].astype(np.float64)
    assert np.abs(gv[:, 3:19] - ref[:, 3:19]).max() <= 5e-3 * np.abs(ref).max() and np.abs(ref[:, 3:19]).max() > 0.1 * np.abs(ref).max()
    assert np.abs(r["g_table"][g["emb_idx"]] - g["emb_grad"]).max() <= 3e-4 * np.abs(g["emb_grad"]).max()
