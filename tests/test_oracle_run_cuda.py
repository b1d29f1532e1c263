"""CPU: the oracle's restatement of the occupancy-grid render chain (NeRFRenderer.run_cuda -- dispatched to by models/instant_nsr.py:362-363, never
defined there; raymarching/raymarching.py:21-188 + update_extra_state :303-356 are what it was meant to chain).  No reference output exists for the
chain itself ("parity unpinned vs the reference": there is nothing to run), so it is pinned piecewise:
  * the marcher / compositor / compaction by the reference's known answers (tests/test_oracle_kat.py);
  * the per-sample field routine (orc_field_samples) against run()'s render core, which IS pinned to the reference's goldens: fed with the mid
    points, directions and section lengths of a run() call it must return that call's alpha / colour / sdf / gradient bit for bit;
  * the chain as a whole against run() on the same field at a variance the occupancy grid was designed for (inv_s = 512, the constant
    update_extra_state hard-codes at :325): two quadratures of the same integral, a few 1e-3 apart."""
import numpy as np
import pytest

from tests.common import make_rays


def test_field_samples_equals_the_render_core_of_run(oracle, oracle_field, golden_params):
    ro, rd = make_rays(12, 12, dist=1.7, f=9.0)
    inv_s = float(golden_params["inv_s"])
    r = oracle.render_rays(oracle_field, ro, rd, 64, 64, 1.6, inv_s)
    z = r["z_vals"]
    N, T = z.shape
    near, far = oracle._near_far_cube(ro, rd, 1.6)
    sample_dist = ((far - near) / np.float32(64)).astype(np.float32)
    delta = np.concatenate([z[:, 1:] - z[:, :-1], sample_dist[:, None]], 1).astype(np.float32)
    zmid = np.concatenate([z[:, :-1] + np.float32(0.5) * delta[:, :-1], z[:, -1:]], 1).astype(np.float32)
    pts = (ro[:, None, :] + rd[:, None, :] * zmid[:, :, None]).astype(np.float32)          # (clamped inside the routine, like new_pts.clamp)
    dirs = np.broadcast_to(rd[:, None, :], pts.shape)
    for stride in (1, 2):
        dl = delta.reshape(-1) if stride == 1 else np.stack([delta.reshape(-1), np.full(N * T, 7.0, np.float32)], 1)
        fs = oracle.field_samples(oracle_field, pts.reshape(-1, 3), np.ascontiguousarray(dirs).reshape(-1, 3), dl, 1.6, 0.005, inv_s, 1.0)
        bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
        assert np.array_equal(bits(fs["alpha"].reshape(N, T)), bits(r["alpha"]))
        assert np.array_equal(bits(fs["rgb"].reshape(N, T, 3)), bits(r["color"]))
        assert np.array_equal(bits(fs["sdf"].reshape(N, T)), bits(r["sdf"]))
        assert np.array_equal(bits(fs["gradient"].reshape(N, T, 3)), bits(r["gradient"]))
    g = fs["gradient"].astype(np.float64)
    n = g / (1e-5 + np.linalg.norm(g, axis=1, keepdims=True))
    assert np.abs(fs["normal"] - n).max() < 1e-6


@pytest.fixture(scope="module")
def grid(oracle, oracle_field):
    g, mean = oracle.update_density_grid(oracle_field, np.zeros((129,) * 3, np.float32), 1.6)
    return g, mean


def test_run_cuda_chain_train_and_eval_agree_with_run(oracle, oracle_field, grid):
    g, mean = grid
    ro, rd = make_rays(16, 16, dist=1.8, f=12.0)
    inv_s = 512.0                                    # the sharpness update_extra_state builds the grid for (:325)
    ref = oracle.render_rays(oracle_field, ro, rd, 64, 64, 1.6, inv_s, extras=False)
    tr = oracle.run_cuda_train(oracle_field, ro, rd, g, mean, 1.6, 0.005, inv_s)
    ev = oracle.run_cuda_eval(oracle_field, ro, rd, g, mean, 1.6, 0.005, inv_s)
    hit = ref["weights_sum"] > 0.5
    assert 0.02 < hit.mean() < 0.9
    for r, tol in ((tr, 5e-3), (ev, 8e-3)):          # eval stops a ray at T < 1e-2 (raymarching.cu:690): up to 1e-2 of opacity left out
        assert np.abs(r["image"] - ref["image"]).max() <= tol and np.abs(r["weights_sum"] - ref["weights_sum"]).max() <= 2 * tol
        dn = np.abs(r["normal_map"] - ref["normal_map"])       # the normal varies fast across the thin shell of this (untrained, rough) field: a
        assert dn.max() <= 0.1 and dn.mean() <= 3e-3             # few silhouette rays weight it differently; the bulk agrees
    assert np.nanmax(np.abs(ev["depth"] - ref["depth"])[hit]) <= 1e-2
    # packed layout: ray n owns samples [offset, offset + count), in ray order; rays that miss the occupied cells own none
    rays, counter = tr["rays"], tr["counter"]
    assert counter[1] == 256 and np.array_equal(rays[:, 0], np.arange(256))
    assert np.array_equal(rays[:, 1], np.concatenate([[0], np.cumsum(rays[:-1, 2])])) and rays[:, 2].sum() == counter[0]
    empty = rays[:, 2] == 0
    assert empty.any() and np.all(tr["weights_sum"][empty] == 0) and ref["weights_sum"][empty].max() < 0.02      # rays past the body march nothing
    assert tr["xyzs"].shape[0] % 128 == 0 and tr["xyzs"].shape[0] >= counter[0]
    # the inference loop: fewer rays alive every round, n_step grows as they die
    a = ev["alive_per_round"]
    assert a[0] == 256 and all(x >= y for x, y in zip(a, a[1:])) and ev["rounds"] == len(a) <= 1024
    assert 0.0 <= tr["gradient_error"] < 10.0


def test_run_cuda_budgeted_march_drops_overflowing_rays(oracle, oracle_field, grid):
    """mean_count > 0 (every epoch after the first): the packed buffers hold mean_count samples rounded up to 128, rays whose samples would not fit are
    dropped by the compositor (raymarching.cu:259-266) -- image 0 + background -- instead of writing out of bounds"""
    g, mean = grid
    ro, rd = make_rays(16, 16, dist=1.8, f=12.0)
    full = oracle.run_cuda_train(oracle_field, ro, rd, g, mean, 1.6, 0.005, 512.0)
    total = int(full["counter"][0])
    tight = oracle.run_cuda_train(oracle_field, ro, rd, g, mean, 1.6, 0.005, 512.0, mean_count=total // 2)
    assert tight["xyzs"].shape[0] == (total // 2) + (128 - (total // 2) % 128)
    M = tight["xyzs"].shape[0]
    fits = (tight["rays"][:, 1] + tight["rays"][:, 2]) < M
    assert fits.any() and (~fits).any()
    assert np.array_equal(tight["image"][fits], full["image"][fits])
    dropped = ~fits & (tight["rays"][:, 2] > 0)
    assert dropped.any() and np.all(tight["weights_sum"][dropped] == 0) and np.all(tight["image"][dropped] == 1.0)
