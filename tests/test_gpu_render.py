"""-m gpu: the fused renderer (ac_render_rays through the C ABI) against the CPU oracle, bit for bit,
and against the reference-generated goldens within tolerance."""
import numpy as np
import pytest
import torch

from tests.common import load_golden, make_rays
from tests.gpu_common import device_field, oracle_field, assert_bitwise

pytestmark = pytest.mark.gpu

FLOAT_KEYS = ["image", "weights_sum", "depth", "normal_map", "eik", "z_vals", "weights", "alpha", "color", "sdf", "gradient"]


@pytest.fixture(scope="module")
def env(oracle):
    assert torch.cuda.is_available(), "these tests need a GPU"
    p = load_golden("nsr_params.npz")
    f, table = device_field(p)
    return dict(p=p, f=f, of=oracle_field(p, table), O=oracle)


def _run_both(env, ro, rd, T0, up, bg=None, noise=None, car=1.0, precision="exact"):
    from avatarcraft_amd import nsr_ops
    d = "cuda:0"
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(d)
    g = nsr_ops.render_rays(env["f"], t(ro), t(rd), T0, up, 1.6, float(env["p"]["inv_s"]), bg=t(bg), noise=t(noise),
                            cos_anneal_ratio=car, extras=True, debug_indices=True, precision=precision)
    torch.cuda.synchronize()
    r = env["O"].render_rays(env["of"], ro, rd, T0, up, 1.6, float(env["p"]["inv_s"]), bg=bg, noise=noise, cos_anneal_ratio=car)
    return g, r


def _compare_bitwise(g, r, up):
    for k in FLOAT_KEYS:
        assert_bitwise(g[k], r[k], k)
    if up:
        assert_bitwise(g["ss_inds"], r["ss_inds"], "ss_inds")
        assert_bitwise(g["sort_index"], r["sort_index"], "sort_index")
    assert_bitwise(g["gradient_error"].reshape(1), np.float32([r["gradient_error"]]), "gradient_error")


@pytest.mark.parametrize("name", ["eval_64_64", "train_64_64", "eval_32_32", "eval_64_0", "eval_edge", "train_edge"])
def test_render_bitwise_vs_oracle_on_golden_inputs(env, name):
    gd = load_golden(f"run_{name}.npz")
    g, r = _run_both(env, gd["rays_o"], gd["rays_d"], int(gd["num_steps"]), int(gd["upsample_steps"]), gd["bg"], gd.get("noise"))
    _compare_bitwise(g, r, int(gd["upsample_steps"]))


@pytest.mark.parametrize("precision", ["exact", "fast"])
@pytest.mark.parametrize("name", ["eval_64_64", "train_64_64", "eval_32_32", "eval_64_0", "eval_edge", "train_edge"])
def test_render_vs_reference_golden(env, name, precision):
    """GPU output against the reference's own run() output (tests/golden/make_golden.py), in both arithmetic modes of the renderer."""
    gd = load_golden(f"run_{name}.npz")
    g, _ = _run_both(env, gd["rays_o"], gd["rays_d"], int(gd["num_steps"]), int(gd["upsample_steps"]), gd["bg"], gd.get("noise"), precision=precision)
    c = lambda k: g[k].cpu().numpy()
    assert np.abs(c("image") - gd["image"]).max() <= 1e-3          # north-star tolerance: RGB within 1e-3 L_inf
    assert np.abs(c("weights_sum") - gd["weights_sum"]).max() <= 1e-3
    assert np.abs(c("depth") - gd["depth"]).max() <= 1e-3
    assert np.abs(c("normal_map") - gd["normal_map"]).max() <= 2e-3
    assert abs(float(g["gradient_error"]) - float(gd["gradient_error"])) <= 1e-4
    up = int(gd["upsample_steps"]) // 16
    if up:   # sample indices bit-exact on identical seeds
        from tests.test_oracle_golden import _indices_match
        _indices_match(c("ss_inds"), gd["ss_inds"], c("sort_index")[:, :up], gd["sort_index"][:, :up], gd["oracle_ss_flips"])
        assert np.abs(c("z_vals") - gd["z_vals"]).max() <= 2e-3


def test_fast_precision_against_exact(env):
    """ac_render_opts.precision = 1 ("fast": the six finite-difference evaluations of a sample as split-bf16 corrections of the centre's layer 1)
    against precision = 0 (every product an fp32 fma, == the CPU oracle): everything that decides WHERE the samples are -- z values,
    searchsorted indices, sort permutations -- and the sdf itself must be identical bit for bit; normals, colours, weights and pixels
    differ by less than fp32 round-off of the exact mode differs from an fp64 evaluation (observed differences are written to gpurun_out/)"""
    from avatarcraft_amd import nsr_ops
    worst = {}
    cases = [("eval_64_64", None), ("train_64_64", None), ("eval_edge", None), ("eval_32_32", None)]
    ro, rd = make_rays(64, 64, dist=1.7, f=50.0)
    cases.append(("view4096", (ro, rd)))
    for name, rays in cases:
        if rays is None:
            gd = load_golden(f"run_{name}.npz")
            args = (gd["rays_o"], gd["rays_d"], int(gd["num_steps"]), int(gd["upsample_steps"]), gd["bg"], gd.get("noise"))
        else:
            args = (rays[0], rays[1], 64, 64, None, None)
        d = "cuda:0"
        t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(d)
        run = lambda prec: nsr_ops.render_rays(env["f"], t(args[0]), t(args[1]), args[2], args[3], 1.6, float(env["p"]["inv_s"]), bg=t(args[4]), noise=t(args[5]),
                                               extras=True, debug_indices=True, precision=prec)
        e, f = run("exact"), run("fast")
        for k in ("z_vals", "sdf") + (("ss_inds", "sort_index") if args[3] else ()):
            assert torch.equal(e[k], f[k]), (name, k)
        for k, tol in (("image", 2e-4), ("weights_sum", 2e-4), ("depth", 2e-4), ("normal_map", 5e-4), ("weights", 3e-4), ("alpha", 3e-4), ("color", 2e-4)):
            dmax = float((e[k] - f[k]).abs().max())
            worst[f"{name}.{k}"] = dmax
            assert dmax <= tol, (name, k, dmax)
        gn = e["gradient"].norm(dim=-1, keepdim=True).clamp_min(1e-3)
        dg = float(((e["gradient"] - f["gradient"]).abs() / gn).max())
        worst[f"{name}.gradient_rel"] = dg
        assert dg <= 2e-3, (name, dg)
        assert abs(float(e["gradient_error"]) - float(f["gradient_error"])) <= 1e-5
    with pytest.raises(RuntimeError, match="precision"):
        from avatarcraft_amd import _lib as L
        import ctypes as C
        op = L.ac_render_opts(8, 32, 32, 1.6, 1.0, 1.0, 0.005, 0, None, None, None, 7, 0)
        o = L.ac_render_out(); z = torch.zeros(64, device="cuda:0")
        for k in ("image", "weights_sum", "depth", "normal_map", "eik"):
            setattr(o, k, z.data_ptr())
        L.check(L.lib().ac_render_rays(C.byref(env["f"].c), C.byref(op), z.data_ptr(), z.data_ptr(), None, None, z.data_ptr(), z.data_ptr(), C.byref(o), None))
    import json, os
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(worst, open("gpurun_out/fast_vs_exact.json", "w"), indent=1)


def test_prepared_field_is_bit_identical(env):
    """ac_field_prepare (the weights pre-arranged in LDS order) changes nothing but the workgroups' prologue: every output identical, both precisions"""
    from avatarcraft_amd import nsr_ops
    from tests.gpu_common import device_field
    gd = load_golden("run_train_64_64.npz")
    f2, _ = device_field(env["p"])
    f2.prepare()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to("cuda:0")
    for prec in ("exact", "fast"):
        run = lambda f: nsr_ops.render_rays(f, t(gd["rays_o"]), t(gd["rays_d"]), 64, 64, 1.6, float(env["p"]["inv_s"]), bg=t(gd["bg"]), noise=t(gd["noise"]),
                                            extras=True, debug_indices=True, precision=prec)
        a, b = run(env["f"]), run(f2)
        for k in FLOAT_KEYS + ["ss_inds", "sort_index"]:
            assert torch.equal(a[k], b[k]), (prec, k)
    assert env["f"].c.prepared is None and f2.c.prepared


def test_render_bitwise_random_rays_4096(env):
    """a full 4096-ray batch (64x64 view), eval and perturbed"""
    ro, rd = make_rays(64, 64, dist=1.7, f=50.0)
    rs = np.random.RandomState(11)
    bg = rs.uniform(0, 1, (4096, 3)).astype(np.float32)
    sel = rs.choice(4096, 192, replace=False)     # the oracle is ~15 ms/ray: check a random subset bit for bit
    from avatarcraft_amd import nsr_ops
    d = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(d)
    noise = rs.uniform(0, 1, (4096, 64)).astype(np.float32)
    for nz in (None, noise):
        g = nsr_ops.render_rays(env["f"], t(ro), t(rd), 64, 64, 1.6, float(env["p"]["inv_s"]), bg=t(bg),
                                noise=None if nz is None else t(nz), extras=True, debug_indices=True)
        torch.cuda.synchronize()
        r = env["O"].render_rays(env["of"], ro[sel], rd[sel], 64, 64, 1.6, float(env["p"]["inv_s"]), bg=bg[sel],
                                 noise=None if nz is None else nz[sel])
        for k in FLOAT_KEYS[:4] + FLOAT_KEYS[5:]:
            assert_bitwise(g[k][torch.from_numpy(sel).to(d)], r[k], k)
        assert_bitwise(g["ss_inds"][torch.from_numpy(sel).to(d)], r["ss_inds"], "ss_inds")
        assert_bitwise(g["sort_index"][torch.from_numpy(sel).to(d)], r["sort_index"], "sort_index")


def test_render_full_baseline_view_vs_oracle(env):
    """BASELINE configuration 2 at full size: EVERY ray of the 256x256 view (16 launches of 4096 rays, 64+64 samples) against the oracle
    (OpenMP over rays: seconds on the GPU box's host cores).  exact mode: every output bit for bit.  fast mode: sample positions, indices and
    sdf bit for bit, pixels / normals within the tolerance DESIGN section 2 states for it."""
    from avatarcraft_amd import nsr_ops
    ro, rd = make_rays(256, 256, dist=1.7, f=200.0)
    d = "cuda:0"
    inv_s = float(env["p"]["inv_s"])
    r = env["O"].render_rays(env["of"], ro, rd, 64, 64, 1.6, inv_s)
    ro_t, rd_t = torch.from_numpy(ro).to(d), torch.from_numpy(rd).to(d)
    worst = {}
    for precision in ("exact", "fast"):
        for i in range(0, 65536, 4096):
            g = nsr_ops.render_rays(env["f"], ro_t[i:i + 4096], rd_t[i:i + 4096], 64, 64, 1.6, inv_s, extras=True, debug_indices=True,
                                    precision=precision)
            torch.cuda.synchronize()
            sl = slice(i, i + 4096)
            exact_keys = FLOAT_KEYS if precision == "exact" else ["z_vals", "sdf"]
            for k in exact_keys:
                assert_bitwise(g[k], r[k][sl], f"{k} [{precision}, rays {i}..]")
            assert_bitwise(g["ss_inds"], r["ss_inds"][sl], "ss_inds")
            assert_bitwise(g["sort_index"], r["sort_index"][sl], "sort_index")
            if precision == "fast":
                for k, tol in (("image", 2e-5), ("weights_sum", 2e-5), ("normal_map", 5e-5), ("weights", 2e-5), ("color", 2e-5)):
                    e = float(np.abs(g[k].cpu().numpy() - r[k][sl]).max())
                    worst[k] = max(worst.get(k, 0.0), e)
                    assert e <= tol, (k, e, i)
    assert float(np.asarray(r["weights_sum"]).max()) > 0.9 and float(np.asarray(r["weights_sum"]).min()) < 0.1      # the view holds body and background


def test_render_bitwise_edge_case_rays(env):
    """rays the slab test and the samplers rarely see: parallel to an axis (a zero direction component: the reference divides by
    d + 1e-15), starting inside the cube, missing the cube (far < near), grazing a face, pointing away, a very long direction"""
    from tests.common import edge_case_rays
    ro, rd = edge_case_rays()
    rs = np.random.RandomState(3)
    noise = rs.uniform(0, 1, (ro.shape[0], 64)).astype(np.float32)
    for nz in (None, noise):
        g, r = _run_both(env, ro, rd, 64, 64, None, nz)
        _compare_bitwise(g, r, 64)
    g, r = _run_both(env, ro, rd, 32, 16, None, None)
    _compare_bitwise(g, r, 16)


def test_render_rough_field_and_anneal(oracle):
    """amplitude-0.5 random table (|grad| ~ 5, worst-case conditioning) and cos_anneal_ratio != 1"""
    p = load_golden("nsr_params.npz")
    f, table = device_field(p, rough=True)
    env = dict(p=p, f=f, of=oracle_field(p, table), O=oracle)
    ro, rd = make_rays(8, 8, dist=1.5, f=5.0, jitter_seed=5)
    g, r = _run_both(env, ro, rd, 48, 32, None, None, car=0.3)
    _compare_bitwise(g, r, 32)


def test_render_properties_full_size(env):
    """size-independent properties at BASELINE size (256x256 = 16 batches of 4096 rays)"""
    from avatarcraft_amd import nsr_ops
    ro, rd = make_rays(256, 256, dist=1.7, f=200.0)
    d = "cuda:0"
    ro_t, rd_t = torch.from_numpy(ro).to(d), torch.from_numpy(rd).to(d)
    outs = []
    for i in range(0, 65536, 4096):
        o = nsr_ops.render_rays(env["f"], ro_t[i:i + 4096], rd_t[i:i + 4096], 64, 64, 1.6, float(env["p"]["inv_s"]), extras=True)
        outs.append({k: v.clone() for k, v in o.items()})
    torch.cuda.synchronize()
    z = torch.cat([o["z_vals"] for o in outs]); w = torch.cat([o["weights"] for o in outs]); a = torch.cat([o["alpha"] for o in outs])
    ws = torch.cat([o["weights_sum"] for o in outs]); img = torch.cat([o["image"] for o in outs])
    assert torch.isfinite(img).all() and torch.isfinite(z).all()
    assert (z[:, 1:] >= z[:, :-1]).all()                                  # sortedness of the merged samples
    assert (a >= 0).all() and (a <= 1).all() and (w >= 0).all()
    assert (ws <= 1 + 1e-4).all() and (img >= -1e-6).all() and (img <= 1 + 1e-4).all()
    # batching invariance: one 8192-ray launch == two 4096-ray launches, bit for bit
    o2 = nsr_ops.render_rays(env["f"], ro_t[:8192], rd_t[:8192], 64, 64, 1.6, float(env["p"]["inv_s"]))
    torch.cuda.synchronize()
    assert torch.equal(o2["image"], torch.cat([outs[0]["image"], outs[1]["image"]]))


def test_render_bad_arguments(env):
    from avatarcraft_amd import nsr_ops
    ro = torch.zeros(4, 3, device="cuda:0"); rd = torch.ones(4, 3, device="cuda:0")
    with pytest.raises(RuntimeError):
        nsr_ops.render_rays(env["f"], ro, rd, 60, 64, 1.6, 1.0)
    with pytest.raises(RuntimeError):
        nsr_ops.render_rays(env["f"], ro, rd, 64, 80, 1.6, 1.0)
    with pytest.raises(RuntimeError):
        nsr_ops.render_rays(env["f"], ro.cpu(), rd, 64, 64, 1.6, 1.0)
    out = nsr_ops.render_rays(env["f"], ro[:0], rd[:0], 64, 64, 1.6, 1.0)   # empty batch is fine
    assert out["image"].shape == (0, 3)
    # the pair launch: same argument checks (sample counts, device), the noise of both copies is mandatory, an empty batch is fine
    nz = torch.rand(2, 4, 64, device="cuda:0")
    with pytest.raises(RuntimeError):
        nsr_ops.render_rays_pair(env["f"], ro, rd, nz[:, :, :60].contiguous(), 60, 64, 1.6, 1.0)
    with pytest.raises(RuntimeError):
        nsr_ops.render_rays_pair(env["f"], ro.cpu(), rd, nz, 64, 64, 1.6, 1.0)
    with pytest.raises(RuntimeError):
        nsr_ops.render_rays_pair(env["f"], ro, rd, nz[:1], 64, 64, 1.6, 1.0)
    pa, pb = nsr_ops.render_rays_pair(env["f"], ro[:0], rd[:0], nz[:, :0], 64, 64, 1.6, 1.0)
    assert pa["image"].shape == (0, 3) and pb["z_vals"].shape == (0, 128)


def test_field_sdf_color_bitwise(env):
    from avatarcraft_amd import nsr_ops
    fp = load_golden("field_points.npz")
    x = torch.from_numpy(fp["pts"]).to("cuda:0")
    s = nsr_ops.field_sdf(env["f"], x, 1.6)
    so = env["of"].sdf(fp["pts"], 1.6)
    assert_bitwise(s, so, "forward_sdf")
    assert np.abs(s.cpu().numpy() - fp["sdf"]).max() < 5e-6          # vs the reference's forward_sdf
    n = torch.from_numpy(fp["normal"]).to("cuda:0")
    c = nsr_ops.field_color(env["f"], x, n, s)
    co = env["of"].color(fp["pts"], fp["normal"], so)
    assert_bitwise(c, co, "forward_color")
    assert np.abs(c.cpu().numpy() - fp["color"]).max() < 5e-6        # vs the reference's forward_color


# ------------------------------------------------------------------ posed-space rendering (render_can=False)
def _warp_both(env, ro, rd, T0, up, guide, noise=None, body=None):
    from avatarcraft_amd import nsr_ops
    from tests.common import make_body
    verts, faces, Ts = body if body is not None else make_body()
    d = "cuda:0"
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(d)
    wm = nsr_ops.WarpMesh(verts, faces, Ts, d, use_mesh_guide=guide)
    g = nsr_ops.render_rays(env["f"], t(ro), t(rd), T0, up, 1.6, float(env["p"]["inv_s"]), noise=t(noise), extras=True, debug_indices=True, warp=wm)
    torch.cuda.synchronize()
    r = env["O"].render_rays(env["of"], ro, rd, T0, up, 1.6, float(env["p"]["inv_s"]), noise=noise,
                             warp=dict(verts=verts, faces=faces, Ts=Ts, use_mesh_guide=guide))
    return g, r


@pytest.mark.parametrize("T0,up,guide,perturb", [(32, 32, True, False), (32, 32, False, True), (64, 64, True, True), (16, 0, True, False)])
def test_warped_render_bitwise_vs_oracle(env, T0, up, guide, perturb):
    ro, rd = make_rays(20, 20, dist=1.8, f=17.0, jitter_seed=9)
    noise = np.random.RandomState(3).rand(ro.shape[0], T0).astype(np.float32) if perturb else None
    g, r = _warp_both(env, ro, rd, T0, up, guide, noise)
    _compare_bitwise(g, r, up)
    assert_bitwise(g["can_mid"].clamp(-1.6, 1.6), r["can_mid"], "can_mid")     # the scratch holds the points before the clamp to the bound
    assert np.array_equal(g["mask"].cpu().numpy(), r["mask"])
    assert 0.02 < r["mask"].mean() < 0.7 and r["weights_sum"].max() > 0.5


@pytest.mark.parametrize("n", [4097, 5000, 8192 + 24, 65536])
def test_render_batching_invariance_large_and_ragged(env, n):
    """one launch of n rays == 4096-ray launches, bit for bit: above 4096 rays the kernel deals the batch to the XCDs in interleaved 512-ray chunks
    (render_fused.hip, AC_XCD_CHUNK), with a partial last chunk for ragged n -- every ray must be rendered exactly once"""
    from avatarcraft_amd import nsr_ops
    ro, rd = make_rays(256, 256, dist=1.7, f=200.0)
    d = "cuda:0"
    ro_t, rd_t = torch.from_numpy(ro[:n]).to(d).contiguous(), torch.from_numpy(rd[:n]).to(d).contiguous()
    keys = ("image", "depth", "weights_sum", "z_vals", "weights")
    parts = []
    for i in range(0, n, 4096):
        o = nsr_ops.render_rays(env["f"], ro_t[i:i + 4096], rd_t[i:i + 4096], 64, 64, 1.6, float(env["p"]["inv_s"]), extras=True)
        parts.append({k: o[k].clone() for k in keys})
    big = nsr_ops.render_rays(env["f"], ro_t, rd_t, 64, 64, 1.6, float(env["p"]["inv_s"]), extras=True)
    torch.cuda.synchronize()
    for k in keys:
        assert torch.equal(big[k], torch.cat([q[k] for q in parts])), k


@pytest.mark.parametrize("skip", [False, True])
def test_warped_render_whole_frame_in_one_batch(env, skip):
    """what drivers.render_animation does by default: the 65 536 rays of a posed 256x256 frame in ONE batch == the reference driver's 8192-ray batches"""
    from avatarcraft_amd import nsr_ops
    from tests.common import make_body
    verts, faces, Ts = make_body(n_lat=83, n_lon=83)
    ro, rd = make_rays(256, 256, dist=1.8, f=0.78125 * 256)
    d = "cuda:0"
    ro_t, rd_t = torch.from_numpy(ro).to(d), torch.from_numpy(rd).to(d)
    wm = nsr_ops.WarpMesh(verts, faces, Ts, d, use_mesh_guide=True)
    kw = dict(warp=wm, skip_masked=skip)
    parts = []
    for i in range(0, 65536, 8192):
        o = nsr_ops.render_rays(env["f"], ro_t[i:i + 8192], rd_t[i:i + 8192], 32, 32, 1.6, float(env["p"]["inv_s"]), **kw)
        parts.append({k: o[k].clone() for k in ("image", "depth", "weights_sum")})
    big = nsr_ops.render_rays(env["f"], ro_t, rd_t, 32, 32, 1.6, float(env["p"]["inv_s"]), **kw)
    torch.cuda.synchronize()
    for k in ("image", "depth", "weights_sum"):
        assert torch.equal(big[k], torch.cat([q[k] for q in parts])), k
    assert 0.05 < float((big["weights_sum"] > 0.5).float().mean()) < 0.6


def test_warped_render_full_batch_vs_oracle(env):
    """BASELINE configuration 4 at its real size: one 8192-ray batch of the 256x256 posed frame (32+32 samples, mesh-guided range, SMPL-sized body of
    6 891 vertices / 13 778 faces) -- every output of the posed-space renderer, the warp mask and the warped mid points, bit for bit against the oracle
    (whose closest-face search is exhaustive, fp64, OpenMP over rays: seconds on the GPU box's host cores)."""
    from tests.common import make_body
    body = make_body(n_lat=83, n_lon=83)
    assert body[0].shape[0] == 6891 and body[1].shape[0] == 13778
    ro, rd = make_rays(256, 256, dist=1.8, f=0.78125 * 256)
    sl = slice(3 * 8192, 4 * 8192)                                  # rows 96..127: body and background
    g, r = _warp_both(env, ro[sl], rd[sl], 32, 32, True, body=body)
    _compare_bitwise(g, r, 32)
    assert_bitwise(g["can_mid"].clamp(-1.6, 1.6), r["can_mid"], "can_mid")
    assert np.array_equal(g["mask"].cpu().numpy(), r["mask"])
    assert 0.02 < r["mask"].mean() < 0.9 and r["weights_sum"].max() > 0.5 and r["weights_sum"].min() < 0.05


@pytest.mark.parametrize("precision", ["exact", "fast"])
def test_warped_render_skip_masked_tiles(env, precision):
    """ac_render_opts.skip_masked: tiles of 16 samples that the warp masks out are not evaluated -- everything that defines the frame (image,
    weights_sum, depth, normal_map) and the per-sample weights / alpha / z must not move by a bit; skipped samples report sdf = colour = 0"""
    from avatarcraft_amd import nsr_ops
    from tests.common import make_body
    body = make_body()
    ro, rd = make_rays(48, 48, dist=1.8, f=0.78125 * 48)
    d = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(d)
    wm = nsr_ops.WarpMesh(*body, d, use_mesh_guide=True)
    rs = np.random.RandomState(2)
    bg = t(rs.uniform(0, 1, (ro.shape[0], 3)))
    outs = []
    for skip in (False, True):
        g = nsr_ops.render_rays(env["f"], t(ro), t(rd), 32, 32, 1.6, float(env["p"]["inv_s"]), bg=bg, extras=True, warp=wm, precision=precision,
                                skip_masked=skip)
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in g.items() if isinstance(v, torch.Tensor)})
    a, b = outs
    for k in ("image", "weights_sum", "depth", "normal_map", "weights", "alpha", "mask"):
        assert torch.equal(a[k], b[k]), k
    lr = a["mask"].bool().any(1)                                      # rays with an unmasked sample: same sample positions; a ray that provably has
    assert torch.equal(a["z_vals"][lr], b["z_vals"][lr])             # none is not sampled at all (coarse z, padded): finite and sorted is all it needs
    assert torch.isfinite(b["z_vals"]).all() and (b["z_vals"][:, 1:] >= b["z_vals"][:, :-1]).all() and 0.05 < float((~lr).float().mean()) < 0.9
    live = a["mask"].bool()                                           # the canonical points of unmasked samples are the exact ones; the search may
    assert torch.equal(a["can_mid"][live], b["can_mid"][live])        # leave out samples its cell grids prove masked (their point is not used)
    assert int((a["can_mid"][~live] != b["can_mid"][~live]).any(-1).sum()) > 0
    m = a["mask"].reshape(-1, 4, 16).bool()                           # [ray, tile, sample]
    dead = ~m.any(-1)                                                 # tiles without a live sample
    assert 0.2 < float(dead.float().mean()) < 0.95
    sdf_b = b["sdf"].reshape(-1, 4, 16)
    assert float(sdf_b[dead].abs().max()) == 0.0 and float(b["color"].reshape(-1, 4, 16, 3)[dead].abs().max()) == 0.0
    assert torch.equal(a["sdf"][live], b["sdf"][live]) and torch.equal(a["color"][live], b["color"][live])       # unmasked samples: untouched


@pytest.mark.parametrize("tag,guide", [("guide", True), ("noguide", False)])
def test_warped_render_vs_reference_golden(env, tag, guide):
    from tests.test_oracle_golden import check_warp_render_vs_golden
    gd = load_golden("warp_render.npz")
    g, _ = _warp_both(env, gd["rays_o"], gd["rays_d"], 32, 32, guide)
    check_warp_render_vs_golden(lambda k: g[k].cpu().numpy(), gd, tag)


def test_warped_render_bad_arguments(env):
    from avatarcraft_amd import nsr_ops, _lib as L
    import ctypes as C
    from tests.common import make_body
    verts, faces, Ts = make_body()
    with pytest.raises(RuntimeError, match="one 4x4 per vertex"):
        nsr_ops.WarpMesh(verts, faces, Ts[:10], "cuda:0")
    wm = nsr_ops.WarpMesh(verts, faces, Ts, "cuda:0")
    ro, rd = make_rays(4, 4)
    t = lambda a: torch.from_numpy(a).to("cuda:0")
    with pytest.raises(RuntimeError, match="unsupported"):
        nsr_ops.render_rays(env["f"], t(ro), t(rd), 24, 32, 1.6, 1.0, warp=wm)
    # scratch too small is refused
    op = L.ac_render_opts(16, 32, 32, 1.6, 1.0, 1.0, 0.005, 0, None, None, None, 0, 0)
    o = L.ac_render_out()
    for k in ("image", "weights_sum", "depth", "normal_map", "eik"):
        setattr(o, k, torch.empty(64, device="cuda:0").data_ptr())
    lz, lu = nsr_ops.linspace_tables(32, torch.device("cuda:0"))
    sc = torch.empty(1024, dtype=torch.uint8, device="cuda:0")
    rc = L.lib().ac_render_rays_warped(C.byref(env["f"].c), C.byref(op), t(ro).data_ptr(), t(rd).data_ptr(), None, None, lz.data_ptr(), lu.data_ptr(),
                                       C.byref(wm.c), sc.data_ptr(), 1024, C.byref(o), None)
    assert rc != 0 and b"scratch" in L.lib().ac_last_error()


@pytest.mark.parametrize("n", [1, 3, 13])
def test_warped_render_small_batches(env, n):
    """ray counts below one workgroup (8 rays) and not a multiple of it; a mesh the culling structure does not cover falls back"""
    ro, rd = make_rays(8, 8, dist=1.8, f=6.0, jitter_seed=4)
    sel = np.arange(n) * 4 + 9
    g, r = _warp_both(env, ro[sel], rd[sel], 32, 32, True)
    _compare_bitwise(g, r, 32)


@pytest.mark.parametrize("T0,up,perturb", [(64, 64, True), (32, 32, False), (16, 0, False)])
def test_sample_rays_equals_render_z(env, T0, up, perturb):
    """the sampling-only entry point returns exactly the z_vals of the full render"""
    from avatarcraft_amd import nsr_ops
    ro, rd = make_rays(24, 24, dist=1.7, f=18.0, jitter_seed=3)
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, np.float32)).to("cuda:0")
    noise = np.random.RandomState(1).rand(ro.shape[0], T0).astype(np.float32) if perturb else None
    full = nsr_ops.render_rays(env["f"], t(ro), t(rd), T0, up, 1.6, float(env["p"]["inv_s"]), noise=t(noise), extras=True)
    z = nsr_ops.sample_rays(env["f"], t(ro), t(rd), T0, up, 1.6, noise=t(noise))
    assert torch.equal(z, full["z_vals"])


@pytest.mark.parametrize("precision", ["exact", "fast"])
@pytest.mark.parametrize("view", ["bench_batch", "sds_view"])
def test_repeat_launch_soak_is_bit_identical(env, view, precision):
    """Soak of the machinery that could make a launch timing dependent -- rays handed out to waves dynamically from per-XCD counters, the
    staggered start, the finite-difference feature slab aliasing the up-sampling buffers in LDS, gathers under exec masks: the first 4096-ray
    batch of the BASELINE view and the stride-4 training view of the SDS step (jittered samples), rendered 50 times each with every optional
    output kept (per-sample arrays, sample indices, the 7 x 32 stencil features of every sample).  Every repeat must equal the first bit for bit,
    in both arithmetic modes."""
    from avatarcraft_amd import nsr_ops
    import bench
    dev = "cuda:0"
    if view == "bench_batch":
        ro, rd = make_rays(256, 256, dist=1.7, f=200.0, yaw=0.0, pitch=0.0)
        ro, rd, noise = ro[:4096], rd[:4096], None
    else:
        ro, rd = bench.sds_view(0)
        noise = torch.rand((ro.shape[0], 64), generator=torch.Generator().manual_seed(7)).to(dev)
    ro, rd = torch.from_numpy(ro).to(dev), torch.from_numpy(rd).to(dev)
    f = env["f"]
    run = lambda: nsr_ops.render_rays(f, ro, rd, 64, 64, 1.6, float(env["p"]["inv_s"]), noise=noise, extras=True, train_extras=True, debug_indices=True,
                                      precision=precision)
    first = {k: v.clone() for k, v in run().items() if isinstance(v, torch.Tensor)}
    assert "feat7" in first and first["feat7"].numel() == 7 * 8 * 4096 * 128 * 4
    for rep in range(1, 50):
        out = run()
        for k, v in first.items():
            assert torch.equal(v, out[k]), f"repeat {rep}: {k} differs from the first launch ({int((v != out[k]).sum())} values)"


@pytest.mark.parametrize("precision", ["exact", "fast"])
@pytest.mark.parametrize("n", [4096, 777])
def test_pair_launch_equals_two_launches(env, n, precision):
    """ac_render_rays_pair (render_val + the training forward of one stylisation step in one launch, the two copies of a ray neighbours in the hand-out
    order): every per-ray output of both copies and every per-sample output of the second equal two separate ac_render_rays launches bit for bit,
    with different backgrounds and noise per copy; repeated launches are identical."""
    from avatarcraft_amd import nsr_ops
    import bench
    dev = "cuda:0"
    ro, rd = bench.sds_view(1)
    ro, rd = torch.from_numpy(ro[:n].copy()).to(dev), torch.from_numpy(rd[:n].copy()).to(dev)
    g = torch.Generator().manual_seed(11)
    noise2 = torch.rand((2, n, 64), generator=g).to(dev)
    bg2 = torch.rand((2, n, 3), generator=g).to(dev)
    f, inv_s = env["f"], float(env["p"]["inv_s"])
    a = nsr_ops.render_rays(f, ro, rd, 64, 64, 1.6, inv_s, bg=bg2[0], noise=noise2[0], precision=precision)
    b = nsr_ops.render_rays(f, ro, rd, 64, 64, 1.6, inv_s, bg=bg2[1], noise=noise2[1], extras=True, train_extras=True, precision=precision)
    for rep in range(3):
        pa, pb = nsr_ops.render_rays_pair(f, ro, rd, noise2, 64, 64, 1.6, inv_s, bg2=bg2, precision=precision, keep_weights=True)
        for k in ("image", "weights_sum", "depth", "normal_map", "eik", "gradient_error"):
            assert torch.equal(pa[k], a[k]), ("copy a", k, rep)
            assert torch.equal(pb[k], b[k]), ("copy b", k, rep)
        for k in ("z_vals", "weights", "alpha", "color", "sdf", "gradient", "sdf_out16", "pts", "feat7", "eik_res"):
            assert torch.equal(pb[k], b[k]), ("copy b", k, rep)
    assert pb.opts[0].n_rays == n


def test_render_on_concurrent_streams(env):
    """launches issued from several streams at once: every stream owns its hand-out scratch (work counters, per-ray flags and state), so renders that
    overlap in time do not disturb each other -- 4 streams x 6 launches of different batches, each equal to its single-stream result bit for bit"""
    from avatarcraft_amd import nsr_ops
    dev = "cuda:0"
    ro, rd = make_rays(256, 256, dist=1.7, f=200.0, yaw=0.0, pitch=0.0)
    f, inv_s = env["f"], float(env["p"]["inv_s"])
    batches = [(torch.from_numpy(ro[k * 4096:(k + 1) * 4096].copy()).to(dev), torch.from_numpy(rd[k * 4096:(k + 1) * 4096].copy()).to(dev)) for k in (0, 5, 9, 14)]
    ref = [nsr_ops.render_rays(f, o, d, 64, 64, 1.6, inv_s) for o, d in batches]
    ref = [{k: r[k].clone() for k in ("image", "weights_sum", "depth", "normal_map", "gradient_error")} for r in ref]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in batches]
    outs = [[] for _ in batches]
    for rep in range(6):
        for s, (o, d), acc in zip(streams, batches, outs):
            with torch.cuda.stream(s):
                acc.append(nsr_ops.render_rays(f, o, d, 64, 64, 1.6, inv_s))
    torch.cuda.synchronize()
    for r, acc in zip(ref, outs):
        for out in acc:
            for k, v in r.items():
                assert torch.equal(out[k], v), k


def test_render_beside_a_foreign_long_kernel(env):
    """The segment hand-off of the render kernel (a taker spins on a per-ray flag that another resident wave sets) under a co-resident FOREIGN
    workload: long fp32 GEMMs (torch / hipBLASLt, the shape of what the SD UNet puts on the device during stylisation) run on a second stream
    and occupy compute units while 200 render launches go through the first.  Every launch must equal the undisturbed result bit for bit -- a lost
    hand-off would show as the NaN the bounded spin poisons a pixel with (render_fused.hip), a stolen slot as a differing pixel."""
    from avatarcraft_amd import nsr_ops
    dev = "cuda:0"
    ro, rd = make_rays(256, 256, dist=1.7, f=200.0, yaw=0.0, pitch=0.0)
    f, inv_s = env["f"], float(env["p"]["inv_s"])
    batches = [(torch.from_numpy(ro[k * 4096:(k + 1) * 4096].copy()).to(dev), torch.from_numpy(rd[k * 4096:(k + 1) * 4096].copy()).to(dev)) for k in (3, 8)]
    keys = ("image", "weights_sum", "depth", "normal_map", "gradient_error")
    ref = [{k: v.clone() for k, v in nsr_ops.render_rays(f, o, d, 64, 64, 1.6, inv_s).items() if k in keys} for o, d in batches]
    torch.cuda.synchronize()
    # both workloads on streams of their own: work on torch's DEFAULT stream was observed to wait for everything queued earlier on a side stream
    # (the renders all ran after the last GEMM: no co-residency at all)
    side, main = torch.cuda.Stream(), torch.cuda.Stream()
    a = torch.randn(8192, 8192, device=dev); b = torch.randn(8192, 8192, device=dev); c = torch.empty_like(a)
    torch.mm(a, b, out=c); torch.cuda.synchronize()           # (library initialisation outside the measurement)
    stats = {}
    for with_gemm in (False, True):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        if with_gemm:
            with torch.cuda.stream(side):
                g0.record()
                for _ in range(60):                            # 60 x 1.1 TFLOP of fp32 matrix work: outlasts the 200 renders
                    torch.mm(a, b, out=c)
                g1.record()
        outs = []
        with torch.cuda.stream(main):
            e0.record()
            for rep in range(200):                             # back to back, no host synchronisation in between: compared afterwards
                o, d = batches[rep % 2]
                outs.append({k: v for k, v in nsr_ops.render_rays(f, o, d, 64, 64, 1.6, inv_s, out={}).items() if k in keys})
            e1.record()
        torch.cuda.synchronize()
        with torch.cuda.stream(main):
            assert nsr_ops.handoff_timeouts(dev) == 0                  # no taker ever gave up waiting for a segment (it would also be a NaN pixel)
        bad = sum(int(not torch.equal(out[k], ref[rep % 2][k])) for rep, out in enumerate(outs) for k in keys)
        assert all(bool(torch.isfinite(out["image"]).all()) for out in outs)
        assert bad == 0, (with_gemm, bad)
        stats["ms_per_render_beside_gemm" if with_gemm else "ms_per_render_alone"] = e0.elapsed_time(e1) / 200
        if with_gemm:
            stats["gemm_ms_each_while_rendering"] = g0.elapsed_time(g1) / 60
            stats["renders_finished_ms_before_the_gemms"] = e1.elapsed_time(g1)
    torch.cuda.synchronize()
    g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g0.record()
    for _ in range(10):
        torch.mm(a, b, out=c)
    g1.record(); torch.cuda.synchronize()
    stats["gemm_ms_each_alone"] = g0.elapsed_time(g1) / 10
    assert torch.isfinite(c).all()
    import json, os
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(stats, open("gpurun_out/render_beside_gemm.json", "w"))
    # the two workloads really shared the device: the renders were slowed down by the GEMMs and / or the GEMMs by the renders
    assert stats["ms_per_render_beside_gemm"] > 1.05 * stats["ms_per_render_alone"] or stats["gemm_ms_each_while_rendering"] > 1.05 * stats["gemm_ms_each_alone"], stats


def test_opacity_only_render_changes_nothing_but_the_image(env):
    """ac_render_opts.opacity_only (what sds_step asks of the frozen avatar, of which it reads weight_sum only): no colour network; weights_sum, depth,
    normal_map and gradient_error bit for bit the full render's, the image is the background over a black body"""
    from avatarcraft_amd import nsr_ops
    ro, rd = make_rays(64, 64, dist=1.7, f=50.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to("cuda:0")
    bg = t(np.random.RandomState(2).uniform(0, 1, (4096, 3)))
    for prec in ("exact", "fast"):
        full = nsr_ops.render_rays(env["f"], t(ro), t(rd), 64, 64, 1.6, float(env["p"]["inv_s"]), bg=bg, precision=prec, out={})
        lean = nsr_ops.render_rays(env["f"], t(ro), t(rd), 64, 64, 1.6, float(env["p"]["inv_s"]), bg=bg, precision=prec, opacity_only=True, out={})
        for k in ("weights_sum", "depth", "normal_map", "gradient_error"):
            assert torch.equal(full[k], lean[k]), (prec, k)
        assert torch.allclose(lean["image"], (1.0 - lean["weights_sum"])[:, None] * bg, atol=1e-6) and not torch.equal(full["image"], lean["image"])


@pytest.mark.parametrize("skip", [False, True])
def test_temporal_seeds_of_the_closest_face_search_change_no_bit(env, skip):
    """round 6: ac_warp_mesh.seed_faces -- every (ray, sample slot) starts its closest-face search from the face the PREVIOUS frame found
    for it.  A five-frame animation of the SMPL-sized body (synthetic.make_body_sequence), 96 x 96 rays: every output of the posed-space renderer, the warp
    mask and the warped mid points of every frame equal the seedless render bit for bit -- with garbage seeds (face ids of another topology, out of range,
    -1), with the previous frame's, and with a view change in between; the searches do less work with the previous frame's seeds than without"""
    from avatarcraft_amd import nsr_ops
    from avatarcraft_amd.synthetic import make_body_sequence
    seq_v, faces, seq_T = make_body_sequence(5, 83, 83)
    ro, rd = make_rays(96, 96, dist=1.8, f=0.78125 * 96)
    ro2, rd2 = make_rays(96, 96, dist=1.8, f=0.78125 * 96, yaw=0.9)
    d = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(d)
    N = ro.shape[0]
    keys = ("image", "weights_sum", "depth", "normal_map", "mask", "can_mid")

    def render(wm, o, dd):
        g = nsr_ops.render_rays(env["f"], t(o), t(dd), 32, 32, 1.6, float(env["p"]["inv_s"]), extras=True, warp=wm, skip_masked=skip)
        torch.cuda.synchronize()
        return {k: g[k].clone() for k in keys}
    seeds = nsr_ops.WarpMesh.new_seed_buffer(N, 32 + 64, d)
    rs = np.random.RandomState(5)
    seeds.copy_(torch.from_numpy(rs.randint(-3, 40000, (N, 96)).astype(np.int32)))       # garbage: valid ids, ids >= F, negatives
    work = {True: 0, False: 0}
    for fi, (v, T_) in enumerate(zip(seq_v, seq_T)):
        o, dd = (ro2, rd2) if fi == 3 else (ro, rd)                  # (frame 3 from another camera: the seeds are still real faces, just worse ones)
        wm_a = nsr_ops.WarpMesh(v, faces, T_, d, use_mesh_guide=True)
        ref = render(wm_a, o, dd)
        work[False] += wm_a.work_counters()["exact_tests"]
        wm_b = nsr_ops.WarpMesh(v, faces, T_, d, use_mesh_guide=True)
        wm_b.bind_seeds(seeds)
        got = render(wm_b, o, dd)
        if fi > 0:
            work[True] += wm_b.work_counters()["exact_tests"]
        else:
            work[False] -= wm_a.work_counters()["exact_tests"]       # (frame 0 ran on garbage seeds: not part of the comparison of work)
        for k in keys:
            assert torch.equal(got[k], ref[k]), (fi, k)
        live = ref["mask"].bool()
        assert 0.02 < float(live.float().mean()) < 0.9
        assert int((seeds >= 0).sum()) > 0 and int(seeds.max()) < faces.shape[0] + 40000
    assert work[True] < work[False], work                            # tighter first bounds: fewer exact point-triangle tests over frames 1 .. 4
    with pytest.raises(RuntimeError):
        nsr_ops.WarpMesh(seq_v[0], faces, seq_T[0], d).bind_seeds(torch.zeros(N, 96, device=d))          # (not int32)
    print(f"temporal seeds (skip_masked={skip}): exact tests over 4 frames {work[False]} -> {work[True]}")


def test_meshes_prepared_on_the_side_stream_render_the_same_frames(env):
    """nsr_ops.warp_mesh_sequence (the loop of drivers.render_animation): the next frame's upload + culling structure are queued on a side stream beside the
    current frame's render.  Six frames of the SMPL-sized animation, no synchronisation between frames (the consumer's stream waits through events only): every
    output equals the frame rendered from a mesh built in line on the consumer's stream, bit for bit; bad faces are refused on the host, before any upload."""
    from avatarcraft_amd import nsr_ops
    from avatarcraft_amd.synthetic import make_body_sequence
    seq_v, faces, seq_T = make_body_sequence(6, 83, 83)
    ro, rd = make_rays(96, 96, dist=1.8, f=0.78125 * 96)
    d = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(d)
    ro_t, rd_t = t(ro), t(rd)
    keys = ("image", "weights_sum", "depth", "normal_map", "mask", "can_mid")
    outs = {}
    for overlap in (True, False, True):
        frames = []
        for wm in nsr_ops.warp_mesh_sequence(zip(seq_v, seq_T), faces, d, overlap=overlap):
            g = nsr_ops.render_rays(env["f"], ro_t, rd_t, 32, 32, 1.6, float(env["p"]["inv_s"]), extras=True, warp=wm, skip_masked=True, out={})
            frames.append({k: g[k].clone() for k in keys})              # (no synchronize: the clones are stream-ordered behind the render)
        torch.cuda.synchronize()
        outs.setdefault(overlap, []).append(frames)
    for run in outs[True]:
        for fi, (a, b) in enumerate(zip(run, outs[False][0])):
            for k in keys:
                assert torch.equal(a[k], b[k]), (fi, k)
    assert not torch.equal(outs[False][0][0]["image"], outs[False][0][5]["image"])       # (the animation moves)
    bad = np.array(faces, copy=True); bad[7, 1] = seq_v[0].shape[0] + 3
    with pytest.raises(RuntimeError, match="one 4x4 per vertex"):
        next(iter(nsr_ops.warp_mesh_sequence(zip(seq_v, seq_T), bad, d)))
    with pytest.raises(RuntimeError, match="one 4x4 per vertex"):
        nsr_ops.WarpMesh(seq_v[0], bad, seq_T[0], d)
