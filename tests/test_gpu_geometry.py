"""-m gpu: csrc/geometry.hip through the C ABI -- the SDF on a grid (ac_field_sdf_grid), marching cubes on the device (ac_marching_cubes_*) and the
one-launch density-grid update (ac_density_grid_update) against the CPU oracle (bit for bit: same table, same order, same double arithmetic) and against
the torch formulation of the reference's chain (models/instant_nsr.py:303-356, 706-764)."""
import numpy as np
import pytest
import torch

from tests.common import load_golden, make_rays
from tests.test_gpu_model import golden_net, DEV
from tests.test_oracle_geometry import mesh_checks, crossing_edges, pad_inside

pytestmark = pytest.mark.gpu


def test_sdf_grid_equals_point_queries_bit_for_bit():
    from avatarcraft_amd import nsr_ops
    net, _ = golden_net()
    f = net._field()
    for shape in ((37, 37, 37), (5, 19, 33), (16, 1, 16)):
        axes = [torch.linspace(-1.6, 1.6, n).to(DEV) if n > 1 else torch.tensor([0.3], device=DEV) for n in shape]
        vol = nsr_ops.field_sdf_grid(f, *axes, 1.6)
        xx, yy, zz = torch.meshgrid(*axes, indexing="ij")
        pts = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1).contiguous()
        ref = nsr_ops.field_sdf(f, pts, 1.6)[:, 0].reshape(shape)
        assert torch.equal(vol, ref)
        neg = nsr_ops.field_sdf_grid(f, *axes, 1.6, negate=True)
        assert torch.equal(neg, -ref)
    # the model's surface: extract_fields (host array, the reference's return type) and the device volume
    u = net.extract_fields(1.6, 33)
    assert u.dtype == np.float32 and np.array_equal(u, net.extract_fields_device(1.6, 33).cpu().numpy())
    assert np.abs(u - load_golden("density_grid.npz")["sdf33"]).max() < 1e-5        # recorded from the reference's extract_fields


def _gpu_mc(u, iso=0.0, **kw):
    from avatarcraft_amd import nsr_ops
    v, t = nsr_ops.marching_cubes(torch.from_numpy(np.ascontiguousarray(u, dtype=np.float32)).to(DEV), iso, **kw)
    return v.cpu().numpy(), t.cpu().numpy()


def test_marching_cubes_equals_oracle_bit_for_bit(oracle):
    n = 40
    ax = np.linspace(-1, 1, n)
    x, y, z = np.meshgrid(ax, ax, ax, indexing="ij")
    sphere = (0.62 - np.sqrt(x * x + y * y + z * z)).astype(np.float32)
    rs = np.random.RandomState(3)
    vols = [(sphere, 0.0, dict(den=n - 1.0, span=[2.0] * 3, lo=[-1.0] * 3)),
            (sphere, 0.11, dict()),
            (pad_inside(rs.uniform(-1, 1, (31, 27, 29)).astype(np.float32), -1.0), 0.0, dict()),          # every configuration, odd sizes: 33 x 29 x 31
            (rs.uniform(-1, 1, (17, 40, 23)).astype(np.float32), 0.2, dict(den=3.0, span=[1.0, 2.0, 3.0], lo=[0.5, -0.5, 0.0])),   # open at the boundary
            (np.full((9, 9, 9), 1.0, np.float32), 0.0, dict())]                                            # no surface at all
    for u, iso, kw in vols:
        v, t = _gpu_mc(u, iso, **kw)
        vo, to = oracle.marching_cubes(u, iso, **kw)
        assert v.shape == vo.shape and t.shape == to.shape
        assert np.array_equal(t, to)
        assert np.array_equal(v.view(np.uint64), vo.view(np.uint64))
        assert len(v) == crossing_edges(u, iso)
    with pytest.raises(RuntimeError):
        _gpu_mc(np.zeros((1, 4, 4), np.float32))


def test_mesh_of_the_field_is_closed_outward_and_close_to_the_tetrahedral_mesh(oracle):
    from avatarcraft_amd import nsr_ops
    from avatarcraft_amd.geometry import marching_tetrahedra
    net, p = golden_net()
    res = 128
    u = net.extract_fields_device(1.6, res, negate=True)
    verts, tris = net.extract_geometry(1.6, res)
    assert verts.dtype == np.float64 and verts.shape[1] == 3 and tris.shape[1] == 3 and len(tris) > 5000 and np.abs(verts).max() <= 1.6
    un = u.cpu().numpy()
    assert len(verts) == crossing_edges(un, 0.0)                           # vertex set == the sign-changing grid edges
    mesh_checks(verts, tris)                                               # closed, one orientation
    vo, to = oracle.marching_cubes(un, 0.0, den=res - 1.0, span=[float(np.float32(1.6) - np.float32(-1.6))] * 3, lo=[float(np.float32(-1.6))] * 3)
    assert np.array_equal(tris, to) and np.array_equal(verts.view(np.uint64), vo.view(np.uint64))
    a, b, c = (verts[tris[:, k]] for k in range(3))
    assert np.einsum("ij,ij->i", a, np.cross(b, c)).sum() > 0              # positive enclosed volume: normals point out of the body
    nrm = np.cross(b - a, c - a)
    cen = torch.from_numpy(((a + b + c) / 3).astype(np.float32)).to(DEV)
    with torch.no_grad():
        gsd = net.gradient(cen, 1.6, 0.005).cpu().numpy()
        sd = net.density(torch.from_numpy(verts.astype(np.float32)).to(DEV), 1.6).cpu().numpy()
    assert ((nrm * gsd).sum(1) > 0).mean() > 0.97                         # ... along the SDF gradient
    assert np.abs(sd).max() < 1e-2                                        # vertices on the zero level set (linear interpolation on a 128^3 grid)
    # Hausdorff distance to the tetrahedral mesh of the same volume <= 1 cell (vertex sets; the cube edges are a subset of the tetrahedra's edges)
    vt, tt = marching_tetrahedra(u, 0.0)
    from scipy.spatial import cKDTree
    vi = (verts - float(np.float32(-1.6))) / float(np.float32(1.6) - np.float32(-1.6)) * (res - 1.0)
    vtn = vt.cpu().numpy().astype(np.float64)
    d_ab = cKDTree(vtn).query(vi)[0]                                        # every marching-cubes vertex IS a tetrahedral vertex (the cube's edges are among theirs)
    d_ba = cKDTree(vi).query(vtn)[0]
    assert float(d_ab.max()) < 1e-3 and float(d_ba.max()) <= 1.0, (float(d_ab.max()), float(d_ba.max()))
    assert len(tt) > 1.5 * len(tris)
    # the other meshers are still there, and the device-tensor form skips the copy
    v2, t2 = net.extract_geometry(1.6, 48, mesher="tetra")
    assert len(t2) > 500
    v3, t3 = net.extract_geometry(1.6, 128, return_torch=True)
    assert v3.is_cuda and t3.is_cuda and np.array_equal(t3.cpu().numpy(), tris)


def test_mesh_export_at_the_references_resolution():
    """the reference's one call: extract_geometry(NSR_BOUND, 512) (stylize.py:267) -- 134 M field evaluations + marching cubes, all on the device"""
    net, _ = golden_net()
    v, t = net.extract_geometry(1.6, 512, return_torch=True)
    assert v.shape[0] > 50000 and t.shape[0] > 100000 and v.dtype == torch.float64 and t.dtype == torch.int32
    assert int(t.max()) == v.shape[0] - 1 and int(t.min()) == 0
    # closed: V - E + F is even and E = 3 F / 2 (every edge shared by two triangles) -- checked through the directed-edge multiset on the device
    e = torch.cat([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]]).long()
    key = e[:, 0] * v.shape[0] + e[:, 1]
    rkey = e[:, 1] * v.shape[0] + e[:, 0]
    ks = torch.sort(key).values
    assert bool((ks[1:] != ks[:-1]).all()) and torch.equal(ks, torch.sort(rkey).values)
    assert float(v.abs().max()) <= 1.6


def test_density_grid_one_launch_equals_the_torch_chain(oracle):
    from avatarcraft_amd.instant_nsr import NeRFNetwork
    from tests.gpu_common import oracle_field as make_of
    src, p = golden_net()

    def make(fused):
        torch.manual_seed(0)
        net = NeRFNetwork(cuda_ray=True)
        net.load_state_dict(src.state_dict(), strict=False)
        net = net.to(DEV).eval()
        net.fused_density_grid = fused
        return net
    a, b = make(True), make(False)
    for rnd in range(3):
        if rnd == 1:
            with torch.no_grad():
                a.sdf_net[1].bias[0] += 0.1; b.sdf_net[1].bias[0] += 0.1          # the surface moves: decay and the maximum merge both matter
        if rnd == 2:
            a.local_step = b.local_step = 3
            a.step_counter[:3, 0] = torch.tensor([300, 500, 100], dtype=torch.int32, device=DEV); b.step_counter.copy_(a.step_counter)
        a.update_extra_state(1.6, decay=0.9); b.update_extra_state(1.6, decay=0.9)
        ga, gb = a.density_grid, b.density_grid
        mx = float(gb.max())
        assert mx > 100 and float((ga - gb).abs().max()) <= 2e-4 * mx, (rnd, float((ga - gb).abs().max()), mx)
        same = float((ga == gb).float().mean())
        assert same > 0.999, (rnd, same)                                   # in fact the same arithmetic: sdf bit-identical, expf / divide of the same library
        assert abs(a.mean_density - b.mean_density) <= 1e-6 * b.mean_density and a.iter_density == b.iter_density == rnd + 1
        assert a.mean_count == b.mean_count and a.local_step == b.local_step == 0
    assert a.mean_count == 300
    # against the oracle's numpy restatement from a zero grid
    c = make(True)
    c.update_extra_state(1.6)
    og, om = oracle.update_density_grid(make_of(p, src.encoder.embeddings.detach().cpu().numpy()), np.zeros((129,) * 3, np.float32), 1.6)
    assert np.abs(c.density_grid.cpu().numpy() - og).max() <= 2e-4 * float(og.max()) and abs(c.mean_density - om) <= 1e-5 * om
    # a non-default grid size through the raw operator: H not a multiple of the brick, one update from a non-zero grid
    from avatarcraft_amd import nsr_ops
    H = 37
    ax = torch.linspace(-1.6, 1.6, H).to(DEV)
    g0 = torch.rand(H, H, H, device=DEV) * 50.0
    g1 = g0.clone()
    mean = nsr_ops.density_grid_update(src._field(), ax, g1, 1.6, 512.0, 0.8)
    xx, yy, zz = torch.meshgrid(ax, ax, ax, indexing="ij")
    sdf = nsr_ops.field_sdf(src._field(), torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1).contiguous(), 1.6)[:, 0]
    e = torch.exp(-512.0 * sdf.abs())
    dens = (512.0 * e / (1 + e)).reshape(H, H, H)
    pool = torch.nn.functional.max_pool3d(torch.nn.functional.pad(dens, (0, 1, 0, 1, 0, 1))[None, None], 2, 1)[0, 0]
    want = torch.maximum(g0 * 0.8, pool)
    assert float((g1 - want).abs().max()) <= 2e-4 * float(want.max()) and abs(float(mean) - float(want.double().mean())) <= 1e-6 * float(want.double().mean())
