"""CPU: the marching-cubes restatement of the mesh export (oracle/ac_oracle_geometry.c, table from tools/gen_mc_table.py) against the DEFINITION of the
algorithm -- PyMCubes, which the reference calls (models/instant_nsr.py:757), is not in this image: "unpinned vs PyMCubes, pinned vs the definition".
What the definition fixes: one vertex per sign-changing grid edge, on that edge at the linear zero crossing; a watertight, consistently oriented
surface with the normals towards u <= iso; the same surface (to within a cell) as any other correct mesher of the same volume."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mesh_checks(verts, tris):
    """closed + consistently oriented: every directed edge appears once and its reverse appears once"""
    e = np.concatenate([tris[:, [0, 1]], tris[:, [1, 2]], tris[:, [2, 0]]]).astype(np.int64)
    key = e[:, 0] * len(verts) + e[:, 1]
    rkey = e[:, 1] * len(verts) + e[:, 0]
    assert len(np.unique(key)) == len(key), "a directed edge is used twice: inconsistent orientation or a duplicated triangle"
    assert np.isin(rkey, key).all(), "an edge without its opposite: the surface has a hole"
    assert (tris[:, 0] != tris[:, 1]).all() and (tris[:, 1] != tris[:, 2]).all() and (tris[:, 0] != tris[:, 2]).all()
    assert len(np.unique(tris)) == len(verts), "every vertex is used"


def crossing_edges(u, iso):
    f = u <= iso
    return int((f[1:] != f[:-1]).sum() + (f[:, 1:] != f[:, :-1]).sum() + (f[:, :, 1:] != f[:, :, :-1]).sum())


def pad_inside(u, value):
    """a one-point frame of `value` around the volume, so that no surface is cut open by the volume's boundary"""
    p = np.full(tuple(s + 2 for s in u.shape), value, dtype=np.float32)
    p[1:-1, 1:-1, 1:-1] = u
    return p


def test_sphere_is_closed_outward_and_on_the_level_set(oracle):
    n = 28
    ax = np.linspace(-1, 1, n)
    x, y, z = np.meshgrid(ax, ax, ax, indexing="ij")
    u = (0.62 - np.sqrt(x * x + y * y + z * z)).astype(np.float32)           # u = -sdf: positive inside
    v, t = oracle.marching_cubes(u, 0.0, den=n - 1.0, span=[2.0] * 3, lo=[-1.0] * 3)
    assert len(v) == crossing_edges(u, 0.0)
    mesh_checks(v, t)
    assert len(v) - 3 * len(t) // 2 + len(t) == 2                            # Euler characteristic of a sphere
    assert np.abs(np.linalg.norm(v, axis=1) - 0.62).max() < 2e-3
    a, b, c = v[t[:, 0]], v[t[:, 1]], v[t[:, 2]]
    vol = np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6
    assert 0.97 * 4 / 3 * np.pi * 0.62 ** 3 < vol < 4 / 3 * np.pi * 0.62 ** 3    # positive: the normals point out of the body (towards u <= iso)
    nrm = np.cross(b - a, c - a)
    assert (np.einsum("ij,ij->i", nrm, (a + b + c) / 3) > 0).all()
    # index space (PyMCubes' own output): den = 1, span = 1, lo = 0
    vi, ti = oracle.marching_cubes(u, 0.0)
    assert np.array_equal(ti, t) and np.abs(vi / (n - 1.0) * 2.0 - 1.0 - v).max() < 1e-15


@pytest.mark.parametrize("seed,shape", [(0, (13, 11, 12)), (1, (9, 17, 8)), (2, (16, 16, 16))])
def test_every_configuration_is_watertight(oracle, seed, shape):
    """white noise: all 256 corner configurations, ambiguous faces and interiors included, many times over.  The surface must still close (the face rule of
    the table looks at the face's own 4 flags only) and keep one orientation; every vertex sits on its grid edge at the linear zero crossing."""
    rs = np.random.RandomState(seed)
    u = pad_inside(rs.uniform(-1, 1, shape).astype(np.float32), -1.0)
    cases = set()
    f = (u <= 0.0).astype(np.int64)
    cs = sum(f[(c & 1):u.shape[0] - 1 + (c & 1), ((c >> 1) & 1):u.shape[1] - 1 + ((c >> 1) & 1), ((c >> 2) & 1):u.shape[2] - 1 + ((c >> 2) & 1)] << c for c in range(8))
    cases.update(np.unique(cs).tolist())
    assert len(cases) >= 240
    v, t = oracle.marching_cubes(u, 0.0)
    assert len(v) == crossing_edges(u, 0.0)
    mesh_checks(v, t)
    frac = v - np.floor(v)
    on_edge = (frac > 0).sum(1)
    assert (on_edge <= 1).all()                                              # two coordinates are grid indices, the third lies inside one edge
    i0 = np.floor(v).astype(np.int64)
    axis = np.argmax(frac, 1)
    i1 = i0.copy(); i1[np.arange(len(v)), axis] += (on_edge == 1)
    ua, ub = u[i0[:, 0], i0[:, 1], i0[:, 2]].astype(np.float64), u[i1[:, 0], i1[:, 1], i1[:, 2]].astype(np.float64)
    m = on_edge == 1
    assert ((ua[m] <= 0) != (ub[m] <= 0)).all()
    assert np.abs(frac[np.arange(len(v)), axis][m] - (0.0 - ua[m]) / (ub[m] - ua[m])).max() < 1e-12
    # the signed volume of the closed surface = the volume of the region u > iso under trilinear-ish interpolation: positive, below the flagged-corner count
    a, b, c = v[t[:, 0]], v[t[:, 1]], v[t[:, 2]]
    assert np.einsum("ij,ij->i", a, np.cross(b, c)).sum() > 0


def test_same_surface_as_the_tetrahedral_mesher(oracle):
    """two blobs and a handle: marching cubes and the marching-tetrahedra mesher of round 3 (avatarcraft_amd/geometry.py) triangulate differently but
    share the vertex positions on common edges; the two surfaces lie within one cell of each other (both ways)."""
    import torch
    from avatarcraft_amd.geometry import marching_tetrahedra
    n = 40
    ax = np.linspace(-1.2, 1.2, n)
    x, y, z = np.meshgrid(ax, ax, ax, indexing="ij")
    s1 = np.sqrt((x + 0.35) ** 2 + y * y + z * z) - 0.45
    s2 = np.sqrt((x - 0.4) ** 2 + (y - 0.1) ** 2 + z * z) - 0.38
    tor = np.sqrt((np.sqrt(y * y + z * z) - 0.7) ** 2 + x * x) - 0.12
    u = (-np.minimum(np.minimum(s1, s2), tor)).astype(np.float32)
    v, t = oracle.marching_cubes(u, 0.0)
    mesh_checks(v, t)
    vt, tt = marching_tetrahedra(torch.from_numpy(u), 0.0)
    vt = vt.numpy().astype(np.float64)
    # the cube's 12 edges are among the tetrahedra's 19: every marching-cubes vertex is a tetrahedral vertex too
    d = np.sqrt(((v[:, None, :] - vt[None, :, :]) ** 2).sum(-1))
    assert d.min(1).max() < 1e-5
    assert d.min(0).max() <= 1.0                                             # and no tetrahedral vertex is further than one cell from the cube mesh
    assert len(tt) > 1.5 * len(t)                                            # (the reason the tetrahedral mesh was replaced: ~2x the triangles)


def test_committed_table_is_what_the_generator_writes():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_mc_table as G
    table = [G.case_triangles(c) for c in range(256)]
    text = open(os.path.join(ROOT, "oracle", "ac_mc_table.h")).read()
    assert text == open(os.path.join(ROOT, "avatarcraft_amd", "csrc", "ac_mc_table.hpp")).read()
    assert "AC_MC_MAXTRI %d" % max(len(t) for t in table) in text
    rows = [l for l in text.splitlines() if l.startswith("    {")]
    assert len(rows) == 256
    for c, row in enumerate(rows):
        vals = [int(x) for x in row.strip().strip("{},").split(",")]
        flat = [e for tri in table[c] for e in tri]
        assert vals[:len(flat)] == flat and all(x == -1 for x in vals[len(flat):])
    # symmetry properties of the definition: complementary configurations cross the same edges; rotating nothing, the triangle count of a configuration
    # and of its complement may differ (the face rule is not complement-symmetric) but stays within the classical bounds
    assert max(len(t) for t in table) == 5 and sum(len(t) for t in table) == 820
