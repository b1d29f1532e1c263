"""Generate the 256-case marching-cubes table of csrc/geometry.hip (and its twin for the CPU oracle) from the DEFINITION of the algorithm
(Lorensen & Cline 1987) instead of typing a published table in: for every sign configuration of a cube's 8 corners

  1. every cube edge whose two corners differ carries one surface vertex;
  2. on each of the 6 faces the crossed edges are joined by segments: 2 crossed edges -> 1 segment; 4 crossed edges (the ambiguous face: the two flagged
     corners are diagonal) -> 2 segments that each cut off ONE FLAGGED corner.  The rule looks only at the face's own 4 corner flags, so the two cubes that
     share a face draw the same segments on it: the surface is watertight by construction (the hole problem of the original 15-case table with complement
     symmetry cannot occur);
  3. the segments close into loops (every crossed edge lies on exactly two faces); every loop is oriented so that the surface normal points from the
     UNFLAGGED corners (u > iso: inside the body, u = -sdf) to the FLAGGED ones (u <= iso: outside) and triangulated as a fan from its first vertex.

Conventions (PyMCubes' `marching_cubes(u, iso)`, which the reference calls on u = -sdf, models/instant_nsr.py:755): corner flagged <=> u <= iso; vertex = linear
zero crossing on the grid edge.  Corner c of cell (i, j, k) is (i + (c & 1), j + ((c >> 1) & 1), k + ((c >> 2) & 1)); edge e = (axis a = e >> 2, the
cube's 4 edges along that axis in the order of the remaining two coordinates): edge e joins corner `EDGE[e][0]` to `EDGE[e][1] = EDGE[e][0] + (1 << a)`.

    python tools/gen_mc_table.py            # writes avatarcraft_amd/csrc/ac_mc_table.hpp and oracle/ac_mc_table.h

Checked here: every case's loops are closed, use every crossed edge exactly once, case 0 / 255 are empty, the table of a configuration and of its
complement have the same vertex set, and (by a brute-force sweep over random volumes in tests/test_oracle_geometry.py) the meshes are closed and oriented."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CORNER = [((c & 1), (c >> 1) & 1, (c >> 2) & 1) for c in range(8)]
# edges: axis 0 (x): corners with x = 0 -> +1; order (y, z) = (0,0), (1,0), (0,1), (1,1); axis 1 (y): (x, z); axis 2 (z): (x, y)
EDGE = []
for a in range(3):
    o = [d for d in range(3) if d != a]
    for q in range(4):
        c0 = [0, 0, 0]
        c0[o[0]] = q & 1; c0[o[1]] = (q >> 1) & 1
        c = c0[0] | (c0[1] << 1) | (c0[2] << 2)
        EDGE.append((c, c | (1 << a)))
EDGE_OF = {frozenset(e): i for i, e in enumerate(EDGE)}

# faces: 4 corners in cyclic order, COUNTER-CLOCKWISE seen from outside the cube (outward normal by the right-hand rule)
FACES = [
    (0, 4, 6, 2),   # x = 0  (normal -x)
    (1, 3, 7, 5),   # x = 1  (+x)
    (0, 1, 5, 4),   # y = 0  (-y)
    (2, 6, 7, 3),   # y = 1  (+y)
    (0, 2, 3, 1),   # z = 0  (-z)
    (4, 5, 7, 6),   # z = 1  (+z)
]


def _check_faces():
    import numpy as np
    for f, (axis, side) in zip(FACES, [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (2, 1)]):
        p = np.array([CORNER[c] for c in f], dtype=float)
        assert all(q[axis] == side for q in p)
        n = np.cross(p[1] - p[0], p[2] - p[1])
        want = np.zeros(3); want[axis] = 1.0 if side else -1.0
        assert np.allclose(n, want), (f, n, want)
        for i in range(4):
            assert frozenset((f[i], f[(i + 1) % 4])) in EDGE_OF


def case_triangles(case):
    flag = [(case >> c) & 1 for c in range(8)]
    # directed segments: on a face seen from outside, walking counter-clockwise, a segment goes from the edge where the boundary walk LEAVES the flagged
    # region ... orientation rule: the segment keeps the flagged corner(s) it cuts off on its LEFT when seen from outside the cube.  Then the loop, seen from
    # the flagged (outside-the-body) side, runs counter-clockwise around the normal pointing towards the flagged corners... verified numerically below.
    nxt = {}
    for f in FACES:
        fl = [flag[c] for c in f]
        crossed = [i for i in range(4) if fl[i] != fl[(i + 1) % 4]]          # face edge i joins f[i] -> f[i+1]
        if not crossed:
            continue
        eid = lambda i: EDGE_OF[frozenset((f[i], f[(i + 1) % 4]))]
        if len(crossed) == 2:
            i, j = crossed
            # walking ccw: edge i is where the walk goes flagged -> unflagged or the reverse.  Segment direction: start at the edge where the walk ENTERS
            # the flagged run (unflagged -> flagged), end where it leaves it: the flagged corners then lie to the left of the directed segment j -> i ...
            enter = i if (fl[i] == 0 and fl[(i + 1) % 4] == 1) else j
            leave = j if enter == i else i
            assert fl[leave] == 1 and fl[(leave + 1) % 4] == 0
            nxt.setdefault(eid(leave), []).append(eid(enter))
        else:
            assert len(crossed) == 4
            # ambiguous face: flagged corners are diagonal; each segment cuts off ONE flagged corner f[k]: from edge k (leaving f[k]) back to edge k-1 (entering it)
            for k in range(4):
                if fl[k] == 1:
                    nxt.setdefault(eid(k), []).append(eid((k - 1) % 4))
    # every crossed edge has exactly one outgoing and one incoming segment
    edges = sorted(nxt)
    for e in edges:
        assert len(nxt[e]) == 1, (case, e, nxt[e])
    succ = {e: v[0] for e, v in nxt.items()}
    assert sorted(succ.values()) == edges, case
    crossed_edges = [i for i, (a, b) in enumerate(EDGE) if flag[a] != flag[b]]
    assert edges == crossed_edges, (case, edges, crossed_edges)
    loops, seen = [], set()
    for e in edges:
        if e in seen:
            continue
        loop = [e]; seen.add(e)
        while succ[loop[-1]] != e:
            loop.append(succ[loop[-1]]); seen.add(loop[-1])
        loops.append(loop)
    tris = []
    for loop in loops:
        assert len(loop) >= 3, (case, loop)
        tris.extend(triangulate(loop, case))
    return tris


FACE_SETS = [frozenset(f) for f in FACES]


def share_face(e1, e2):
    """both cube edges lie in one face of the cube: a straight line between points on them runs INSIDE that face"""
    c = set(EDGE[e1]) | set(EDGE[e2])
    return any(c <= f for f in FACE_SETS)


def triangulate(loop, case):
    """Triangles of one loop (a polygon of 3 .. 7 surface vertices, given by their cube edges).  A diagonal between two vertices whose cube edges share a
    face would lie IN that face -- where the neighbouring cell draws its own segments: the two cells' triangles could then overlap or share an edge
    twice (a non-manifold edge).  So: the first triangulation, in the order of the recursive enumeration below (ear (0, i, n-1) splits), none of whose
    diagonals lies in a face.  One exists for every loop of every configuration (asserted)."""
    n = len(loop)

    def rec(lo, hi):
        """all triangulations of the sub-polygon lo, lo+1, ..., hi (indices into loop) that contains the edge (lo, hi)"""
        if hi - lo < 2:
            yield []
            return
        for k in range(lo + 1, hi):
            ok = True
            for (a, b) in ((lo, k), (k, hi)):
                if b - a >= 2 and not (a == 0 and b == n - 1) and share_face(loop[a], loop[b]):
                    ok = False
            if not ok:
                continue
            for left in rec(lo, k):
                for right in rec(k, hi):
                    yield left + [(loop[lo], loop[k], loop[hi])] + right
    for tri in rec(0, n - 1):
        return tri
    raise AssertionError(("no triangulation without an in-face diagonal", case, loop))


def _orientation_ok():
    """numerical check of the orientation rule: for the 8 single-corner cases the triangle normal must point TOWARDS the flagged corner (from inside the
    body, u > iso, to outside, u <= iso), and for every case every triangle's normal must have a positive component along (flagged centroid - unflagged centroid)
    when the case has a single loop of 3 vertices"""
    import numpy as np
    mid = lambda e: (np.array(CORNER[EDGE[e][0]], float) + np.array(CORNER[EDGE[e][1]], float)) / 2
    for c in range(8):
        tris = case_triangles(1 << c)
        assert len(tris) == 1
        a, b, d = (mid(e) for e in tris[0])
        n = np.cross(b - a, d - a)
        to_flagged = np.array(CORNER[c], float) - (a + b + d) / 3
        assert np.dot(n, to_flagged) > 0, c
        # the complement: 7 flagged corners, the normal points away from the single unflagged one
        tris = case_triangles(255 ^ (1 << c))
        assert len(tris) == 1
        a, b, d = (mid(e) for e in tris[0])
        n = np.cross(b - a, d - a)
        assert np.dot(n, to_flagged) < 0, c


def main():
    _check_faces()
    _orientation_ok()
    table = [case_triangles(c) for c in range(256)]
    assert table[0] == [] and table[255] == []
    for c in range(256):
        assert sorted({e for t in table[c] for e in t}) == sorted({e for t in table[255 ^ c] for e in t})
    maxt = max(len(t) for t in table)
    assert maxt <= 7, maxt
    rows = []
    for t in table:
        flat = [e for tri in t for e in tri]
        rows.append(flat + [-1] * (3 * maxt - len(flat)))
    body = []
    body.append("// GENERATED by tools/gen_mc_table.py -- do not edit.  Marching-cubes triangle table derived from the algorithm's definition (see the generator):")
    body.append("// corner c of a cell = (i + (c & 1), j + ((c >> 1) & 1), k + ((c >> 2) & 1)); case bit c set <=> u[corner c] <= iso; edge e: axis e >> 2, joins")
    body.append("// corner AC_MC_EDGE[e][0] to AC_MC_EDGE[e][1]; AC_MC_NTRI[case] triangles, vertices on edges AC_MC_TRI[case][3 t .. 3 t + 2], normals towards u <= iso.")
    body.append("#pragma once")
    body.append("#ifndef AC_MC_CONST")
    body.append("#define AC_MC_CONST static const        /* the HIP translation unit defines it as `static __constant__ const` */")
    body.append("#endif")
    body.append(f"#define AC_MC_MAXTRI {maxt}")
    body.append("AC_MC_CONST unsigned char AC_MC_EDGE[12][2] = {" + ", ".join("{%d, %d}" % e for e in EDGE) + "};")
    body.append("AC_MC_CONST unsigned char AC_MC_NTRI[256] = {" + ", ".join(str(len(t)) for t in table) + "};")
    body.append(f"AC_MC_CONST signed char AC_MC_TRI[256][{3 * maxt}] = {{")
    for r in rows:
        body.append("    {" + ", ".join("%2d" % v for v in r) + "},")
    body.append("};")
    text = "\n".join(body) + "\n"
    for path in (os.path.join(ROOT, "avatarcraft_amd", "csrc", "ac_mc_table.hpp"), os.path.join(ROOT, "oracle", "ac_mc_table.h")):
        with open(path, "w") as f:
            f.write(text)
        print("wrote", path, "max triangles per cell", maxt, "total triangles", sum(len(t) for t in table))


if __name__ == "__main__":
    main()
