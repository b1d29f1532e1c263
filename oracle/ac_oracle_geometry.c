/* oracle/ac_oracle_geometry.c -- CPU restatement of the mesh export's marching cubes.  TEST INFRASTRUCTURE ONLY (see ac_oracle.c): imported by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline legs, never by the product.
 *
 * What it restates: mcubes.marching_cubes(u, threshold) as the reference calls it (models/instant_nsr.py:757, on u = -sdf) followed by the scaling of the
 * vertices to world units (:760-762).  PyMCubes is a third-party dependency of the reference (environment.yml: `pymcubes`), not vendored and not installed in
 * this image, so the algorithm is restated from its published definition (Lorensen & Cline 1987: a cell's 8 corner flags select one of 256 triangle
 * configurations; one vertex per sign-changing cell edge, at the linear zero crossing) with PyMCubes' conventions: a corner is flagged when u <= isovalue,
 * the float32 samples are evaluated in double.  The case table (oracle/ac_mc_table.h) is generated from the definition by tools/gen_mc_table.py.
 * PARITY UNPINNED against PyMCubes itself (triangle order, vertex order and the choice on ambiguous faces are implementation details of that package);
 * pinned against the definition by tests/test_oracle_geometry.py: watertight, consistently oriented, one vertex per sign-changing edge, on the edge.
 *
 * Output order (a plain serial loop; the HIP kernels reproduce it bit for bit): vertices by owning grid point (linear index, z fastest), then axis x, y, z;
 * triangles by cell (linear index), then table position. */
#include <stdint.h>
#include <stdlib.h>
#include "ac_mc_table.h"

#define ORC_API __attribute__((visibility("default")))

static int flagged(float u, float iso) { return u <= iso; }

/* counts[0] = vertices, counts[1] = triangles.  verts / tris may be NULL (count only).  Returns 0, or 1 when the scratch allocation fails. */
ORC_API int orc_marching_cubes(const float *vol, uint32_t nx, uint32_t ny, uint32_t nz, float iso, double den, const double *span, const double *lo,
                               double *verts, int32_t *tris, uint32_t *counts)
{
    const size_t npts = (size_t)nx * ny * nz, sx = (size_t)ny * nz, sy = nz;
    uint32_t *voff = (uint32_t *)malloc(npts * sizeof(uint32_t));
    unsigned char *vmask = (unsigned char *)malloc(npts);
    if (!voff || !vmask) { free(voff); free(vmask); return 1; }
    uint32_t nv = 0, nt = 0;
    for (uint32_t i = 0; i < nx; ++i)
        for (uint32_t j = 0; j < ny; ++j)
            for (uint32_t k = 0; k < nz; ++k) {
                const size_t p = (size_t)i * sx + (size_t)j * sy + k;
                const int f0 = flagged(vol[p], iso);
                const size_t st[3] = { sx, sy, 1 };
                const int has[3] = { i + 1 < nx, j + 1 < ny, k + 1 < nz };
                unsigned m = 0;
                voff[p] = nv;
                for (int a = 0; a < 3; ++a) {
                    if (!has[a] || flagged(vol[p + st[a]], iso) == f0) continue;
                    m |= 1u << a;
                    if (verts) {
                        const double va = (double)vol[p], vb = (double)vol[p + st[a]];
                        const double t = ((double)iso - va) / (vb - va);
                        double c[3] = { (double)i, (double)j, (double)k };
                        c[a] += t;
                        for (int q = 0; q < 3; ++q) verts[3 * (size_t)nv + q] = c[q] / den * span[q] + lo[q];
                    }
                    ++nv;
                }
                vmask[p] = (unsigned char)m;
            }
    for (uint32_t i = 0; i + 1 < nx; ++i)
        for (uint32_t j = 0; j + 1 < ny; ++j)
            for (uint32_t k = 0; k + 1 < nz; ++k) {
                const size_t p = (size_t)i * sx + (size_t)j * sy + k;
                unsigned cs = 0;
                for (int c = 0; c < 8; ++c)
                    if (flagged(vol[p + (size_t)(c & 1) * sx + (size_t)((c >> 1) & 1) * sy + (size_t)((c >> 2) & 1)], iso)) cs |= 1u << c;
                const int n = AC_MC_NTRI[cs];
                for (int t = 0; t < n; ++t) {
                    if (tris)
                        for (int q = 0; q < 3; ++q) {
                            const int e = AC_MC_TRI[cs][3 * t + q];
                            const unsigned c0 = AC_MC_EDGE[e][0], a = (unsigned)e >> 2;
                            const size_t pq = p + (size_t)(c0 & 1) * sx + (size_t)((c0 >> 1) & 1) * sy + (size_t)((c0 >> 2) & 1);
                            const unsigned below = vmask[pq] & ((1u << a) - 1u);
                            tris[3 * (size_t)nt + q] = (int32_t)(voff[pq] + (uint32_t)__builtin_popcount(below));
                        }
                    ++nt;
                }
            }
    counts[0] = nv; counts[1] = nt;
    free(voff); free(vmask);
    return 0;
}
