// avatarcraft_amd/csrc/warp.hip -- SMPL-guided sample warp and mesh-guided near/far for gfx950.
//
// Replaces two CPU stages of the reference's animation path (render_warp.py -> NeRFRenderer.run with render_can=False):
//   geometry_guided_near_far_torch (utils/ray_utils.py:277-294): O(N*V) with three [N,V,3] temporaries (0.68 GB each at
//     N=8192) -> mesh_near_far_kernel: lane = ray, 16 waves share 64 rays and split the vertices (LDS tiles), no temporaries;
//   warp_samples_to_canonical (utils/ray_utils.py:62-90): libigl closest-point query + numpy fp64 4x4 inverse on the CPU
//     with two PCIe round trips per ray batch (models/instant_nsr.py:166-172,198-203) -> warp_samples_kernel: one lane
//     per sample, the triangle soup streamed through LDS tiles, exact closest point / barycentric blend / 4x4 inverse in
//     fp64 on the device (fp64 vector rate of MI355X: 78 TFLOP/s).
// Arithmetic order follows oracle/ac_oracle_ops.c (orc_mesh_near_far, orc_warp_samples) operation for operation
// (-ffp-contract=off), so results are bit-identical to the CPU oracle.  warp_samples_kernel is the exhaustive search;
// warp_samples_accel_kernel (further down) returns the same bits from an exact culled search, one wave per sample.
#include "ac_common.hpp"

namespace {

constexpr int VT = 1024;   // vertices per LDS tile (12 KB)
constexpr int FT = 512;    // faces per LDS tile (9 floats each, 18 KB)

// one workgroup = 64 rays x 16 waves: every wave scans a 1/16 slice of each vertex tile for the same 64 rays (lane = ray), the
// per-wave min / max meet in LDS.  min / max are exact and order independent: the result does not depend on the split.
constexpr int NF_WAVES = 16;
__global__ __launch_bounds__(NF_WAVES * 64) void mesh_near_far_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                                      const float *__restrict__ verts, uint32_t N, uint32_t V, float r2,
                                                                      float *__restrict__ near, float *__restrict__ far)
{
    __shared__ float sv[VT * 3];
    __shared__ float sg[VT / 32][4];                      // bounding sphere of every run of 32 consecutive vertices of the tile (centre, padded radius)
    __shared__ float snr[NF_WAVES][64], sfr[NF_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = blockIdx.x * 64 + lane;
    const bool live = n < N;
    const uint32_t nn = live ? n : 0;
    const float ox = rays_o[3 * nn], oy = rays_o[3 * nn + 1], oz = rays_o[3 * nn + 2];
    const float dx = rays_d[3 * nn], dy = rays_d[3 * nn + 1], dz_ = rays_d[3 * nn + 2];
    float nr = __builtin_inff(), fr = -__builtin_inff();
    const float rad = __builtin_sqrtf(r2) * (1.0f + 1e-6f);
    // gridDim.y > 1: this workgroup scans ONE vertex tile and the tiles' results meet in near / far through atomic min / max (both exact and order
    // independent; the host presets +inf / -inf).  A 8192-ray batch is only 128 workgroups otherwise -- half the device idle, and seven dependent
    // load -> barrier -> scan rounds each.
    const uint32_t v_begin = gridDim.y > 1 ? blockIdx.y * (uint32_t)VT : 0u;
    const uint32_t v_end = gridDim.y > 1 ? (v_begin + (uint32_t)VT < V ? v_begin + (uint32_t)VT : V) : V;
    // the run test below bounds a DISTANCE; the per-vertex formula (the reference's) is one only for unit directions: with |d|^2 = 1 +- 2e-6 the two
    // differ by < 2e-6 z0^2, inside the 1e-5 s2 slack of the test.  Other rays (never produced by the drivers): every run is scanned.
    const bool cull_ok = __all(!live || __builtin_fabsf(((dx * dx + dy * dy) + dz_ * dz_) - 1.0f) <= 2e-6f) != 0;
    for (uint32_t v0 = v_begin; v0 < v_end; v0 += VT) {
        const uint32_t cnt = (v_end - v0 < (uint32_t)VT) ? v_end - v0 : (uint32_t)VT;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < cnt * 3; i += blockDim.x) sv[i] = verts[(size_t)v0 * 3 + i];
        __syncthreads();
        // mesh vertex order is spatially coherent (body parts): a run of 32 vertices sits in a small sphere, and the 64 rays of a wave are a thin fan
        // (neighbouring pixels) -- most runs are missed by the whole fan and cost one test instead of 32
        const uint32_t ngr = (cnt + 31u) >> 5;
        if (threadIdx.x < ngr) {
            const uint32_t b0 = threadIdx.x * 32u, b1 = (b0 + 32u < cnt) ? b0 + 32u : cnt;
            float cx = 0.0f, cy = 0.0f, cz = 0.0f;
            for (uint32_t v = b0; v < b1; ++v) { cx += sv[3 * v]; cy += sv[3 * v + 1]; cz += sv[3 * v + 2]; }
            const float inv = 1.0f / (float)(b1 - b0);
            cx *= inv; cy *= inv; cz *= inv;
            float m2 = 0.0f;
            for (uint32_t v = b0; v < b1; ++v) {
                const float ex = sv[3 * v] - cx, ey = sv[3 * v + 1] - cy, ez = sv[3 * v + 2] - cz, e2 = (ex * ex + ey * ey) + ez * ez;
                m2 = e2 > m2 ? e2 : m2;                    // NaN vertices compare false: they are handled (ignored) by the per-vertex code below
            }
            sg[threadIdx.x][0] = cx; sg[threadIdx.x][1] = cy; sg[threadIdx.x][2] = cz;
            sg[threadIdx.x][3] = __builtin_sqrtf(m2) * (1.0f + 1e-5f) + 1e-6f;
        }
        __syncthreads();
        for (uint32_t gi = wave; gi < ngr; gi += NF_WAVES) {
            {   // the whole run: a ray farther than R + r from the centre (as a line, like the per-vertex formula) touches none of its spheres
                const float x = sg[gi][0] - ox, y = sg[gi][1] - oy, z = sg[gi][2] - oz;
                const float z0 = (x * dx + y * dy) + z * dz_, s2 = (x * x + y * y) + z * z;
                const float rr = sg[gi][3] + rad;
                // (a NaN centre -- a NaN vertex in the run -- compares false everywhere: `!(... > ...)` keeps the run)
                if (cull_ok && !__any(!((s2 - z0 * z0) > rr * rr * (1.0f + 1e-5f) + 1e-5f * s2 + 1e-10f))) continue;
            }
            const uint32_t e1 = (gi * 32u + 32u < cnt) ? gi * 32u + 32u : cnt;
        for (uint32_t v = gi * 32u; v < e1; ++v) {
            const float x = sv[3 * v] - ox, y = sv[3 * v + 1] - oy, z = sv[3 * v + 2] - oz;
            const float z0 = (x * dx + y * dy) + z * dz_;
            const float s2 = (x * x + y * y) + z * z;
            // the sphere of this vertex is missed for sure (negative radicand -> NaN -> ignored below) when the squared distance of the
            // vertex from the ray exceeds r^2 by more than the rounding of nrm * nrm; most vertices are far from all 64 rays of the
            // wave, and then both square roots are skipped (wave-uniform branch; the arithmetic of the kept path is unchanged)
            if (!__any((s2 - z0 * z0) - r2 <= 1e-6f * s2 + 1e-12f)) continue;
            const float nrm = __builtin_sqrtf(s2);
            const float dz = __builtin_sqrtf(r2 - (nrm * nrm - z0 * z0));
            const float a = z0 - dz, b = z0 + dz;
            if (a == a && a < nr) nr = a;
            if (b == b && b > fr) fr = b;
        }
        }
    }
    snr[wave][lane] = nr; sfr[wave][lane] = fr;
    __syncthreads();
    if (wave == 0 && live) {
#pragma unroll
        for (int w = 1; w < NF_WAVES; ++w) { const float a = snr[w][lane], b = sfr[w][lane]; nr = a < nr ? a : nr; fr = b > fr ? b : fr; }
        if (gridDim.y == 1) { near[n] = nr; far[n] = fr; }
        else {
            uint32_t *pn = reinterpret_cast<uint32_t *>(near + n), *pf = reinterpret_cast<uint32_t *>(far + n);
            for (uint32_t old = *pn; nr < __uint_as_float(old);) { const uint32_t was = atomicCAS(pn, old, __float_as_uint(nr)); if (was == old) break; old = was; }
            for (uint32_t old = *pf; fr > __uint_as_float(old);) { const uint32_t was = atomicCAS(pf, old, __float_as_uint(fr)); if (was == old) break; old = was; }
        }
    }
}

#define DOT3(u, v) ((u)[0] * (v)[0] + (u)[1] * (v)[1] + (u)[2] * (v)[2])

// Ericson, Real-Time Collision Detection 5.1.5 (same branch order as the oracle)
__device__ __forceinline__ void closest_pt_tri(const double (&p)[3], const double (&a)[3], const double (&b)[3], const double (&c)[3],
                                               double (&out)[3])
{
    double ab[3], ac[3], ap[3], bp[3], cp[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; ap[i] = p[i] - a[i]; }
    const double d1 = DOT3(ab, ap), d2 = DOT3(ac, ap);
    if (d1 <= 0.0 && d2 <= 0.0) { out[0] = a[0]; out[1] = a[1]; out[2] = a[2]; return; }
#pragma unroll
    for (int i = 0; i < 3; i++) bp[i] = p[i] - b[i];
    const double d3 = DOT3(ab, bp), d4 = DOT3(ac, bp);
    if (d3 >= 0.0 && d4 <= d3) { out[0] = b[0]; out[1] = b[1]; out[2] = b[2]; return; }
    const double vc = d1 * d4 - d3 * d2;
    if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) {
        const double v = d1 / (d1 - d3);
#pragma unroll
        for (int i = 0; i < 3; i++) out[i] = a[i] + v * ab[i];
        return;
    }
#pragma unroll
    for (int i = 0; i < 3; i++) cp[i] = p[i] - c[i];
    const double d5 = DOT3(ab, cp), d6 = DOT3(ac, cp);
    if (d6 >= 0.0 && d5 <= d6) { out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; return; }
    const double vb = d5 * d2 - d1 * d6;
    if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {
        const double w = d2 / (d2 - d6);
#pragma unroll
        for (int i = 0; i < 3; i++) out[i] = a[i] + w * ac[i];
        return;
    }
    const double va = d3 * d6 - d5 * d4;
    if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
        const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
#pragma unroll
        for (int i = 0; i < 3; i++) out[i] = b[i] + w * (c[i] - b[i]);
        return;
    }
    const double denom = 1.0 / (va + vb + vc), v = vb * denom, w = vc * denom;
#pragma unroll
    for (int i = 0; i < 3; i++) out[i] = a[i] + ab[i] * v + ac[i] * w;
}

// 4x4 inverse: Gauss-Jordan with partial pivoting on an augmented [4][8] system (oracle: inv4)
__device__ __forceinline__ bool inv4(const double (&m)[16], double (&out)[16])
{
    double a[4][8];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) { a[i][j] = m[4 * i + j]; a[i][4 + j] = (i == j) ? 1.0 : 0.0; }
#pragma unroll
    for (int col = 0; col < 4; col++) {
        int piv = col; double best = __builtin_fabs(a[col][col]);
#pragma unroll
        for (int r = col + 1; r < 4; r++) { const double t = __builtin_fabs(a[r][col]); if (t > best) { best = t; piv = r; } }
        if (best == 0.0) return false;
#pragma unroll
        for (int r = col + 1; r < 4; r++)            // swap row `piv` into place without dynamic register indexing
            if (piv == r) {
#pragma unroll
                for (int j = 0; j < 8; j++) { const double t = a[col][j]; a[col][j] = a[r][j]; a[r][j] = t; }
            }
        const double ip = 1.0 / a[col][col];
#pragma unroll
        for (int j = 0; j < 8; j++) a[col][j] *= ip;
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (r != col) {
                const double f = a[r][col];
                if (f != 0.0) {
#pragma unroll
                    for (int j = 0; j < 8; j++) a[r][j] -= f * a[col][j];
                }
            }
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) out[4 * i + j] = a[i][4 + j];
    return true;
}

// barycentric coordinates of the closest point, blend of the three per-vertex 4x4, inverse, application  (ray_utils.py:77-88)
__device__ __forceinline__ void finish_sample(uint32_t i, const double (&p)[3], const double (&bc)[3], double best, int bf,
                                              const float *__restrict__ verts, const int32_t *__restrict__ faces, const double *__restrict__ T,
                                              double threshold, double *__restrict__ can_pts, float *__restrict__ can_pts_f32,
                                              double *__restrict__ closest, double *__restrict__ dist2, int32_t *__restrict__ face_id,
                                              uint8_t *__restrict__ mask)
{
    const int32_t f0v = faces[3 * (size_t)bf], f1v = faces[3 * (size_t)bf + 1], f2v = faces[3 * (size_t)bf + 2];
    double a[3], b[3], c[3], v0[3], v1[3], v2[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        a[k] = (double)verts[3 * (size_t)f0v + k]; b[k] = (double)verts[3 * (size_t)f1v + k]; c[k] = (double)verts[3 * (size_t)f2v + k];
        v0[k] = b[k] - a[k]; v1[k] = c[k] - a[k]; v2[k] = bc[k] - a[k];
    }
    const double d00 = DOT3(v0, v0), d01 = DOT3(v0, v1), d11 = DOT3(v1, v1), d20 = DOT3(v2, v0), d21 = DOT3(v2, v1);
    const double den = d00 * d11 - d01 * d01;
    const double bv = (d11 * d20 - d01 * d21) / den, bw = (d00 * d21 - d01 * d20) / den, bu = 1.0 - bv - bw;
    double M[16], Mi[16];
#pragma unroll
    for (int e = 0; e < 16; e++) M[e] = T[16 * (size_t)f0v + e] * bu + T[16 * (size_t)f1v + e] * bv + T[16 * (size_t)f2v + e] * bw;
    inv4(M, Mi);
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const double v = Mi[4 * r] * p[0] + Mi[4 * r + 1] * p[1] + Mi[4 * r + 2] * p[2] + Mi[4 * r + 3];
        if (can_pts) can_pts[3 * (size_t)i + r] = v;
        if (can_pts_f32) can_pts_f32[3 * (size_t)i + r] = (float)v;
    }
    if (closest) { closest[3 * (size_t)i] = bc[0]; closest[3 * (size_t)i + 1] = bc[1]; closest[3 * (size_t)i + 2] = bc[2]; }
    if (dist2) dist2[i] = best;
    if (face_id) face_id[i] = bf;
    mask[i] = best < threshold ? 1 : 0;
}

__global__ __launch_bounds__(256) void warp_samples_kernel(const float *__restrict__ pts, const float *__restrict__ verts,
                                                           const int32_t *__restrict__ faces, const double *__restrict__ T, uint32_t P,
                                                           uint32_t F, double threshold, double *__restrict__ can_pts,
                                                           float *__restrict__ can_pts_f32, double *__restrict__ closest,
                                                           double *__restrict__ dist2, int32_t *__restrict__ face_id,
                                                           uint8_t *__restrict__ mask)
{
    __shared__ float st[FT * 9];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < P;
    const uint32_t ii = live ? i : 0;
    const double p[3] = { (double)pts[3 * ii], (double)pts[3 * ii + 1], (double)pts[3 * ii + 2] };
    double best = __builtin_inf(), bc[3] = { 0.0, 0.0, 0.0 };
    int bf = 0;
    for (uint32_t f0 = 0; f0 < F; f0 += FT) {
        const uint32_t cnt = (F - f0 < (uint32_t)FT) ? F - f0 : (uint32_t)FT;
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < cnt * 3; e += blockDim.x) {       // one (face, corner) per step
            const int32_t vi = faces[(size_t)f0 * 3 + e];
            st[3 * e] = verts[3 * (size_t)vi]; st[3 * e + 1] = verts[3 * (size_t)vi + 1]; st[3 * e + 2] = verts[3 * (size_t)vi + 2];
        }
        __syncthreads();
        for (uint32_t f = 0; f < cnt; ++f) {
            const float *t = st + 9 * f;
            const double a[3] = { (double)t[0], (double)t[1], (double)t[2] }, b[3] = { (double)t[3], (double)t[4], (double)t[5] },
                         c[3] = { (double)t[6], (double)t[7], (double)t[8] };
            double q[3];
            closest_pt_tri(p, a, b, c, q);
            const double ex = p[0] - q[0], ey = p[1] - q[1], ez = p[2] - q[2], d2 = ex * ex + ey * ey + ez * ez;
            if (d2 < best) { best = d2; bf = (int)(f0 + f); bc[0] = q[0]; bc[1] = q[1]; bc[2] = q[2]; }
        }
    }
    if (!live) return;
    finish_sample(i, p, bc, best, bf, verts, faces, T, threshold, can_pts, can_pts_f32, closest, dist2, face_id, mask);
}

// ---- exact closest-face search with culling ---------------------------------------------------------------------------------
// Per frame (ac_warp_accel_build): faces sorted along a Morton curve of their centroids (one workgroup, bitonic sort of
// 16384 64-bit keys in 128 KB of LDS), cut into tiles of TILE_F faces with an oriented box (axis 0 = mean normal), four sub-boxes
// (groups of 8 consecutive faces, in the tile's frame), a bounding disc per face and one representative vertex; then two cell
// grids whose cells list the tiles that matter for any point inside them (below).
// Per sample (warp_samples_accel_kernel): a wave owns 64 consecutive samples.  A sample's candidate tiles come from its cell's list -- since round 4
// every lane walks the list of ITS sample's cell, four entries per trip, box-testing them against the sample's current bound -- (or, without a cell,
// from a pass over all tile boxes, one sample at a time) and the rest is a pipeline of packed LDS queues shared by the wave's samples:
//   (sample, tile) -> sub-box test -> (sample, tile, group) -> disc test -> (sample, face) -> fp64 Ericson routine -> running minimum of the sample
// with lanes = 16 pairs x 4 groups, 8 triples x 8 faces and 64 pairs respectively, so the lanes never diverge and every step is full.
// Every test before the last is a conservative lower bound; the exact routine is the brute-force kernel's; ties -> lowest face id.
// The barycentric blend / 4x4 inverse epilogue runs lane-parallel for the 64 samples.  Bit-identical to warp_samples_kernel.
#ifndef AC_LIST_CLASSES
#define AC_LIST_CLASSES 3        // a cell's tile list is written in this many distance classes, nearest first (1: in tile order)
#endif
#ifndef AC_TILE_F
#define AC_TILE_F 32
#endif
constexpr int TILE_F = AC_TILE_F;   // faces per tile (the pair encodings of the queues assume 32)
constexpr int MAX_TILES = 16384 / TILE_F;
constexpr int NIT = MAX_TILES / 64; // bounding-pass iterations of 64 lanes
constexpr int TPB = 256 / TILE_F;   // tiles per block of the tile builder
constexpr uint32_t MAX_ACCEL_FACES = MAX_TILES * TILE_F;     // 16384
constexpr int NB = 18;              // floats of bounds per tile
#ifndef AC_WARP_WAVES
#define AC_WARP_WAVES 4                // waves per SIMD the search kernel is compiled for (<= 128 VGPRs)
#endif

// Cell grids around the body (round 2): two axis-aligned grids -- a fine one hugging the mesh and a coarse one for the rest of the scene; every
// cell knows, for ALL points inside it, a superset of the tiles that can hold their closest face (<= K of them, else the cell is marked OVERFLOW)
// and one face near its centre (the "seed": its exact distance is an upper bound for any sample of the cell).  The search then tests the listed
// boxes of a sample in one to three lane-parallel steps instead of all tiles (431 for SMPL) in seven, and needs no seed search.  Samples without a
// usable cell (outside both grids, overflow) take the full bounding pass.
constexpr int GRID_LEVELS = 2;
#ifndef AC_LVL0_LOG2
#define AC_LVL0_LOG2 15                     // (round 4, with the lane = sample front end: 2^15 fine cells of ~4 cm; 2^17 of 2.5 cm cost 0.13 ms more per frame in the build than they saved the search)
#endif
#ifndef AC_LVL1_LOG2
#define AC_LVL1_LOG2 16
#endif
constexpr uint32_t LVL_CELLS[GRID_LEVELS] = { 1u << AC_LVL0_LOG2, 1u << AC_LVL1_LOG2 };     // capacity in cells
#ifndef AC_LVL0_K
#define AC_LVL0_K 128
#endif
constexpr uint32_t LVL_K[GRID_LEVELS] = { AC_LVL0_K, 192 };             // listed tiles per cell
constexpr uint32_t LVL_CELL0[GRID_LEVELS] = { 0, LVL_CELLS[0] };        // first cell of the level in AccelView::cell
constexpr uint32_t LVL_CTL0[GRID_LEVELS] = { 0, LVL_CELLS[0] * LVL_K[0] };   // first entry of the level in AccelView::ctl
constexpr uint32_t CTL_ENTRIES = LVL_CELLS[0] * LVL_K[0] + LVL_CELLS[1] * LVL_K[1];
constexpr uint32_t CELL_OVERFLOW = 0xffffu;      // count field of a cell without a list
#ifndef AC_GRID_MARGIN0
#define AC_GRID_MARGIN0 0.25f                    // metres of fine grid around the mesh's bounding box (round 4: wider than sqrt(0.05) = 0.224, the reach of the
                                                 // warp mask, so that every sample that can be unmasked lies in a fine cell with a face list)
#endif
#ifndef AC_GRID_MARGIN1
#define AC_GRID_MARGIN1 1.25f                    // ... of coarse grid
#endif
constexpr int HDR_WORDS = 64;
constexpr int HDR_LVL = 8, HDR_LVL_STRIDE = 12;
constexpr int WORK_SLOTS = 256;                  // AccelView::work: [WORK_SLOTS][4] 64-bit work counters of the searches since the build (exact point-triangle
                                                 // tests, bounding-disc tests, sub-box tests, tile-box tests), a wave adds to slot (its index % WORK_SLOTS): spread
                                                 // over slots because ~10^5 waves adding to ONE address cost the frame 1.4 ms (round 4, measured); zeroed by the build
// hdr words: [0] tiles, [1] F, [4..7] debug counters (64-bit x 2), level l at [8 + 12 l]: grid origin (3 floats), 1 / cell size, 2 x padded half
// diagonal of a cell (float), nx, ny, nz, cells
struct AccelView {                   // pointers into the caller's accel buffer
    uint32_t *hdr;                   // [HDR_WORDS]
    uint32_t *sorted;                // [16384] face ids along the curve
    float *tri;                      // [MAX_ACCEL_FACES][9]
    int32_t *oid;                    // [MAX_ACCEL_FACES] original face id of each slot
    float *box;                      // [NB][MAX_TILES]: oriented box: axes u0 (mean normal), u1, u2 (9), lo (3), hi (3); representative vertex (3)
    float4 *sph;                     // [MAX_ACCEL_FACES][2] bounding disc of each slot's face: (centre, padded radius), (unit normal or 0, -)
    uint32_t *cell;                  // [cells of all levels] (count << 16) | seed slot; count = CELL_OVERFLOW: no list
    uint16_t *ctl;                   // [CTL_ENTRIES] the cells' candidate tiles
    float *sub;                      // [MAX_TILES * SUBS][6] lo (3), hi (3) of each group of TILE_F / SUBS consecutive faces of a tile, in the tile's frame
    float *cfar;                     // [cells of all levels] a conservative LOWER bound of dist(q, mesh)^2 over the points q of the cell
    // round 4: per FINE cell the faces (slots) that can hold the closest face -- or one at equal distance -- of any point of the cell
    uint32_t *fl_off;                // [LVL_CELLS[0]] first entry of the cell's list in fl_pool (a multiple of 8)
    uint32_t *fl_cnt;                // [LVL_CELLS[0]] entries (0: the cell has no face list: too far, tile-list overflow, pool exhausted, degenerate seed)
    uint16_t *fl_pool;               // [FL_POOL] slots; hdr[2] = entries handed out, hdr[3] = cells that did not fit
    unsigned long long *work;        // [WORK_SLOTS][4] work counters (see WORK_SLOTS)
};
constexpr int SUBS = 4, SUB_F = TILE_F / SUBS;   // sub-boxes per tile, faces per sub-box
constexpr uint32_t FL_POOL = 48u << 20;          // 48 M entries (96 MB): ~ 80 K listed cells x 250 faces for an SMPL-sized body, twice over
#ifndef AC_FLIST_MAXD
#define AC_FLIST_MAXD 0.25f                      // fine cells whose every point is provably farther than this from the mesh get no face list
#endif
constexpr uint8_t FL_TODO = 2;                   // mask value of a sample the face-list kernel leaves to the tile-walk kernel (fixup pass)
constexpr int ACCEL_SEGS = 14;
__host__ __device__ inline size_t accel_offsets(size_t (&o)[ACCEL_SEGS])
{
    size_t off = 0;
    const size_t sz[ACCEL_SEGS] = { HDR_WORDS * 4, MAX_ACCEL_FACES * 4, (size_t)MAX_ACCEL_FACES * 36, (size_t)MAX_ACCEL_FACES * 4, (size_t)NB * MAX_TILES * 4,
                                    (size_t)MAX_ACCEL_FACES * 32, ((size_t)LVL_CELLS[0] + LVL_CELLS[1]) * 4, (size_t)CTL_ENTRIES * 2,
                                    (size_t)MAX_TILES * SUBS * 6 * 4, ((size_t)LVL_CELLS[0] + LVL_CELLS[1]) * 4,
                                    (size_t)LVL_CELLS[0] * 4, (size_t)LVL_CELLS[0] * 4, (size_t)WORK_SLOTS * 4 * 8, (size_t)FL_POOL * 2 };
    for (int i = 0; i < ACCEL_SEGS; ++i) { o[i] = off; off += (sz[i] + 255) & ~(size_t)255; }
    return off;
}
__host__ __device__ inline AccelView accel_view(void *base)
{
    size_t o[ACCEL_SEGS]; accel_offsets(o);
    char *b = static_cast<char *>(base);
    AccelView v;
    v.hdr = reinterpret_cast<uint32_t *>(b + o[0]); v.sorted = reinterpret_cast<uint32_t *>(b + o[1]);
    v.tri = reinterpret_cast<float *>(b + o[2]); v.oid = reinterpret_cast<int32_t *>(b + o[3]); v.box = reinterpret_cast<float *>(b + o[4]);
    v.sph = reinterpret_cast<float4 *>(b + o[5]);
    v.cell = reinterpret_cast<uint32_t *>(b + o[6]); v.ctl = reinterpret_cast<uint16_t *>(b + o[7]); v.sub = reinterpret_cast<float *>(b + o[8]); v.cfar = reinterpret_cast<float *>(b + o[9]);
    v.fl_off = reinterpret_cast<uint32_t *>(b + o[10]); v.fl_cnt = reinterpret_cast<uint32_t *>(b + o[11]);
    v.work = reinterpret_cast<unsigned long long *>(b + o[12]); v.fl_pool = reinterpret_cast<uint16_t *>(b + o[13]);
    return v;
}

__device__ __forceinline__ uint32_t spread10(uint32_t v)      // 10 bits -> every third bit
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu; v = (v | (v << 8)) & 0x0300f00fu; v = (v | (v << 4)) & 0x030c30c3u; v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// Face order along a Morton curve.  Keys ((30-bit code of the centroid, quantised in the vertex bounding box that accel_grid_setup_kernel left in
// the header) << 32 | face id) are unique, so the position of a face is the number of smaller keys: accel_keys_kernel writes the keys, the
// (face block, key slice) grid of accel_rank_kernel counts, and the last slice of a face block to finish writes its part of the order.
constexpr int HDR_BBOX = 32;                   // hdr[32..37]: vertex bounding box lo (3), hi (3)
constexpr uint32_t SORT_SLICE = 1024;          // keys per slice (staged in LDS)
struct SortScratch { unsigned long long *keys; uint32_t *rank, *done; };     // lives in the cfar segment until accel_cells_kernel overwrites it
__host__ __device__ inline SortScratch sort_scratch(const AccelView &av)
{
    SortScratch s;
    s.keys = reinterpret_cast<unsigned long long *>(av.cfar);
    s.rank = reinterpret_cast<uint32_t *>(s.keys + MAX_ACCEL_FACES);
    s.done = s.rank + MAX_ACCEL_FACES;
    return s;
}
static_assert(((size_t)LVL_CELLS[0] + LVL_CELLS[1]) * 4 >= (size_t)MAX_ACCEL_FACES * 12 + 4 * (MAX_ACCEL_FACES / 256), "sort scratch must fit the cfar segment");

__global__ __launch_bounds__(256) void accel_keys_kernel(const float *__restrict__ verts, const int32_t *__restrict__ faces, uint32_t F, AccelView av)
{
    const uint32_t f = blockIdx.x * 256 + threadIdx.x;
    const SortScratch ss = sort_scratch(av);
    if (threadIdx.x == 0) ss.done[blockIdx.x] = 0;
    if (f >= MAX_ACCEL_FACES) return;
    ss.rank[f] = 0;
    if (f >= F) return;
    uint32_t q[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float lo = __builtin_bit_cast(float, av.hdr[HDR_BBOX + k]), ext = __builtin_bit_cast(float, av.hdr[HDR_BBOX + 3 + k]) - lo;
        const float inv = ext > 0.0f ? 1023.0f / (3.0f * ext) : 0.0f;
        const float c = (verts[3 * (size_t)faces[3 * f] + k] + verts[3 * (size_t)faces[3 * f + 1] + k]) + verts[3 * (size_t)faces[3 * f + 2] + k];
        float g = (c - 3.0f * lo) * inv;
        g = g < 0.0f ? 0.0f : (g > 1023.0f ? 1023.0f : g);
        q[k] = (uint32_t)g;
    }
    const uint32_t m = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
    ss.keys[f] = ((unsigned long long)m << 32) | f;
}

__global__ __launch_bounds__(256) void accel_rank_kernel(uint32_t F, AccelView av)
{
    __shared__ __attribute__((aligned(16))) unsigned long long sk[SORT_SLICE];
    __shared__ uint32_t last;
    const SortScratch ss = sort_scratch(av);
    const uint32_t f = blockIdx.x * 256 + threadIdx.x;
    const uint32_t k0 = blockIdx.y * SORT_SLICE;
    for (uint32_t i = threadIdx.x; i < SORT_SLICE; i += 256) sk[i] = k0 + i < F ? ss.keys[k0 + i] : ~0ull;      // (no key is below a padding key)
    __syncthreads();
    if (f < F) {
        const unsigned long long mine = ss.keys[f];
        const ulonglong2 *sk2 = reinterpret_cast<const ulonglong2 *>(sk);
        uint32_t cnt = 0;
#pragma unroll 16
        for (uint32_t i = 0; i < SORT_SLICE / 2; ++i) { const ulonglong2 v = sk2[i]; cnt += (v.x < mine ? 1u : 0u) + (v.y < mine ? 1u : 0u); }
        if (cnt) atomicAdd(&ss.rank[f], cnt);
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(&ss.done[blockIdx.x], 1u) == gridDim.y - 1 ? 1u : 0u;
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (f < F) av.sorted[__hip_atomic_load(&ss.rank[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = f;
    // the slots behind the last face (the tile builder repeats the last face there and never reads them)
    for (uint32_t p = F + f; p < MAX_ACCEL_FACES; p += gridDim.x * 256) av.sorted[p] = 0xffffffffu;
}

// identity order (meshes the sort kernel does not cover are not accelerated at all; kept for tests of the tile builder)
__global__ __launch_bounds__(256) void accel_tiles_kernel(const float *__restrict__ verts, const int32_t *__restrict__ faces, uint32_t F,
                                                          AccelView av)
{
    __shared__ float sb[256][9];
    const uint32_t slot = blockIdx.x * 256 + threadIdx.x;            // TPB tiles per block
    const uint32_t nt = (F + TILE_F - 1) / TILE_F;
    const uint32_t src = slot < F ? slot : F - 1;                     // the tail of the last tile repeats the last face
    const uint32_t f = av.sorted[src];
    float v[9];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) v[3 * c + k] = verts[3 * (size_t)faces[3 * f + c] + k];
    if (slot < nt * TILE_F) {
#pragma unroll
        for (int e = 0; e < 9; ++e) av.tri[(size_t)slot * 9 + e] = v[e];
        av.oid[slot] = (int32_t)f;
        // bounding disc: the face lies in the disc (centre c = centroid, radius r = largest vertex distance from the STORED c) of its
        // own plane, so dist(q, face)^2 >= pd^2 + max(0, rho - r)^2 with pd = n.(q - c) and rho^2 = |q - c|^2 - pd^2.  Built in fp64
        // from the fp32 vertices and rounded to fp32; the search pads for that rounding.  A (nearly) degenerate face gets n = 0:
        // the bound then falls back to the bounding sphere.
        const double A[3] = { (double)v[0], (double)v[1], (double)v[2] }, Bv[3] = { (double)v[3], (double)v[4], (double)v[5] },
                     Cv[3] = { (double)v[6], (double)v[7], (double)v[8] };
        float cf[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) cf[k] = (float)((A[k] + Bv[k] + Cv[k]) / 3.0);
        double r2 = 0.0;
#pragma unroll
        for (int cnr = 0; cnr < 3; ++cnr) {
            const double ex = (double)v[3 * cnr] - (double)cf[0], ey = (double)v[3 * cnr + 1] - (double)cf[1], ez = (double)v[3 * cnr + 2] - (double)cf[2];
            const double d = ex * ex + ey * ey + ez * ez;
            r2 = d > r2 ? d : r2;
        }
        const double e0[3] = { Bv[0] - A[0], Bv[1] - A[1], Bv[2] - A[2] }, e1[3] = { Cv[0] - A[0], Cv[1] - A[1], Cv[2] - A[2] };
        const double nx = e0[1] * e1[2] - e0[2] * e1[1], ny = e0[2] * e1[0] - e0[0] * e1[2], nz = e0[0] * e1[1] - e0[1] * e1[0];
        const double nl = __builtin_sqrt(nx * nx + ny * ny + nz * nz);
        const double l0 = __builtin_sqrt(DOT3(e0, e0)), l1 = __builtin_sqrt(DOT3(e1, e1));
        const bool flat = nl > 1e-6 * l0 * l1 && nl > 0.0;           // sin(angle at A) > 1e-6: the normal of a sliver is not trustworthy
        av.sph[2 * (size_t)slot] = make_float4(cf[0], cf[1], cf[2], (float)(__builtin_sqrt(r2) * (1.0 + 1e-6) + 1e-7));
        av.sph[2 * (size_t)slot + 1] = flat ? make_float4((float)(nx / nl), (float)(ny / nl), (float)(nz / nl), 0.0f) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
#pragma unroll
    for (int e = 0; e < 9; ++e) sb[threadIdx.x][e] = v[e];
    __syncthreads();
    if (threadIdx.x < TPB) {
        const uint32_t tile = blockIdx.x * TPB + threadIdx.x;
        if (tile < MAX_TILES) {
            // oriented box of the tile: axis 0 = area-weighted mean normal of its faces, axes 1, 2 = a tangent basis; a surface patch
            // is thin along its normal, so this box is tight where an axis-aligned one is loose (by the patch size) -- and the
            // looseness of the lower bound is what decides how many tiles a sample has to test
            float ax[3][3] = { { 1.0f, 0.0f, 0.0f }, { 0.0f, 1.0f, 0.0f }, { 0.0f, 0.0f, 1.0f } };
            float lo[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() }, hi[3] = { -__builtin_inff(), -__builtin_inff(), -__builtin_inff() };
            float rep[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() };
            if (tile < nt) {
                float nrm[3] = { 0.0f, 0.0f, 0.0f };
                for (int j = 0; j < TILE_F; ++j) {
                    const float *t = sb[threadIdx.x * TILE_F + j];
                    const float e0[3] = { t[3] - t[0], t[4] - t[1], t[5] - t[2] }, e1[3] = { t[6] - t[0], t[7] - t[1], t[8] - t[2] };
                    nrm[0] += e0[1] * e1[2] - e0[2] * e1[1]; nrm[1] += e0[2] * e1[0] - e0[0] * e1[2]; nrm[2] += e0[0] * e1[1] - e0[1] * e1[0];
                }
                const float len = __builtin_sqrtf(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
                if (len > 0.0f) {
                    const float n0 = nrm[0] / len, n1 = nrm[1] / len, n2 = nrm[2] / len;
                    // tangent: the coordinate axis least aligned with n, made orthogonal to it
                    const float a0 = __builtin_fabsf(n0), a1 = __builtin_fabsf(n1), a2 = __builtin_fabsf(n2);
                    float h0 = 0.0f, h1 = 0.0f, h2 = 0.0f;
                    if (a0 <= a1 && a0 <= a2) h0 = 1.0f; else if (a1 <= a2) h1 = 1.0f; else h2 = 1.0f;
                    const float dp = h0 * n0 + h1 * n1 + h2 * n2;
                    float t0 = h0 - dp * n0, t1 = h1 - dp * n1, t2 = h2 - dp * n2;
                    const float tl = __builtin_sqrtf(t0 * t0 + t1 * t1 + t2 * t2);
                    t0 /= tl; t1 /= tl; t2 /= tl;
                    ax[0][0] = n0; ax[0][1] = n1; ax[0][2] = n2;
                    ax[1][0] = t0; ax[1][1] = t1; ax[1][2] = t2;
                    ax[2][0] = n1 * t2 - n2 * t1; ax[2][1] = n2 * t0 - n0 * t2; ax[2][2] = n0 * t1 - n1 * t0;
                }
                for (int gI = 0; gI < SUBS; ++gI) {                   // per group of SUB_F consecutive faces (compact along the curve): its own box in the tile's frame
                    float slo[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() }, shi[3] = { -__builtin_inff(), -__builtin_inff(), -__builtin_inff() };
                    for (int j = gI * SUB_F; j < (gI + 1) * SUB_F; ++j)
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float *t = sb[threadIdx.x * TILE_F + j] + 3 * c;
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                const float d = ax[k][0] * t[0] + ax[k][1] * t[1] + ax[k][2] * t[2];
                                slo[k] = d < slo[k] ? d : slo[k]; shi[k] = d > shi[k] ? d : shi[k];
                            }
                        }
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        lo[k] = slo[k] < lo[k] ? slo[k] : lo[k]; hi[k] = shi[k] > hi[k] ? shi[k] : hi[k];
                        const float pad = 1e-5f * (1.0f + __builtin_fabsf(slo[k]) + __builtin_fabsf(shi[k]));      // as for the tile's box below
                        av.sub[((size_t)tile * SUBS + gI) * 6 + k] = slo[k] - pad; av.sub[((size_t)tile * SUBS + gI) * 6 + 3 + k] = shi[k] + pad;
                    }
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {                        // fp32 dot products and axes: stay conservative
                    const float pad = 1e-5f * (1.0f + __builtin_fabsf(lo[k]) + __builtin_fabsf(hi[k]));
                    lo[k] -= pad; hi[k] += pad;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) rep[k] = sb[threadIdx.x * TILE_F][k];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
#pragma unroll
                for (int c = 0; c < 3; ++c) av.box[(3 * k + c) * MAX_TILES + tile] = ax[k][c];
                av.box[(9 + k) * MAX_TILES + tile] = lo[k]; av.box[(12 + k) * MAX_TILES + tile] = hi[k]; av.box[(15 + k) * MAX_TILES + tile] = rep[k];
            }
        }
    }
    if (slot == 0) { av.hdr[0] = nt; av.hdr[1] = F; }
}

__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// wave-wide minima without LDS traffic: inclusive min scan inside every 16-lane row (DPP row_shr 1, 2, 4, 8), the rows are joined with
// row_bcast15 / row_bcast31, the total sits in lane 63 and is broadcast through an SGPR.  Lanes without a DPP source keep their own value.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int dpp_keep(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false); }
#define AC_WAVE_MIN_STEPS(STEP) STEP(0x111, 0xf) STEP(0x112, 0xf) STEP(0x114, 0xf) STEP(0x118, 0xf) STEP(0x142, 0xa) STEP(0x143, 0xc)
__device__ __forceinline__ float wave_min_f32(float v)
{
#define STEP(C, M) { const float o = __builtin_bit_cast(float, dpp_keep<C, M>(__builtin_bit_cast(int, v))); v = o < v ? o : v; }
    AC_WAVE_MIN_STEPS(STEP)
#undef STEP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ int wave_min_i32(int v)
{
#define STEP(C, M) { const int o = dpp_keep<C, M>(v); v = o < v ? o : v; }
    AC_WAVE_MIN_STEPS(STEP)
#undef STEP
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ double lane_f64(double v, int l)        // l wave-uniform
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_min_f64(double v)
{
#define STEP(C, M) { const unsigned long long b = __builtin_bit_cast(unsigned long long, v); \
                     const unsigned lo = (unsigned)dpp_keep<C, M>((int)(unsigned)b), hi = (unsigned)dpp_keep<C, M>((int)(unsigned)(b >> 32)); \
                     const double o = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo); v = o < v ? o : v; }
    AC_WAVE_MIN_STEPS(STEP)
#undef STEP
    return lane_f64(v, 63);
}
#undef AC_WAVE_MIN_STEPS
__device__ __forceinline__ float lane_f32(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

#ifdef AC_PROFILE_WARP         // s_memtime per phase, summed over waves: tools/warp_profile.py
__device__ unsigned long long g_warp_prof[8];
#define WP_T0() unsigned long long wp_t_ = __builtin_amdgcn_s_memtime(), wp_acc_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }
#define WP_TICK(S) { const unsigned long long t2_ = __builtin_amdgcn_s_memtime(); wp_acc_[S] += t2_ - wp_t_; wp_t_ = t2_; }
#define WP_END() if (lane == 0) { for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_warp_prof[i_], wp_acc_[i_]); }
#else
#define WP_T0()
#define WP_TICK(S)
#define WP_END()
#endif
// ---- pieces shared by the search kernel and the builder of the cell grid -----------------------------------------------------------------
#ifdef AC_WARP_SEED_DEBUG
__device__ const double *g_warp_seed_d2 = nullptr;
#endif
// The tiles' bounds in LDS, two layouts.  Row-major [NB][ntp] (TM = false): lane = tile reads are conflict-free -- the structure build
// (accel_cells_kernel: every cell runs the full bounding pass).  Tile-major [ntp][NBT] (TM = true, round 4): one tile's 15 numbers are four 16-byte
// reads for a lane that looks at ITS OWN tile -- the search, whose front end is lane = sample since round 4 (15 scattered dword reads with 4 - 8-way
// bank conflicts were a third of its time: tools/warp_profile.py).
typedef float f32x4w __attribute__((ext_vector_type(4)));
constexpr int NBT = 20;            // floats per tile in the tile-major layout: axes (9) | lo (3) | hi (3) | representative vertex (3) | pad (2)
template <bool TM> __device__ __forceinline__ float sbox_at(const float *sbox_raw, uint32_t ntp, int row, int tl)
{
    return TM ? sbox_raw[tl * NBT + row] : sbox_raw[(uint32_t)row * ntp + (uint32_t)tl];
}
#define SBOX(ROW, TL) sbox_at<TM>(sbox_raw, ntp, (ROW), (int)(TL))
template <bool TM = false>
__device__ __forceinline__ void load_boxes(float *sbox_raw, const AccelView &av, uint32_t ntp)
{
    if (TM) {
        for (uint32_t e = threadIdx.x; e < (uint32_t)NBT * ntp; e += blockDim.x) {
            const uint32_t tl = e / NBT, r = e % NBT;
            sbox_raw[e] = r < (uint32_t)NB ? av.box[r * MAX_TILES + tl] : 0.0f;
        }
    } else {
        for (uint32_t e = threadIdx.x; e < NB * ntp; e += blockDim.x) sbox_raw[e] = av.box[(e / ntp) * MAX_TILES + e % ntp];
    }
}

// conservative fp32 lower bound of dist(q, box of tile tl)^2 (every rounding padded to the safe side; +inf for padding tiles)
template <bool TM = false>
__device__ __forceinline__ float box_lower_bound(const float *sbox_raw, uint32_t ntp, int tl, const float (&qf)[3], float padq)
{
    float ax[9], blo[3], bhi[3];
    if (TM) {
        const f32x4w *rec = reinterpret_cast<const f32x4w *>(sbox_raw + tl * NBT);
        const f32x4w a0 = rec[0], a1 = rec[1], a2 = rec[2], a3 = rec[3];
        ax[0] = a0[0]; ax[1] = a0[1]; ax[2] = a0[2]; ax[3] = a0[3]; ax[4] = a1[0]; ax[5] = a1[1]; ax[6] = a1[2]; ax[7] = a1[3]; ax[8] = a2[0];
        blo[0] = a2[1]; blo[1] = a2[2]; blo[2] = a2[3]; bhi[0] = a3[0]; bhi[1] = a3[1]; bhi[2] = a3[2];
    } else {
#pragma unroll
        for (int r = 0; r < 9; ++r) ax[r] = SBOX(r, tl);
#pragma unroll
        for (int k = 0; k < 3; ++k) { blo[k] = SBOX(9 + k, tl); bhi[k] = SBOX(12 + k, tl); }
    }
    float l = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float sk = qf[0] * ax[3 * k] + qf[1] * ax[3 * k + 1] + qf[2] * ax[3 * k + 2];
        const float lo = blo[k] - sk, hi = sk - bhi[k];
        float d = (lo > hi ? lo : hi) - padq;                      // the box itself is padded by its builder
        d = d > 0.0f ? d : 0.0f;
        l += d * d;
    }
    return l * (1.0f - 1e-5f);                                     // axes orthonormal up to fp32 rounding
}

// 1. of the full search: lower bound of every tile (lb[], lane = tile), upper bound from the representative vertices, and the two most
// promising tiles: tA = nearest representative vertex, tB = smallest lower bound
template <bool TM = false>
__device__ __forceinline__ float bounding_pass(const float *sbox_raw, uint32_t ntp, uint32_t nit, int lane, const float (&qf)[3], float (&lb)[NIT],
                                               int &tA, int &tB)
{
    const float padq = 4e-7f * ((__builtin_fabsf(qf[0]) + __builtin_fabsf(qf[1])) + __builtin_fabsf(qf[2]));      // >= the error of q . axis
    float ubl = __builtin_inff();
    int tbest = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        lb[it] = __builtin_inff();
        if ((uint32_t)it >= nit) continue;                         // wave-uniform
        const int tl = it * 64 + lane;
        const float ex = qf[0] - SBOX(15, tl), ey = qf[1] - SBOX(16, tl), ez = qf[2] - SBOX(17, tl);
        const float u = (ex * ex + ey * ey + ez * ez) * (1.0f + 1e-6f);      // >= |q - representative vertex|^2
        if (u < ubl) { ubl = u; tbest = tl; }                      // padding tiles hold +inf
        lb[it] = box_lower_bound<TM>(sbox_raw, ntp, tl, qf, padq);
    }
    const float ub = wave_min_f32(ubl);
    float lmin = lb[0];
    int tlow = lane;
#pragma unroll
    for (int it = 1; it < NIT; ++it) if (lb[it] < lmin) { lmin = lb[it]; tlow = it * 64 + lane; }
    const float lminw = wave_min_f32(lmin);
    tA = __builtin_amdgcn_readlane(tbest, __builtin_ctzll(__ballot(ubl == ub)));
    tB = __builtin_amdgcn_readlane(tlow, __builtin_ctzll(__ballot(lmin == lminw)));
    return ub;
}

// seed: the faces of tile tA (lanes 0..31) and of tile tB (lanes 32..63) through the exact routine; each lane keeps (best, bid, bc, its slot)
__device__ __forceinline__ void seed_test(const AccelView &av, const double (&q)[3], int tA, int tB, int lane, double &best, int &bid, double (&bc)[3],
                                          uint32_t &myslot)
{
    myslot = 0;
    if (lane < 2 * TILE_F) {
        const int tmine = lane < TILE_F ? tA : tB;
        const uint32_t slot = (uint32_t)tmine * TILE_F + (uint32_t)(lane & (TILE_F - 1));
        myslot = slot;
        const float *tp = av.tri + (size_t)slot * 9;
        const double a[3] = { (double)tp[0], (double)tp[1], (double)tp[2] }, b[3] = { (double)tp[3], (double)tp[4], (double)tp[5] },
                     c[3] = { (double)tp[6], (double)tp[7], (double)tp[8] };
        double cq[3];
        closest_pt_tri(q, a, b, c, cq);
        const double ex = q[0] - cq[0], ey = q[1] - cq[1], ez = q[2] - cq[2], d2 = ex * ex + ey * ey + ez * ez;
        // same acceptance rule as everywhere else: a degenerate face (two equal corners: 0 / 0 in the edge regions) yields NaN and is
        // never accepted -- an unconditional assignment would poison this lane's running minimum for the rest of the sample
        if (d2 < best) { best = d2; bid = av.oid[slot]; bc[0] = cq[0]; bc[1] = cq[1]; bc[2] = cq[2]; }
    }
}

// grid parameters of both levels from the vertex bounding box (one workgroup)
__global__ __launch_bounds__(1024) void accel_grid_setup_kernel(const float *__restrict__ verts, uint32_t V, AccelView av)
{
    __shared__ float red[6][1024];
    const uint32_t t = threadIdx.x;
    float lo[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() }, hi[3] = { -__builtin_inff(), -__builtin_inff(), -__builtin_inff() };
    for (uint32_t v = t; v < V; v += 1024)
#pragma unroll
        for (int k = 0; k < 3; ++k) { const float c = verts[3 * (size_t)v + k]; lo[k] = c < lo[k] ? c : lo[k]; hi[k] = c > hi[k] ? c : hi[k]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { red[k][t] = lo[k]; red[3 + k][t] = hi[k]; }
    __syncthreads();
    for (uint32_t s = 512; s > 0; s >>= 1) {
        if (t < s) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                red[k][t] = red[k][t + s] < red[k][t] ? red[k][t + s] : red[k][t];
                red[3 + k][t] = red[3 + k][t + s] > red[3 + k][t] ? red[3 + k][t + s] : red[3 + k][t];
            }
        }
        __syncthreads();
    }
    if (t < 6) av.hdr[HDR_BBOX + t] = __builtin_bit_cast(uint32_t, red[t][0]);
    if (t == 6) { av.hdr[2] = 0u; av.hdr[3] = 0u; }                  // face-list pool: nothing handed out yet
    for (uint32_t e = t; e < (uint32_t)WORK_SLOTS * 4u; e += 1024) av.work[e] = 0ull;      // work counters of the searches on this structure
    if (t < (uint32_t)GRID_LEVELS) {
        const int l = (int)t;
        const float margin = l == 0 ? AC_GRID_MARGIN0 : AC_GRID_MARGIN1;
        const uint32_t cap = LVL_CELLS[l];
        float org[3], ext[3];
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            org[k] = red[k][0] - margin; ext[k] = (red[3 + k][0] - red[k][0]) + 2.0f * margin;
            ok = ok && ext[k] > 0.0f && ext[k] < 1e6f;           // NaN / inf vertices: no grid, every sample takes the full bounding pass
        }
        uint32_t n[3] = { 0, 0, 0 };
        float cs = 1.0f;
        if (ok) {
            cs = cbrtf(ext[0] * ext[1] * ext[2] / (float)cap);
            cs = cs > 0.005f ? cs : 0.005f;                        // cells below 5 mm buy nothing (faces are ~1 cm)
            for (int trial = 0; trial < 64; ++trial) {
#pragma unroll
                for (int k = 0; k < 3; ++k) n[k] = (uint32_t)(ext[k] / cs) + 1u;
                if ((unsigned long long)n[0] * n[1] * n[2] <= cap) break;
                cs *= 1.02f;
            }
            if ((unsigned long long)n[0] * n[1] * n[2] > cap) { n[0] = n[1] = n[2] = 0; }
        }
        uint32_t *h = av.hdr + HDR_LVL + HDR_LVL_STRIDE * l;
#pragma unroll
        for (int k = 0; k < 3; ++k) { h[k] = __builtin_bit_cast(uint32_t, org[k]); h[5 + k] = n[k]; }
        h[3] = __builtin_bit_cast(uint32_t, 1.0f / cs);
        // |q - centre| <= h for every q that the search maps to the cell: half diagonal, plus the rounding of the index computation
        // ((q - org) * inv: relative 2^-23 of a value < 2^10 cells) and of the centre itself
        const float hd = 0.8660254f * cs * (1.0f + 1e-3f) + 1e-6f;
        h[4] = __builtin_bit_cast(uint32_t, 2.0f * hd);
        h[8] = n[0] * n[1] * n[2];
    }
}

struct GridParams { float org[3], inv, h2; uint32_t n[3], cells; };
__device__ __forceinline__ GridParams grid_params(const AccelView &av, int l)
{
    GridParams g;
    const uint32_t *h = av.hdr + HDR_LVL + HDR_LVL_STRIDE * l;
#pragma unroll
    for (int k = 0; k < 3; ++k) { g.org[k] = __builtin_bit_cast(float, h[k]); g.n[k] = h[5 + k]; }
    g.inv = __builtin_bit_cast(float, h[3]); g.h2 = __builtin_bit_cast(float, h[4]); g.cells = h[8];
    return g;
}
// cell of a point in level g, or ~0u (outside; NaN coordinates fail every comparison)
__device__ __forceinline__ uint32_t grid_cell(const GridParams &g, const float (&pf)[3])
{
    const float gx = (pf[0] - g.org[0]) * g.inv, gy = (pf[1] - g.org[1]) * g.inv, gz = (pf[2] - g.org[2]) * g.inv;
    const bool inside = g.cells != 0 && gx >= 0.0f && gy >= 0.0f && gz >= 0.0f && gx < (float)g.n[0] && gy < (float)g.n[1] && gz < (float)g.n[2];
    return inside ? ((uint32_t)gz * g.n[1] + (uint32_t)gy) * g.n[0] + (uint32_t)gx : ~0u;
}

// Lower-bound test of one face against a bound: false only if the face's bounding disc (accel_tiles_kernel) proves dist(q, face)^2 > limf.
// fp32 with every rounding padded towards "pass" (a face that passes wrongly only costs an exact test) -- the arithmetic of disc_trip below.
__device__ __forceinline__ bool disc_pass(float q0, float q1, float q2, const float4 &sp, const float4 &sn, float limf)
{
    const float ex = q0 - sp.x, ey = q1 - sp.y, ez = q2 - sp.z;
    const float e1 = (__builtin_fabsf(ex) + __builtin_fabsf(ey)) + __builtin_fabsf(ez);
    const float e2 = (ex * ex + ey * ey) + ez * ez;
    const float apd = __builtin_fabsf((ex * sn.x + ey * sn.y) + ez * sn.z);
    const float err = 1e-6f * e1 + 1e-6f;
    const float pdl = apd > err ? apd - err : 0.0f, pdh = apd + err;
    const float rem = (limf - pdl * pdl) + 1e-6f * (limf + pdl * pdl);
    const float rho2 = (e2 - pdh * pdh) - 2e-6f * (e2 + pdh * pdh);
    const float rr = (sp.w + __builtin_sqrtf(rem > 0.0f ? rem : 0.0f) * 1.000001f) + 1e-12f;
    return rem >= 0.0f && rho2 <= rr * rr * 1.000001f;
}

// one wave per cell (strided): the full bounding pass + seed test at the cell centre c, then the list of tiles t with
//   sqrt(lb_t(c)) <= sqrt(d2(c, seed face)) + 2 h:   for q in the cell, boxdist(q, t) >= boxdist(c, t) - h and dist(q, mesh) <= dist(c, seed face) + h,
// so a tile outside the list cannot hold the closest face (nor one at equal distance) of any q of the cell.
__global__ __launch_bounds__(256) void accel_cells_kernel(AccelView av, int build_flists)
{
    const int lane = threadIdx.x & 63;
    const uint32_t nt = av.hdr[0];
    const uint32_t nit = (nt + 63) >> 6;
    extern __shared__ __attribute__((aligned(16))) float sbox_raw[];
    __shared__ uint16_t s_tl[4][LVL_K[0]];                       // per wave: the cell's listed tiles (fine level: face-list build)
    __shared__ unsigned long long s_pm[4][LVL_K[0] / 2];         // ... and which faces of each pair of tiles are candidates
    const uint32_t ntp = nit * 64;
    load_boxes(sbox_raw, av, ntp);
    __syncthreads();
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    const int wv = threadIdx.x >> 6;
    static_assert(TILE_F == 32, "two tiles per wave step in the face-list build");
#pragma unroll 1
    for (int l = 0; l < GRID_LEVELS; ++l) {
        const GridParams g = grid_params(av, l);
        const float cs = 1.0f / g.inv;
        const uint32_t K = LVL_K[l];
        for (uint32_t cell = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); cell < g.cells; cell += nwaves) {
            const uint32_t ix = cell % g.n[0], iy = (cell / g.n[0]) % g.n[1], iz = cell / (g.n[0] * g.n[1]);
            const float qf[3] = { g.org[0] + ((float)ix + 0.5f) * cs, g.org[1] + ((float)iy + 0.5f) * cs, g.org[2] + ((float)iz + 0.5f) * cs };
            const double q[3] = { (double)qf[0], (double)qf[1], (double)qf[2] };
            float lb[NIT];
            int tA, tB;
            float cfar_cell = 0.0f;
            (void)bounding_pass(sbox_raw, ntp, nit, lane, qf, lb, tA, tB);
            {   // every face lies in its tile's box: dist(q, mesh) >= min_t boxdist(c, t) - h for all q of the cell
                float mn = lb[0];
#pragma unroll
                for (int it = 1; it < NIT; ++it) mn = lb[it] < mn ? lb[it] : mn;
                mn = wave_min_f32(mn);
                const float dl = __builtin_sqrtf(mn) * (1.0f - 1e-6f) - 0.5f * g.h2;
                cfar_cell = dl > 0.0f ? dl * dl * (1.0f - 1e-6f) : 0.0f;
                if (lane == 0) av.cfar[LVL_CELL0[l] + cell] = cfar_cell;
            }
            double best = __builtin_inf(), bc[3] = { 0.0, 0.0, 0.0 };
            int bid = 0x7fffffff;
            uint32_t myslot;
            seed_test(av, q, tA, tB, lane, best, bid, bc, myslot);
            const double seed = wave_min_f64(best);
            uint32_t info = CELL_OVERFLOW << 16;
            uint32_t fcnt = 0, foff = 0;
            if (seed < 1e30) {                                           // false for inf / NaN: a cell next to nothing but degenerate faces
                const uint32_t sslot = (uint32_t)__builtin_amdgcn_readlane((int)myslot, __builtin_ctzll(__ballot(best == seed)));
                const float su = __builtin_sqrtf((float)seed * (1.0f + 1e-6f)) * (1.0f + 1e-6f) + g.h2 * (1.0f + 1e-6f);      // >= sqrt(seed) + 2 h
                uint32_t cnt = 0;
                unsigned long long cand[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    cand[it] = 0;
                    if ((uint32_t)it >= nit) continue;
                    cand[it] = __ballot(__builtin_sqrtf(lb[it]) * (1.0f - 1e-6f) <= su);      // <= sqrt(lb): a superset
                    cnt += (uint32_t)__builtin_popcountll(cand[it]);
                }
                if (cnt <= K) {
                    uint16_t *dst = av.ctl + LVL_CTL0[l] + (size_t)cell * K;
                    uint32_t at = 0;
                    // Listed NEAR TILES FIRST (round 4): the search walks a list front to back against a bound that falls with every exact test, so
                    // the tiles most likely to hold the closest face should come before the ones that are only in the list because of the 2 h band.
                    // Three classes by the box distance from the cell centre: <= half, <= the whole, > the distance of the seed face.
                    const float seedf = (float)seed;
#pragma unroll 1
                    for (int cls = 0; cls < AC_LIST_CLASSES; ++cls) {
#pragma unroll
                        for (int it = 0; it < NIT; ++it) {
                            if ((uint32_t)it >= nit) continue;
                            const int mycls = AC_LIST_CLASSES == 1 ? 0 : (lb[it] <= 0.25f * seedf ? 0 : (lb[it] <= seedf ? 1 : 2));
                            const unsigned long long m = cand[it] & __ballot(mycls == cls);
                            if ((m >> lane) & 1ull) {
                                const uint32_t at_l = at + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                                dst[at_l] = (uint16_t)(it * 64 + lane);
                                if (l == 0) s_tl[wv][at_l] = (uint16_t)(it * 64 + lane);
                            }
                            at += (uint32_t)__builtin_popcountll(m);
                        }
                    }
                    info = (cnt << 16) | sslot;
                    // round 4, fine level: the FACES of the listed tiles that can hold the closest face (or one at equal distance) of a point of the cell:
                    //   lb_f(c) <= sqrt(d2(c, seed face)) + 2 h   with lb_f the face's bounding-disc lower bound (disc_pass against su^2) --
                    // dist(q, f) >= dist(c, f) - h >= lb_f(c) - h and dist(q, mesh) <= dist(c, seed face) + h for every q of the cell.  Lane = (tile of a
                    // pair, face); candidate masks are kept per pair, the list is reserved with one atomic and written in a second sweep.
                    if (l == 0 && build_flists && cfar_cell < AC_FLIST_MAXD * AC_FLIST_MAXD) {
                        wave_sync_lds();
                        const float limf = su * su * (1.0f + 1e-6f);
                        const uint32_t steps = (cnt + 1u) >> 1;
                        for (uint32_t st = 0; st < steps; ++st) {
                            const uint32_t ti = 2u * st + (uint32_t)(lane >> 5);
                            const bool have = ti < cnt;
                            const uint32_t slot = (uint32_t)s_tl[wv][have ? ti : 0u] * TILE_F + (uint32_t)(lane & 31);
                            const float4 sp = av.sph[2 * (size_t)slot], sn = av.sph[2 * (size_t)slot + 1];
                            const unsigned long long pm = __ballot(have && disc_pass(qf[0], qf[1], qf[2], sp, sn, limf));
                            if (lane == 0) s_pm[wv][st] = pm;
                            fcnt += (uint32_t)__builtin_popcountll(pm);
                        }
                        wave_sync_lds();
                        if (fcnt > 0xffffu) fcnt = 0;                                                     // (cannot happen: <= 128 tiles x 32 faces)
                        if (fcnt) {
                            uint32_t o = 0;
                            if (lane == 0) o = atomicAdd(av.hdr + 2, (fcnt + 7u) & ~7u);                  // lists start on 16-byte boundaries
                            foff = (uint32_t)__builtin_amdgcn_readfirstlane((int)o);
                            if (foff + ((fcnt + 7u) & ~7u) > FL_POOL) { if (lane == 0) atomicAdd(av.hdr + 3, 1u); fcnt = 0; foff = 0; }
                        }
                        if (fcnt) {
                            uint32_t at2 = 0;
                            for (uint32_t st = 0; st < steps; ++st) {
                                const unsigned long long pm = s_pm[wv][st];
                                const uint32_t ti = 2u * st + (uint32_t)(lane >> 5);
                                if ((pm >> lane) & 1ull)
                                    av.fl_pool[(size_t)foff + at2 + (uint32_t)__builtin_popcountll(pm & ((1ull << lane) - 1ull))] =
                                        (uint16_t)((uint32_t)s_tl[wv][ti] * TILE_F + (uint32_t)(lane & 31));
                                at2 += (uint32_t)__builtin_popcountll(pm);
                            }
                            const uint32_t padn = ((fcnt + 7u) & ~7u) - fcnt;                              // the tail of the last 8-entry group: never read as entries
                            if ((uint32_t)lane < padn) av.fl_pool[(size_t)foff + fcnt + (uint32_t)lane] = 0;
                        }
                        wave_sync_lds();
                    }
                }
            }
            if (lane == 0) {
                av.cell[LVL_CELL0[l] + cell] = info;
                if (l == 0) { av.fl_off[cell] = foff; av.fl_cnt[cell] = fcnt; }
            }
        }
    }
}

// conservative lower bound of dist(q, mesh)^2 from the cell grids (0 when nothing is known; the coarse margin outside both grids)
__device__ __forceinline__ float grid_far_bound(const AccelView &av, const float (&pf)[3])
{
#pragma unroll
    for (int l = 0; l < GRID_LEVELS; ++l) {
        const uint32_t cell = grid_cell(grid_params(av, l), pf);
        if (cell != ~0u) return av.cfar[LVL_CELL0[l] + cell];
    }
    return av.hdr[HDR_LVL + HDR_LVL_STRIDE * (GRID_LEVELS - 1) + 8] != 0u ? AC_GRID_MARGIN1 * AC_GRID_MARGIN1 * (1.0f - 1e-5f) : 0.0f;
}

// skip_masked rendering: a ray none of whose samples can come within the mask threshold of the mesh renders to the background whatever the field
// says.  pts = the coarse samples [N, T0, 3] (posed space); every later sample of the ray (up-sampled z, mid points) lies within half a coarse step
// of one of them, so the ray is DEAD if  sqrt(far_bound(p_i)) - step_i / 2 >= sqrt(threshold)  for every coarse sample.  One lane per ray.
__global__ __launch_bounds__(256) void ray_cull_kernel(const float *__restrict__ pts, uint32_t N, uint32_t T0, AccelView av, float thr,
                                                       uint8_t *__restrict__ ray_dead)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= N) return;
    const float rt = __builtin_sqrtf(thr) * (1.0f + 1e-6f);
    bool dead = true;
    float prev[3] = { 0.0f, 0.0f, 0.0f }, dprev = 0.0f;
    for (uint32_t i = 0; i < T0 && dead; ++i) {
        const float *pp = pts + ((size_t)r * T0 + i) * 3;
        const float p[3] = { pp[0], pp[1], pp[2] };
        float dnext = 0.0f;
        if (i + 1 < T0) { const float ex = pp[3] - p[0], ey = pp[4] - p[1], ez = pp[5] - p[2]; dnext = __builtin_sqrtf((ex * ex + ey * ey) + ez * ez); }
        const float step = (dprev > dnext ? dprev : dnext) * (1.0f + 1e-5f);
        const float lo = __builtin_sqrtf(grid_far_bound(av, p)) * (1.0f - 1e-6f) - 0.5f * step;
        dead = lo >= rt;                                               // NaN points / steps compare false: the ray is searched
        dprev = dnext; prev[0] = p[0];
    }
    (void)prev;
    ray_dead[r] = dead ? 1 : 0;
}

// ---- the search: one wave owns 64 consecutive samples -------------------------------------------------------------------------------------------
// The work of the wave's samples is PACKED: (sample, tile) pairs whose box can hold the answer go to a queue; a disc trip takes 8 pairs of whatever
// samples and tests their 8 x 32 faces against the bounding discs; the surviving (sample, face) pairs go to a second queue and through the fp64
// Ericson routine 64 at a time.  Every trip and every exact batch is full (up to the wave's last one) whatever the number of candidates of a single
// sample is -- with one sample at a time a batch held 17 faces on average.  Running minima live in LDS: best[s] as the bit pattern of the (non-negative)
// fp64 distance^2 under an unsigned 64-bit minimum, the lowest face id among equal distances under a second minimum; the result does not depend on the
// order of the pairs, and equals the exhaustive kernel's bit for bit.
constexpr uint32_t TQ = 1024, GQ = 256, FQ = 256;     // pair queues (rings): a sample adds <= NIT x 64 = 512 (sample, tile) pairs to < 32 left over; a group trip
                                                       // <= 128 (sample, tile, group) triples to < 16; a disc trip <= 128 (sample, face) pairs to < 64
#ifndef AC_WARP_LANE_LISTS
#define AC_WARP_LANE_LISTS 1     // the search's front end as lane = sample (every lane walks its own cell's tile list); 0: one sample at a time, lane = list entry
#endif
#ifndef AC_GSTEPS
#define AC_GSTEPS 2
#endif
#ifndef AC_DSTEPS
#define AC_DSTEPS 3
#endif
constexpr int GSTEPS = AC_GSTEPS, DSTEPS = AC_DSTEPS;  // steps per group trip (16 tile pairs each) and per disc trip (8 triples each)
static_assert(TQ >= NIT * 64 + 16 * GSTEPS && GQ >= 64 * GSTEPS + 8 * DSTEPS && FQ >= 64 * DSTEPS + 64 && MAX_TILES <= 512 && SUBS == 4 && SUB_F == 8,
              "queue capacities / pair encoding");
constexpr int PK_WAVES = 8;                     // waves per workgroup (they share the 32 KB of boxes)
constexpr uint32_t PK_WAVE_BYTES = 64 * 8 + 64 * 4 + 3 * 64 * 4 + FQ * 4 + GQ * 4 + TQ * 2;

__global__ __launch_bounds__(PK_WAVES * 64, AC_WARP_WAVES) void warp_samples_accel_kernel(const float *__restrict__ pts, const float *__restrict__ verts,
                                                                 const int32_t *__restrict__ faces, const double *__restrict__ T, uint32_t P,
                                                                 double threshold, AccelView av, double *__restrict__ can_pts,
                                                                 float *__restrict__ can_pts_f32, double *__restrict__ closest,
                                                                 double *__restrict__ dist2, int32_t *__restrict__ face_id,
                                                                 uint8_t *__restrict__ mask, float skip_thr,
                                                                 const uint8_t *__restrict__ ray_dead, uint32_t spr, uint32_t perm_mul, int fixup,
                                                                 int32_t *__restrict__ tseeds, uint32_t tseed_stride, uint32_t tseed_off, uint32_t n_faces)
{
    const int lane = threadIdx.x & 63;
    // fixup != 0 (round 4): warp_samples_flist_kernel ran first and left mask[i] == FL_TODO on the samples it could not resolve (no fine cell / no
    // face list); only those are searched here, every other lane is idle and writes nothing, and a workgroup without any returns at once
    bool todo_lane = true;
    if (fixup) {
        const uint32_t i0 = (uint32_t)(((unsigned long long)((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * perm_mul) % ((P + 63u) >> 6)) * 64u + (uint32_t)lane;
        todo_lane = i0 < P && ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) < ((P + 63u) >> 6) && mask[i0] == FL_TODO;
        if (!__syncthreads_or(todo_lane ? 1 : 0)) return;
    }
    // which 64 samples this wave owns: consecutive waves take chunks perm_mul apart (a bijection: gcd(perm_mul, chunks) = 1, chosen by the host), so
    // that the eight waves of a workgroup -- and the two workgroups of a compute unit -- hold a mix of cheap chunks (rays far from the body, samples the
    // caller lets the search skip) and expensive ones instead of 16 neighbouring rays of the same kind
    const uint32_t wave = (uint32_t)(((unsigned long long)((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * perm_mul) % ((P + 63u) >> 6));
    const uint32_t nt = av.hdr[0];
    const uint32_t nit = (nt + 63) >> 6;
    extern __shared__ __attribute__((aligned(16))) float sbox_raw[];
    const uint32_t ntp = nit * 64;                                     // tiles rounded up to whole bounding-pass iterations: the LDS row length
    constexpr bool TM = true;                                          // tile-major bounds (see load_boxes)
    char *wl = reinterpret_cast<char *>(sbox_raw + NBT * ntp) + (threadIdx.x >> 6) * PK_WAVE_BYTES;
    unsigned long long *sbest = reinterpret_cast<unsigned long long *>(wl);      // [64] bits of the running minimum distance^2 of sample s
    uint32_t *sbid = reinterpret_cast<uint32_t *>(wl + 512);                     // [64] lowest face id at that distance
    float *sq = reinterpret_cast<float *>(wl + 768);                             // [3][64] the samples
    uint32_t *fq = reinterpret_cast<uint32_t *>(wl + 1536);                      // [FQ] (sample << 14) | slot
    uint32_t *gq = reinterpret_cast<uint32_t *>(wl + 1536 + FQ * 4);             // [GQ] (sample << 11) | (tile << 2) | group
    uint16_t *tq = reinterpret_cast<uint16_t *>(wl + 1536 + FQ * 4 + GQ * 4);    // [TQ] (sample << 9) | tile
    load_boxes<TM>(sbox_raw, av, ntp);
    __syncthreads();
    if (((blockIdx.x * blockDim.x + threadIdx.x) >> 6) >= ((P + 63u) >> 6)) return;       // (after the barrier) a wave without samples
    const uint32_t i = wave * 64 + lane;
    const bool live = i < P;
    const uint32_t ii = live ? i : P - 1;
    const float pf[3] = { pts[3 * (size_t)ii], pts[3 * (size_t)ii + 1], pts[3 * (size_t)ii + 2] };
    const double p[3] = { (double)pf[0], (double)pf[1], (double)pf[2] };
    const uint32_t npts = (P - wave * 64 < 64u) ? P - wave * 64 : 64u;            // wave-uniform
    WP_T0();
    // 0. lane = sample: the sample's cell (finest level that has a list), and the exact distance^2 to the cell's seed face: an upper bound of the
    // result and a real face's distance, so the running minimum may start from it
    uint32_t mybase = 0, mycnt = CELL_OVERFLOW;
    double myseed = __builtin_inf();
    // skip_thr >= 0 (the caller only wants mask and the canonical points of UNMASKED samples): a sample whose cell proves dist^2 >= threshold is
    // masked out whatever its closest face is -- it is not searched at all (dead).  Outside both grids the mesh is >= the coarse margin away.
    bool dead = ray_dead ? ray_dead[ii / spr] != 0 : false;              // skip_masked: the sample's whole ray is provably masked out (ray_cull_kernel)
    if (!todo_lane) dead = true;                                         // fixup mode: already resolved (nothing is written for it below)
#ifndef AC_ABL_NOGRID
    if (dead) mycnt = 0;
    else if (skip_thr >= 0.0f) {
        bool in_any = false;
#pragma unroll
        for (int l = 0; l < GRID_LEVELS; ++l) {
            const uint32_t cell = grid_cell(grid_params(av, l), pf);
            if (cell != ~0u && !in_any) { in_any = true; dead = av.cfar[LVL_CELL0[l] + cell] >= skip_thr; }
        }
        if (!in_any && av.hdr[HDR_LVL + HDR_LVL_STRIDE * (GRID_LEVELS - 1) + 8] != 0u)
            dead = AC_GRID_MARGIN1 * AC_GRID_MARGIN1 * (1.0f - 1e-5f) >= skip_thr;
        if (dead) mycnt = 0;                                           // no list, no full pass
    }
#pragma unroll
    for (int l = 0; l < GRID_LEVELS; ++l) {
        if (mycnt != CELL_OVERFLOW) continue;
        const GridParams g = grid_params(av, l);
        const uint32_t cell = grid_cell(g, pf);
        if (cell != ~0u) {
            const uint32_t info = av.cell[LVL_CELL0[l] + cell];
            if ((info >> 16) != CELL_OVERFLOW) {
                const uint32_t slot = info & 0xffffu;
                const float *tp = av.tri + (size_t)slot * 9;
                const double a[3] = { (double)tp[0], (double)tp[1], (double)tp[2] }, b[3] = { (double)tp[3], (double)tp[4], (double)tp[5] },
                             c[3] = { (double)tp[6], (double)tp[7], (double)tp[8] };
                double cq[3];
                closest_pt_tri(p, a, b, c, cq);
                const double ex = p[0] - cq[0], ey = p[1] - cq[1], ez = p[2] - cq[2], d2 = ex * ex + ey * ey + ez * ez;
                if (d2 < 1e30) {                                       // NaN from a degenerate seed face in this sample's region: next level / full pass
                    myseed = d2; mycnt = info >> 16; mybase = LVL_CTL0[l] + cell * LVL_K[l];
                }
            }
        }
    }
#endif
    // TEMPORAL SEED (round 6): the caller keeps, per (ray, sample slot), the face the PREVIOUS frame's search found (tseeds; -1 = none) -- an animation's
    // body moves little between frames, so that face is usually the closest one again or next to it.  Its exact distance in THIS frame's pose is a real
    // face's distance, i.e. a valid first bound like the cell's seed face, and usually a much tighter one: the walk prunes against it from the first
    // tile on.  Results: the same face, the same bits (the bound only removes candidates that are provably farther).
    double tseed = __builtin_inf();
    const size_t tsi = tseeds ? (size_t)(ii / spr) * tseed_stride + tseed_off + (ii % spr) : 0;
    if (tseeds && live && !dead) {
        const int32_t tf = tseeds[tsi];
        if (tf >= 0 && (uint32_t)tf < n_faces) {
            const int32_t f0v = faces[3 * (size_t)tf], f1v = faces[3 * (size_t)tf + 1], f2v = faces[3 * (size_t)tf + 2];
            double a[3], b[3], c[3], cq[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { a[k] = (double)verts[3 * (size_t)f0v + k]; b[k] = (double)verts[3 * (size_t)f1v + k]; c[k] = (double)verts[3 * (size_t)f2v + k]; }
            closest_pt_tri(p, a, b, c, cq);
            const double ex = p[0] - cq[0], ey = p[1] - cq[1], ez = p[2] - cq[2], d2 = ex * ex + ey * ey + ez * ez;
            if (d2 < 1e30) tseed = d2;
        }
    }
    if (tseed < myseed && mycnt != CELL_OVERFLOW) myseed = tseed;
#ifdef AC_WARP_SEED_DEBUG   // ceiling experiment (tools/warp_seed_probe.py): start every sample from its TRUE distance^2 (taken from a previous run)
    if (g_warp_seed_d2 && live && mycnt != CELL_OVERFLOW) { const double t = g_warp_seed_d2[ii] * (1.0 + 1e-12); myseed = t < myseed ? t : myseed; }
#endif
    // skip_thr >= 0 (round 5): the caller reads mask and the canonical points of UNMASKED samples only, i.e. a closest face matters only if it is closer than
    // the mask's threshold -- so the running bound never needs to start above it.  A sample whose seed face is farther away (the outer part of the
    // shell the cell grids cannot prove masked: the samples with the LONGEST candidate lists) walks its list against threshold (1 + 1e-6) instead; if no
    // face comes in under that, the sample is masked out and leaves like a dead one.  Unmasked samples: the same face, the same bits.
    if (skip_thr >= 0.0f && myseed > (double)skip_thr) myseed = (double)skip_thr;
    sbest[lane] = __builtin_bit_cast(unsigned long long, myseed);
    sbid[lane] = 0x7fffffffu;
#pragma unroll
    for (int k = 0; k < 3; ++k) sq[k * 64 + lane] = pf[k];
    wave_sync_lds();
    WP_TICK(7)
    uint32_t th = 0, tt = 0, gh = 0, gt = 0, fh = 0, ft = 0;             // wave-uniform ring positions: tile pairs, group triples, face pairs
    // work counters of this wave (wave-uniform; one atomic each at the end -> AccelView::work: what bench.py prices the search with)
    uint32_t n_exact = 0, n_disc = 0, n_sub = 0, n_box = 0;

    // exact distances of n queued (sample, face) pairs, folded into the samples' running minima
    auto exact_batch = [&](uint32_t n) {
        wave_sync_lds();
        bool act = (uint32_t)lane < n;
#ifdef AC_ABL_NOBATCH
        act = false;
#endif
        const uint32_t e = fq[(fh + (act ? (uint32_t)lane : 0u)) & (FQ - 1)];
        const uint32_t s = e >> 14, slot = e & 16383u;
        unsigned long long d2b = ~0ull;
        int id = 0x7fffffff;
        if (act) {
            const double q[3] = { (double)sq[s], (double)sq[64 + s], (double)sq[128 + s] };
            const float *tp = av.tri + (size_t)slot * 9;
            const double a[3] = { (double)tp[0], (double)tp[1], (double)tp[2] }, b[3] = { (double)tp[3], (double)tp[4], (double)tp[5] },
                         c[3] = { (double)tp[6], (double)tp[7], (double)tp[8] };
            double cq[3];
            closest_pt_tri(q, a, b, c, cq);
            const double ex = q[0] - cq[0], ey = q[1] - cq[1], ez = q[2] - cq[2], d2 = ex * ex + ey * ey + ez * ez;
            id = av.oid[slot];
            act = d2 == d2;                                              // a degenerate face (0 / 0 in an edge region) is never accepted
            d2b = __builtin_bit_cast(unsigned long long, d2);             // d2 >= +0: the bit patterns order like the values
        }
        n_exact += n;
        const unsigned long long prev = sbest[s];
        wave_sync_lds();
        if (act) atomicMin(&sbest[s], d2b);
        wave_sync_lds();
        const unsigned long long cur = sbest[s];
        if (act && d2b == cur && cur < prev) sbid[s] = 0x7fffffffu;      // a strictly better distance: ids recorded for the old one are void
        wave_sync_lds();
        if (act && d2b == cur) atomicMin(&sbid[s], (uint32_t)id);
        fh += n;
        wave_sync_lds();
    };
    // up to GSTEPS x 16 queued (sample, tile) pairs: the four sub-boxes of the tile (groups of 8 consecutive faces, boxes in the tile's frame) against
    // the sample's current bound, lane = (pair, group); the survivors are queued as (sample, tile, group)
    auto group_trip = [&]() {
        wave_sync_lds();
        const uint32_t np = (tt - th) < (uint32_t)(16 * GSTEPS) ? (tt - th) : (uint32_t)(16 * GSTEPS);
#pragma unroll
        for (int u = 0; u < GSTEPS; ++u) {
            if ((uint32_t)(16 * u) >= np) break;                         // wave-uniform
            const uint32_t pi = (uint32_t)(16 * u + (lane >> 2));
            const bool have = pi < np;
            const uint32_t e = tq[(th + (have ? pi : 0u)) & (TQ - 1)];
            const uint32_t smp = e >> 9, tile = e & 511u, tg = tile * SUBS + (uint32_t)(lane & 3);
            const float2 *sb2 = reinterpret_cast<const float2 *>(av.sub + (size_t)tg * 6);
            const float2 b0 = sb2[0], b1 = sb2[1], b2 = sb2[2];          // lo.x lo.y | lo.z hi.x | hi.y hi.z
            const float q0 = sq[smp], q1 = sq[64 + smp], q2 = sq[128 + smp];
            const float padq = 4e-7f * ((__builtin_fabsf(q0) + __builtin_fabsf(q1)) + __builtin_fabsf(q2));
            const float limf = (float)(__builtin_bit_cast(double, sbest[smp]) * (1.0 + 1e-9)) * 1.000001f;      // >= the bound (+inf stays +inf)
            const float blo[3] = { b0.x, b0.y, b1.x }, bhi[3] = { b1.y, b2.x, b2.y };
            const f32x4w *rec = reinterpret_cast<const f32x4w *>(sbox_raw + tile * NBT);     // the tile's axes: three 16-byte reads (tile-major bounds)
            const f32x4w a0 = rec[0], a1 = rec[1], a2 = rec[2];
            const float ax[9] = { a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3], a2[0] };
            float l = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float sk = q0 * ax[3 * k] + q1 * ax[3 * k + 1] + q2 * ax[3 * k + 2];
                const float lo = blo[k] - sk, hi = sk - bhi[k];
                float d = (lo > hi ? lo : hi) - padq;
                d = d > 0.0f ? d : 0.0f;
                l += d * d;
            }
            const bool pass = have && l * (1.0f - 1e-5f) <= limf;
            const unsigned long long pm = __ballot(pass);
            if (pass) gq[(gt + (uint32_t)__builtin_popcountll(pm & ((1ull << lane) - 1ull))) & (GQ - 1)] = (smp << 11) | tg;
            gt += (uint32_t)__builtin_popcountll(pm);
        }
        th += np;
        n_sub += 4u * np;
        WP_TICK(3)
    };
    // up to DSTEPS x 8 queued (sample, tile, group) triples: the group's 8 faces against their bounding discs (a lower bound of a face's distance)
    // under the sample's current bound, lane = (triple, face); survivors are queued for the exact routine
    auto disc_trip = [&]() {
        wave_sync_lds();
        const uint32_t ne = (gt - gh) < (uint32_t)(8 * DSTEPS) ? (gt - gh) : (uint32_t)(8 * DSTEPS);
        uint32_t slot[DSTEPS], smp[DSTEPS]; bool have[DSTEPS];
        float4 sp[DSTEPS], sn[DSTEPS];
#pragma unroll
        for (int u = 0; u < DSTEPS; ++u) {
            const uint32_t ei = (uint32_t)(8 * u + (lane >> 3));
            have[u] = ei < ne;
            const uint32_t e = gq[(gh + (have[u] ? ei : 0u)) & (GQ - 1)];
            smp[u] = e >> 11;
            slot[u] = (e & 2047u) * SUB_F + (uint32_t)(lane & 7);        // (tile * 4 + group) * 8 + face = tile * 32 + ...
            sp[u] = av.sph[2 * (size_t)slot[u]]; sn[u] = av.sph[2 * (size_t)slot[u] + 1];
        }
        gh += ne;
        n_disc += 8u * ne;
#pragma unroll
        for (int u = 0; u < DSTEPS; ++u) {
            if ((uint32_t)(8 * u) >= ne) break;                          // wave-uniform
            // lower bound of the face's distance from its bounding disc (accel_tiles_kernel), in fp32: q, the disc and the normal are
            // fp32 data, and every rounding below is padded towards "pass" (a face that passes wrongly only costs an exact test)
            const float q0 = sq[smp[u]], q1 = sq[64 + smp[u]], q2 = sq[128 + smp[u]];
            const double lim = __builtin_bit_cast(double, sbest[smp[u]]) * (1.0 + 1e-9);
            const float limf = (float)lim * 1.000001f;                   // >= lim (+inf stays +inf)
            const float ex = q0 - sp[u].x, ey = q1 - sp[u].y, ez = q2 - sp[u].z;
            const float e1 = (__builtin_fabsf(ex) + __builtin_fabsf(ey)) + __builtin_fabsf(ez);
            const float e2 = (ex * ex + ey * ey) + ez * ez;
            const float apd = __builtin_fabsf((ex * sn[u].x + ey * sn[u].y) + ez * sn[u].z);
            const float err = 1e-6f * e1 + 1e-6f;                              // rounding of the dot product, of the stored normal and centre
            const float pdl = apd > err ? apd - err : 0.0f, pdh = apd + err;   // plane distance of q: lower / upper bound
            const float rem = (limf - pdl * pdl) + 1e-6f * (limf + pdl * pdl); // >= what the bound leaves for the in-plane distance^2
            const float rho2 = (e2 - pdh * pdh) - 2e-6f * (e2 + pdh * pdh);    // <= (in-plane distance of q from the disc centre)^2
            const float rr = (sp[u].w + __builtin_sqrtf(rem > 0.0f ? rem : 0.0f) * 1.000001f) + 1e-12f;
            const bool pass = have[u] && rem >= 0.0f && rho2 <= rr * rr * 1.000001f;
            const unsigned long long pm = __ballot(pass);
#ifdef AC_COUNT_CAND
            if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(av.hdr + 6), (unsigned long long)__builtin_popcountll(pm));
#endif
            if (pass) fq[(ft + (uint32_t)__builtin_popcountll(pm & ((1ull << lane) - 1ull))) & (FQ - 1)] = (smp[u] << 14) | slot[u];
            ft += (uint32_t)__builtin_popcountll(pm);
        }
        WP_TICK(4)
    };
    // queue the tiles in `cand` (lane = position, tile id `tl`) for sample j
    auto push_tiles = [&](uint32_t j, unsigned long long cand, uint32_t tl) {
        if ((cand >> lane) & 1ull) tq[(tt + (uint32_t)__builtin_popcountll(cand & ((1ull << lane) - 1ull))) & (TQ - 1)] = (uint16_t)((j << 9) | tl);
        tt += (uint32_t)__builtin_popcountll(cand);
#ifdef AC_COUNT_CAND
        if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(av.hdr + 4), (unsigned long long)__builtin_popcountll(cand));
#endif
    };

    // ONE inlined copy of each downstream stage; downstream first (that bounds the queues): full trips / batches while candidates are still coming, the
    // leftovers at the end (last)
    auto drain = [&](bool last) {
        for (;;) {
            const uint32_t ntq = tt - th, ngq = gt - gh, nfq = ft - fh;
            const bool do_e = nfq >= 64u || (last && !ntq && !ngq && nfq);
            const bool do_d = !do_e && (ngq >= (uint32_t)(8 * DSTEPS) || (last && !ntq && ngq));
            const bool do_g = !do_e && !do_d && (ntq >= (uint32_t)(16 * GSTEPS) || (last && ntq));
            if (do_e) { exact_batch(nfq < 64u ? nfq : 64u); WP_TICK(5) }
            else if (do_d) disc_trip();
            else if (do_g) group_trip();
            else break;
        }
    };
#if AC_WARP_LANE_LISTS
    // Round 4, the front end as lane = sample: every lane walks the tile list of ITS cell, four entries per trip -- the box test of an entry against the lane's
    // CURRENT bound (the running minimum the exact batches keep lowering, not only the seed), survivors compacted into the pair queue.  A trip costs what
    // one sample's step cost before (15 LDS reads + ~40 vector instructions) and serves up to 64 samples; the per-sample version broadcast the sample to
    // all lanes, waited for its list and left most lanes idle (35 listed tiles per sample on average, 64 lanes).  Trips = the longest list in the wave.
    {
        const bool has_list = (uint32_t)lane < npts && mycnt != CELL_OVERFLOW && mycnt != 0u;
        const uint32_t kmax = (uint32_t)(-wave_min_i32(has_list ? -(int)mycnt : 0));
        const float padq_l = 4e-7f * ((__builtin_fabsf(pf[0]) + __builtin_fabsf(pf[1])) + __builtin_fabsf(pf[2]));
        // four list entries per trip (one 8-byte load: lists start on multiples of four entries), requested one trip ahead; the bound is read once per trip
        constexpr int BU = 4;
        static_assert(LVL_K[0] % BU == 0 && LVL_K[1] % BU == 0 && LVL_CTL0[1] % BU == 0, "8-byte aligned list chunks");
        auto chunk = [&](uint32_t k) -> uint2 {
            return (has_list && k < mycnt) ? *reinterpret_cast<const uint2 *>(av.ctl + (size_t)mybase + k) : make_uint2(0u, 0u);
        };
        uint2 nxt = chunk(0u);
        for (uint32_t k = 0; k < kmax; k += BU) {
            const uint2 cur = nxt;
            nxt = chunk(k + (uint32_t)BU);
            const uint32_t tl[BU] = { cur.x & 0xffffu, cur.x >> 16, cur.y & 0xffffu, cur.y >> 16 };
            const float limf = (float)(__builtin_bit_cast(double, sbest[lane]) * (1.0 + 1e-9)) * 1.000001f;      // >= the bound (+inf stays +inf)
#pragma unroll
            for (int u = 0; u < BU; ++u) {
                const bool mine = has_list && k + (uint32_t)u < mycnt;
                const float l = box_lower_bound<TM>(sbox_raw, ntp, (int)(mine ? tl[u] : 0u), pf, padq_l);
                n_box += (uint32_t)__builtin_popcountll(__ballot(mine));
                const unsigned long long cand = __ballot(mine && l <= limf);
                if (cand) push_tiles((uint32_t)lane, cand, tl[u]);
            }
            WP_TICK(0)
            drain(false);
        }
    }
#endif
    // the first 64 entries of a sample's list are requested PF samples ahead (a sample's front end is ~100 instructions: one sample of distance
    // leaves the load's latency exposed); further chunks of a coarse-level list are requested together when the sample's turn comes
    constexpr int PF = 4;
    uint32_t tlq[PF];
    auto list_head = [&](uint32_t jj) -> uint32_t {
        if (AC_WARP_LANE_LISTS || jj >= npts) return 0u;
        const uint32_t cn = (uint32_t)__builtin_amdgcn_readlane((int)mycnt, (int)jj);
        return cn != CELL_OVERFLOW ? (uint32_t)av.ctl[(size_t)__builtin_amdgcn_readlane((int)mybase, (int)jj) + lane] : 0u;
    };
#pragma unroll
    for (int d = 0; d < PF; ++d) tlq[d] = list_head((uint32_t)d);
    for (uint32_t j = 0; j <= npts; ++j) {                                 // j == npts: drain the queues
        const bool last = j == npts;
#if AC_WARP_LANE_LISTS
        // (samples with a list were served above: only the few without one -- outside both grids, an overflowing cell -- take a turn here)
        if (!last && (uint32_t)__builtin_amdgcn_readlane((int)mycnt, (int)j) != CELL_OVERFLOW) continue;
#endif
        if (!last) {
        const float qf[3] = { lane_f32(pf[0], (int)j), lane_f32(pf[1], (int)j), lane_f32(pf[2], (int)j) };
        const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)mycnt, (int)j);          // wave-uniform
        const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)mybase, (int)j);
        uint32_t tl_mine = tlq[0];
#pragma unroll
        for (int d = 0; d + 1 < PF; ++d) tlq[d] = tlq[d + 1];
        tlq[PF - 1] = list_head(j + (uint32_t)PF);
        uint32_t tl_c1 = 0, tl_c2 = 0;
        if (cnt != CELL_OVERFLOW && cnt > 64u) {
            tl_c1 = av.ctl[(size_t)base + 64u + lane];
            if (cnt > 128u) tl_c2 = av.ctl[(size_t)base + 128u + lane];
        }
        if (cnt != CELL_OVERFLOW) {
            // 1'. the sample's cell lists the only tiles that matter: lane-parallel box tests against the seed bound
            const float padq = 4e-7f * ((__builtin_fabsf(qf[0]) + __builtin_fabsf(qf[1])) + __builtin_fabsf(qf[2]));
            const double lim0 = lane_f64(myseed, (int)j) * (1.0 + 1e-9);
            const float lim0f = (float)lim0 * 1.000001f;
            n_box += cnt;
            for (uint32_t c0 = 0; c0 < cnt; c0 += 64) {                  // one step for the fine grid, up to three for the coarse one
                if (c0) tl_mine = c0 == 64u ? tl_c1 : tl_c2;
                const bool mine = c0 + (uint32_t)lane < cnt;
                const uint32_t tl = mine ? tl_mine : 0u;
                const float l = box_lower_bound<TM>(sbox_raw, ntp, (int)tl, qf, padq);
                const unsigned long long cand = __ballot(mine && l <= lim0f);
                push_tiles(j, cand, tl);
            }
            WP_TICK(0)
        } else {
            // 1. no cell list: lower bound of every tile (fp32 with every rounding padded to the safe side), exact test of the two most promising
            // tiles; the best of their faces is the sample's first bound (a real face's distance)
            float lb[NIT];
            int tA, tB;
            (void)bounding_pass<TM>(sbox_raw, ntp, nit, lane, qf, lb, tA, tB);
            WP_TICK(1)
            const double q[3] = { (double)qf[0], (double)qf[1], (double)qf[2] };
            double best = __builtin_inf(), bc[3] = { 0.0, 0.0, 0.0 };
            int bid = 0x7fffffff;
            uint32_t myslot;
            seed_test(av, q, tA, tB, lane, best, bid, bc, myslot);
            double seed = wave_min_f64(best);                            // +inf if every seed face is degenerate
            { const double ts = lane_f64(tseed, (int)j); if (ts < seed) seed = ts; }        // (the previous frame's face, see stage 0)
            if (skip_thr >= 0.0f && seed > (double)skip_thr) seed = (double)skip_thr;       // (as above: only faces under the mask's threshold matter)
            n_box += nt; n_exact += 2u * TILE_F;
            if (lane == 0) sbest[j] = __builtin_bit_cast(unsigned long long, seed);
            WP_TICK(2)
            const float lim0f = (float)(seed * (1.0 + 1e-9)) * 1.000001f;
            // 2. the candidate tiles (box distance^2 <= bound): the seed tiles among them -- their faces go through the queues like all others
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if ((uint32_t)it >= nit) break;                            // wave-uniform
                unsigned long long cand = __ballot(lb[it] <= lim0f);       // fp32 compare against the bound rounded up: a superset
#ifdef AC_ABL_NOCAND
                cand = 0;
#endif
                push_tiles(j, cand, (uint32_t)(it * 64 + lane));
            }
#ifdef AC_COUNT_CAND
            if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(av.hdr + 4) + 1, 1ull << 40);      // samples through the full pass: high bits of counter 1
#endif
            WP_TICK(3)
        }
        }
        drain(last);
    }
#undef SBOX
    WP_TICK(5)
    if (live && !dead && todo_lane && skip_thr >= 0.0f && sbid[lane] == 0x7fffffffu) dead = true;      // no face under the mask's threshold: masked out
    if (live && !todo_lane) {
    } else if (live && dead) {                                           // certainly masked out: no closest face was looked for
        mask[i] = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            if (can_pts) can_pts[3 * (size_t)i + r] = p[r];
            if (can_pts_f32) can_pts_f32[3 * (size_t)i + r] = pf[r];
            if (closest) closest[3 * (size_t)i + r] = p[r];
        }
        if (dist2) dist2[i] = __builtin_inf();
        if (face_id) face_id[i] = 0;
    } else if (live) {
        // lane = sample again: the closest point on the winning face (the same routine on the same operands as in the batch that found it)
        const double rbest = __builtin_bit_cast(double, sbest[lane]);
        uint32_t rb = sbid[lane];
        double rbc[3] = { 0.0, 0.0, 0.0 };
        if (rb == 0x7fffffffu) rb = 0;                                   // no face with a distance (all degenerate): what the exhaustive kernel reports
        else {
            const int32_t f0v = faces[3 * (size_t)rb], f1v = faces[3 * (size_t)rb + 1], f2v = faces[3 * (size_t)rb + 2];
            double a[3], b[3], c[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { a[k] = (double)verts[3 * (size_t)f0v + k]; b[k] = (double)verts[3 * (size_t)f1v + k]; c[k] = (double)verts[3 * (size_t)f2v + k]; }
            closest_pt_tri(p, a, b, c, rbc);
        }
        finish_sample(i, p, rbc, rbest, (int)rb, verts, faces, T, threshold, can_pts, can_pts_f32, closest, dist2, face_id, mask);
        if (tseeds && sbid[lane] != 0x7fffffffu) tseeds[tsi] = (int32_t)rb;                  // the next frame's seed of this (ray, slot)
    }
    const uint32_t seeds = (uint32_t)__builtin_popcountll(__ballot(myseed < 1e30)) + (uint32_t)__builtin_popcountll(__ballot(tseed < 1e30));   // exact tests of the seed faces (stage 0)
    if (lane == 0) {
        unsigned long long *wk = av.work + 4 * (((blockIdx.x * blockDim.x + threadIdx.x) >> 6) % (uint32_t)WORK_SLOTS);
        atomicAdd(wk, (unsigned long long)(n_exact + seeds)); atomicAdd(wk + 1, (unsigned long long)n_disc);
        atomicAdd(wk + 2, (unsigned long long)n_sub); atomicAdd(wk + 3, (unsigned long long)n_box);
    }
    WP_TICK(6)
    WP_END()
}

// ---- round 4: the search over per-cell FACE lists, lane = sample -------------------------------------------------------------------------------
// A sample in a fine cell with a face list (accel_cells_kernel) needs no tile walk, no queues and no cross-lane traffic: its first bound is the exact
// distance to the cell's seed face, its candidates are the cell's listed faces.  Per 8 list entries (one 16-byte load) the lane tests the bounding
// discs against its running bound and remembers the survivors in a bit mask; the survivors of all 64 lanes then go through the fp64 Ericson routine
// together, one per lane and trip (re-tested against the lane's bound of the moment first), so a trip is as full as the lanes' survivor counts allow.
// (d2, face id) is a lexicographic minimum over a superset of the possible winners: the exhaustive kernel's answer bit for bit, ties -> lowest id.
// Samples that have no list are marked FL_TODO in `mask` and resolved by warp_samples_accel_kernel's fixup pass.
__device__ __forceinline__ uint32_t fl_entry(const uint4 &e, int k)      // k = 0..7: the k-th 16-bit entry
{
    const uint32_t w = k < 4 ? (k < 2 ? e.x : e.y) : (k < 6 ? e.z : e.w);
    return (k & 1) ? (w >> 16) : (w & 0xffffu);
}
__global__ __launch_bounds__(256) void warp_samples_flist_kernel(const float *__restrict__ pts, const float *__restrict__ verts, const int32_t *__restrict__ faces,
                                                                 const double *__restrict__ T, uint32_t P, double threshold, AccelView av,
                                                                 double *__restrict__ can_pts, float *__restrict__ can_pts_f32, double *__restrict__ closest,
                                                                 double *__restrict__ dist2, int32_t *__restrict__ face_id, uint8_t *__restrict__ mask,
                                                                 float skip_thr, const uint8_t *__restrict__ ray_dead, uint32_t spr)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < P;
    const uint32_t ii = live ? i : P - 1;
    const float pf[3] = { pts[3 * (size_t)ii], pts[3 * (size_t)ii + 1], pts[3 * (size_t)ii + 2] };
    const double p[3] = { (double)pf[0], (double)pf[1], (double)pf[2] };
    bool dead = ray_dead ? ray_dead[ii / spr] != 0 : false;
    if (!dead && skip_thr >= 0.0f) {                                     // as in the tile-walk kernel: the cell grids may prove the sample masked out
        bool in_any = false;
#pragma unroll
        for (int l = 0; l < GRID_LEVELS; ++l) {
            const uint32_t cell = grid_cell(grid_params(av, l), pf);
            if (cell != ~0u && !in_any) { in_any = true; dead = av.cfar[LVL_CELL0[l] + cell] >= skip_thr; }
        }
        if (!in_any && av.hdr[HDR_LVL + HDR_LVL_STRIDE * (GRID_LEVELS - 1) + 8] != 0u)
            dead = AC_GRID_MARGIN1 * AC_GRID_MARGIN1 * (1.0f - 1e-5f) >= skip_thr;
    }
    uint32_t off = 0, cnt = 0;
    double best = __builtin_inf();
    if (live && !dead) {
        const uint32_t cell = grid_cell(grid_params(av, 0), pf);
        if (cell != ~0u) {
            cnt = av.fl_cnt[cell];
            if (cnt) {
                off = av.fl_off[cell];
                const uint32_t slot = av.cell[cell] & 0xffffu;           // the seed face: a real face, its exact distance is the first bound
                const float *tp = av.tri + (size_t)slot * 9;
                const double a[3] = { (double)tp[0], (double)tp[1], (double)tp[2] }, b[3] = { (double)tp[3], (double)tp[4], (double)tp[5] },
                             c[3] = { (double)tp[6], (double)tp[7], (double)tp[8] };
                double cq[3];
                closest_pt_tri(p, a, b, c, cq);
                const double ex = p[0] - cq[0], ey = p[1] - cq[1], ez = p[2] - cq[2], d2 = ex * ex + ey * ey + ez * ez;
                if (d2 < 1e30) best = d2; else cnt = 0;                  // a degenerate seed for this sample: the tile walk handles it
            }
        }
    }
    const bool scan = cnt != 0;
    uint32_t bid = 0x7fffffffu;
    for (uint32_t k0 = 0; __ballot(scan && k0 < cnt) != 0ull; k0 += 8) {
        const bool mine = scan && k0 < cnt;
        uint4 ent = make_uint4(0u, 0u, 0u, 0u);
        if (mine) ent = *reinterpret_cast<const uint4 *>(av.fl_pool + (size_t)off + k0);
        const uint32_t left = mine ? cnt - k0 : 0u;
        float limf = (float)(best * (1.0 + 1e-9)) * 1.000001f;
        uint32_t pass = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t slot = fl_entry(ent, e);
            const float4 sp = av.sph[2 * (size_t)slot], sn = av.sph[2 * (size_t)slot + 1];
            if ((uint32_t)e < left && disc_pass(pf[0], pf[1], pf[2], sp, sn, limf)) pass |= 1u << e;
        }
        while (__ballot(pass != 0u) != 0ull) {
            if (pass) {
                const int e = __builtin_ctz(pass);
                pass &= pass - 1u;
                const uint32_t slot = fl_entry(ent, e);
                limf = (float)(best * (1.0 + 1e-9)) * 1.000001f;         // the bound may have dropped since the disc test
                const float4 sp = av.sph[2 * (size_t)slot], sn = av.sph[2 * (size_t)slot + 1];
                if (disc_pass(pf[0], pf[1], pf[2], sp, sn, limf)) {
                    const float *tp = av.tri + (size_t)slot * 9;
                    const double a[3] = { (double)tp[0], (double)tp[1], (double)tp[2] }, b[3] = { (double)tp[3], (double)tp[4], (double)tp[5] },
                                 c[3] = { (double)tp[6], (double)tp[7], (double)tp[8] };
                    double cq[3];
                    closest_pt_tri(p, a, b, c, cq);
                    const double ex = p[0] - cq[0], ey = p[1] - cq[1], ez = p[2] - cq[2], d2 = ex * ex + ey * ey + ez * ez;
                    const uint32_t id = (uint32_t)av.oid[slot];
                    if (d2 < best || (d2 == best && id < bid)) { best = d2; bid = id; }      // NaN (degenerate face) compares false: never accepted
                }
            }
        }
    }
    if (!live) return;
    if (dead) {                                                          // certainly masked out: no closest face was looked for
        mask[i] = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            if (can_pts) can_pts[3 * (size_t)i + r] = p[r];
            if (can_pts_f32) can_pts_f32[3 * (size_t)i + r] = pf[r];
            if (closest) closest[3 * (size_t)i + r] = p[r];
        }
        if (dist2) dist2[i] = __builtin_inf();
        if (face_id) face_id[i] = 0;
    } else if (!scan) {
        mask[i] = FL_TODO;
    } else {
        uint32_t rb = bid;
        double rbc[3] = { 0.0, 0.0, 0.0 };
        if (rb == 0x7fffffffu) rb = 0;
        else {
            const int32_t f0v = faces[3 * (size_t)rb], f1v = faces[3 * (size_t)rb + 1], f2v = faces[3 * (size_t)rb + 2];
            double a[3], b[3], c[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { a[k] = (double)verts[3 * (size_t)f0v + k]; b[k] = (double)verts[3 * (size_t)f1v + k]; c[k] = (double)verts[3 * (size_t)f2v + k]; }
            closest_pt_tri(p, a, b, c, rbc);
        }
        finish_sample(i, p, rbc, best, (int)rb, verts, faces, T, threshold, can_pts, can_pts_f32, closest, dist2, face_id, mask);
    }
}

}  // namespace

#ifdef AC_WARP_SEED_DEBUG
AC_API void ac_debug_warp_seed(const double *d2)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_warp_seed_d2), &d2, sizeof(d2));
}
#endif

#ifdef AC_PROFILE_WARP
AC_API void ac_debug_warp_prof(unsigned long long *out, int reset)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_warp_prof), sizeof(unsigned long long) * 8);
    if (reset) { static unsigned long long z[8] = { 0 }; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_warp_prof), z, sizeof(z)); }
}
#endif

AC_API int ac_mesh_near_far(const float *rays_o, const float *rays_d, const float *verts, uint32_t N, uint32_t V, float geo_threshold,
                            float *near, float *far, ac_stream_t stream)
{
    if (N == 0) return AC_OK;
    if (!rays_o || !rays_d || !verts || !near || !far || V == 0) { ac::set_error("mesh_near_far: NULL buffer or empty mesh"); return AC_ERR_BAD_ARG; }
    const float r2 = (float)((double)geo_threshold * (double)geo_threshold);
    uint32_t tiles = (V + VT - 1) / VT;
    if ((N + 63) / 64 >= ac::cu_count()) tiles = 1;                  // enough ray blocks to fill the device: one workgroup walks all vertex tiles
    if (tiles > 1) {                                                 // one workgroup per (64 rays, vertex tile): results meet through atomic min / max
        (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(near), 0x7f800000, N, (hipStream_t)stream);      // +inf
        (void)hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(far), (int)0xff800000u, N, (hipStream_t)stream); // -inf
    }
    hipLaunchKernelGGL(mesh_near_far_kernel, dim3((N + 63) / 64, tiles), dim3(NF_WAVES * 64), 0, (hipStream_t)stream, rays_o, rays_d, verts, N, V, r2, near, far);
    return ac::check_launch("mesh_near_far");
}

AC_API int ac_warp_samples(const float *pts, const float *verts, const int32_t *faces, const double *T, uint32_t P, uint32_t V, uint32_t F,
                           double threshold, double *can_pts, float *can_pts_f32, double *closest, double *dist2, int32_t *face_id,
                           uint8_t *mask, ac_stream_t stream)
{
    (void)V;
    if (P == 0) return AC_OK;
    if (!pts || !verts || !faces || !T || !mask || F == 0 || (!can_pts && !can_pts_f32)) { ac::set_error("warp_samples: NULL buffer or empty mesh"); return AC_ERR_BAD_ARG; }
    hipLaunchKernelGGL(warp_samples_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, pts, verts, faces, T, P, F, threshold,
                       can_pts, can_pts_f32, closest, dist2, face_id, mask);
    return ac::check_launch("warp_samples");
}

// The face-list search (warp_samples_flist_kernel + the lists accel_cells_kernel builds for it) is an opt-in experiment: AC_WARP_FLIST=1 in the
// environment of the process, read once.  Measured on the bench frame (profiles/r04_experiments.txt): same bits, slower -- see DESIGN.md 5.3.
static int warp_flist_enabled()
{
    static const int on = []() { const char *e = getenv("AC_WARP_FLIST"); return (e && e[0] == '1') ? 1 : 0; }();
    return on;
}

AC_API int ac_warp_accel_work(const void *accel, unsigned long long out[4], ac_stream_t stream)
{
    if (!accel || !out) { ac::set_error("warp_accel_work: NULL argument"); return AC_ERR_BAD_ARG; }
    const AccelView av = accel_view(const_cast<void *>(accel));
    static unsigned long long host[WORK_SLOTS * 4];
    if (hipMemcpyAsync(host, av.work, sizeof(host), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess) { ac::set_error("warp_accel_work: copy failed"); return AC_ERR_LAUNCH; }
    for (int k = 0; k < 4; ++k) { out[k] = 0; for (int sl = 0; sl < WORK_SLOTS; ++sl) out[k] += host[4 * sl + k]; }
    return AC_OK;
}

AC_API size_t ac_warp_accel_bytes(uint32_t F)
{
    if (F == 0 || F > MAX_ACCEL_FACES) return 0;
    size_t o[ACCEL_SEGS];
    const size_t all = accel_offsets(o);
    return warp_flist_enabled() ? all : o[ACCEL_SEGS - 1];          // the face-list pool (the last segment, 96 MB) only when the experiment is on
}

AC_API int ac_warp_accel_build(const float *verts, const int32_t *faces, uint32_t V, uint32_t F, void *accel, size_t accel_bytes,
                               ac_stream_t stream)
{
    const size_t need = ac_warp_accel_bytes(F);
    if (need == 0) { ac::set_error("warp_accel_build: %u faces not supported (1..%u); use ac_warp_samples", F, MAX_ACCEL_FACES); return AC_ERR_BAD_ARG; }
    if (!verts || !faces || !accel || accel_bytes < need) { ac::set_error("warp_accel_build: NULL buffer or accel buffer smaller than %zu bytes", need); return AC_ERR_BAD_ARG; }
    const AccelView av = accel_view(accel);
    // grid parameters and the vertex bounding box (one workgroup), the Morton order of the faces (keys, then ranks), the tiles, then one wave per
    // cell (strided over a grid that fills the device)
    hipLaunchKernelGGL(accel_grid_setup_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, verts, V, av);
    hipLaunchKernelGGL(accel_keys_kernel, dim3(MAX_ACCEL_FACES / 256), dim3(256), 0, (hipStream_t)stream, verts, faces, F, av);
    hipLaunchKernelGGL(accel_rank_kernel, dim3((F + 255) / 256, (F + SORT_SLICE - 1) / SORT_SLICE), dim3(256), 0, (hipStream_t)stream, F, av);
    hipLaunchKernelGGL(accel_tiles_kernel, dim3(MAX_TILES / TPB), dim3(256), 0, (hipStream_t)stream, verts, faces, F, av);
    const size_t ntp = (((size_t)F + TILE_F - 1) / TILE_F + 63) / 64 * 64;
    const size_t lds_c = (size_t)NB * ntp * sizeof(float);
    static uint64_t seen_c = 0;
    ac::allow_dynamic_lds(seen_c, reinterpret_cast<const void *>(accel_cells_kernel), (size_t)NB * MAX_TILES * sizeof(float));
    hipLaunchKernelGGL(accel_cells_kernel, dim3(4 * (unsigned)ac::cu_count()), dim3(256), lds_c, (hipStream_t)stream, av, warp_flist_enabled());
    return ac::check_launch("warp_accel_build");
}

// skip_far: samples that the cell grids prove to be masked out (dist^2 >= threshold) are not searched; their canonical point is reported as the
// sample itself (ac_render_rays_warped with skip_masked: the final pass never evaluates them)
int ac::warp_samples_accel_impl(const float *pts, const float *verts, const int32_t *faces, const double *T, uint32_t P, uint32_t V,
                                uint32_t F, double threshold, const void *accel, double *can_pts, float *can_pts_f32, double *closest,
                                double *dist2, int32_t *face_id, uint8_t *mask, ac_stream_t stream, int skip_far, const uint8_t *ray_dead,
                                uint32_t samples_per_ray, int32_t *tseeds, uint32_t tseed_stride, uint32_t tseed_off)
{
    (void)V;
    if (P == 0) return AC_OK;
    if (!pts || !verts || !faces || !T || !mask || !accel || F == 0 || F > MAX_ACCEL_FACES || (!can_pts && !can_pts_f32)) {
        ac::set_error("warp_samples_accel: NULL buffer, empty mesh or more than %u faces", MAX_ACCEL_FACES); return AC_ERR_BAD_ARG;
    }
    const AccelView av = accel_view(const_cast<void *>(accel));
    const uint32_t waves = (P + 63) / 64;
    const size_t ntp = (((size_t)F + TILE_F - 1) / TILE_F + 63) / 64 * 64;        // as in the kernel: tiles rounded up to 64
    const size_t lds = (size_t)NBT * ntp * sizeof(float) + (size_t)PK_WAVES * PK_WAVE_BYTES;
    static uint64_t seen = 0;        // the limit for the largest mesh the search supports; a launch asks for what its mesh needs
    ac::allow_dynamic_lds(seen, reinterpret_cast<const void *>(warp_samples_accel_kernel), (size_t)NBT * MAX_TILES * sizeof(float) + (size_t)PK_WAVES * PK_WAVE_BYTES);
    uint32_t perm_mul = 1;
    for (uint32_t m : { 37u, 41u, 43u, 47u, 53u }) {
        uint32_t a = waves, b = m;
        while (b) { const uint32_t t = a % b; a = b; b = t; }
        if (a == 1u) { perm_mul = m; break; }
    }
    // round 4: samples in fine cells with a face list are resolved by the lane-per-sample kernel; the tile-walk kernel follows as a fixup pass over
    // the samples it marked (AC_WARP_FLIST=0 in the environment: the tile walk alone, as in rounds 2 - 3 -- same results, for A/B timing)
    const int use_flist = warp_flist_enabled();
    const float skip_thr = skip_far ? (float)threshold * (1.0f + 1e-6f) : -1.0f;
    const uint32_t spr = samples_per_ray ? samples_per_ray : 1u;
    if (use_flist)
        hipLaunchKernelGGL(warp_samples_flist_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, pts, verts, faces, T, P, threshold, av, can_pts,
                           can_pts_f32, closest, dist2, face_id, mask, skip_thr, ray_dead, spr);
    hipLaunchKernelGGL(warp_samples_accel_kernel, dim3((waves + PK_WAVES - 1) / PK_WAVES), dim3(PK_WAVES * 64), lds, (hipStream_t)stream, pts, verts, faces, T, P,
                       threshold, av, can_pts, can_pts_f32, closest, dist2, face_id, mask, skip_thr, ray_dead, spr, perm_mul, use_flist,
                       use_flist ? nullptr : tseeds, tseed_stride, tseed_off, F);
    return ac::check_launch("warp_samples_accel");
}

int ac::warp_ray_cull(const float *coarse_pts, uint32_t N, uint32_t T0, double threshold, const void *accel, uint8_t *ray_dead, ac_stream_t stream)
{
    if (N == 0) return AC_OK;
    const AccelView av = accel_view(const_cast<void *>(accel));
    hipLaunchKernelGGL(ray_cull_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, coarse_pts, N, T0, av, (float)threshold * (1.0f + 1e-6f), ray_dead);
    return ac::check_launch("warp_ray_cull");
}

AC_API int ac_warp_samples_accel(const float *pts, const float *verts, const int32_t *faces, const double *T, uint32_t P, uint32_t V,
                                 uint32_t F, double threshold, const void *accel, double *can_pts, float *can_pts_f32, double *closest,
                                 double *dist2, int32_t *face_id, uint8_t *mask, ac_stream_t stream)
{
    return ac::warp_samples_accel_impl(pts, verts, faces, T, P, V, F, threshold, accel, can_pts, can_pts_f32, closest, dist2, face_id, mask, stream, 0, nullptr, 1,
                                       nullptr, 0, 0);
}
