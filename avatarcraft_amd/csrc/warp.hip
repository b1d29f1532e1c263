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
    __shared__ float snr[NF_WAVES][64], sfr[NF_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = blockIdx.x * 64 + lane;
    const bool live = n < N;
    const uint32_t nn = live ? n : 0;
    const float ox = rays_o[3 * nn], oy = rays_o[3 * nn + 1], oz = rays_o[3 * nn + 2];
    const float dx = rays_d[3 * nn], dy = rays_d[3 * nn + 1], dz_ = rays_d[3 * nn + 2];
    float nr = __builtin_inff(), fr = -__builtin_inff();
    for (uint32_t v0 = 0; v0 < V; v0 += VT) {
        const uint32_t cnt = (V - v0 < (uint32_t)VT) ? V - v0 : (uint32_t)VT;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < cnt * 3; i += blockDim.x) sv[i] = verts[(size_t)v0 * 3 + i];
        __syncthreads();
        for (uint32_t v = wave; v < cnt; v += NF_WAVES) {
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
    snr[wave][lane] = nr; sfr[wave][lane] = fr;
    __syncthreads();
    if (wave == 0 && live) {
#pragma unroll
        for (int w = 1; w < NF_WAVES; ++w) { const float a = snr[w][lane], b = sfr[w][lane]; nr = a < nr ? a : nr; fr = b > fr ? b : fr; }
        near[n] = nr; far[n] = fr;
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
// 16384 64-bit keys in 128 KB of LDS), cut into tiles of TILE_F faces with an oriented box (axis 0 = mean normal) and one
// representative vertex.
// Per sample (warp_samples_accel_kernel): one WAVE searches for one sample at a time -- lane = tile in the bounding pass
// (bounds of all tiles in LDS), lane = face in the exact pass, so the lanes never diverge:
//   1. ub = min over tiles of |p - representative vertex|^2          (a point of the mesh: upper bound of the answer)
//   2. tiles whose oriented-box distance^2 <= bound (1 + 1e-9) are the candidates (a lower bound for every face inside)
//   3. the faces of two candidate tiles at a time go through the same fp64 Ericson routine as the brute-force kernel; each lane
//      keeps its own best (d2, face id), one wave reduction per sample, ties -> lowest face id (order independent).
// A wave owns 64 consecutive samples: the search runs sample by sample, the result of sample j parks in lane j, and the
// barycentric blend / 4x4 inverse epilogue runs lane-parallel for the 64 samples.  Bit-identical to warp_samples_kernel.
#ifndef AC_TILE_F
#define AC_TILE_F 32
#endif
constexpr int TILE_F = AC_TILE_F;   // faces per tile; 16 was tried: twice the boxes to bound costs more than the smaller candidates save (6.0 vs 4.3 ms / 1 M samples)
constexpr int MAX_TILES = 16384 / TILE_F;
constexpr int NIT = MAX_TILES / 64; // bounding-pass iterations of 64 lanes
constexpr int GROUPS = 64 / TILE_F; // tiles tested per exact step
constexpr int TPB = 256 / TILE_F;   // tiles per block of the tile builder
constexpr uint32_t MAX_ACCEL_FACES = MAX_TILES * TILE_F;     // 16384
constexpr int NB = 18;              // floats of bounds per tile
#ifndef AC_WARP_WAVES
#define AC_WARP_WAVES 4                // waves per SIMD the search kernel is compiled for (<= 128 VGPRs; 29.5 instead of 31.6 ms per posed frame)
#endif
#ifndef AC_WARP_STEPS
#define AC_WARP_STEPS 4
#endif
constexpr int STEPS = AC_WARP_STEPS;            // trips of the face loop handle STEPS x GROUPS tiles
constexpr uint32_t RING = 512;      // per-wave ring of faces that passed the disc test: < 64 left over + STEPS x 64 new ones per trip

// Cell grid over the body (round 2): an axis-aligned grid of <= MAX_CELLS cells around the mesh; every cell knows, for ALL points inside it, a
// superset of the tiles that can hold their closest face (<= CELL_K of them, else the cell is marked OVERFLOW and its samples take the full
// bounding pass) and one face near its centre (the "seed", an upper bound for any sample of the cell).  The search then tests the <= 64 listed
// boxes of a sample in ONE lane-parallel step instead of all (431 for SMPL) in seven, and needs no seed search.
constexpr uint32_t MAX_CELLS = 1u << 19;
constexpr int CELL_K = 64;                       // listed tiles per cell (one bounding step of 64 lanes)
constexpr uint32_t CELL_OVERFLOW = 0xffffu;      // count field of a cell without a list
#ifndef AC_GRID_MARGIN
#define AC_GRID_MARGIN 0.15f                     // metres of grid around the mesh's bounding box; samples outside take the full bounding pass
#endif
constexpr int HDR_WORDS = 64;
// hdr words: [0] tiles, [1] F, [2] cells, [4..7] debug counters (64-bit x 2), [8..10] grid origin (float), [11] 1 / cell size, [12] 2 x padded half
// diagonal of a cell (float), [13..15] nx, ny, nz
struct AccelView {                   // pointers into the caller's accel buffer
    uint32_t *hdr;                   // [HDR_WORDS]
    uint32_t *sorted;                // [16384] face ids along the curve
    float *tri;                      // [MAX_ACCEL_FACES][9]
    int32_t *oid;                    // [MAX_ACCEL_FACES] original face id of each slot
    float *box;                      // [NB][MAX_TILES]: oriented box: axes u0 (mean normal), u1, u2 (9), lo (3), hi (3); representative vertex (3)
    float4 *sph;                     // [MAX_ACCEL_FACES][2] bounding disc of each slot's face: (centre, padded radius), (unit normal or 0, -)
    uint32_t *cell;                  // [MAX_CELLS] (count << 16) | seed slot; count = CELL_OVERFLOW: no list
    uint16_t *ctl;                   // [MAX_CELLS][CELL_K] the cell's candidate tiles
};
constexpr int ACCEL_SEGS = 8;
__host__ __device__ inline size_t accel_offsets(size_t (&o)[ACCEL_SEGS])
{
    size_t off = 0;
    const size_t sz[ACCEL_SEGS] = { HDR_WORDS * 4, MAX_ACCEL_FACES * 4, (size_t)MAX_ACCEL_FACES * 36, (size_t)MAX_ACCEL_FACES * 4, (size_t)NB * MAX_TILES * 4,
                                    (size_t)MAX_ACCEL_FACES * 32, (size_t)MAX_CELLS * 4, (size_t)MAX_CELLS * CELL_K * 2 };
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
    v.cell = reinterpret_cast<uint32_t *>(b + o[6]); v.ctl = reinterpret_cast<uint16_t *>(b + o[7]);
    return v;
}

__device__ __forceinline__ uint32_t spread10(uint32_t v)      // 10 bits -> every third bit
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu; v = (v | (v << 8)) & 0x0300f00fu; v = (v | (v << 4)) & 0x030c30c3u; v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// one workgroup of 1024 threads: centroid bounds, Morton keys, bitonic sort (key << 32 | face id) in LDS
__global__ __launch_bounds__(1024) void accel_sort_kernel(const float *__restrict__ verts, const int32_t *__restrict__ faces, uint32_t F,
                                                          uint32_t *__restrict__ sorted)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];       // [16384]
    __shared__ float red[6][1024];
    const uint32_t t = threadIdx.x;
    float lo[3] = { __builtin_inff(), __builtin_inff(), __builtin_inff() }, hi[3] = { -__builtin_inff(), -__builtin_inff(), -__builtin_inff() };
    for (uint32_t f = t; f < F; f += 1024) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float c = (verts[3 * (size_t)faces[3 * f] + k] + verts[3 * (size_t)faces[3 * f + 1] + k]) + verts[3 * (size_t)faces[3 * f + 2] + k];
            lo[k] = c < lo[k] ? c : lo[k]; hi[k] = c > hi[k] ? c : hi[k];
        }
    }
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
    float org[3], inv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { org[k] = red[k][0]; const float ext = red[3 + k][0] - red[k][0]; inv[k] = ext > 0.0f ? 1023.0f / ext : 0.0f; }
    for (uint32_t f = t; f < MAX_ACCEL_FACES; f += 1024) {
        unsigned long long key = ~0ull;
        if (f < F) {
            uint32_t q[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float c = (verts[3 * (size_t)faces[3 * f] + k] + verts[3 * (size_t)faces[3 * f + 1] + k]) + verts[3 * (size_t)faces[3 * f + 2] + k];
                float g = (c - org[k]) * inv[k];
                g = g < 0.0f ? 0.0f : (g > 1023.0f ? 1023.0f : g);
                q[k] = (uint32_t)g;
            }
            const uint32_t m = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
            key = ((unsigned long long)m << 32) | f;
        }
        keys[f] = key;
    }
    __syncthreads();
    for (uint32_t k = 2; k <= MAX_ACCEL_FACES; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = t; i < MAX_ACCEL_FACES; i += 1024) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const unsigned long long a = keys[i], b = keys[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[l] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t f = t; f < MAX_ACCEL_FACES; f += 1024) sorted[f] = (uint32_t)(keys[f] & 0xffffffffull);
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
                for (int j = 0; j < TILE_F; ++j)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float *t = sb[threadIdx.x * TILE_F + j] + 3 * c;
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const float d = ax[k][0] * t[0] + ax[k][1] * t[1] + ax[k][2] * t[2];
                            lo[k] = d < lo[k] ? d : lo[k]; hi[k] = d > hi[k] ? d : hi[k];
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
#define SBOX(ROW, TL) sbox_raw[(ROW) * ntp + (TL)]
__device__ __forceinline__ void load_boxes(float *sbox_raw, const AccelView &av, uint32_t ntp)
{
    for (uint32_t e = threadIdx.x; e < NB * ntp; e += blockDim.x) sbox_raw[e] = av.box[(e / ntp) * MAX_TILES + e % ntp];
}

// conservative fp32 lower bound of dist(q, box of tile tl)^2 (every rounding padded to the safe side; +inf for padding tiles)
__device__ __forceinline__ float box_lower_bound(const float *sbox_raw, uint32_t ntp, int tl, const float (&qf)[3], float padq)
{
    float l = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float sk = qf[0] * SBOX(3 * k, tl) + qf[1] * SBOX(3 * k + 1, tl) + qf[2] * SBOX(3 * k + 2, tl);
        const float lo = SBOX(9 + k, tl) - sk, hi = sk - SBOX(12 + k, tl);
        float d = (lo > hi ? lo : hi) - padq;                      // the box itself is padded by its builder
        d = d > 0.0f ? d : 0.0f;
        l += d * d;
    }
    return l * (1.0f - 1e-5f);                                     // axes orthonormal up to fp32 rounding
}

// 1. of the full search: lower bound of every tile (lb[], lane = tile), upper bound from the representative vertices, and the two most
// promising tiles: tA = nearest representative vertex, tB = smallest lower bound
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
        lb[it] = box_lower_bound(sbox_raw, ntp, tl, qf, padq);
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

// grid parameters from the vertex bounding box (one workgroup)
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
    if (t == 0) {
        float org[3], ext[3];
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            org[k] = red[k][0] - AC_GRID_MARGIN; ext[k] = (red[3 + k][0] - red[k][0]) + 2.0f * AC_GRID_MARGIN;
            ok = ok && ext[k] > 0.0f && ext[k] < 1e6f;           // NaN / inf vertices: no grid, every sample takes the full bounding pass
        }
        uint32_t n[3] = { 0, 0, 0 };
        float cs = 1.0f;
        if (ok) {
            cs = cbrtf(ext[0] * ext[1] * ext[2] / (float)MAX_CELLS);
            cs = cs > 0.005f ? cs : 0.005f;                        // cells below 5 mm buy nothing (faces are ~1 cm)
            for (int trial = 0; trial < 64; ++trial) {
#pragma unroll
                for (int k = 0; k < 3; ++k) n[k] = (uint32_t)(ext[k] / cs) + 1u;
                if ((unsigned long long)n[0] * n[1] * n[2] <= MAX_CELLS) break;
                cs *= 1.02f;
            }
            if ((unsigned long long)n[0] * n[1] * n[2] > MAX_CELLS) { n[0] = n[1] = n[2] = 0; }
        }
        av.hdr[2] = n[0] * n[1] * n[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) { av.hdr[8 + k] = __builtin_bit_cast(uint32_t, org[k]); av.hdr[13 + k] = n[k]; }
        av.hdr[11] = __builtin_bit_cast(uint32_t, 1.0f / cs);
        // |q - centre| <= h for every q that the search maps to the cell: half diagonal, plus the rounding of the index computation
        // ((q - org) * inv: relative 2^-23 of a value < 2^10 cells) and of the centre itself
        const float h = 0.8660254f * cs * (1.0f + 1e-3f) + 1e-6f;
        av.hdr[12] = __builtin_bit_cast(uint32_t, 2.0f * h);
    }
}

struct GridParams { float org[3], inv, h2; uint32_t n[3], cells; };
__device__ __forceinline__ GridParams grid_params(const AccelView &av)
{
    GridParams g;
#pragma unroll
    for (int k = 0; k < 3; ++k) { g.org[k] = __builtin_bit_cast(float, av.hdr[8 + k]); g.n[k] = av.hdr[13 + k]; }
    g.inv = __builtin_bit_cast(float, av.hdr[11]); g.h2 = __builtin_bit_cast(float, av.hdr[12]); g.cells = av.hdr[2];
    return g;
}

// one wave per cell (strided): the full bounding pass + seed test at the cell centre c, then the list of tiles t with
//   sqrt(lb_t(c)) <= sqrt(d2(c, seed face)) + 2 h:   for q in the cell, boxdist(q, t) >= boxdist(c, t) - h and dist(q, mesh) <= dist(c, seed face) + h,
// so a tile outside the list cannot hold the closest face (nor one at equal distance) of any q of the cell.
__global__ __launch_bounds__(256) void accel_cells_kernel(AccelView av)
{
    const int lane = threadIdx.x & 63;
    const uint32_t nt = av.hdr[0];
    const uint32_t nit = (nt + 63) >> 6;
    extern __shared__ __attribute__((aligned(16))) float sbox_raw[];
    const uint32_t ntp = nit * 64;
    load_boxes(sbox_raw, av, ntp);
    __syncthreads();
    const GridParams g = grid_params(av);
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    const float cs = 1.0f / g.inv;
    for (uint32_t cell = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); cell < g.cells; cell += nwaves) {
        const uint32_t ix = cell % g.n[0], iy = (cell / g.n[0]) % g.n[1], iz = cell / (g.n[0] * g.n[1]);
        const float qf[3] = { g.org[0] + ((float)ix + 0.5f) * cs, g.org[1] + ((float)iy + 0.5f) * cs, g.org[2] + ((float)iz + 0.5f) * cs };
        const double q[3] = { (double)qf[0], (double)qf[1], (double)qf[2] };
        float lb[NIT];
        int tA, tB;
        (void)bounding_pass(sbox_raw, ntp, nit, lane, qf, lb, tA, tB);
        double best = __builtin_inf(), bc[3] = { 0.0, 0.0, 0.0 };
        int bid = 0x7fffffff;
        uint32_t myslot;
        seed_test(av, q, tA, tB, lane, best, bid, bc, myslot);
        const double seed = wave_min_f64(best);
        uint32_t info = CELL_OVERFLOW << 16;
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
            if (cnt <= (uint32_t)CELL_K) {
                uint32_t at = 0;
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    if ((uint32_t)it >= nit) continue;
                    if ((cand[it] >> lane) & 1ull)
                        av.ctl[(size_t)cell * CELL_K + at + (uint32_t)__builtin_popcountll(cand[it] & ((1ull << lane) - 1ull))] = (uint16_t)(it * 64 + lane);
                    at += (uint32_t)__builtin_popcountll(cand[it]);
                }
                info = (cnt << 16) | sslot;
            }
        }
        if (lane == 0) av.cell[cell] = info;
    }
}

__global__ __launch_bounds__(256, AC_WARP_WAVES) void warp_samples_accel_kernel(const float *__restrict__ pts, const float *__restrict__ verts,
                                                                 const int32_t *__restrict__ faces, const double *__restrict__ T, uint32_t P,
                                                                 double threshold, AccelView av, double *__restrict__ can_pts,
                                                                 float *__restrict__ can_pts_f32, double *__restrict__ closest,
                                                                 double *__restrict__ dist2, int32_t *__restrict__ face_id,
                                                                 uint8_t *__restrict__ mask)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nt = av.hdr[0];
    const uint32_t nit = (nt + 63) >> 6;
    // bounds of all tiles in LDS (28 KB), lane = tile in the bounding pass
    extern __shared__ __attribute__((aligned(16))) float sbox_raw[];
    const uint32_t ntp = nit * 64;                                     // tiles rounded up to whole bounding-pass iterations: the LDS row length
    uint16_t *ring = reinterpret_cast<uint16_t *>(sbox_raw + NB * ntp) + (threadIdx.x >> 6) * (RING + ntp);   // this wave's candidate faces (slots)
    uint16_t *tlist = ring + RING;                                                                            // ... and candidate tiles
    load_boxes(sbox_raw, av, ntp);
    __syncthreads();
    const uint32_t i = wave * 64 + lane;
    const bool live = i < P;
    const uint32_t ii = live ? i : P - 1;
    const float pf[3] = { pts[3 * (size_t)ii], pts[3 * (size_t)ii + 1], pts[3 * (size_t)ii + 2] };
    const double p[3] = { (double)pf[0], (double)pf[1], (double)pf[2] };
    double rbest = __builtin_inf(), rbc[3] = { 0.0, 0.0, 0.0 };
    int rbf = 0;
    const uint32_t npts = (P - wave * 64 < 64u) ? P - wave * 64 : 64u;            // wave-uniform
    WP_T0();
    // 0. lane = sample: the sample's cell, its tile count and the exact distance^2 to the cell's seed face (an upper bound of the result)
    uint32_t mycell = 0, mycnt = CELL_OVERFLOW;
    double myseed = __builtin_inf();
#ifndef AC_ABL_NOGRID
    {
        const GridParams g = grid_params(av);
        const float gx = (pf[0] - g.org[0]) * g.inv, gy = (pf[1] - g.org[1]) * g.inv, gz = (pf[2] - g.org[2]) * g.inv;
        // written so that NaN coordinates fail
        const bool inside = g.cells != 0 && gx >= 0.0f && gy >= 0.0f && gz >= 0.0f && gx < (float)g.n[0] && gy < (float)g.n[1] && gz < (float)g.n[2];
        if (inside) {
            mycell = ((uint32_t)gz * g.n[1] + (uint32_t)gy) * g.n[0] + (uint32_t)gx;
            const uint32_t info = av.cell[mycell];
            mycnt = info >> 16;
            if (mycnt != CELL_OVERFLOW) {
                const uint32_t slot = info & 0xffffu;
                const float *tp = av.tri + (size_t)slot * 9;
                const double a[3] = { (double)tp[0], (double)tp[1], (double)tp[2] }, b[3] = { (double)tp[3], (double)tp[4], (double)tp[5] },
                             c[3] = { (double)tp[6], (double)tp[7], (double)tp[8] };
                double cq[3];
                closest_pt_tri(p, a, b, c, cq);
                const double ex = p[0] - cq[0], ey = p[1] - cq[1], ez = p[2] - cq[2];
                myseed = ex * ex + ey * ey + ez * ez;
                if (!(myseed < 1e30)) mycnt = CELL_OVERFLOW;         // NaN from a degenerate seed face in this sample's region: full pass
            }
        }
    }
#endif
    WP_TICK(7)
    uint32_t tl_next = (uint32_t)__builtin_amdgcn_readlane((int)mycnt, 0) != CELL_OVERFLOW
                           ? av.ctl[(size_t)__builtin_amdgcn_readlane((int)mycell, 0) * CELL_K + lane] : 0u;
    for (uint32_t j = 0; j < npts; ++j) {
        const float qf[3] = { lane_f32(pf[0], (int)j), lane_f32(pf[1], (int)j), lane_f32(pf[2], (int)j) };
        const double q[3] = { (double)qf[0], (double)qf[1], (double)qf[2] };
        const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)mycnt, (int)j);          // wave-uniform
        const uint32_t tl_mine = tl_next;
        if (j + 1 < npts) {                                                                     // the next sample's list is requested a sample ahead
            const uint32_t cn = (uint32_t)__builtin_amdgcn_readlane((int)mycnt, (int)(j + 1));
            if (cn != CELL_OVERFLOW) tl_next = av.ctl[(size_t)__builtin_amdgcn_readlane((int)mycell, (int)(j + 1)) * CELL_K + lane];
        }
        double best = __builtin_inf(), bc[3] = { 0.0, 0.0, 0.0 };
        int bid = 0x7fffffff;
        double lim;
        uint32_t ntl = 0;                                              // wave-uniform
        if (cnt != CELL_OVERFLOW) {
            // 1'. the sample's cell lists the only tiles that matter: one lane-parallel box test against the seed bound
            const float padq = 4e-7f * ((__builtin_fabsf(qf[0]) + __builtin_fabsf(qf[1])) + __builtin_fabsf(qf[2]));
            const bool mine = (uint32_t)lane < cnt;
            const float l = box_lower_bound(sbox_raw, ntp, mine ? (int)tl_mine : 0, qf, padq);
            lim = lane_f64(myseed, (int)j) * (1.0 + 1e-9);
            const float lim0f = (float)lim * 1.000001f;
            const unsigned long long cand = __ballot(mine && l <= lim0f);
            if ((cand >> lane) & 1ull) tlist[(uint32_t)__builtin_popcountll(cand & ((1ull << lane) - 1ull))] = (uint16_t)tl_mine;
            ntl = (uint32_t)__builtin_popcountll(cand);
            WP_TICK(0)
        } else {
            // 1. upper bound from the representative vertices, lower bound of every tile (kept in registers).  Both are bounds, not results:
            // fp32 with every rounding padded to the safe side (the vector fp32 rate is twice the fp64 rate, and the boxes are fp32)
            float lb[NIT];
            int tA, tB;
            const float ub = bounding_pass(sbox_raw, ntp, nit, lane, qf, lb, tA, tB);
            WP_TICK(1)
            // seed: the faces of the tile with the nearest representative vertex (lanes 0..31) and of the tile with the smallest lower
            // bound (lanes 32..63) are tested first; their exact distances replace the vertex distance as the bound
            uint32_t myslot;
            seed_test(av, q, tA, tB, lane, best, bid, bc, myslot);
            WP_TICK(2)
            const double seed = wave_min_f64(best);
            const double lim0 = (seed < (double)ub ? seed : (double)ub) * (1.0 + 1e-9);
            // 2. the candidate tiles (box distance^2 <= bound) are listed in LDS
            lim = lim0;
            const float lim0f = (float)lim0 * 1.000001f;                   // >= lim0
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if ((uint32_t)it >= nit) break;                            // wave-uniform
                unsigned long long cand = __ballot(lb[it] <= lim0f);           // fp32 compare against the bound rounded up: a superset
                if (it == (tA >> 6)) cand &= ~(1ull << (tA & 63));         // the seed tiles are done
                if (it == (tB >> 6)) cand &= ~(1ull << (tB & 63));
#ifdef AC_ABL_NOCAND
                cand = 0;
#endif
                if ((cand >> lane) & 1ull) tlist[ntl + (uint32_t)__builtin_popcountll(cand & ((1ull << lane) - 1ull))] = (uint16_t)(it * 64 + lane);
                ntl += (uint32_t)__builtin_popcountll(cand);
            }
#ifdef AC_COUNT_CAND
            if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(av.hdr + 4) + 1, 1ull << 40);      // samples through the full pass: high bits of counter 1
#endif
        }
#ifdef AC_COUNT_CAND
        if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(av.hdr + 4), (unsigned long long)ntl);
#endif
        wave_sync_lds();
        WP_TICK(3)
        // 3. their faces, STEPS x 2 tiles per trip (the loads of a trip are in flight together): every face is first tested against its
        // bounding disc (a lower bound of its distance); the survivors are compacted into a ring in LDS and go through the fp64
        // Ericson routine 64 at a time
        uint32_t head = 0, tail = 0;                                   // wave-uniform ring positions
        float limf = (float)lim * 1.000001f;                           // >= lim
        auto exact_batch = [&](uint32_t n) {
            wave_sync_lds();
#ifdef AC_ABL_NOBATCH       // timing ablation: survivors of the disc test are dropped
            if ((uint32_t)lane > 1000u) {
#else
            if ((uint32_t)lane < n) {
#endif
                const uint32_t slot = ring[(head + (uint32_t)lane) & (RING - 1)];
                const float *tp = av.tri + (size_t)slot * 9;
                const double a[3] = { (double)tp[0], (double)tp[1], (double)tp[2] }, b[3] = { (double)tp[3], (double)tp[4], (double)tp[5] },
                             c[3] = { (double)tp[6], (double)tp[7], (double)tp[8] };
                double cq[3];
                closest_pt_tri(q, a, b, c, cq);
                const double ex = q[0] - cq[0], ey = q[1] - cq[1], ez = q[2] - cq[2], d2 = ex * ex + ey * ey + ez * ez;
                const int id = av.oid[slot];
                if (d2 < best || (d2 == best && id < bid)) { best = d2; bid = id; bc[0] = cq[0]; bc[1] = cq[1]; bc[2] = cq[2]; }
            }
            head += n;
            const double nb = wave_min_f64(best) * (1.0 + 1e-9);       // a better bound prunes the faces still to come
            if (nb < lim) { lim = nb; limf = (float)nb * 1.000001f; }
            wave_sync_lds();
        };
        for (uint32_t t0 = 0; t0 < ntl; t0 += GROUPS * STEPS) {
            uint32_t slot[STEPS]; bool have[STEPS];
            float4 sp[STEPS], sn[STEPS];
#pragma unroll
            for (int u = 0; u < STEPS; ++u) {
                const uint32_t ti = t0 + (uint32_t)(GROUPS * u + lane / TILE_F);
                have[u] = ti < ntl;
                slot[u] = (uint32_t)tlist[have[u] ? ti : t0] * TILE_F + (uint32_t)(lane & (TILE_F - 1));
                sp[u] = av.sph[2 * (size_t)slot[u]]; sn[u] = av.sph[2 * (size_t)slot[u] + 1];
            }
#pragma unroll
            for (int u = 0; u < STEPS; ++u) {
                // lower bound of the face's distance from its bounding disc (accel_tiles_kernel), in fp32: q, the disc and the normal are
                // fp32 data, and every rounding below is padded towards "pass" (a face that passes wrongly only costs an exact test)
                const float ex = qf[0] - sp[u].x, ey = qf[1] - sp[u].y, ez = qf[2] - sp[u].z;
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
                if (pass) ring[(tail + (uint32_t)__builtin_popcountll(pm & ((1ull << lane) - 1ull))) & (RING - 1)] = (uint16_t)slot[u];
                tail += (uint32_t)__builtin_popcountll(pm);
            }
            WP_TICK(4)
            while (tail - head >= 64u) exact_batch(64u);
            WP_TICK(5)
        }
        WP_TICK(4)
        if (tail != head) exact_batch(tail - head);
        WP_TICK(5)
        const double wbest = wave_min_f64(best);
        const int wid = wave_min_i32(best == wbest ? bid : 0x7fffffff);
        const unsigned long long win = __ballot(best == wbest && bid == wid);
        const int wl = __builtin_ctzll(win);
        const double w0 = lane_f64(bc[0], wl), w1 = lane_f64(bc[1], wl), w2 = lane_f64(bc[2], wl);
        if (lane == (int)j) { rbest = wbest; rbf = wid; rbc[0] = w0; rbc[1] = w1; rbc[2] = w2; }
        WP_TICK(6)
    }
#undef SBOX
    if (!live) return;
    finish_sample(i, p, rbc, rbest, rbf, verts, faces, T, threshold, can_pts, can_pts_f32, closest, dist2, face_id, mask);
    WP_TICK(7)
    WP_END()
}

}  // namespace

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
    hipLaunchKernelGGL(mesh_near_far_kernel, dim3((N + 63) / 64), dim3(NF_WAVES * 64), 0, (hipStream_t)stream, rays_o, rays_d, verts, N, V, r2, near, far);
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

AC_API size_t ac_warp_accel_bytes(uint32_t F)
{
    if (F == 0 || F > MAX_ACCEL_FACES) return 0;
    size_t o[ACCEL_SEGS];
    return accel_offsets(o);
}

AC_API int ac_warp_accel_build(const float *verts, const int32_t *faces, uint32_t V, uint32_t F, void *accel, size_t accel_bytes,
                               ac_stream_t stream)
{
    const size_t need = ac_warp_accel_bytes(F);
    if (need == 0) { ac::set_error("warp_accel_build: %u faces not supported (1..%u); use ac_warp_samples", F, MAX_ACCEL_FACES); return AC_ERR_BAD_ARG; }
    if (!verts || !faces || !accel || accel_bytes < need) { ac::set_error("warp_accel_build: NULL buffer or accel buffer smaller than %zu bytes", need); return AC_ERR_BAD_ARG; }
    const AccelView av = accel_view(accel);
    static uint64_t seen = 0;
    const size_t lds = (size_t)MAX_ACCEL_FACES * 8;
    ac::allow_dynamic_lds(seen, reinterpret_cast<const void *>(accel_sort_kernel), lds);
    hipLaunchKernelGGL(accel_sort_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, verts, faces, F, av.sorted);
    hipLaunchKernelGGL(accel_tiles_kernel, dim3(MAX_TILES / TPB), dim3(256), 0, (hipStream_t)stream, verts, faces, F, av);
    // the cell grid: parameters from the vertex bounding box, then one wave per cell (strided over a grid that fills the device)
    hipLaunchKernelGGL(accel_grid_setup_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, verts, V, av);
    const size_t ntp = (((size_t)F + TILE_F - 1) / TILE_F + 63) / 64 * 64;
    const size_t lds_c = (size_t)NB * ntp * sizeof(float);
    static uint64_t seen_c = 0;
    ac::allow_dynamic_lds(seen_c, reinterpret_cast<const void *>(accel_cells_kernel), (size_t)NB * MAX_TILES * sizeof(float));
    hipLaunchKernelGGL(accel_cells_kernel, dim3(4 * (unsigned)ac::cu_count()), dim3(256), lds_c, (hipStream_t)stream, av);
    return ac::check_launch("warp_accel_build");
}

AC_API int ac_warp_samples_accel(const float *pts, const float *verts, const int32_t *faces, const double *T, uint32_t P, uint32_t V,
                                 uint32_t F, double threshold, const void *accel, double *can_pts, float *can_pts_f32, double *closest,
                                 double *dist2, int32_t *face_id, uint8_t *mask, ac_stream_t stream)
{
    (void)V;
    if (P == 0) return AC_OK;
    if (!pts || !verts || !faces || !T || !mask || !accel || F == 0 || F > MAX_ACCEL_FACES || (!can_pts && !can_pts_f32)) {
        ac::set_error("warp_samples_accel: NULL buffer, empty mesh or more than %u faces", MAX_ACCEL_FACES); return AC_ERR_BAD_ARG;
    }
    const AccelView av = accel_view(const_cast<void *>(accel));
    const uint32_t waves = (P + 63) / 64;
    const size_t ntp = (((size_t)F + TILE_F - 1) / TILE_F + 63) / 64 * 64;        // as in the kernel: tiles rounded up to 64
    const size_t lds = (size_t)NB * ntp * sizeof(float) + 4 * (RING + ntp) * sizeof(uint16_t);
    static uint64_t seen = 0;        // the limit for the largest mesh the search supports; a launch asks for what its mesh needs
    ac::allow_dynamic_lds(seen, reinterpret_cast<const void *>(warp_samples_accel_kernel), (size_t)NB * MAX_TILES * sizeof(float) + 4 * (RING + MAX_TILES) * sizeof(uint16_t));
    hipLaunchKernelGGL(warp_samples_accel_kernel, dim3((waves + 3) / 4), dim3(256), lds, (hipStream_t)stream, pts, verts, faces, T, P, threshold, av,
                       can_pts, can_pts_f32, closest, dist2, face_id, mask);
    return ac::check_launch("warp_samples_accel");
}
