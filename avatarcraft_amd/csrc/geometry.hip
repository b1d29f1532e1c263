// avatarcraft_amd/csrc/geometry.hip -- the two consumers of the learned SDF outside the ray path:
//
//   mesh export   NeRFNetwork.extract_geometry (models/instant_nsr.py:706-764; the reference's one call: stylize.py:267, resolution 512):
//                 forward_sdf on a resolution^3 grid over [-bound, bound]^3 (extract_fields :728-745, 256^3-point blocks through query_func, assembled
//                 on the HOST) -> u = -sdf -> PyMCubes marching_cubes(u, threshold) -> vertices scaled to world units.
//                 Here: ac_field_sdf_grid (grid coordinates formed in the kernel from the three axis tables, the volume stays on the device) and
//                 ac_marching_cubes_count / _emit (classify -> scan -> emit, shared-edge vertex indexing, no atomics: a deterministic mesh).
//   density grid  NeRFRenderer.update_extra_state (:303-356): forward_sdf on the 129^3 grid -> logistic density -> zero pad + 2^3 max pool ->
//                 maximum(grid * decay, new) -> mean.  Here two launches (ac_density_grid_update): the density of every grid point on the x-tiles of
//                 ac_field_sdf_grid into a scratch volume, then one streaming pass that pools, merges into the running grid in place and forms the mean.
//
// The SDF itself is the renderer's own tile code (nsr_device.hpp: sdf_tile = hash gather + MFMA SDF network), so every value is bit-identical to
// ac_field_sdf / density() and therefore to the CPU oracle's orc_field_sdf.
//
// Marching cubes: the 256-case table is GENERATED from the algorithm's definition (tools/gen_mc_table.py -> ac_mc_table.hpp): corner flagged <=> u <= iso
// (PyMCubes' convention), one vertex per sign-changing grid edge at the linear zero crossing (formed in double like PyMCubes: it evaluates its float32
// input in double), ambiguous faces always cut off the flagged corners (a rule of the face's own 4 flags: watertight by construction), triangles oriented
// with the normal towards u <= iso = out of the body for u = -sdf.  PyMCubes is not in this image: "unpinned vs PyMCubes, pinned vs the definition".
// Order of the output (what makes the mesh reproducible bit for bit, and equal to the CPU oracle's serial loop): vertices by owning grid point (linear
// index, z fastest) then by axis x, y, z; triangles by cell (linear index) then by table position.
#include "nsr_device.hpp"

#include <stdlib.h>
#ifndef AC_GRID_ORDER_DEFAULT
#define AC_GRID_ORDER_DEFAULT 1
#endif
#ifndef AC_GRID_ROUND_DEFAULT
#define AC_GRID_ROUND_DEFAULT 2
#endif
#define AC_MC_CONST static __constant__ const
#include "ac_mc_table.hpp"

namespace {

// ---------------------------------------------------------------------------------------------------------------- SDF on a regular grid
// sdf_tile (nsr_device.hpp) with the number of hash levels a lane requests per memory round trip as a parameter: the same arithmetic in the same order
// (bit-identical values), more gathers in flight per wave
template <int ROUND>
__device__ __forceinline__ f32x4 sdf_tile_r(const float *__restrict__ lds, const FieldCtx &fc, int lane, float px, float py, float pz)
{
    const int g = lane >> 4;
    float f[4][2];
    encode4<ROUND>(lds, fc.table, g, fc.jmode, px, py, pz, fc.bound, fc.two_bound, f, fc.inv_tb);
    __builtin_amdgcn_sched_barrier(0);
    return sdf_mlp(lds, lane, sel4(g, px, py, pz, 0.0f), f);
}

// update_extra_state's density (:331-337): inv_s e^(-inv_s sdf) / (1 + e^(-inv_s sdf)) for sdf > 0, the mirrored form for sdf <= 0 -- the two
// overflow-free branches, every operation in the reference's order (mul, exp, mul | add, div) in fp32
__device__ __forceinline__ float dg_density(float sdf, float inv_s)
{
    const float e = expf(sdf > 0.0f ? -inv_s * sdf : inv_s * sdf);
    return (inv_s * e) / (1.0f + e);
}

constexpr uint32_t GB_X = 16, GB_Y = 4, GB_Z = 16;           // a brick of grid points: 64 tiles of 16 points along X

// BRICK = false: tiles of 16 consecutive points in the volume's linear order (z fastest), dealt to the waves round robin (round 5's first version).
// BRICK = true : the volume is cut into bricks of 16 x 4 x 16 points; a workgroup evaluates one brick at a time, its 8 waves take 8 tiles each, and a
//   tile is 16 consecutive points ALONG X.  Why x: the spatial hash of the fine levels is x ^ y P1 ^ z P2 (hashencoder.cu:54-70) -- linear in x, so the eight cells
//   x in [8k, 8k + 8) of one (y, z) row are the eight entries of ONE 64-byte sector of the table (and consecutive entries on the dense levels).  A gather
//   instruction serves one corner of the 16 points of a tile: along x those 16 entries sit in 2 .. 9 sectors, along y or z in 16 (the primes scatter them
//   over the level's 4 MB).  The per-CU gather path and the fabric behind L2 -- what this kernel is short of -- see a fraction of the requests.
//   The results are staged in LDS and written with z fastest (64-byte runs) -- the volume keeps the reference's [x][y][z] layout.
//   Every XCD (workgroup index mod 8) walks through its own contiguous eighth of the bricks: coarse and middle levels are re-used from its L2.
template <int ROUND, bool BRICK>
__global__ __launch_bounds__(BLOCK) void field_sdf_grid_kernel(const RenderArgs a, const float *__restrict__ ax, const float *__restrict__ ay,
                                                               const float *__restrict__ az, uint32_t nx, uint32_t ny, uint32_t nz, int mode, float inv_s,
                                                               float *__restrict__ vol)
{
    // mode: what is stored per grid point -- 0 the sdf, 1 its negative (the mesh export's u), 2 update_extra_state's logistic density of it (dg_density)
    auto value = [&](float sdf) { return mode == 2 ? dg_density(sdf, inv_s) : (mode == 1 ? -sdf : sdf); };
    extern __shared__ __attribute__((aligned(16))) float lds[];
    fill_lds_sdf(lds, a);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    const FieldCtx fc = make_ctx(a);
    if constexpr (!BRICK) {
        const uint32_t B = nx * ny * nz, ntiles = (B + 15u) / 16u;
        for (uint32_t tile = blockIdx.x * WAVES_PER_BLOCK + wave; tile < ntiles; tile += gridDim.x * WAVES_PER_BLOCK) {
            const uint32_t b = tile * 16u + (uint32_t)n, bb = b < B ? b : B - 1u;
            const uint32_t iz = bb % nz, t = bb / nz, iy = t % ny, ix = t / ny;
            const f32x4 o = sdf_tile_r<ROUND>(lds, fc, lane, ax[ix], ay[iy], az[iz]);
            if (b < B && g == 0) vol[b] = value(o[0]);
        }
    } else {
        float *stage = lds + OFF_WAVE;                                                               // [GB_X][GB_Y][GB_Z]
        const uint32_t nbz = (nz + GB_Z - 1) / GB_Z, nby = (ny + GB_Y - 1) / GB_Y, nbx = (nx + GB_X - 1) / GB_X;
        const uint32_t nbricks = nbx * nby * nbz, per_xcd = (nbricks + 7u) / 8u;
        const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;        // (the grid is a multiple of 8 workgroups)
        for (uint32_t bi = slot; bi < per_xcd; bi += slots) {
            const uint32_t brick = xcd * per_xcd + bi;
            if (brick >= nbricks) break;                                                             // (workgroup-uniform)
            const uint32_t bz = brick % nbz, bt = brick / nbz, by = bt % nby, bx = bt / nby;
#pragma unroll 1
            for (uint32_t t = (uint32_t)wave; t < GB_Y * GB_Z; t += WAVES_PER_BLOCK) {
                const uint32_t ly = t >> 4, lz = t & 15u;
                const uint32_t ix = bx * GB_X + (uint32_t)n, iy = by * GB_Y + ly, iz = bz * GB_Z + lz;
                if (iy >= ny || iz >= nz) continue;                                                  // (wave-uniform)
                const uint32_t cx = ix < nx ? ix : nx - 1u;
                const f32x4 o = sdf_tile_r<ROUND>(lds, fc, lane, ax[cx], ay[iy], az[iz]);
                if (g == 0) stage[((uint32_t)n * GB_Y + ly) * GB_Z + lz] = value(o[0]);
            }
            __syncthreads();
            for (uint32_t o = threadIdx.x; o < GB_X * GB_Y * GB_Z; o += BLOCK) {
                const uint32_t lz = o & 15u, ly = (o >> 4) & 3u, lx = o >> 6;
                const uint32_t ix = bx * GB_X + lx, iy = by * GB_Y + ly, iz = bz * GB_Z + lz;
                if (ix < nx && iy < ny && iz < nz) vol[((size_t)ix * ny + iy) * nz + iz] = stage[o];
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- marching cubes
constexpr int MC_PTS = 1024;                     // grid points per workgroup (256 threads x 4 consecutive points: their four count bytes are one dword)
struct McDims { uint32_t nx, ny, nz, npts; };

__device__ __forceinline__ bool mc_flag(float u, float iso) { return u <= iso; }

// The corner flags of the whole volume as a bit array (1 bit per grid point: 16.8 MB at 512^3, L2-resident), written by ONE coalesced pass over the
// volume: a wave takes 64 consecutive points per step, the ballot of `u <= iso` is their 64 flags.  The classification then reads bits: its eight
// look-ups per point (four rows of the volume) were 2.1 GB of L2 traffic on floats (0.95 ms at 512^3), on bits they are nothing.
__global__ __launch_bounds__(256) void mc_flags_kernel(const float *__restrict__ vol, uint32_t npts, float iso, unsigned long long *__restrict__ bits64)
{
    const uint32_t lane = threadIdx.x & 63u, nchunks = (npts + 1023u) >> 10;            // a wave's unit of work: 1024 points = 16 steps
    for (uint32_t chunk = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; chunk < nchunks; chunk += (gridDim.x * blockDim.x) >> 6) {
        const uint32_t base = chunk << 10;
        unsigned long long mine = 0ull;
#pragma unroll
        for (uint32_t it = 0; it < 16; ++it) {
            const uint32_t p = base + it * 64u + lane;
            const unsigned long long m = __ballot(p < npts && mc_flag(vol[p < npts ? p : npts - 1u], iso));
            if (lane == it) mine = m;
        }
        if (lane < 16u && base + lane * 64u < npts) bits64[(base >> 6) + lane] = mine;     // 16 consecutive 8-byte words
    }
}
__device__ __forceinline__ bool mc_bit(const uint32_t *__restrict__ bits, uint32_t q) { return (bits[q >> 5] >> (q & 31u)) & 1u; }

// count byte of a grid point: bits 0-2 = which of its three OWNED edges (towards +x, +y, +z) change sign; bits 3-5 = triangles of the cell it is the origin of
__device__ __forceinline__ uint32_t mc_classify(const uint32_t *__restrict__ bits, const McDims d, uint32_t p, uint32_t *case_out = nullptr)
{
    const uint32_t k = p % d.nz, t = p / d.nz, j = t % d.ny, i = t / d.ny;
    const bool hx = i + 1u < d.nx, hy = j + 1u < d.ny, hz = k + 1u < d.nz;
    const uint32_t sx = d.ny * d.nz, sy = d.nz;
    const bool f0 = mc_bit(bits, p);
    uint32_t vmask = 0, cs = f0 ? 1u : 0u;
    bool f1 = false, f2 = false, f4 = false;
    if (hx) { f1 = mc_bit(bits, p + sx); vmask |= (f1 != f0) ? 1u : 0u; }
    if (hy) { f2 = mc_bit(bits, p + sy); vmask |= (f2 != f0) ? 2u : 0u; }
    if (hz) { f4 = mc_bit(bits, p + 1u); vmask |= (f4 != f0) ? 4u : 0u; }
    uint32_t nt = 0;
    if (hx && hy && hz) {
        cs |= (f1 ? 2u : 0u) | (f2 ? 4u : 0u) | (f4 ? 16u : 0u);
        cs |= mc_bit(bits, p + sx + sy) ? 8u : 0u;
        cs |= mc_bit(bits, p + sx + 1u) ? 32u : 0u;
        cs |= mc_bit(bits, p + sy + 1u) ? 64u : 0u;
        cs |= mc_bit(bits, p + sx + sy + 1u) ? 128u : 0u;
        nt = AC_MC_NTRI[cs];
    } else cs = 0u;
    if (case_out) *case_out = cs;
    return vmask | (nt << 3);
}

// workgroup-wide exclusive scan of one value per thread (256 threads), in thread order; total -> every thread
__device__ __forceinline__ uint32_t block_exscan_256(uint32_t v, uint32_t *sh /* [8] */, uint32_t &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, d); if (lane >= d) inc += o; }
    __syncthreads();                              // (sh may still be read by a previous call)
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) if (w < wave) base += sh[w];
    total = sh[0] + sh[1] + sh[2] + sh[3];
    return base + inc - v;
}

__global__ __launch_bounds__(256) void mc_classify_kernel(const uint32_t *__restrict__ bits, const McDims d, uint32_t *__restrict__ cnt4,
                                                          uint32_t *__restrict__ bsum)
{
    __shared__ uint32_t sh[8];
    const uint32_t p0 = blockIdx.x * MC_PTS + threadIdx.x * 4u;
    uint32_t packed = 0, nv = 0, nt = 0;
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) {
        const uint32_t p = p0 + r;
        uint32_t c = 0;
        if (p < d.npts) c = mc_classify(bits, d, p);
        packed |= c << (8 * r);
        nv += (uint32_t)__builtin_popcount(c & 7u); nt += c >> 3;
    }
    if (p0 < d.npts) cnt4[p0 >> 2] = packed;      // (the count array is padded to a multiple of 4 points)
    uint32_t tot;
    (void)block_exscan_256(nv | (nt << 16), sh, tot);      // 1024 points: at most 3072 vertices / 5120 triangles per workgroup -- 16 bits each
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// exclusive scan of the per-workgroup totals (vertices in the low, triangles in the high 16 bits) -> boff[2 * b], boff[2 * b + 1]; totals -> counts[0..1].
// One workgroup of 16 waves; a wave owns a contiguous range of the totals and walks it 64 at a time (coalesced loads, shuffle scan, running carry);
// the waves' totals are combined through LDS and added in a second coalesced pass (131 072 totals at 512^3: 128 steps per wave and pass).
__device__ __forceinline__ uint32_t wave_incscan(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)v, d); if (lane >= d) v += o; }
    return v;
}
__global__ __launch_bounds__(1024) void mc_scan_kernel(const uint32_t *__restrict__ bsum, uint32_t nblk, uint32_t *__restrict__ boff, uint32_t *__restrict__ counts)
{
    __shared__ uint32_t wv[16], wt[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t per = (((nblk + 15u) / 16u) + 63u) & ~63u, lo = (uint32_t)wave * per, hi = lo + per < nblk ? lo + per : nblk;
    uint32_t cv = 0, ct = 0;
    for (uint32_t b0 = lo; b0 < hi; b0 += 64u) {
        const uint32_t b = b0 + (uint32_t)lane;
        const uint32_t sv = b < hi ? bsum[b] : 0u;
        const uint32_t v = sv & 0xffffu, t = sv >> 16;
        const uint32_t iv = wave_incscan(v, lane), it = wave_incscan(t, lane);
        if (b < hi) { boff[2 * b] = cv + iv - v; boff[2 * b + 1] = ct + it - t; }
        cv += (uint32_t)__shfl((int)iv, 63); ct += (uint32_t)__shfl((int)it, 63);
    }
    if (lane == 0) { wv[wave] = cv; wt[wave] = ct; }
    __syncthreads();
    uint32_t bv = 0, bt = 0;
    for (int w = 0; w < wave; ++w) { bv += wv[w]; bt += wt[w]; }
    if (bv | bt)
        for (uint32_t b = lo + (uint32_t)lane; b < hi; b += 64u) { boff[2 * b] += bv; boff[2 * b + 1] += bt; }
    if (threadIdx.x == 1023) { counts[0] = bv + cv; counts[1] = bt + ct; }
}

struct McXform { double den, span[3], lo[3]; };   // world = index / den * span + lo, the reference's three operations in its order (instant_nsr.py:760-762)

__global__ __launch_bounds__(256) void mc_vertices_kernel(const float *__restrict__ vol, const McDims d, float iso, const uint32_t *__restrict__ cnt4,
                                                          const uint32_t *__restrict__ bsum, const uint32_t *__restrict__ boff, uint32_t *__restrict__ voff,
                                                          const McXform xf, double *__restrict__ verts, uint32_t n_verts)
{
    __shared__ uint32_t sh[8];
    if ((bsum[blockIdx.x] & 0xffffu) == 0u) return;
    const uint32_t p0 = blockIdx.x * MC_PTS + threadIdx.x * 4u;
    const uint32_t packed = p0 < d.npts ? cnt4[p0 >> 2] : 0u;
    uint32_t nv = 0;
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) nv += (uint32_t)__builtin_popcount((packed >> (8 * r)) & 7u);
    uint32_t tot;
    uint32_t vid = boff[2 * blockIdx.x] + block_exscan_256(nv, sh, tot);
    const uint32_t stride[3] = { d.ny * d.nz, d.nz, 1u };
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) {
        const uint32_t m = (packed >> (8 * r)) & 7u;
        if (!m) continue;
        const uint32_t p = p0 + r;
        voff[p] = vid;
        const uint32_t k = p % d.nz, t = p / d.nz, j = t % d.ny, i = t / d.ny;
        const double va = (double)vol[p];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (!((m >> a) & 1u)) continue;
            const double vb = (double)vol[p + stride[a]];
            const double tt = ((double)iso - va) / (vb - va);            // the linear zero crossing between the two grid points, 0 <= tt <= 1
            double c[3] = { (double)i, (double)j, (double)k };
            c[a] += tt;
            if (vid < n_verts) {
                verts[3 * (size_t)vid] = c[0] / xf.den * xf.span[0] + xf.lo[0];
                verts[3 * (size_t)vid + 1] = c[1] / xf.den * xf.span[1] + xf.lo[1];
                verts[3 * (size_t)vid + 2] = c[2] / xf.den * xf.span[2] + xf.lo[2];
            }
            ++vid;
        }
    }
}

__global__ __launch_bounds__(256) void mc_triangles_kernel(const uint32_t *__restrict__ bits, const McDims d, const uint32_t *__restrict__ cnt4,
                                                           const uint32_t *__restrict__ bsum, const uint32_t *__restrict__ boff, const uint32_t *__restrict__ voff,
                                                           int32_t *__restrict__ tris, uint32_t n_tris)
{
    __shared__ uint32_t sh[8];
    if ((bsum[blockIdx.x] >> 16) == 0u) return;
    const uint32_t p0 = blockIdx.x * MC_PTS + threadIdx.x * 4u;
    const uint32_t packed = p0 < d.npts ? cnt4[p0 >> 2] : 0u;
    uint32_t nt = 0;
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) nt += (packed >> (8 * r + 3)) & 7u;
    uint32_t tot;
    uint32_t tid = boff[2 * blockIdx.x + 1] + block_exscan_256(nt, sh, tot);
    const uint8_t *cnt8 = reinterpret_cast<const uint8_t *>(cnt4);
    const uint32_t sx = d.ny * d.nz, sy = d.nz;
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) {
        const uint32_t n = (packed >> (8 * r + 3)) & 7u;
        if (!n) continue;
        const uint32_t p = p0 + r;
        uint32_t cs;
        (void)mc_classify(bits, d, p, &cs);
        for (uint32_t t = 0; t < n; ++t) {
            int32_t id[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int e = AC_MC_TRI[cs][3 * t + q];
                const uint32_t c0 = AC_MC_EDGE[e][0], a = (uint32_t)e >> 2;
                const uint32_t pq = p + (c0 & 1u) * sx + ((c0 >> 1) & 1u) * sy + ((c0 >> 2) & 1u);     // the grid point that owns edge e of this cell
                const uint32_t m = cnt8[pq] & 7u;
                id[q] = (int32_t)(voff[pq] + (uint32_t)__builtin_popcount(m & ((1u << a) - 1u)));
            }
            if (tid < n_tris) { tris[3 * (size_t)tid] = id[0]; tris[3 * (size_t)tid + 1] = id[1]; tris[3 * (size_t)tid + 2] = id[2]; }
            ++tid;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- density grid of the ray marcher
// Round 6: two launches instead of one.  Round 5 gave every workgroup a brick of 16 x 8 x 8 outputs PLUS the one-point halo the 2^3 max pool reads
// (17 x 9 x 9 = 1377 evaluations for 1024 outputs, 3.58 M for the 2.15 M points of the 129^3 grid) and cut them into tiles of 16 consecutive evaluations of
// a 17-wide row -- tiles that straddle two rows, i.e. none of what makes field_sdf_grid_kernel fast (16 points along x = 2 .. 9 sectors per gather).
//   1. field_sdf_grid_kernel<.., true> in mode 2: the density of every grid point ONCE, on x-tiles, into a scratch volume (8.6 MB at 129^3: L2-resident);
//   2. density_pool_kernel: zero pad + 2^3 max pool of that volume, maximum(grid * decay, new) in place, the mean -- one short streaming pass.
// 0.94 -> 0.3x ms per update (bench.py: density_grid_update); the same values (sdf bits, expf / divide, max).
struct DgArgs {
    const float *dens;        // [H,H,H] the densities of pass 1
    uint32_t H;
    float decay;
    float *grid;              // [H,H,H] in / out
    double *partials;         // [number of workgroups]
    uint32_t *ticket;         // [1], zero between launches (the last workgroup re-arms it)
    double *mean_out;         // [1]
};
constexpr int DP_BLOCK = 256;

__global__ __launch_bounds__(DP_BLOCK) void density_pool_kernel(const DgArgs g_)
{
    __shared__ double red[DP_BLOCK / 64];
    __shared__ uint32_t last;
    const uint32_t H = g_.H, N = H * H * H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double part = 0.0;
#pragma unroll 4
    for (uint32_t p = blockIdx.x * DP_BLOCK + threadIdx.x; p < N; p += gridDim.x * DP_BLOCK) {       // z fastest: the grid's own order
        const uint32_t gz = p % H, t = p / H, gy = t % H, gx = t / H;
        float m = 0.0f;                                                      // (F.max_pool3d over the zero-padded tmp_grid (:341); densities are >= 0)
#pragma unroll
        for (uint32_t dx = 0; dx < 2; ++dx)
#pragma unroll
            for (uint32_t dy = 0; dy < 2; ++dy)
#pragma unroll
                for (uint32_t dz = 0; dz < 2; ++dz)
                    if (gx + dx < H && gy + dy < H && gz + dz < H) m = fmaxf(m, g_.dens[((size_t)(gx + dx) * H + (gy + dy)) * H + (gz + dz)]);
        const float v = fmaxf(g_.grid[p] * g_.decay, m);                     // torch.maximum(density_grid * decay, tmp_grid)  (:345)
        g_.grid[p] = v;
        part += (double)v;
    }
    // mean: per-wave shuffle tree -> per-workgroup sum (fixed order) -> the last workgroup adds the workgroups' partials (fixed order); in double, where the
    // reference's torch.mean reduces in fp32 -- the two agree to ~1e-7 relative (tests: <= 1e-6), not bit for bit: mean_density is a threshold of the marcher
    // (density > min(mean, 0.01 ...)), compared against values that differ from it by orders of magnitude
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < DP_BLOCK / 64; ++w) s += red[w];
        g_.partials[blockIdx.x] = s;
        __threadfence();
        last = atomicAdd(g_.ticket, 1u) == gridDim.x - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    double s = 0.0;
    for (uint32_t b = threadIdx.x; b < gridDim.x; b += DP_BLOCK)             // (agent-scope loads: served by L2, never by a stale vector-L1 line)
        s += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long *>(g_.partials) + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
    __syncthreads();
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int w = 0; w < DP_BLOCK / 64; ++w) tot += red[w];
        g_.mean_out[0] = tot / ((double)H * (double)H * (double)H);
        *g_.ticket = 0u;
    }
}

int geo_args(RenderArgs &a, const ac_field *field, float bound)
{
    if (int rc = fill_args(a, field, bound)) return rc;
    a.T0 = 0; a.lin_z = nullptr; a.lin_u = nullptr;
    return AC_OK;
}

struct McLayout { size_t cnt, voff, bsum, boff, bits, total; uint32_t nblk; };
McLayout mc_layout(uint32_t nx, uint32_t ny, uint32_t nz)
{
    McLayout l{};
    const uint64_t npts = (uint64_t)nx * ny * nz;
    l.nblk = (uint32_t)((npts + MC_PTS - 1) / MC_PTS);
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t o = 0;
    l.cnt = o; o += al((size_t)l.nblk * MC_PTS);
    l.voff = o; o += al((size_t)npts * 4);
    l.bsum = o; o += al((size_t)l.nblk * 4);
    l.boff = o; o += al((size_t)l.nblk * 8);
    l.bits = o; o += al(((size_t)l.nblk * MC_PTS / 64 + 2) * 8);
    l.total = o;
    return l;
}

int mc_check(const char *who, const float *vol, uint32_t nx, uint32_t ny, uint32_t nz, const void *scratch, size_t scratch_bytes, McLayout &l)
{
    if (!vol || !scratch) { ac::set_error("%s: NULL buffer", who); return AC_ERR_BAD_ARG; }
    if (nx < 2 || ny < 2 || nz < 2 || (uint64_t)nx * ny * nz >= (1ull << 31)) { ac::set_error("%s: grid %u x %u x %u unsupported (2 <= n, fewer than 2^31 points)", who, nx, ny, nz); return AC_ERR_BAD_ARG; }
    l = mc_layout(nx, ny, nz);
    if (scratch_bytes < l.total) { ac::set_error("%s: scratch of %zu bytes needed, %zu given", who, l.total, scratch_bytes); return AC_ERR_BAD_ARG; }
    return AC_OK;
}

// the regular-grid launch shared by ac_field_sdf_grid and ac_density_grid_update (mode: see field_sdf_grid_kernel)
int launch_grid(const RenderArgs &a, const float *axis_x, const float *axis_y, const float *axis_z, uint32_t nx, uint32_t ny, uint32_t nz, int mode, float inv_s,
                float *volume, hipStream_t stream, const char *what)
{
    const size_t lds_bytes = (OFF_WAVE + GB_X * GB_Y * GB_Z) * sizeof(float);
    const uint32_t B = nx * ny * nz, ntiles = (B + 15u) / 16u;
    // experiment switches (same values whatever they say): AC_GRID_ORDER = 0 linear tiles | 1 bricks per XCD; AC_GRID_ROUND = 2 | 4 levels per gather round
    static const int order = []() { const char *e = getenv("AC_GRID_ORDER"); return e ? atoi(e) : AC_GRID_ORDER_DEFAULT; }();
    static const int round_ = []() { const char *e = getenv("AC_GRID_ROUND"); return (e && atoi(e) == 4) ? 4 : (e && atoi(e) == 2 ? 2 : AC_GRID_ROUND_DEFAULT); }();
    using kern_t = void (*)(const RenderArgs, const float *, const float *, const float *, uint32_t, uint32_t, uint32_t, int, float, float *);
    const kern_t kern = order ? (round_ == 4 ? (kern_t)field_sdf_grid_kernel<4, true> : (kern_t)field_sdf_grid_kernel<2, true>)
                              : (round_ == 4 ? (kern_t)field_sdf_grid_kernel<4, false> : (kern_t)field_sdf_grid_kernel<2, false>);
    uint32_t blocks = order ? ((nx + GB_X - 1) / GB_X) * ((ny + GB_Y - 1) / GB_Y) * ((nz + GB_Z - 1) / GB_Z) : (ntiles + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    // persistent workgroups, exactly as many as the device keeps resident at once (43 KB of LDS each; the registers decide): a workgroup that had to
    // wait for a slot would start its share of the tiles when the others are done
    static int per_cu[4] = { 0, 0, 0, 0 };
    int &pc = per_cu[(order ? 2 : 0) + (round_ == 4 ? 1 : 0)];
    if (!pc) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, BLOCK, lds_bytes) != hipSuccess || n < 1) n = 1;
        pc = n;
    }
    const uint32_t cap = (uint32_t)pc * ac::cu_count();
    if (blocks > cap) blocks = cap;
    if (order) blocks = (blocks + 7u) & ~7u;                          // every XCD the same number of workgroups
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(BLOCK), lds_bytes, stream, a, axis_x, axis_y, axis_z, nx, ny, nz, mode, inv_s, volume);
    return ac::check_launch(what);
}

}  // namespace

AC_API int ac_field_sdf_grid(const ac_field *field, const float *axis_x, const float *axis_y, const float *axis_z, uint32_t nx, uint32_t ny, uint32_t nz,
                             float bound, int negate, float *volume, ac_stream_t stream)
{
    if (nx == 0 || ny == 0 || nz == 0) return AC_OK;
    if (!axis_x || !axis_y || !axis_z || !volume) { ac::set_error("field_sdf_grid: NULL buffer"); return AC_ERR_BAD_ARG; }
    if ((uint64_t)nx * ny * nz >= (1ull << 32) - 16) { ac::set_error("field_sdf_grid: more than 2^32 grid points"); return AC_ERR_BAD_ARG; }
    RenderArgs a{};
    if (int rc = geo_args(a, field, bound)) return rc;
    return launch_grid(a, axis_x, axis_y, axis_z, nx, ny, nz, negate ? 1 : 0, 0.0f, volume, (hipStream_t)stream, "field_sdf_grid");
}

AC_API size_t ac_marching_cubes_scratch(uint32_t nx, uint32_t ny, uint32_t nz)
{
    if (nx < 2 || ny < 2 || nz < 2 || (uint64_t)nx * ny * nz >= (1ull << 31)) return 0;
    return mc_layout(nx, ny, nz).total;
}

AC_API int ac_marching_cubes_count(const float *volume, uint32_t nx, uint32_t ny, uint32_t nz, float iso, void *scratch, size_t scratch_bytes,
                                   uint32_t *counts, ac_stream_t stream)
{
    McLayout l;
    if (int rc = mc_check("marching_cubes_count", volume, nx, ny, nz, scratch, scratch_bytes, l)) return rc;
    if (!counts) { ac::set_error("marching_cubes_count: NULL counts"); return AC_ERR_BAD_ARG; }
    char *sc = static_cast<char *>(scratch);
    const McDims d{ nx, ny, nz, nx * ny * nz };
    {
        const uint32_t nchunks = (d.npts + 1023u) >> 10;
        uint32_t blocks = (nchunks + 3u) / 4u;
        const uint32_t cap = 32u * ac::cu_count();
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(mc_flags_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, volume, d.npts, iso, reinterpret_cast<unsigned long long *>(sc + l.bits));
    }
    hipLaunchKernelGGL(mc_classify_kernel, dim3(l.nblk), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint32_t *>(sc + l.bits), d,
                       reinterpret_cast<uint32_t *>(sc + l.cnt), reinterpret_cast<uint32_t *>(sc + l.bsum));
    hipLaunchKernelGGL(mc_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, reinterpret_cast<const uint32_t *>(sc + l.bsum), l.nblk,
                       reinterpret_cast<uint32_t *>(sc + l.boff), counts);
    return ac::check_launch("marching_cubes_count");
}

AC_API int ac_marching_cubes_emit(const float *volume, uint32_t nx, uint32_t ny, uint32_t nz, float iso, void *scratch, size_t scratch_bytes,
                                  double den, const double span[3], const double lo[3], double *vertices, uint32_t n_vertices, int32_t *triangles,
                                  uint32_t n_triangles, ac_stream_t stream)
{
    McLayout l;
    if (int rc = mc_check("marching_cubes_emit", volume, nx, ny, nz, scratch, scratch_bytes, l)) return rc;
    if ((n_vertices && !vertices) || (n_triangles && !triangles) || !span || !lo || !(den != 0.0)) { ac::set_error("marching_cubes_emit: NULL buffer or den == 0"); return AC_ERR_BAD_ARG; }
    char *sc = static_cast<char *>(scratch);
    const McDims d{ nx, ny, nz, nx * ny * nz };
    McXform xf{ den, { span[0], span[1], span[2] }, { lo[0], lo[1], lo[2] } };
    const uint32_t *cnt4 = reinterpret_cast<const uint32_t *>(sc + l.cnt), *bsum = reinterpret_cast<const uint32_t *>(sc + l.bsum);
    const uint32_t *boff = reinterpret_cast<const uint32_t *>(sc + l.boff);
    uint32_t *voff = reinterpret_cast<uint32_t *>(sc + l.voff);
    if (n_vertices)
        hipLaunchKernelGGL(mc_vertices_kernel, dim3(l.nblk), dim3(256), 0, (hipStream_t)stream, volume, d, iso, cnt4, bsum, boff, voff, xf, vertices, n_vertices);
    if (n_triangles)
        hipLaunchKernelGGL(mc_triangles_kernel, dim3(l.nblk), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint32_t *>(sc + l.bits), d, cnt4, bsum, boff, voff,
                           triangles, n_triangles);
    return ac::check_launch("marching_cubes_emit");
}

static uint32_t dp_blocks(uint32_t H)
{
    // few workgroups, several points per thread: every workgroup ends with a device-scope fence + a ticket (the mean's fixed-order reduction), and with one
    // point per thread (8403 workgroups at 129^3) those epilogues were the pass: 223 us against 65 us with 2048 workgroups
    const uint64_t n = ((uint64_t)H * H * H + DP_BLOCK - 1) / DP_BLOCK;
    const uint32_t cap = 4u * ac::cu_count();
    return (uint32_t)(n < cap ? n : cap);
}
static size_t dg_dens_offset(uint32_t H) { return (256 + (size_t)dp_blocks(H) * sizeof(double) + 255) & ~(size_t)255; }

AC_API size_t ac_density_grid_update_scratch(uint32_t H)
{
    if (H < 2 || H > 1024) return 0;
    return dg_dens_offset(H) + (size_t)H * H * H * sizeof(float);           // ticket | per-workgroup partial sums | the densities of pass 1
}

AC_API int ac_density_grid_update(const ac_field *field, const float *axis, uint32_t H, float bound, float inv_s, float decay, float *grid,
                                  double *mean_out, void *scratch, size_t scratch_bytes, ac_stream_t stream)
{
    if (!axis || !grid || !mean_out || !scratch) { ac::set_error("density_grid_update: NULL buffer"); return AC_ERR_BAD_ARG; }
    const size_t need = ac_density_grid_update_scratch(H);
    if (!need || scratch_bytes < need) { ac::set_error("density_grid_update: H = %u unsupported or scratch of %zu bytes needed, %zu given", H, need, scratch_bytes); return AC_ERR_BAD_ARG; }
    RenderArgs a{};
    if (int rc = geo_args(a, field, bound)) return rc;
    float *dens = reinterpret_cast<float *>(static_cast<char *>(scratch) + dg_dens_offset(H));
    if (int rc = launch_grid(a, axis, axis, axis, H, H, H, 2, inv_s, dens, (hipStream_t)stream, "density_grid_update")) return rc;
    DgArgs g{};
    g.dens = dens; g.H = H; g.decay = decay; g.grid = grid;
    g.ticket = static_cast<uint32_t *>(scratch);
    g.partials = reinterpret_cast<double *>(static_cast<char *>(scratch) + 256);
    g.mean_out = mean_out;
    hipLaunchKernelGGL(density_pool_kernel, dim3(dp_blocks(H)), dim3(DP_BLOCK), 0, (hipStream_t)stream, g);
    return ac::check_launch("density_grid_update");
}
