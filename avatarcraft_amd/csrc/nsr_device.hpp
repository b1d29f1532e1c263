// avatarcraft_amd/csrc/nsr_device.hpp -- device-side building blocks of the Instant-NSR field on gfx950, shared by the fused
// renderer (render_fused.hip) and the fused training operators (sdf_train.hip): LDS layout, MFMA-ordered weight fragments,
// hash-grid gathers (single point and 7-point finite-difference stencil), SDF / colour MLP tiles, DPP scans.
// Everything lives in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include "ac_common.hpp"
#include "ac_devmath.hpp"
#include "ac_sp_table.hpp"
#include "ac_sh16.hpp"

using namespace acdev;

typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifdef AC_ABL_MFMA     // timing ablation: replace every MFMA by one VALU fma per accumulator register
static __device__ __forceinline__ f32x4 abl_mfma(float a, float b, f32x4 c) { c[0] = __builtin_fmaf(a, b, c[0]); return c; }
#define __builtin_amdgcn_mfma_f32_16x16x4f32(A, B, C, X, Y, Z) abl_mfma(A, B, C)
#endif
#ifdef AC_ABL_GATHER   // timing ablation: every gather reads table entry 0 (perfectly cached, fully coalesced)
#define AC_GOFF(X) ((X) & 0u)
#elif defined(AC_ABL_L0)   // timing ablation (round 3, "what could level 0 in LDS save at most?"): gathers below byte AC_ABL_L0 of the table -- 39304 = the
// dense 17^3 level 0 -- go out of range (dropped by the descriptor's bounds check: free).  The threshold is a mutable device global so that the control
// build (-DAC_ABL_L0=0) runs exactly the same instructions and drops nothing.  Results are wrong by construction; timing only.
static __device__ uint32_t g_abl_l0_thr = AC_ABL_L0;
#define AC_GOFF(X) (((X) < g_abl_l0_thr) ? 0xfffffff8u : (X))
#else
#define AC_GOFF(X) (X)
#endif

namespace {

#ifndef AC_WPB
#define AC_WPB 8
#endif
constexpr int WAVES_PER_BLOCK = AC_WPB;
constexpr int BLOCK = WAVES_PER_BLOCK * 64;
constexpr int MAXT = 128;
constexpr int SEG_STATE = MAXT + 16 + 64;   // floats per ray handed from one segment of a ray to the next: z [128] | cT, 10 running sums | pad | (view directions) the ray's layer-1 bias [64]
#ifndef AC_FINE_BATCH
#define AC_FINE_BATCH 3     // fine stencil levels: offset-point pairs gathered per memory round trip (1: x | y | z, 2: x+y | z, 3: all six: 221 VGPRs, -2 % time)
#endif
#ifndef AC_FACE_VALUE
#define AC_FACE_VALUE 0     // fast precision: coarse-level features of the six offset points from bilinear face values (see coarse_finish_fv)
#endif
#ifndef AC_FV_SIGNS
#define AC_FV_SIGNS 3       // (debug) bit 0: +eps, bit 1: -eps along z use the face-value form
#endif
#ifndef AC_FV_AXES
#define AC_FV_AXES 7        // (debug) bit k: the face-value form is used for the offset points along axis k
#endif
#ifndef AC_SENTINEL_LOADS
#define AC_SENTINEL_LOADS 0  // 1: lanes of a coarse level that need no new face send an out-of-range offset instead of being masked out (round 1 / 2; see coarse_issue)
#endif
#ifndef AC_ENC_ROUND
#define AC_ENC_ROUND 2      // hash levels gathered per round per lane (registers vs loads in flight)
#endif

// ---- LDS layout (floats) -------------------------------------------------------------------------
constexpr int OFF_W1F = 0;                       // [4 tiles][9 ksteps][64]
constexpr int OFF_W2F = OFF_W1F + 4 * 9 * 64;    // [16 ksteps][64]
constexpr int OFF_C1F = OFF_W2F + 16 * 64;       // [4][6][64]
constexpr int OFF_C2F = OFF_C1F + 4 * 6 * 64;    // [4][16][64]
constexpr int OFF_C3F = OFF_C2F + 4 * 16 * 64;   // [16][64]
constexpr int OFF_B1 = OFF_C3F + 16 * 64;        // [64]
constexpr int OFF_B2 = OFF_B1 + 64;              // [16]
constexpr int OFF_LVL = OFF_B2 + 16;             // [16 levels][8 words]
constexpr int OFF_LIN = OFF_LVL + 16 * 8;        // lin_z[64] + lin_u[16]
constexpr int SPQ_FLOATS = 129 * 4;              // softplus G table [129][4] (row 128 = 0)
constexpr int OFF_SPQ = OFF_LIN + 80;
constexpr int OFF_WAVE = OFF_SPQ + SPQ_FLOATS;        // end of the weights every kernel shares; per-wave slabs of the training kernels start here
constexpr int FE_SLAB = 6 * 8 * 64;                        // hash features of the 6 finite-difference points: [e-1][2j+c][lane]
// the renderer only: layer 1 of sdf_net for the "fast" precision (ac_render_opts.precision = 1) -- the 32 feature columns of W1 split into
// bf16 hi + lo in the A-fragment order of v_mfma_f32_16x16x32_bf16 (lane (unit m, group kk) holds k = 8 kk .. 8 kk + 7 = the eight
// features lane group kk gathers: 16 bytes per lane and tile), and the three coordinate columns as plain fp32
constexpr int OFF_W1H = OFF_WAVE;                // [4 tiles][64 lanes][4 dwords]
constexpr int OFF_W1L = OFF_W1H + 4 * 64 * 4;
constexpr int OFF_W1C = OFF_W1L + 4 * 64 * 4;    // [3 axes][64 units]
// ... and the colour network in split bf16 (fast precision): 14 A-fragments [64 lanes][4 dwords] (layer 1: 4 output tiles; layer 2: 4 tiles x 2 k-steps
// of 32; layer 3: 2 k-steps), hi and lo parts.  They OVERLAY the fp32 colour fragments [OFF_C1F, OFF_B1) -- 14 hi + the first 12 lo fill it exactly --
// and the last two lo fragments (layer 3) sit behind W1C.
constexpr int CF_FRAGS = 14, CF_FRAG = 64 * 4;
constexpr int OFF_CFH = OFF_C1F;                            // hi fragments 0..13
constexpr int OFF_CFL = OFF_C1F + CF_FRAGS * CF_FRAG;       // lo fragments 0..11
constexpr int OFF_C3L = OFF_W1C + 3 * 64;                   // lo fragments 12, 13
static_assert(OFF_CFL + 12 * CF_FRAG == OFF_B1, "the bf16 colour fragments fill the fp32 colour region exactly");
constexpr int OFF_RWAVE = OFF_C3L + 2 * CF_FRAG;  // per-wave slabs of the renderer
constexpr int CF_OVERLAY = OFF_B1 - OFF_C1F;      // floats of the overlay: the prepared image keeps it behind the exact image
// per-wave slab of the renderer: the final z values (zs0) + ONE region that holds the up-sampling state (second z buffer, the two sdf
// buffers, cdf, new samples) until the sampling of the ray is finished and the finite-difference feature slab afterwards
constexpr int UPS_FLOATS = MAXT + 2 * MAXT + MAXT + 32;              // zs1[128], sd[2][128], cdf[128], znew[16] + pad
constexpr int SLAB_ACC = MAXT + (FE_SLAB > UPS_FLOATS ? FE_SLAB : UPS_FLOATS);     // the ray's ten running sums (compositing, eikonal): [16] floats, kept by lane 15
constexpr int SLAB_SHB = SLAB_ACC + 16;                                           // use_viewdirs: the ray's layer-1 bias of the colour network [64] + sh(d) [16]
constexpr int WAVE_SLAB = SLAB_SHB + 80;
constexpr int LDS_FLOATS = OFF_RWAVE + WAVES_PER_BLOCK * WAVE_SLAB;
static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget: one workgroup per CU");
static_assert(OFF_WAVE % 4 == 0 && OFF_RWAVE % 4 == 0 && WAVE_SLAB % 4 == 0, "16-byte aligned slabs");

// per-level launch constants: index = (hashed ? x ^ y*my ^ z*mz : x + y*my + z*mz) & mask [% wsize if wsize]
// (my,mz) = (P1,P2) for hashed levels, (res+1, (res+1)^2) for dense ones; mask = size-1 for power-of-two hashed
// levels, ~0 otherwise (a dense index is < size by construction); wsize = size only for a hashed level whose size
// is not a power of two (never the case for tables allocated by HashEncoder, handled for completeness).
#ifdef AC_PROFILE
#define AC_T0() unsigned long long t_prof_ = __builtin_amdgcn_s_memtime(); const unsigned long long ray_r0_ = __builtin_amdgcn_s_memrealtime()
#define AC_TICK(SLOT) { const unsigned long long t2_ = __builtin_amdgcn_s_memtime(); prof_acc[SLOT] += t2_ - t_prof_; t_prof_ = t2_; }
#else
#define AC_T0()
#define AC_TICK(SLOT)
#endif

struct LevelRec { float scale; uint32_t my, mz, offset, mask, hashed, wsize, pad; };

struct RenderArgs {
    const float *table;
    uint32_t table_bytes;
    const float *W1, *b1, *W2, *b2, *Wc1, *Wc2, *Wc3;
    const float *Wsh;           // ac_field.Wc1_sh: [64,16] colour layer-1 columns of the 16 spherical harmonics of the view direction, or NULL (no view directions)
    const float *prepared;      // ac_field.prepared: the renderer's LDS image [0, OFF_RWAVE) of these parameters, or NULL
    const float *rays_o, *rays_d, *bg, *noise, *lin_z, *lin_u;
    ac_render_out out;
    LevelRec lvl[16];
    int n_rays, T0, nup;
    int jmode[4];          // per gather round j (levels 4j..4j+3): 0 all dense, 1 all hashed, 2 mixed
    int jfine[4];          // per round: 1 if the FD offset eps can span >= 1 cell on any of its levels
    float bound, two_bound, inv_s, car, one_m_car, eps;
    float inv_tb;          // RN(1 / two_bound) if the divisor is one of the exhaustively verified ones (unit_div), else 0: IEEE division
    const float *inv_s_dev;     // non-NULL: inv_s is read from device memory (ac_render_opts.inv_s_dev)
    int fast;                   // ac_render_opts.precision
    int skip_masked;            // ac_render_opts.skip_masked (MODE_FINAL only)
    int opacity_only;           // ac_render_opts.opacity_only: no colour network (rgb = 0)
    int perturb;
    unsigned long long *prof;   // AC_PROFILE builds only: [n_waves][10]: 8 per-phase s_memtime counters, whole-wave s_memtime and s_memrealtime (100 MHz)
    // posed-space rendering (render_can=False) only: see ac_render_rays_warped
    const float *near_m, *far_m;   // [N] mesh-guided range (+-inf where the ray misses the body) or NULL
    const float *ext_pts;          // MODE_UPSAMPLE: warped coarse points [N,T0,3]; MODE_FINAL: warped mid points [N,T,3]
    const uint8_t *mask;           // MODE_FINAL: [N,T] alpha mask
    uint32_t *ray_counter;         // AC_DYNAMIC_RAYS: [8 XCDs][8 segments] work counters (zeroed before the launch): waves fetch their next (ray, segment) instead of owning fixed rays
    uint32_t *seg_flags;           // [N] number of finished segments of each ray (zeroed before the launch), or NULL when seg_n == 1
    float *seg_state;              // [N][SEG_STATE] what a ray's next segment continues from: z values + running sums (library scratch)
    uint64_t seg_cb;               // first tile of segment s in bits 4s .. 4s+3, s = 0 .. seg_n
    int seg_n;                     // segments per ray (1 .. 8)
    // pair launches (ac_render_rays_pair): the same pair_n rays twice, n_rays = 2 * pair_n work items handed out as a0 b0 a1 b1 ...; copy a = rows
    // [0, pair_n), copy b = rows [pair_n, 2 pair_n) of noise, bg and the per-ray outputs; rays_o / rays_d / near_m / far_m have pair_n rows.  0 = off.
    int pair_n;
    // launch hygiene (round 3): the last workgroup to finish (ticket from done_counter) reduces gradient_error into eik_red -- [1 or 2 (pair)][2] floats:
    // (sum relax * err / (sum relax + 1e-5), sum relax + 1e-5) in eikonal_reduce_kernel's fixed order, bit for bit -- and re-arms the work counters for the
    // slot's next launch; seg_flags carry the launch generation (gen << 4 | finished segments), so they need no clearing either: no memset and no reduction
    // launch around the render kernel.
    uint32_t *done_counter;
    uint32_t *handoff_timeouts;    // device word counting segment hand-offs that timed out (ac_render_handoff_timeouts), or NULL
    float *eik_red;
    uint32_t gen;
    int ex_from;                   // the per-sample outputs (EX kernels) are kept for rays >= ex_from only, at row ray - ex_from of arrays with ex_rows rows
    int ex_rows;
    const uint8_t *ray_dead;       // MODE_UPSAMPLE, skip_masked: [N] rays that cannot hold an unmasked sample (no field evaluation, coarse z only)
    float *zbuf;                   // [N,T] final z values: written by MODE_UPSAMPLE, read by MODE_FINAL
    float *mid_pts;                // MODE_UPSAMPLE: posed-space mid points [N,T,3]
};

// render_rays_kernel<MODE>: MODE_FULL is the fused canonical-space renderer.  With the SMPL inverse warp between the phases
// (a global closest-point search per sample: csrc/warp.hip) the same code is instantiated as its two halves:
// MODE_UPSAMPLE = coarse sdf at externally warped points + the NeuS up-sampling, writes z and the posed-space mid points;
// MODE_FINAL = render core at externally warped mid points, alpha multiplied by the mask.
enum { MODE_FULL = 0, MODE_UPSAMPLE = 2, MODE_FINAL = 3 };

// near_far_from_bound (cube)  instant_nsr.py:58-77
__device__ __forceinline__ void cube_near_far(float ox, float oy, float oz, float dx, float dy, float dz, float bound, float &near, float &far)
{
    const float ex = dx + 1e-15f, ey = dy + 1e-15f, ez = dz + 1e-15f;
    const float ax = (-bound - ox) / ex, bx = (bound - ox) / ex;
    const float ay = (-bound - oy) / ey, by = (bound - oy) / ey;
    const float az = (-bound - oz) / ez, bz = (bound - oz) / ez;
    const float lx = ax < bx ? ax : bx, hx = ax > bx ? ax : bx;
    const float ly = ay < by ? ay : by, hy = ay > by ? ay : by;
    const float lz = az < bz ? az : bz, hz = az > bz ? az : bz;
    near = lx; if (ly > near) near = ly; if (lz > near) near = lz;
    far = hx; if (hy < far) far = hy; if (hz < far) far = hz;
    if (near < 0.05f) near = 0.05f;
}
__device__ __forceinline__ bool is_inf(float v) { return __builtin_fabsf(v) == __builtin_inff(); }

// ---- wave-level helpers -----------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int OFF> __device__ __forceinline__ float dpp_shr(float ident, float v)
{
    // lanes n >= OFF of every 16-lane row receive v[n-OFF]; the others keep `ident`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ident), __builtin_bit_cast(int, v),
                                                                  0x110 + OFF, 0xf, 0xf, false));
}
// Kogge-Stone inclusive scan inside every 16-lane row
template <bool MUL> __device__ __forceinline__ float row_scan(float v)
{
    const float id = MUL ? 1.0f : 0.0f;
    float s;
    s = dpp_shr<1>(id, v); v = MUL ? s * v : s + v;
    s = dpp_shr<2>(id, v); v = MUL ? s * v : s + v;
    s = dpp_shr<4>(id, v); v = MUL ? s * v : s + v;
    s = dpp_shr<8>(id, v); v = MUL ? s * v : s + v;
    return v;
}
__device__ __forceinline__ float lane_bcast(float v, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float sel4(int g, float a, float b, float c, float d)
{
    return g == 0 ? a : (g == 1 ? b : (g == 2 ? c : d));
}

// ---- workgroup prologue: weights -> MFMA A-fragment order in LDS -------------------------------------
// the SDF side (sdf_net fragments, biases, level records, sampling tables, softplus table) and the colour side are laid out separately: the
// SDF-query training kernels never touch the colour fragments and use their 26 KB for their own
__device__ __forceinline__ void fill_lds_sdf(float *lds, const RenderArgs &a)
{
    for (int e = threadIdx.x; e < 4 * 9 * 64; e += blockDim.x) {          // sdf_net.0 [64,35]
        int l = e & 63, fs = e >> 6, t = fs / 9, s = fs % 9;
        int u = 16 * t + (l & 15), g = l >> 4;
        int col = (s == 0) ? (g < 3 ? g : -1) : 3 + 2 * (4 * ((s - 1) >> 1) + g) + ((s - 1) & 1);
        lds[OFF_W1F + e] = col < 0 ? 0.0f : a.W1[u * 35 + col];
    }
    for (int e = threadIdx.x; e < 16 * 64; e += blockDim.x) {             // sdf_net.1 [16,64]
        int l = e & 63, kk = e >> 6, t = kk >> 2, r = kk & 3;
        lds[OFF_W2F + e] = a.W2[(l & 15) * 64 + 16 * t + 4 * (l >> 4) + r];
    }
    for (int e = threadIdx.x; e < 64; e += blockDim.x) lds[OFF_B1 + e] = a.b1[e];
    for (int e = threadIdx.x; e < 16; e += blockDim.x) lds[OFF_B2 + e] = a.b2[e];
    {   // level records: static kernarg indexing only (a dynamic index would copy the struct to scratch)
        uint32_t *lw = reinterpret_cast<uint32_t *>(lds) + OFF_LVL;
#pragma unroll
        for (int l = 0; l < 16; ++l) {
            if (threadIdx.x == (unsigned)l) {
                lw[8 * l + 0] = __float_as_uint(a.lvl[l].scale); lw[8 * l + 1] = a.lvl[l].my;
                lw[8 * l + 2] = a.lvl[l].mz; lw[8 * l + 3] = a.lvl[l].offset;
                lw[8 * l + 4] = a.lvl[l].mask; lw[8 * l + 5] = a.lvl[l].hashed;
                lw[8 * l + 6] = a.lvl[l].wsize; lw[8 * l + 7] = 0u;
            }
        }
    }
    for (int e = threadIdx.x; e < 64; e += blockDim.x) lds[OFF_LIN + e] = e < a.T0 ? a.lin_z[e] : 0.0f;
    for (int e = threadIdx.x; e < 16; e += blockDim.x) lds[OFF_LIN + 64 + e] = a.lin_u ? a.lin_u[e] : 0.0f;
    for (int e = threadIdx.x; e < SPQ_FLOATS; e += blockDim.x) lds[OFF_SPQ + e] = AC_SP_G[e >> 2][e & 3];
}

__device__ __forceinline__ void fill_lds_color(float *lds, const RenderArgs &a)
{
    for (int e = threadIdx.x; e < 4 * 6 * 64; e += blockDim.x) {          // color_net.0 [64,21] = [x(3), n(3), feat(15)]
        int l = e & 63, fs = e >> 6, t = fs / 6, s = fs % 6;
        int u = 16 * t + (l & 15), g = l >> 4;
        float v;
        if (s < 4) { int o = 4 * g + s; v = (o == 0) ? 0.0f : a.Wc1[u * 21 + 6 + (o - 1)]; }
        else if (s == 4) v = g < 3 ? a.Wc1[u * 21 + g] : 0.0f;
        else v = g < 3 ? a.Wc1[u * 21 + 3 + g] : 0.0f;
        lds[OFF_C1F + e] = v;
    }
    for (int e = threadIdx.x; e < 4 * 16 * 64; e += blockDim.x) {         // color_net.1 [64,64]
        int l = e & 63, fs = e >> 6, to = fs >> 4, kk = fs & 15, t = kk >> 2, r = kk & 3;
        lds[OFF_C2F + e] = a.Wc2[(16 * to + (l & 15)) * 64 + 16 * t + 4 * (l >> 4) + r];
    }
    for (int e = threadIdx.x; e < 16 * 64; e += blockDim.x) {             // color_net.2 [3,64]
        int l = e & 63, kk = e >> 6, t = kk >> 2, r = kk & 3, o = l & 15;
        lds[OFF_C3F + e] = o < 3 ? a.Wc3[o * 64 + 16 * t + 4 * (l >> 4) + r] : 0.0f;
    }
}

__device__ __forceinline__ void fill_lds(float *lds, const RenderArgs &a)
{
    fill_lds_sdf(lds, a);
    fill_lds_color(lds, a);
}

// ---- "fast" precision: bf16 hi / lo fragments of W1's feature columns, fp32 coordinate columns ----------
__device__ __forceinline__ uint32_t bf16_rne_bits(float f)       // finite inputs (weights): round to nearest even
{
    const uint32_t u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
template <int W1H = OFF_W1H, int W1L = OFF_W1L, int W1C = OFF_W1C>
__device__ __forceinline__ void fill_lds_fast(float *lds, const RenderArgs &a)
{
    uint32_t *lw = reinterpret_cast<uint32_t *>(lds);
    for (int e = threadIdx.x; e < 4 * 64 * 4; e += blockDim.x) {          // dword q of lane l, tile t: slots i = 2q, 2q + 1 of lane group kk
        const int q = e & 3, l = (e >> 2) & 63, t = e >> 8;
        const int u = 16 * t + (l & 15), kk = l >> 4;
        uint32_t hi2 = 0, lo2 = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = 2 * q + h;                                      // feature slot 2 j + c of lane group kk = level 4 j + kk, channel c
            const float w = a.W1[u * 35 + 3 + 2 * (4 * (i >> 1) + kk) + (i & 1)];
            const uint32_t hb = bf16_rne_bits(w);
            const uint32_t lb = bf16_rne_bits(w - __uint_as_float(hb << 16));
            hi2 |= hb << (16 * h); lo2 |= lb << (16 * h);
        }
        lw[W1H + e] = hi2; lw[W1L + e] = lo2;
    }
    for (int e = threadIdx.x; e < 3 * 64; e += blockDim.x) lds[W1C + e] = a.W1[(e & 63) * 35 + (e >> 6)];
}

// colour network, fast precision.  B operand of v_mfma_f32_16x16x32_bf16: lane (sample n, group g) supplies k = 8g .. 8g+7; the eight slots of a lane
// are values it already holds, so no data moves between lanes:
//   layer 1: slots 0..3 = sdf_out[4g + i] (the SDF network's output rows of this lane; row 0, the sdf itself, gets weight 0), slots 4..6 = x (g = 0) or
//            the normal (g = 1), everything else 0;
//   layer 2 / 3, k-step s: slots i = the lane's rows r = i & 3 of the previous layer's output tiles 2s + (i >> 2), i.e. units 16 (2s + (i >> 2)) + 4g + r.
// The weights (A operand, lane (row m, group g)) are permuted to match and split hi + lo by round-to-nearest; the activations are split by
// truncation (split8_bf16); three of the four partial products are formed.  Weight of fragment f, lane l, slot i:
__device__ __forceinline__ float color_fast_weight(const RenderArgs &a, int f, int l, int i)
{
    const int m = l & 15, g = l >> 4;
    if (f < 4) {                                                          // layer 1, output tile f
        const int u = 16 * f + m;
        if (i < 4) { const int o = 4 * g + i; return o == 0 ? 0.0f : a.Wc1[u * 21 + 5 + o]; }
        if (i < 7 && g < 2) return a.Wc1[u * 21 + 3 * g + (i - 4)];
        return 0.0f;
    }
    const int s = (f - 4) & 1, unit = 16 * (2 * s + (i >> 2)) + 4 * g + (i & 3);
    if (f < 12) return a.Wc2[(16 * ((f - 4) >> 1) + m) * 64 + unit];    // layer 2, output tile (f - 4) / 2, k-step s
    return m < 3 ? a.Wc3[m * 64 + unit] : 0.0f;                           // layer 3, k-step s
}
template <bool OVERLAY, bool TAIL>
__device__ __forceinline__ void fill_lds_color_fast(float *lds, const RenderArgs &a)
{
    uint32_t *lw = reinterpret_cast<uint32_t *>(lds);
    for (int e = threadIdx.x; e < CF_FRAGS * CF_FRAG; e += blockDim.x) {
        const int q = e & 3, l = (e >> 2) & 63, f = e >> 8;
        if (!(f < 12 ? OVERLAY : (OVERLAY || TAIL))) continue;
        uint32_t hi2 = 0, lo2 = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float w = color_fast_weight(a, f, l, 2 * q + h);
            const uint32_t hb = bf16_rne_bits(w);
            const uint32_t lb = bf16_rne_bits(w - __uint_as_float(hb << 16));
            hi2 |= hb << (16 * h); lo2 |= lb << (16 * h);
        }
        if (OVERLAY) lw[OFF_CFH + e] = hi2;
        if (f < 12) { if (OVERLAY) lw[OFF_CFL + e] = lo2; }
        else if (TAIL) lw[OFF_C3L + (e - 12 * CF_FRAG)] = lo2;
    }
}

// ---- hash-grid features of this lane's 4 levels (HashEncoder.forward + kernel_grid) -------------------
// p: world position (clamped to +-bound); returns f[j][c] for level 4j+g.  The gathers go through a
// buffer descriptor (32-bit byte offsets, hardware bounds check) and are issued ROUND levels at a time.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32q __attribute__((ext_vector_type(4)));
#ifndef AC_DENSE_PAIRS
#define AC_DENSE_PAIRS 0    // 1: on dense levels the corners (x, x + 1) of a cell come through ONE 16-byte gather (round 4 experiment: bit-identical, 8.7 % fewer
                            // gather instructions on the all-dense round, and NO gain -- render 0.7438 vs 0.7464 ms, the frozen SDS render 0.969 vs 0.959:
                            // the texture-address path is paced by bytes per instruction, a 64-lane dwordx4 costs it what two dwordx2 cost; r04_experiments 7b)
#endif

#ifndef AC_PK_INTERP
#define AC_PK_INTERP 0      // 1: the trilinear interpolation on the packed fp32 instructions (v_pk_mul_f32 / v_pk_fma_f32) -- round 6 experiment: bit-identical, 6.8 % fewer vector
                            // instructions (6813 -> 6349 static, 10 spilled dwords fewer) and SLOWER: 0.746 -> 0.781 ms (profiles/r06_experiments.txt section 4d)
#endif
typedef float f32p __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void interp8(const u32x2 (&v)[8], float qx, float qy, float qz, bool oob, float &f0, float &f1)
{
#if AC_PK_INTERP
    // The same 12 multiplications and 16 fused multiply-adds per level as below, in the same order per accumulator -- the same bits -- as 6 + 8 packed
    // instructions: the weights of the corners (x, x + 1) travel as a pair, a corner's two channels are the pair the gather delivered, and a weight is
    // broadcast to both channels by the instruction's operand selects (no move).
    const f32p wx = { 1.0f - qx, qx };
    const float wy0 = 1.0f - qy, wz0 = 1.0f - qz;
    const f32p w0 = wx * f32p{ wy0, wy0 }, w1 = wx * f32p{ qy, qy };           // (w00, w10), (w01, w11)
    f32p acc = { 0.0f, 0.0f };
#pragma unroll
    for (int c = 0; c < 8; c += 2) {
        const float z = (c & 4) ? qz : wz0;
        const f32p wp = ((c & 2) ? w1 : w0) * f32p{ z, z };
        const f32p a = { __uint_as_float(v[c].x), __uint_as_float(v[c].y) }, b = { __uint_as_float(v[c + 1].x), __uint_as_float(v[c + 1].y) };
#if AC_PK_INTERP == 2       // the weights packed, the accumulation as two independent scalar chains
        acc[0] = fma_(wp[0], a[0], acc[0]); acc[1] = fma_(wp[0], a[1], acc[1]);
        acc[0] = fma_(wp[1], b[0], acc[0]); acc[1] = fma_(wp[1], b[1], acc[1]);
#else
        acc = __builtin_elementwise_fma(__builtin_shufflevector(wp, wp, 0, 0), a, acc);
        acc = __builtin_elementwise_fma(__builtin_shufflevector(wp, wp, 1, 1), b, acc);
#endif
    }
    f0 = oob ? 0.0f : acc[0];
    f1 = oob ? 0.0f : acc[1];
#else
    const float wx0 = 1.0f - qx, wy0 = 1.0f - qy, wz0 = 1.0f - qz;
    const float w00 = wx0 * wy0, w10 = qx * wy0, w01 = wx0 * qy, w11 = qx * qy;
    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float wxy = (c & 2) ? ((c & 1) ? w11 : w01) : ((c & 1) ? w10 : w00);
        const float w = wxy * ((c & 4) ? qz : wz0);
        a0 = fma_(w, __uint_as_float(v[c].x), a0);
        a1 = fma_(w, __uint_as_float(v[c].y), a1);
    }
    f0 = oob ? 0.0f : a0;
    f1 = oob ? 0.0f : a1;
#endif
}
template <int ROUND>
__device__ __forceinline__ void encode4(const float *__restrict__ lds, rsrc_t table, int g, const int (&jmode)[4],
                                        float px, float py, float pz, float bound, float two_bound, float (&f)[4][2], float inv_tb = 0.0f)
{
    float ux, uy, uz;
    if (inv_tb != 0.0f) { ux = unit_div(px + bound, two_bound, inv_tb); uy = unit_div(py + bound, two_bound, inv_tb); uz = unit_div(pz + bound, two_bound, inv_tb); }
    else { ux = (px + bound) / two_bound; uy = (py + bound) / two_bound; uz = (pz + bound) / two_bound; }
    const bool oob = (ux < 0.0f) | (ux > 1.0f) | (uy < 0.0f) | (uy > 1.0f) | (uz < 0.0f) | (uz > 1.0f);
#pragma unroll
    for (int j0 = 0; j0 < 4; j0 += ROUND) {
        float q[ROUND][3];
        u32x2 v[ROUND][8];
#pragma unroll
        for (int jj = 0; jj < ROUND; ++jj) {
            const int j = j0 + jj;
            const uint4 r0 = *reinterpret_cast<const uint4 *>(lds + OFF_LVL + (4 * j + g) * 8);
            const uint4 r1 = *reinterpret_cast<const uint4 *>(lds + OFF_LVL + (4 * j + g) * 8 + 4);
            const float scale = __uint_as_float(r0.x);
            const uint32_t my = r0.y, mz = r0.z, offset = r0.w, mask = r1.x, hashed = r1.y;
            float qx = fma_(ux, scale, 0.5f), qy = fma_(uy, scale, 0.5f), qz = fma_(uz, scale, 0.5f);
            const uint32_t gx = (uint32_t)__builtin_floorf(qx), gy = (uint32_t)__builtin_floorf(qy), gz = (uint32_t)__builtin_floorf(qz);
            q[jj][0] = qx - (float)gx; q[jj][1] = qy - (float)gy; q[jj][2] = qz - (float)gz;
            const uint32_t ax0 = gx, ax1 = gx + 1u, ay0 = gy * my, ay1 = ay0 + my, az0 = gz * mz, az1 = az0 + mz;
            const int mode = jmode[j];                      // wave-uniform: one code path per gather round
            uint32_t idx[8];
#if AC_DENSE_PAIRS
            if (mode == 0) {                                // all four levels of this round are dense: the corners (x, x + 1) are ADJACENT entries -- one
#pragma unroll                                              // 16-byte gather per pair (the per-CU gather path is what the kernel is short of: r04_experiments 7)
                for (int p = 0; p < 4; ++p) {
                    const u32q w = __builtin_amdgcn_raw_buffer_load_b128(table, AC_GOFF((offset + (ax0 + ((p & 1) ? ay1 : ay0) + ((p & 2) ? az1 : az0))) * 8u), 0, 0);
                    v[jj][2 * p] = u32x2{ w.x, w.y }; v[jj][2 * p + 1] = u32x2{ w.z, w.w };
                }
                continue;
            }
#endif
            if (mode == 0) {                                // all four levels of this round are dense
#pragma unroll
                for (int c = 0; c < 8; ++c) idx[c] = ((c & 1) ? ax1 : ax0) + ((c & 2) ? ay1 : ay0) + ((c & 4) ? az1 : az0);
            } else if (mode == 1) {                         // all hashed
#pragma unroll
                for (int c = 0; c < 8; ++c) idx[c] = (((c & 1) ? ax1 : ax0) ^ ((c & 2) ? ay1 : ay0) ^ ((c & 4) ? az1 : az0)) & mask;
            } else {                                        // mixed round (levels 4..7 of the default model)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const uint32_t tx = (c & 1) ? ax1 : ax0, ty = (c & 2) ? ay1 : ay0, tz = (c & 4) ? az1 : az0;
                    idx[c] = (hashed ? (tx ^ ty ^ tz) : (tx + ty + tz)) & mask;
                }
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) v[jj][c] = __builtin_amdgcn_raw_buffer_load_b64(table, AC_GOFF((offset + idx[c]) * 8u), 0, 0);
        }
#pragma unroll
        for (int jj = 0; jj < ROUND; ++jj)
            interp8(v[jj], q[jj][0], q[jj][1], q[jj][2], oob, f[j0 + jj][0], f[j0 + jj][1]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- forward_sdf for a tile of 16 points: returns the 16 outputs as D2^T fragment (o = 4g+r) ----------
struct FieldCtx { rsrc_t table; int jmode[4]; int jfine[4]; float bound, two_bound, inv_tb; };
__device__ __forceinline__ rsrc_t table_of(const FieldCtx &fc) { return fc.table; }

// SDF MLP 35-64-16 on the tile's features (f[j][c] = level 4j+g, channel c; bxyz = this lane group's coordinate),
// split into layer 1 (36 MFMA) and softplus + layer 2 (16 x ~40 VALU + 16 MFMA) so that the caller can overlap
// layer 1 of the NEXT evaluation (matrix pipe) with the softplus of the current one (vector pipe).
struct Acc4 { f32x4 a[4]; };

__device__ __forceinline__ Acc4 sdf_l1(const float *__restrict__ lds, int lane, float bxyz, const float (&f)[4][2])
{
    const int g = lane >> 4;
    Acc4 acc;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc.a[t] = *reinterpret_cast<const f32x4 *>(lds + OFF_B1 + 16 * t + 4 * g);
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        const float b = (s == 0) ? bxyz : f[(s - 1) >> 1][(s - 1) & 1];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            acc.a[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(lds[OFF_W1F + (t * 9 + s) * 64 + lane], b, acc.a[t], 0, 0, 0);
    }
    return acc;
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// layer 1 of one finite-difference evaluation from the centre's layer 1 (acc0, exact fp32):
//   l1(x +- eps e_k) = acc0 + W1[:, features] (fe - fe0) + W1[:, k] (p_off - p)
// the middle product on the bf16 matrix pipe with both factors split hi + lo (3 of the 4 partial products; the dropped lo x lo is
// 2^-16 of a term that is itself ~1e-2 of l1).  The differences are split by truncation: d = hi + (d - hi) exactly, lo = the top 16
// bits of (d - hi).  B operand: this lane's eight differences = k 8g .. 8g+7, the order fill_lds_fast gives the A rows.
// 8 fp32 values -> their bf16 hi parts and the bf16 hi parts of the remainders, packed in B-operand order (value 2q in the low half of dword q)
__device__ __forceinline__ void split8_bf16(const float (&d)[8], u32x4 &bh, u32x4 &bl)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t u0 = __float_as_uint(d[2 * q]), u1 = __float_as_uint(d[2 * q + 1]);
        const float r0 = d[2 * q] - __uint_as_float(u0 & 0xffff0000u), r1 = d[2 * q + 1] - __uint_as_float(u1 & 0xffff0000u);
        bh[q] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);               // (top half of d[2q+1]) << 16 | top half of d[2q]
        bl[q] = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
    }
}

template <int W1H = OFF_W1H, int W1L = OFF_W1L, int W1C = OFF_W1C>
__device__ __forceinline__ Acc4 sdf_l1_delta(const float *__restrict__ lds, int lane, const Acc4 &acc0, const float (&fe)[4][2],
                                             const float (&fe0)[4][2], int kn, float dcoord)
{
    const int g = lane >> 4;
    u32x4 bh, bl;
    float dlt[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) { dlt[2 * q] = fe[q][0] - fe0[q][0]; dlt[2 * q + 1] = fe[q][1] - fe0[q][1]; }
    split8_bf16(dlt, bh, bl);
    const bf16x8 Bh = __builtin_bit_cast(bf16x8, bh), Bl = __builtin_bit_cast(bf16x8, bl);
    Acc4 r;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const bf16x8 Ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(lds + W1H + (t * 64 + lane) * 4));
        const bf16x8 Al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(lds + W1L + (t * 64 + lane) * 4));
        f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bh, acc0.a[t], 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, Bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bl, c, 0, 0, 0);
        const f32x4 wc = *reinterpret_cast<const f32x4 *>(lds + W1C + kn * 64 + 16 * t + 4 * g);     // W1[16t + 4g + r][kn]
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) c[rr] = fma_(dcoord, wc[rr], c[rr]);
        r.a[t] = c;
    }
    return r;
}

#ifndef AC_SP_BATCH
#define AC_SP_BATCH 4        // softplus values per LDS round trip (table rows in flight: 4 x 4 registers)
#endif
__device__ __forceinline__ f32x4 sdf_l2(const float *__restrict__ lds, int lane, const Acc4 &acc)
{
    const int g = lane >> 4;
    f32x4 o2 = *reinterpret_cast<const f32x4 *>(lds + OFF_B2 + 4 * g);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float h[4];
#ifdef AC_ABL_SOFTPLUS
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = acc.a[t][r] * 0.5f;
#else
        const float xin[4] = { acc.a[t][0], acc.a[t][1], acc.a[t][2], acc.a[t][3] };
        dv_softplus100_n<4>(lds + OFF_SPQ, xin, h);
#endif
#pragma unroll
        for (int r = 0; r < 4; ++r)
            o2 = __builtin_amdgcn_mfma_f32_16x16x4f32(lds[OFF_W2F + (4 * t + r) * 64 + lane], h[r], o2, 0, 0, 0);
    }
    return o2;
}

// Layer 2 for an evaluation whose only consumer is the sdf value (the six finite-difference points of a sample): output row 0 as
// a dot product on the vector pipe instead of 16 MFMA that would also produce the 15 unused feature rows (fp32 MFMA and fp32
// VALU share the datapath: 16 x 32 clocks against ~20 x 4).  Every lane sums its own 16 hidden units (16t + 4g + r: t outer, r inner),
// the four lane groups are joined as ((p0 + p1) + (p2 + p3)) + b2[0]; all lanes of a sample receive the same bits.
// oracle/ac_oracle.c: orc_sdf_mlp_sdf.
struct W2Row0 { float w[4][4]; };
__device__ __forceinline__ W2Row0 load_w2_row0(const float *__restrict__ lds, int lane)
{
    W2Row0 r;                                         // fragment (4t + r) holds W2[l & 15][16t + 4(l >> 4) + r]: row 0 sits in lane 16 g
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) r.w[t][q] = lds[OFF_W2F + (4 * t + q) * 64 + (lane & 48)];
    return r;
}
__device__ __forceinline__ float sdf_l2_sdf(const float *__restrict__ lds, const Acc4 &acc, const W2Row0 &w0)
{
    float p = 0.0f;
#pragma unroll
    for (int t0 = 0; t0 < 4; t0 += AC_SP_BATCH / 4) {
        constexpr int NB_ = AC_SP_BATCH;
        float xin[NB_], h[NB_];
#pragma unroll
        for (int i = 0; i < NB_; ++i) xin[i] = acc.a[t0 + (i >> 2)][i & 3];
#ifdef AC_ABL_SOFTPLUS
#pragma unroll
        for (int i = 0; i < NB_; ++i) h[i] = xin[i] * 0.5f;
#else
        dv_softplus100_n<NB_>(lds + OFF_SPQ, xin, h);
#endif
#pragma unroll
        for (int i = 0; i < NB_; ++i) p = fma_(w0.w[t0 + (i >> 2)][i & 3], h[i], p);
    }
    const float other = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, p), (0x10 << 10) | 0x1f));   // lane ^ 16
    // v_permlane32_swap exchanges lanes 32..63 of its first register with lanes 0..31 of its second: with s1 in both, the first ends up
    // holding the lower-half sums in both halves and the second the upper-half sums.  (Inline assembly: the builtin of this compiler
    // returns the first register for both results.  The s_nop covers the VALU write -> permlane read wait states.)
    int lo = __builtin_bit_cast(int, p + other), hi = lo;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
    return (__builtin_bit_cast(float, lo) + __builtin_bit_cast(float, hi)) + lds[OFF_B2];
}

__device__ __forceinline__ f32x4 sdf_mlp(const float *__restrict__ lds, int lane, float bxyz, const float (&f)[4][2])
{
    const Acc4 acc = sdf_l1(lds, lane, bxyz, f);
    __builtin_amdgcn_sched_barrier(0);
    return sdf_l2(lds, lane, acc);
}

__device__ __forceinline__ f32x4 sdf_tile(const float *__restrict__ lds, const FieldCtx &fc, int lane, float px, float py, float pz)
{
    const int g = lane >> 4;
    float f[4][2];
    encode4<AC_ENC_ROUND>(lds, fc.table, g, fc.jmode, px, py, pz, fc.bound, fc.two_bound, f, fc.inv_tb);
    __builtin_amdgcn_sched_barrier(0);
    return sdf_mlp(lds, lane, sel4(g, px, py, pz, 0.0f), f);
}

// ---- finite-difference stencil: hash features of the centre point and of p +- eps*e_k (k = x,y,z) ------------------
// The 7 evaluations of finite_difference_normals_approximator (instant_nsr.py:687-704) share most grid corners
// on the coarse levels (eps*scale < 1 cell up to level 11).  Every evaluation still computes ITS OWN position,
// cell and interpolation weights exactly as a stand-alone evaluation would (bit-identical features); only the
// memory fetches are shared: a face of the offset point's cell that coincides with a face of the centre cell
// re-uses the centre's 4 corner values, other faces are gathered under an exec mask.
// fe[e][j][c]: e = 0 centre, 1..6 = +x,-x,+y,-y,+z,-z ; pe[e] = the offset coordinate (clamped) of evaluation e.
struct LvlC { float scale; uint32_t my, mz, offset, mask, hashed; };

// GM (wave-uniform, RenderArgs::jmode of the level group): 1 = all four levels of the group are hashed -- xor only, which the compiler folds with the mask
// into one three-input bit operation per corner; anything else = the per-lane choice between the dense and the hashed index
template <int GM>
__device__ __forceinline__ uint32_t gidx(const LvlC &L, uint32_t tx, uint32_t ty, uint32_t tz)
{
    if constexpr (GM == 1) return (tx ^ ty ^ tz) & L.mask;
    else if constexpr (GM == 0) return tx + ty + tz;                       // a dense round (every lane's level): index < level size by construction
    else return (L.hashed ? (tx ^ ty ^ tz) : (tx + ty + tz)) & L.mask;
}
// corner index (0..7) of the i-th corner (i = 0..3, other two axes in increasing order) on face b of axis K
template <int K> __device__ __forceinline__ constexpr int face_corner(int b, int i)
{
    return K == 0 ? (b | (i << 1)) : (K == 1 ? ((i & 1) | (b << 1) | ((i >> 1) << 2)) : (i | (b << 2)));
}

// ---- coarse level (eps*scale < 1 cell): an offset point lies in the centre cell or in the adjacent one, so it needs at
// most ONE face the centre does not have (coordinate g+2 for +eps, g-1 for -eps).  All 8 + 6*4 gathers of the level
// are issued as one batch (lanes that need nothing send an out-of-range offset: dropped by the descriptor's bounds
// check), then the 7 interpolations run from registers: one memory round trip per level instead of seven.  (As compiled, the compiler folds
// coarse_finish's select into the offset points' loads -- register pre-set to the centre's corner, load under an exec mask, skipped when no lane of the
// wave needs it -- so they leave after the centre's have returned: two round trips.  Forcing one batch was measured and changes nothing, the other wave
// of the SIMD covers the second trip: profiles/r02_experiments.txt.)
template <int K, int SIGN>   // SIGN 0: +eps, 1: -eps
struct AxisGeo { float qk; bool need, oob; };

template <int K, int SIGN, int GM>
__device__ __forceinline__ AxisGeo<K, SIGN> coarse_issue(rsrc_t table, const LvlC &L, const uint32_t (&gc)[3], const uint32_t (&tx)[2],
                                                       const uint32_t (&ty)[2], const uint32_t (&tz)[2], bool oob_c, float u,
                                                       u32x2 (&w)[4])
{
    AxisGeo<K, SIGN> a;
    a.oob = oob_c | (u < 0.0f) | (u > 1.0f);
    const float pos = fma_(u, L.scale, 0.5f);
    const uint32_t gk = (uint32_t)__builtin_floorf(pos);
    a.qk = pos - (float)gk;
    a.need = gk != gc[K];                                   // shifted by exactly one cell (host guarantees |shift| <= 1)
    const uint32_t mk = K == 0 ? 1u : (K == 1 ? L.my : L.mz);
    const uint32_t base = K == 0 ? tx[0] : (K == 1 ? ty[0] : tz[0]);
    const uint32_t tk = SIGN == 0 ? base + 2u * mk : base - mk;   // coordinate g+2 / g-1 along K
    if constexpr (GM == 0 && K != 0 && !AC_SENTINEL_LOADS) {      // dense round, a y- or z-face: its corners come in x-adjacent pairs
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int c = face_corner<K>(0, 2 * pp);
            const uint32_t ay = K == 1 ? tk : ty[(c >> 1) & 1], az = K == 2 ? tk : tz[(c >> 2) & 1];
            w[2 * pp] = u32x2{ 0u, 0u }; w[2 * pp + 1] = u32x2{ 0u, 0u };
            if (a.need) {
                const u32q q = __builtin_amdgcn_raw_buffer_load_b128(table, AC_GOFF((L.offset + (tx[0] + ay + az)) * 8u), 0, 0);
                w[2 * pp] = u32x2{ q.x, q.y }; w[2 * pp + 1] = u32x2{ q.z, q.w };
            }
        }
        return a;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = face_corner<K>(0, i);
        const uint32_t ax = K == 0 ? tk : tx[c & 1], ay = K == 1 ? tk : ty[(c >> 1) & 1], az = K == 2 ? tk : tz[(c >> 2) & 1];
#if AC_SENTINEL_LOADS
        const uint32_t off = a.need ? (L.offset + gidx<GM>(L, ax, ay, az)) * 8u : 0xfffffff8u;
        w[i] = __builtin_amdgcn_raw_buffer_load_b64(table, AC_GOFF(off), 0, 0);
#else
        w[i] = u32x2{ 0u, 0u };
        if (a.need) w[i] = __builtin_amdgcn_raw_buffer_load_b64(table, AC_GOFF((L.offset + gidx<GM>(L, ax, ay, az)) * 8u), 0, 0);
#endif
    }
    return a;
}

template <int K, int SIGN>
__device__ __forceinline__ void coarse_finish(const AxisGeo<K, SIGN> &a, const u32x2 (&vc)[8], const u32x2 (&w)[4], const float (&qc)[3],
                                              float &f0, float &f1)
{
    constexpr int NEWBIT = SIGN == 0 ? 1 : 0;               // +eps: the new face is face 1 of the shifted cell
    u32x2 v2[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int bk = (c >> K) & 1;
        const int i = K == 0 ? (c >> 1) : (K == 1 ? ((c & 1) | ((c >> 2) << 1)) : (c & 3));
        if (bk == NEWBIT) { v2[c].x = a.need ? w[i].x : vc[c].x; v2[c].y = a.need ? w[i].y : vc[c].y; }
        else { v2[c].x = a.need ? vc[c ^ (1 << K)].x : vc[c].x; v2[c].y = a.need ? vc[c ^ (1 << K)].y : vc[c].y; }
    }
    interp8(v2, K == 0 ? a.qk : qc[0], K == 1 ? a.qk : qc[1], K == 2 ? a.qk : qc[2], a.oob, f0, f1);
}

// ---- fast precision only (AC_FACE_VALUE): an offset point differs from the centre along ONE axis, so its feature is the linear
// interpolation, along that axis, of two bilinear FACE values -- (A, B) of the centre cell, (B, N) one cell up, (N, A) one cell down, N = the
// gathered face -- with the centre's weights for the other two axes: 4 + 3 x (4 + 4 x 8) + 6 x 11 vector instructions per level instead of
// 7 full 8-corner interpolations and their corner selects.  Same function, different rounding (not the oracle's order): the offset features
// only feed the split-bf16 correction of layer 1, whose own error is 2^-16 of the difference.
template <int K> __device__ __forceinline__ void face_weights(const float (&qc)[3], float (&fw)[4])
{
    const float qa = K == 0 ? qc[1] : qc[0], qb = K == 2 ? qc[1] : qc[2];        // the two other axes, in increasing order
    const float a0 = 1.0f - qa, b0 = 1.0f - qb;
    fw[0] = a0 * b0; fw[1] = qa * b0; fw[2] = a0 * qb; fw[3] = qa * qb;
}
template <int K> __device__ __forceinline__ void face_value(const u32x2 (&vc)[8], int b, const float (&fw)[4], float (&f)[2])
{
    float s0 = fw[0] * __uint_as_float(vc[face_corner<K>(b, 0)].x), s1 = fw[0] * __uint_as_float(vc[face_corner<K>(b, 0)].y);
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        s0 = fma_(fw[i], __uint_as_float(vc[face_corner<K>(b, i)].x), s0);
        s1 = fma_(fw[i], __uint_as_float(vc[face_corner<K>(b, i)].y), s1);
    }
    f[0] = s0; f[1] = s1;
}
template <int K, int SIGN>
__device__ __forceinline__ void coarse_finish_fv(const AxisGeo<K, SIGN> &a, const float (&A)[2], const float (&B)[2], const u32x2 (&w)[4],
                                                 const float (&fw)[4], float &f0, float &f1)
{
    float n0 = fw[0] * __uint_as_float(w[0].x), n1 = fw[0] * __uint_as_float(w[0].y);
#pragma unroll
    for (int i = 1; i < 4; ++i) { n0 = fma_(fw[i], __uint_as_float(w[i].x), n0); n1 = fma_(fw[i], __uint_as_float(w[i].y), n1); }
    const float x0 = a.need ? (SIGN == 0 ? B[0] : n0) : A[0], x1 = a.need ? (SIGN == 0 ? B[1] : n1) : A[1];
    const float y0 = a.need ? (SIGN == 0 ? n0 : A[0]) : B[0], y1 = a.need ? (SIGN == 0 ? n1 : A[1]) : B[1];
    const float r0 = fma_(a.qk, y0 - x0, x0), r1 = fma_(a.qk, y1 - x1, x1);
    f0 = a.oob ? 0.0f : r0;
    f1 = a.oob ? 0.0f : r1;
}

// ---- fine level (eps spans one cell or more): every offset point gathers its own 8 corners ---------------------------
template <int K, int GM>
__device__ __forceinline__ void fine_issue(rsrc_t table, const LvlC &L, const uint32_t (&tx)[2], const uint32_t (&ty)[2],
                                           const uint32_t (&tz)[2], bool oob_c, float u, u32x2 (&v)[8], float &qk, bool &oob)
{
    oob = oob_c | (u < 0.0f) | (u > 1.0f);
    const float pos = fma_(u, L.scale, 0.5f);
    const uint32_t gk = (uint32_t)__builtin_floorf(pos);
    qk = pos - (float)gk;
    const uint32_t mk = K == 0 ? 1u : (K == 1 ? L.my : L.mz);
    const uint32_t t0 = gk * mk, t1 = t0 + mk;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint32_t tk = ((c >> K) & 1) ? t1 : t0;
        const uint32_t ax = K == 0 ? tk : tx[c & 1], ay = K == 1 ? tk : ty[(c >> 1) & 1], az = K == 2 ? tk : tz[(c >> 2) & 1];
        v[c] = __builtin_amdgcn_raw_buffer_load_b64(table, AC_GOFF((L.offset + gidx<GM>(L, ax, ay, az)) * 8u), 0, 0);
    }
}

#define AC_FSTORE(E, F0, F1) { fslab[((E - 1) * 8 + 2 * j) * 64 + lane] = F0; fslab[((E - 1) * 8 + 2 * j + 1) * 64 + lane] = F1; }

#ifndef AC_STENCIL_SPECIALIZE
#define AC_STENCIL_SPECIALIZE 1      // a second copy of the stencil code for level groups that are hashed throughout (xor-only index arithmetic)
#endif
// one group of four levels (4j + g) of the stencil: centre features in c0 / c1, the six offset points' features to the slab
template <int GM, int FV = 0>
__device__ __forceinline__ void stencil_levels(const float *__restrict__ lds, float *__restrict__ fslab, rsrc_t table, int lane, int g, int j, bool fine,
                                               float ux, float uy, float uz, bool oob, float xp, float xm, float yp, float ym, float zp, float zm,
                                               float &c0, float &c1)
{
    const uint4 r0 = *reinterpret_cast<const uint4 *>(lds + OFF_LVL + (4 * j + g) * 8);
    const uint4 r1 = *reinterpret_cast<const uint4 *>(lds + OFF_LVL + (4 * j + g) * 8 + 4);
    LvlC L; L.scale = __uint_as_float(r0.x); L.my = r0.y; L.mz = r0.z; L.offset = r0.w; L.mask = r1.x; L.hashed = r1.y;
    float qc[3]; uint32_t gc[3];
    {
        const float qx = fma_(ux, L.scale, 0.5f), qy = fma_(uy, L.scale, 0.5f), qz = fma_(uz, L.scale, 0.5f);
        gc[0] = (uint32_t)__builtin_floorf(qx); gc[1] = (uint32_t)__builtin_floorf(qy); gc[2] = (uint32_t)__builtin_floorf(qz);
        qc[0] = qx - (float)gc[0]; qc[1] = qy - (float)gc[1]; qc[2] = qz - (float)gc[2];
    }
    uint32_t tx[2], ty[2], tz[2];
    tx[0] = gc[0]; tx[1] = gc[0] + 1u; ty[0] = gc[1] * L.my; ty[1] = ty[0] + L.my; tz[0] = gc[2] * L.mz; tz[1] = tz[0] + L.mz;
    u32x2 vc[8];
    if constexpr (GM == 0) {                                // dense round: x-adjacent corners through one 16-byte gather each
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const u32q w = __builtin_amdgcn_raw_buffer_load_b128(table, AC_GOFF((L.offset + (tx[0] + ty[p & 1] + tz[p >> 1])) * 8u), 0, 0);
            vc[2 * p] = u32x2{ w.x, w.y }; vc[2 * p + 1] = u32x2{ w.z, w.w };
        }
    } else {
#pragma unroll
    for (int c = 0; c < 8; ++c)
        vc[c] = __builtin_amdgcn_raw_buffer_load_b64(table, AC_GOFF((L.offset + gidx<GM>(L, tx[c & 1], ty[(c >> 1) & 1], tz[c >> 2])) * 8u), 0, 0);
    }
    if (GM == 0 || !fine) {
        u32x2 w0[4], w1[4], w2[4], w3[4], w4[4], w5[4];
        const auto a0 = coarse_issue<0, 0, GM>(table, L, gc, tx, ty, tz, oob, xp, w0);
        const auto a1 = coarse_issue<0, 1, GM>(table, L, gc, tx, ty, tz, oob, xm, w1);
        const auto a2 = coarse_issue<1, 0, GM>(table, L, gc, tx, ty, tz, oob, yp, w2);
        const auto a3 = coarse_issue<1, 1, GM>(table, L, gc, tx, ty, tz, oob, ym, w3);
        const auto a4 = coarse_issue<2, 0, GM>(table, L, gc, tx, ty, tz, oob, zp, w4);
        const auto a5 = coarse_issue<2, 1, GM>(table, L, gc, tx, ty, tz, oob, zm, w5);
        __builtin_amdgcn_sched_barrier(0);
        interp8(vc, qc[0], qc[1], qc[2], oob, c0, c1);
        float f0, f1;
        if constexpr (FV != 0) {
            float fw[4], A[2], B[2];
            if constexpr ((AC_FV_AXES & 1) != 0) {
                face_weights<0>(qc, fw); face_value<0>(vc, 0, fw, A); face_value<0>(vc, 1, fw, B);
                coarse_finish_fv<0, 0>(a0, A, B, w0, fw, f0, f1); AC_FSTORE(1, f0, f1)
                coarse_finish_fv<0, 1>(a1, A, B, w1, fw, f0, f1); AC_FSTORE(2, f0, f1)
            } else {
                coarse_finish<0, 0>(a0, vc, w0, qc, f0, f1); AC_FSTORE(1, f0, f1)
                coarse_finish<0, 1>(a1, vc, w1, qc, f0, f1); AC_FSTORE(2, f0, f1)
            }
            if constexpr ((AC_FV_AXES & 2) != 0) {
                face_weights<1>(qc, fw); face_value<1>(vc, 0, fw, A); face_value<1>(vc, 1, fw, B);
                coarse_finish_fv<1, 0>(a2, A, B, w2, fw, f0, f1); AC_FSTORE(3, f0, f1)
                coarse_finish_fv<1, 1>(a3, A, B, w3, fw, f0, f1); AC_FSTORE(4, f0, f1)
            } else {
                coarse_finish<1, 0>(a2, vc, w2, qc, f0, f1); AC_FSTORE(3, f0, f1)
                coarse_finish<1, 1>(a3, vc, w3, qc, f0, f1); AC_FSTORE(4, f0, f1)
            }
            if constexpr ((AC_FV_AXES & 4) != 0) {
#ifdef AC_FV_DEBUG_NOP
                asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
#endif
                face_weights<2>(qc, fw); face_value<2>(vc, 0, fw, A); face_value<2>(vc, 1, fw, B);
                if constexpr ((AC_FV_SIGNS & 1) != 0) { coarse_finish_fv<2, 0>(a4, A, B, w4, fw, f0, f1); } else { coarse_finish<2, 0>(a4, vc, w4, qc, f0, f1); }
                AC_FSTORE(5, f0, f1)
                if constexpr ((AC_FV_SIGNS & 2) != 0) { coarse_finish_fv<2, 1>(a5, A, B, w5, fw, f0, f1); } else { coarse_finish<2, 1>(a5, vc, w5, qc, f0, f1); }
                AC_FSTORE(6, f0, f1)
            } else {
                coarse_finish<2, 0>(a4, vc, w4, qc, f0, f1); AC_FSTORE(5, f0, f1)
                coarse_finish<2, 1>(a5, vc, w5, qc, f0, f1); AC_FSTORE(6, f0, f1)
            }
        } else {
        coarse_finish<0, 0>(a0, vc, w0, qc, f0, f1); AC_FSTORE(1, f0, f1)
        coarse_finish<0, 1>(a1, vc, w1, qc, f0, f1); AC_FSTORE(2, f0, f1)
        coarse_finish<1, 0>(a2, vc, w2, qc, f0, f1); AC_FSTORE(3, f0, f1)
        coarse_finish<1, 1>(a3, vc, w3, qc, f0, f1); AC_FSTORE(4, f0, f1)
        coarse_finish<2, 0>(a4, vc, w4, qc, f0, f1); AC_FSTORE(5, f0, f1)
        coarse_finish<2, 1>(a5, vc, w5, qc, f0, f1); AC_FSTORE(6, f0, f1)
        }
    } else {
#if AC_FINE_BATCH == 3     // all 48 gathers of the six offset points in flight at once: one memory round trip instead of three
        u32x2 va[8], vb[8], vc2[8], vd[8], ve[8], vf[8];
        float qa, qb, qc2, qd, qe, qf, f0, f1; bool oa, ob, oc2, od, oe, of;
        fine_issue<0, GM>(table, L, tx, ty, tz, oob, xp, va, qa, oa);
        fine_issue<0, GM>(table, L, tx, ty, tz, oob, xm, vb, qb, ob);
        fine_issue<1, GM>(table, L, tx, ty, tz, oob, yp, vc2, qc2, oc2);
        fine_issue<1, GM>(table, L, tx, ty, tz, oob, ym, vd, qd, od);
        fine_issue<2, GM>(table, L, tx, ty, tz, oob, zp, ve, qe, oe);
        fine_issue<2, GM>(table, L, tx, ty, tz, oob, zm, vf, qf, of);
        __builtin_amdgcn_sched_barrier(0);
        interp8(vc, qc[0], qc[1], qc[2], oob, c0, c1);
        interp8(va, qa, qc[1], qc[2], oa, f0, f1); AC_FSTORE(1, f0, f1)
        interp8(vb, qb, qc[1], qc[2], ob, f0, f1); AC_FSTORE(2, f0, f1)
        interp8(vc2, qc[0], qc2, qc[2], oc2, f0, f1); AC_FSTORE(3, f0, f1)
        interp8(vd, qc[0], qd, qc[2], od, f0, f1); AC_FSTORE(4, f0, f1)
        interp8(ve, qc[0], qc[1], qe, oe, f0, f1); AC_FSTORE(5, f0, f1)
        interp8(vf, qc[0], qc[1], qf, of, f0, f1); AC_FSTORE(6, f0, f1)
#elif AC_FINE_BATCH == 2   // x and y offsets in one round trip, z offsets in a second one
        u32x2 va[8], vb[8], vc2[8], vd[8];
        float qa, qb, qc2, qd, f0, f1; bool oa, ob, oc2, od;
        fine_issue<0, GM>(table, L, tx, ty, tz, oob, xp, va, qa, oa);
        fine_issue<0, GM>(table, L, tx, ty, tz, oob, xm, vb, qb, ob);
        fine_issue<1, GM>(table, L, tx, ty, tz, oob, yp, vc2, qc2, oc2);
        fine_issue<1, GM>(table, L, tx, ty, tz, oob, ym, vd, qd, od);
        __builtin_amdgcn_sched_barrier(0);
        interp8(vc, qc[0], qc[1], qc[2], oob, c0, c1);
        interp8(va, qa, qc[1], qc[2], oa, f0, f1); AC_FSTORE(1, f0, f1)
        interp8(vb, qb, qc[1], qc[2], ob, f0, f1); AC_FSTORE(2, f0, f1)
        __builtin_amdgcn_sched_barrier(0);
        fine_issue<2, GM>(table, L, tx, ty, tz, oob, zp, va, qa, oa);
        fine_issue<2, GM>(table, L, tx, ty, tz, oob, zm, vb, qb, ob);
        __builtin_amdgcn_sched_barrier(0);
        interp8(vc2, qc[0], qc2, qc[2], oc2, f0, f1); AC_FSTORE(3, f0, f1)
        interp8(vd, qc[0], qd, qc[2], od, f0, f1); AC_FSTORE(4, f0, f1)
        __builtin_amdgcn_sched_barrier(0);
        interp8(va, qc[0], qc[1], qa, oa, f0, f1); AC_FSTORE(5, f0, f1)
        interp8(vb, qc[0], qc[1], qb, ob, f0, f1); AC_FSTORE(6, f0, f1)
#else
        u32x2 va[8], vb[8];
        float qa, qb, f0, f1; bool oa, ob;
        fine_issue<0, GM>(table, L, tx, ty, tz, oob, xp, va, qa, oa);
        fine_issue<0, GM>(table, L, tx, ty, tz, oob, xm, vb, qb, ob);
        __builtin_amdgcn_sched_barrier(0);
        interp8(vc, qc[0], qc[1], qc[2], oob, c0, c1);
        interp8(va, qa, qc[1], qc[2], oa, f0, f1); AC_FSTORE(1, f0, f1)
        interp8(vb, qb, qc[1], qc[2], ob, f0, f1); AC_FSTORE(2, f0, f1)
        __builtin_amdgcn_sched_barrier(0);
        fine_issue<1, GM>(table, L, tx, ty, tz, oob, yp, va, qa, oa);
        fine_issue<1, GM>(table, L, tx, ty, tz, oob, ym, vb, qb, ob);
        __builtin_amdgcn_sched_barrier(0);
        interp8(va, qc[0], qa, qc[2], oa, f0, f1); AC_FSTORE(3, f0, f1)
        interp8(vb, qc[0], qb, qc[2], ob, f0, f1); AC_FSTORE(4, f0, f1)
        __builtin_amdgcn_sched_barrier(0);
        fine_issue<2, GM>(table, L, tx, ty, tz, oob, zp, va, qa, oa);
        fine_issue<2, GM>(table, L, tx, ty, tz, oob, zm, vb, qb, ob);
        __builtin_amdgcn_sched_barrier(0);
        interp8(va, qc[0], qc[1], qa, oa, f0, f1); AC_FSTORE(5, f0, f1)
        interp8(vb, qc[0], qc[1], qb, ob, f0, f1); AC_FSTORE(6, f0, f1)
#endif
    }
}

template <int FV = 0>
__device__ __forceinline__ void encode_stencil(const float *__restrict__ lds, float *__restrict__ fslab, const FieldCtx &fc, int lane,
                                               float px, float py, float pz, float eps, float (&fe0)[4][2])
{
    const int g = lane >> 4;
    const rsrc_t table = fc.table;
    const float bound = fc.bound, two_bound = fc.two_bound, inv_tb = fc.inv_tb;
    // normalised coordinates of the centre and of the six offset points (one quotient each, hoisted out of the level loop): unit_div for a verified divisor
    const float ax = px + bound, ay = py + bound, az = pz + bound;
    const float axp = clampf(px + eps, -bound, bound) + bound, axm = clampf(px + (-eps), -bound, bound) + bound;
    const float ayp = clampf(py + eps, -bound, bound) + bound, aym = clampf(py + (-eps), -bound, bound) + bound;
    const float azp = clampf(pz + eps, -bound, bound) + bound, azm = clampf(pz + (-eps), -bound, bound) + bound;
    float ux, uy, uz, xp, xm, yp, ym, zp, zm;
    if (inv_tb != 0.0f) {                                   // wave-uniform
        ux = unit_div(ax, two_bound, inv_tb); uy = unit_div(ay, two_bound, inv_tb); uz = unit_div(az, two_bound, inv_tb);
        xp = unit_div(axp, two_bound, inv_tb); xm = unit_div(axm, two_bound, inv_tb); yp = unit_div(ayp, two_bound, inv_tb);
        ym = unit_div(aym, two_bound, inv_tb); zp = unit_div(azp, two_bound, inv_tb); zm = unit_div(azm, two_bound, inv_tb);
    } else {
        ux = ax / two_bound; uy = ay / two_bound; uz = az / two_bound;
        xp = axp / two_bound; xm = axm / two_bound; yp = ayp / two_bound; ym = aym / two_bound; zp = azp / two_bound; zm = azm / two_bound;
    }
    const bool oob = (ux < 0.0f) | (ux > 1.0f) | (uy < 0.0f) | (uy > 1.0f) | (uz < 0.0f) | (uz > 1.0f);
    // jmode (2 bits each) and jfine (1 bit each) of the four level groups in one scalar register: indexed by the loop counter as arrays they
    // would live in scratch memory, and the load of jfine[j] would sit between the centre's gathers and the offset points' (one more round trip)
    const uint32_t jbits = (uint32_t)fc.jmode[0] | ((uint32_t)fc.jmode[1] << 2) | ((uint32_t)fc.jmode[2] << 4) | ((uint32_t)fc.jmode[3] << 6)
                         | ((fc.jfine[0] ? 1u : 0u) << 8) | ((fc.jfine[1] ? 1u : 0u) << 9) | ((fc.jfine[2] ? 1u : 0u) << 10) | ((fc.jfine[3] ? 1u : 0u) << 11);
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {                           // one copy of each code path; results go to LDS / a rotating fe0
        float c0, c1;
        const bool fine = (jbits >> (8 + j)) & 1u, hashed4 = ((jbits >> (2 * j)) & 3u) == 1u, dense4 = ((jbits >> (2 * j)) & 3u) == 0u;
        if (AC_STENCIL_SPECIALIZE && hashed4) stencil_levels<1, FV>(lds, fslab, table, lane, g, j, fine, ux, uy, uz, oob, xp, xm, yp, ym, zp, zm, c0, c1);
        else if (AC_DENSE_PAIRS && dense4 && !fine) stencil_levels<0, FV>(lds, fslab, table, lane, g, j, false, ux, uy, uz, oob, xp, xm, yp, ym, zp, zm, c0, c1);
        else stencil_levels<2, FV>(lds, fslab, table, lane, g, j, fine, ux, uy, uz, oob, xp, xm, yp, ym, zp, zm, c0, c1);
        // rotate the centre features into place: after the 4th iteration fe0[j] holds level 4j+g
        fe0[0][0] = fe0[1][0]; fe0[0][1] = fe0[1][1]; fe0[1][0] = fe0[2][0]; fe0[1][1] = fe0[2][1];
        fe0[2][0] = fe0[3][0]; fe0[2][1] = fe0[3][1]; fe0[3][0] = c0; fe0[3][1] = c1;
        __builtin_amdgcn_sched_barrier(0);
    }
}
#undef AC_FSTORE

// ---- forward_color for a tile: rgb (post-sigmoid) valid in lanes g==0 ---------------------------------
// ---- use_viewdirs = True (models/instant_nsr.py:565-569, 644-653): h = cat[x, sh(d), n, geo_feat] ----------------------------------------
// The 16 spherical harmonics (degree 4, encoder/shencoder) of the RAW ray direction enter layer 1 of the colour network only.  The direction is constant
// along a ray, so Wsh sh(d) is a per-ray BIAS of that layer: bias[u] = fma chain over j = 0..15 of Wsh[u][j] * sh_j(d), and the accumulator of unit u
// starts from bias[u] instead of 0 -- zero cost per sample.  sh_j: the value table of the stand-alone encoder (shencoder.hip), same operations.
// one ray per wave: shb [80] floats of the wave's LDS slab -> shb[u] = bias of unit u (u = lane); KEEP_SH: shb[64 + j] = sh_j(d) as well (ac_sh_bias)
template <bool KEEP_SH = false>
__device__ __forceinline__ void ray_sh_bias(float *__restrict__ shb, const float *__restrict__ Wsh, float dx, float dy, float dz, int lane)
{
    float sh[16];
    ac_sh16(dx, dy, dz, sh);                                  // (ac_sh16.hpp: the stand-alone encoder's operations, written out)
    const f32x4 *w = reinterpret_cast<const f32x4 *>(Wsh + lane * 16);
    float acc = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 wv = w[q];
        acc = fma_(wv[0], sh[4 * q], acc); acc = fma_(wv[1], sh[4 * q + 1], acc);
        acc = fma_(wv[2], sh[4 * q + 2], acc); acc = fma_(wv[3], sh[4 * q + 3], acc);
    }
    wave_sync();                                              // (the previous work item's readers of the slab are done)
    shb[lane] = acc;
    if constexpr (KEEP_SH) {
#pragma unroll
        for (int j = 0; j < 16; ++j) if (lane == j) shb[64 + j] = sh[j];
    }
    wave_sync();
}
// packed samples (a tile of 16 samples of whatever rays): lane (n, g) computes the 16 biases it needs -- units 16 t + 4 g + r of ITS sample's direction --
// into slab[(t * 64 + lane) * 4 + r] ([4][64][4] floats of the wave's LDS: the finite-difference feature slab, free once the normals are formed)
__device__ __forceinline__ void sample_sh_bias(float *__restrict__ slab, const float *__restrict__ Wsh, float dx, float dy, float dz, int lane)
{
    const int g = lane >> 4;
    float sh[16];
    ac_sh16(dx, dy, dz, sh);
#pragma unroll 1
    for (int t = 0; t < 4; ++t) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x4 *w = reinterpret_cast<const f32x4 *>(Wsh + (16 * t + 4 * g + r) * 16);
            float acc = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 wv = w[q];
                acc = fma_(wv[0], sh[4 * q], acc); acc = fma_(wv[1], sh[4 * q + 1], acc); acc = fma_(wv[2], sh[4 * q + 2], acc); acc = fma_(wv[3], sh[4 * q + 3], acc);
            }
            o[r] = acc;
        }
        *reinterpret_cast<f32x4 *>(slab + (t * 64 + lane) * 4) = o;
    }
    wave_sync();
}

#ifndef AC_SH_VALUE_SELECT
#define AC_SH_VALUE_SELECT 1
#endif
// The view-direction bias of layer 1 enters unit u's fma chain at ONE fixed position: after the three coordinates, in the k = 3 slot of the MFMA that
// carries (x, y, z, -) -- a slot that multiplies 0 by 0 without view directions.  acc = fma(bias[u], 1, acc) there, in both forms:
//   shb1 (the renderer: one ray per wave): the ray's bias row [64] in the wave's slab becomes that slot's A operand on the lanes of group 3 (B = 1 there):
//        no extra instruction but a select -- the renderer's tile loop has neither registers nor issue slots to spare;
//   shb  (packed samples, a direction per sample): this lane's bias quadruples -- (sample slab) + 4 lane, tstride 256 -- added to the accumulator right
//        after that MFMA (acc + bias, one rounding: the same bits as the slot's fma).
__device__ __forceinline__ void color_tile(const float *__restrict__ lds, int lane, float px, float py, float pz,
                                           float nx, float ny, float nz, f32x4 sdfout, float (&rgb)[3], const float *__restrict__ shb = nullptr,
                                           int tstride = 16, const float *__restrict__ shb1 = nullptr)
{
    const int g = lane >> 4;
    f32x4 h1[4], h2[4];
    const float bxyz = sel4(g, px, py, pz, shb1 ? 1.0f : 0.0f), bn = sel4(g, nx, ny, nz, 0.0f);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const float b = s < 4 ? sdfout[s] : (s == 4 ? bxyz : bn);
            float wv = lds[OFF_C1F + (t * 6 + s) * 64 + lane];
#if AC_SH_VALUE_SELECT
            // the ray's bias row for the lanes of group 3 as a second, plainly addressed LDS read and a select of the VALUE (round 6: the address select this
            // replaces kept a per-lane pointer alive across the tile, which the register allocator answered with a scratch reload inside this block)
            if (s == 4 && shb1) { const float wb = shb1[16 * t + (lane & 15)]; wv = g == 3 ? wb : wv; }
#else
            if (s == 4 && shb1) wv = *(g == 3 ? shb1 + 16 * t + (lane & 15) : lds + OFF_C1F + (t * 6 + s) * 64 + lane);      // (an address select: still one LDS read per lane)
#endif
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, b, acc, 0, 0, 0);
            if (s == 4 && shb) {
                const f32x4 bq = *reinterpret_cast<const f32x4 *>(shb + t * tstride);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = acc[r] + bq[r];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = acc[r] > 0.0f ? acc[r] : 0.0f;
        h1[t] = acc;
    }
#pragma unroll
    for (int to = 0; to < 4; ++to) {
        f32x4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(lds[OFF_C2F + (to * 16 + kk) * 64 + lane], h1[kk >> 2][kk & 3], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = acc[r] > 0.0f ? acc[r] : 0.0f;
        h2[to] = acc;
    }
    f32x4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(lds[OFF_C3F + kk * 64 + lane], h2[kk >> 2][kk & 3], acc, 0, 0, 0);
    rgb[0] = dv_sigmoid(acc[0]); rgb[1] = dv_sigmoid(acc[1]); rgb[2] = dv_sigmoid(acc[2]);
}

// the same network in split bf16 (fast precision; fragments of fill_lds_color_fast): 42 MFMA of 16 clocks instead of 104 of 32
__device__ __forceinline__ f32x4 cf_mma(const float *__restrict__ lds, int f, int lane, const u32x4 &bh, const u32x4 &bl, f32x4 acc)
{
    const bf16x8 Ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(lds + OFF_CFH + (f * 64 + lane) * 4));
    const bf16x8 Al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(lds + (f < 12 ? OFF_CFL + f * CF_FRAG : OFF_C3L + (f - 12) * CF_FRAG) + lane * 4));
    const bf16x8 Bh = __builtin_bit_cast(bf16x8, bh), Bl = __builtin_bit_cast(bf16x8, bl);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, Bh, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bl, acc, 0, 0, 0);
}
__device__ __forceinline__ void color_tile_fast(const float *__restrict__ lds, int lane, float px, float py, float pz,
                                                float nx, float ny, float nz, f32x4 sdfout, float (&rgb)[3], const float *__restrict__ shb = nullptr,
                                                int tstride = 16, const float *__restrict__ shb1 = nullptr)
{
    const int g = lane >> 4;
    u32x4 bh, bl;
    {
        const float in[8] = { sdfout[0], sdfout[1], sdfout[2], sdfout[3], g == 0 ? px : (g == 1 ? nx : 0.0f), g == 0 ? py : (g == 1 ? ny : 0.0f),
                              g == 0 ? pz : (g == 1 ? nz : 0.0f), 0.0f };
        split8_bf16(in, bh, bl);
    }
    f32x4 h1[4], h2[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x4 acc0 = { 0.0f, 0.0f, 0.0f, 0.0f };
        if (shb) acc0 = *reinterpret_cast<const f32x4 *>(shb + t * tstride);      // (the view-direction bias stays fp32 in either precision)
        if (shb1) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(shb1[16 * t + (lane & 15)], g == 0 ? 1.0f : 0.0f, acc0, 0, 0, 0);
        f32x4 acc = cf_mma(lds, t, lane, bh, bl, acc0);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = acc[r] > 0.0f ? acc[r] : 0.0f;
        h1[t] = acc;
    }
    u32x4 ch[2], cl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float in[8] = { h1[2 * s][0], h1[2 * s][1], h1[2 * s][2], h1[2 * s][3], h1[2 * s + 1][0], h1[2 * s + 1][1], h1[2 * s + 1][2], h1[2 * s + 1][3] };
        split8_bf16(in, ch[s], cl[s]);
    }
#pragma unroll
    for (int to = 0; to < 4; ++to) {
        f32x4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
        for (int s = 0; s < 2; ++s) acc = cf_mma(lds, 4 + 2 * to + s, lane, ch[s], cl[s], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = acc[r] > 0.0f ? acc[r] : 0.0f;
        h2[to] = acc;
    }
    f32x4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const float in[8] = { h2[2 * s][0], h2[2 * s][1], h2[2 * s][2], h2[2 * s][3], h2[2 * s + 1][0], h2[2 * s + 1][1], h2[2 * s + 1][2], h2[2 * s + 1][3] };
        u32x4 dh, dl;
        split8_bf16(in, dh, dl);
        acc = cf_mma(lds, 12 + s, lane, dh, dl, acc);
    }
    rgb[0] = dv_sigmoid(acc[0]); rgb[1] = dv_sigmoid(acc[1]); rgb[2] = dv_sigmoid(acc[2]);
}

// ---- one 64-bin chunk of the per-bin scans of up_sample: row-local inclusive scan + cross-row carry ----
// v: this lane's element (identity-padded); carry/first: state entering the chunk; returns the inclusive
// tile-scan value and the value entering this lane's row (rc), updates carry/first.
template <bool MUL>
__device__ __forceinline__ float chunk_scan(float v, int lane, float &carry, bool &first, float &loc, float &row_in, bool &row_first)
{
    loc = row_scan<MUL>(v);
    const float t0 = lane_bcast(loc, 15), t1 = lane_bcast(loc, 31), t2 = lane_bcast(loc, 47), t3 = lane_bcast(loc, 63);
    const float c0 = carry; const bool f0 = first;
    const float c1 = f0 ? t0 : (MUL ? c0 * t0 : c0 + t0);
    const float c2 = MUL ? c1 * t1 : c1 + t1;
    const float c3 = MUL ? c2 * t2 : c2 + t2;
    const float c4 = MUL ? c3 * t3 : c3 + t3;
    const int g = lane >> 4;
    row_in = sel4(g, c0, c1, c2, c3);
    row_first = f0 && g == 0;
    carry = c4; first = false;
    return row_first ? loc : (MUL ? row_in * loc : row_in + loc);
}

__device__ __forceinline__ FieldCtx make_ctx(const RenderArgs &a)
{
    FieldCtx fc;
    fc.table = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.table), 0, a.table_bytes, 0x00020000);
    fc.jmode[0] = a.jmode[0]; fc.jmode[1] = a.jmode[1]; fc.jmode[2] = a.jmode[2]; fc.jmode[3] = a.jmode[3];
    fc.jfine[0] = a.jfine[0]; fc.jfine[1] = a.jfine[1]; fc.jfine[2] = a.jfine[2]; fc.jfine[3] = a.jfine[3];
    fc.bound = a.bound; fc.two_bound = a.two_bound; fc.inv_tb = a.inv_tb;
    return fc;
}

// =====================================================================================================

int fill_args(RenderArgs &a, const ac_field *f, float bound)
{
    if (!f || !f->table || !f->W1 || !f->b1 || !f->W2 || !f->b2 || !f->Wc1 || !f->Wc2 || !f->Wc3) {
        ac::set_error("ac_field: NULL parameter pointer"); return AC_ERR_BAD_ARG;
    }
    ac::LevelTable lt; ac::make_level_table(lt, 16, 3, f->S, f->H, f->offsets);
    for (int l = 0; l < 16; ++l) {
        const bool hashed = lt.hashed[l] != 0, pow2 = lt.pow2mask[l] != 0;
        a.lvl[l].scale = lt.scale[l]; a.lvl[l].offset = lt.offset[l]; a.lvl[l].hashed = hashed ? 1u : 0u;
        a.lvl[l].my = hashed ? 2654435761u : lt.stride1[l];
        a.lvl[l].mz = hashed ? 805459861u : lt.stride1[l] * lt.stride1[l];
        a.lvl[l].mask = (hashed && pow2) ? lt.pow2mask[l] : 0xffffffffu;
        a.lvl[l].wsize = (hashed && !pow2) ? lt.size[l] : 0u;
        a.lvl[l].pad = 0;
        if (a.lvl[l].wsize) {
            ac::set_error("ac_field: level %d is hashed with a non power-of-two size %u; the fused renderer supports tables "
                          "allocated by HashEncoder only (use ac_hash_encode_forward for arbitrary layouts)", l, lt.size[l]);
            return AC_ERR_BAD_ARG;
        }
        if (lt.size[l] == 0) { ac::set_error("ac_field: level %d has zero size", l); return AC_ERR_BAD_ARG; }
    }
    for (int j = 0; j < 4; ++j) {
        int nh = 0;
        for (int g = 0; g < 4; ++g) nh += lt.hashed[4 * j + g] ? 1 : 0;
        a.jmode[j] = nh == 0 ? 0 : (nh == 4 ? 1 : 2);
    }
    a.prepared = static_cast<const float *>(f->prepared);
    a.Wsh = f->Wc1_sh;
    a.table = f->table; a.table_bytes = (uint32_t)f->offsets[16] * 8u; a.W1 = f->W1; a.b1 = f->b1; a.W2 = f->W2; a.b2 = f->b2; a.Wc1 = f->Wc1; a.Wc2 = f->Wc2; a.Wc3 = f->Wc3;
    a.bound = bound; a.two_bound = (float)(2.0 * (double)bound);
    a.inv_tb = ac::verified_reciprocal(a.two_bound);
    return AC_OK;
}


}  // namespace
