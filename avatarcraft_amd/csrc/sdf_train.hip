// avatarcraft_amd/csrc/sdf_train.hip -- fused SDF query of the differentiable render core (training path), forward and backward.
//
// One SDF query of the render core (reference models/instant_nsr.py:205-215) is forward_sdf at x (:627-642: hash encoder ->
// cat[x, h] -> WN-Linear 35->64 -> Softplus(100) -> WN-Linear 64->16) plus finite_difference_normals_approximator at
// clamp(x +- eps e_k) (:687-704): 7 encoder + MLP evaluations, ~40 PyTorch kernels forward and ~90 backward, every
// intermediate ([7B,35], [7B,64] x 2, [7B,16]) round-tripping HBM.  Here:
//   sdf_stencil_fwd_kernel : x -> sdf_out[B,16], gradient[B,3]     (the renderer's own stencil gather + MFMA MLP tiles,
//                                                                    bit-identical to the fused renderer and the oracle)
//   sdf_stencil_bwd_kernel : (x, d sdf_out, d gradient) -> d features [7,16,B,2] (consumed by hash_stencil_bwd_kernel, which
//                            owns the atomic-bound table scatter) and per-wave partial sums of dW1, db1, dW2, db2.
// The backward recomputes the forward per tile of 16 samples (gathers are L2/MALL hits, MFMA is cheap) instead of storing
// activations, and runs every product on the matrix pipe, tile layout lane = (sample n = lane & 15, group g = lane >> 4):
//   ga   = W2^T d2         16 MFMA   B operand = d2 in the register layout the forward produced it in (o = 4g + s)
//   d1   = ga * softplus'  (derivative of the SAME table polynomial the forward evaluates)
//   dinp = W1^T d1         32 MFMA   row order chosen so that every lane receives the gradient of ITS OWN 8 features
//   dW2 += d2 a^T          16 MFMA   } K = the 16 samples of the tile: operands transposed through a per-wave LDS slab,
//   dW1 += d1 inp^T        48 MFMA   } accumulators live in registers for the whole kernel (a column of ones gives db1)
// Weight-gradient partials are written per wave (no atomics) and summed by sdf_partials_reduce_kernel (deterministic).
#include <atomic>
#include "nsr_device.hpp"
#include "rm_device.hpp"

namespace {

#ifndef AC_SDFBWD_RANK1
#define AC_SDFBWD_RANK1 1     // offset evaluations of the backward: rank-1 shortcuts instead of 32 of their 148 MFMA (0: every evaluation alike)
#endif
#ifndef AC_SDFBWD_PREFETCH
#define AC_SDFBWD_PREFETCH 1  // sdf_stencil_bwd_kernel requests the next tile's inputs while it computes the current one
#endif
constexpr int TW = 4;                              // waves per workgroup of the backward kernels (296 / 284 VGPRs: one wave per SIMD)
constexpr int TBLOCK = TW * 64;
constexpr int FW = 8;                              // waves per workgroup of the forward SDF query (224 VGPRs: two waves per SIMD, like the renderer)
constexpr int FBLOCK = FW * 64;
#ifndef AC_TLD
#define AC_TLD 17
#endif
constexpr int TLD = AC_TLD;                         // padded leading dimension of the transpose slabs
// The SDF-query backward never evaluates the colour network: the 26 KB the shared layout reserves for its fragments hold this kernel's
// own ones instead (fill_lds_sdf leaves that region alone).
constexpr int OFF_W2T = OFF_C1F;                   // [4 tiles][4 ksteps][64]   fp32 A fragments of W2^T (the centre evaluation's ga = W2^T d2)
constexpr int OFF_W1TH = OFF_W2T + 16 * 64;        // [2 tiles][2 k blocks][64 lanes][4 dwords]  bf16 hi A fragments of W1^T (feature rows), 16x16x32 order
constexpr int OFF_W1TL = OFF_W1TH + 2 * 2 * 64 * 4;   // ... and the bf16 lo parts
constexpr int OFF_BW1H = OFF_W1TL + 2 * 2 * 64 * 4;   // layer-1 recomputation of the six offset evaluations (sdf_l1_delta): W1 hi / lo / coordinate columns
constexpr int OFF_BW1L = OFF_BW1H + 4 * 64 * 4;
constexpr int OFF_BW1C = OFF_BW1L + 4 * 64 * 4;
static_assert(OFF_BW1C + 3 * 64 <= OFF_B1, "the backward's fragments fit the colour region");
constexpr int OFF_TW = OFF_WAVE;                   // per-wave slabs
constexpr int TS_T2 = FE_SLAB;                     // d2  [16][TLD]
constexpr int TS_TA = TS_T2 + 16 * TLD;            // a   [64][TLD]
constexpr int TS_TD = TS_TA + 64 * TLD;            // d1  [64][TLD]
constexpr int TS_TI = TS_TD + 64 * TLD;            // inp [48][TLD]  rows 0..34 inputs, 35 = 1 (bias column), 36..47 = 0
constexpr int TRAIN_SLAB = ((TS_TI + 48 * TLD + 3) / 4) * 4;
constexpr int FWD_LDS_FLOATS = OFF_WAVE + FW * FE_SLAB;
static_assert(FWD_LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
constexpr int BWD_LDS_FLOATS = OFF_TW + TW * TRAIN_SLAB;
static_assert(BWD_LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
// Round 6: the SDF-query backward on SAVED stencil features (the render core's backward) at TWO waves per SIMD.  Round 4 / 5 ran it at one (296 registers: 66 of
// them the next tile's prefetched inputs, 56 of those the 14 x 16 bytes of saved features, which then also occupied a 12 KB slab per wave): every LDS round
// trip, every matrix-to-vector dependency of the serial chain was exposed (0.635 ms per 4096-ray patch, VALU + MFMA pipe time ~0.4 of that).  Now an evaluation's
// eight features are requested from the forward's buffer one evaluation ahead (2 x 16 bytes per lane): no feature slab.  What that buys:
//   * NOT a second wave per SIMD (AC_SDFBWD_WAVES=8: 256 registers, occupancy 2 -- and 148 spilled registers, 368 bytes of scratch per lane: the kernel's
//     natural live set is ~370 registers; backward of the SDS step 2.24 -> 3.10 ms);
//   * room for a second buffer of the d1 / input transposes, so that the weight-gradient products of evaluation e (48 fp32 MFMA of 32 clocks, a quarter of
//     the evaluation's time, with nothing beside them on the wave's only instruction stream) are issued INSIDE evaluation e + 1's recomputation
//     (softplus, splits: vector work) instead of behind a barrier at the end of their own evaluation.
#ifndef AC_SDFBWD_WAVES
#define AC_SDFBWD_WAVES 4
#endif
#ifndef AC_SDFBWD_PIPE
#define AC_SDFBWD_PIPE 1      // 1: the weight-gradient products of evaluation e run inside evaluation e + 1 on the bf16 matrix pipe (SAVED variant); 0: fp32, behind their own evaluation
#endif
constexpr int TW_S = AC_SDFBWD_WAVES;                                  // waves per workgroup of sdf_stencil_bwd_kernel<SAVED = true>
// PIPE: the transposes of d1 [64 units][16 samples] and inp [48 columns][16 samples] as bf16 hi | lo parts, rows of 16 samples = 32 bytes padded to 48
// (conflict-free 16-byte reads, 16-byte aligned): one buffer = d1 hi 3072 | d1 lo 3072 | inp hi 2304 | inp lo 2304 bytes, two buffers per wave
constexpr int BF_ROW = 48, BF_D1H = 0, BF_D1L = 64 * BF_ROW, BF_INH = 2 * 64 * BF_ROW, BF_INL = BF_INH + 48 * BF_ROW, BF_BYTES = BF_INL + 48 * BF_ROW;
constexpr int TRAIN_SLAB_PIPE = ((TS_TD - FE_SLAB) + 2 * BF_BYTES / 4 + 3) / 4 * 4;       // T2 | TA | two bf16 buffers
constexpr int TRAIN_SLAB_NOPIPE = TRAIN_SLAB - FE_SLAB;
constexpr int TRAIN_SLAB_S = TRAIN_SLAB_PIPE > TRAIN_SLAB_NOPIPE ? TRAIN_SLAB_PIPE : TRAIN_SLAB_NOPIPE;    // the SAVED variant's per-wave slab: no feature region
constexpr int BWD_LDS_FLOATS_S = OFF_TW + TW_S * TRAIN_SLAB_S;
static_assert(BWD_LDS_FLOATS_S * 4 <= 160 * 1024, "LDS budget");
constexpr int NPART = 64 * 36 + 16 * 64 + 16;      // dW1 [64][36] (column 35 = db1), dW2 [16][64], db2 [16]

// softplus_100 and its derivative from the same table row: d/dx [max(x,0) + q(fract(|400 x|))] = [x > 0] + sign(x) 400 q'(v)
__device__ __forceinline__ void softplus100_vg(const float *__restrict__ spg, float x, float &val, float &der)
{
    const float a4 = __builtin_fminf(__builtin_fabsf(x * 400.0f), 128.0f);
    const uint32_t idx = (uint32_t)a4;
    const float v = __builtin_amdgcn_fractf(a4);
    const float4 c = *reinterpret_cast<const float4 *>(spg + idx * 4);
    float q = c.w;
    q = fma_(q, v, c.z); q = fma_(q, v, c.y); q = fma_(q, v, c.x);
    float dq = 3.0f * c.w;
    dq = fma_(dq, v, 2.0f * c.z); dq = fma_(dq, v, c.y);
    const bool pos = x > 0.0f;
    val = fma_(0.5f, __builtin_fabsf(x), fma_(0.5f, x, q));
    der = (pos ? 1.0f : 0.0f) + (pos ? 400.0f : -400.0f) * dq;
}

// four values at once: the table rows are requested together (one LDS round trip per batch, see dv_softplus100_n)
__device__ __forceinline__ void softplus100_vg4(const float *__restrict__ spg, const f32x4 &x, f32x4 &val, f32x4 &der)
{
    float v[4]; float4 c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a4 = __builtin_fminf(__builtin_fabsf(x[i] * 400.0f), 128.0f);
        const uint32_t idx = (uint32_t)a4;
        v[i] = __builtin_amdgcn_fractf(a4);
        c[i] = *reinterpret_cast<const float4 *>(spg + idx * 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float q = c[i].w;
        q = fma_(q, v[i], c[i].z); q = fma_(q, v[i], c[i].y); q = fma_(q, v[i], c[i].x);
        float dq = 3.0f * c[i].w;
        dq = fma_(dq, v[i], 2.0f * c[i].z); dq = fma_(dq, v[i], c[i].y);
        const bool pos = x[i] > 0.0f;
        val[i] = fma_(0.5f, __builtin_fabsf(x[i]), fma_(0.5f, x[i], q));
        der[i] = (pos ? 1.0f : 0.0f) + (pos ? 400.0f : -400.0f) * dq;
    }
}

// the 7 evaluations of one tile: centre outputs (o = 4g + r) and the finite-difference gradient (the same in all four lanes of a sample)
__device__ __forceinline__ void fd_forward(const float *__restrict__ lds, const float *__restrict__ fsl, int lane, float px, float py, float pz,
                                           float eps, float bound, const float (&fe0)[4][2], f32x4 &oc, float (&gr)[3])
{
    const int g = lane >> 4;
    const float pc0 = sel4(g, px, py, pz, 0.0f);
    oc = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
    gr[0] = gr[1] = gr[2] = 0.0f;
    float spos = 0.0f;
    const W2Row0 w2r0 = load_w2_row0(lds, lane);
    Acc4 acc = sdf_l1(lds, lane, pc0, fe0);
#pragma unroll 1
    for (int e = 0; e < 7; ++e) {
        Acc4 accn = acc;
        if (e < 6) {                                               // layer 1 of the next evaluation
            const int kn = e >> 1;
            float fe[4][2];
#pragma unroll
            for (int q_ = 0; q_ < 8; ++q_) fe[q_ >> 1][q_ & 1] = fsl[(e * 8 + q_) * 64 + lane];
            const float pk = kn == 0 ? px : (kn == 1 ? py : pz);
            const float poff = clampf(pk + ((e & 1) ? -eps : eps), -bound, bound);
            accn = sdf_l1(lds, lane, g == kn ? poff : pc0, fe);
        }
        if (e == 0) oc = sdf_l2(lds, lane, acc);                   // the centre: all 16 outputs
        else {                                                     // the six offset points: the sdf alone (same arithmetic as the renderer)
            const float s_e = sdf_l2_sdf(lds, acc, w2r0);
            const int k = (e - 1) >> 1;
            if (e & 1) spos = s_e;
            else {
                const float gk = 0.5f * (spos - s_e) / eps;
                if (k == 0) gr[0] = gk; else if (k == 1) gr[1] = gk; else gr[2] = gk;
            }
        }
        acc = accn;
    }
}

__global__ __launch_bounds__(FBLOCK) void sdf_stencil_fwd_kernel(const RenderArgs a, const float *__restrict__ x, uint32_t B, float eps,
                                                                 float *__restrict__ out16, float *__restrict__ grad)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    fill_lds(lds, a);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    float *fsl = lds + OFF_WAVE + wave * FE_SLAB;
    const FieldCtx fc = make_ctx(a);
    const uint32_t ntiles = (B + 15) / 16;
    for (uint32_t tile = blockIdx.x * FW + wave; tile < ntiles; tile += gridDim.x * FW) {
        const uint32_t b = tile * 16 + n, bb = b < B ? b : B - 1;
        const float px = x[3 * (size_t)bb], py = x[3 * (size_t)bb + 1], pz = x[3 * (size_t)bb + 2];
        float fe0[4][2];
        encode_stencil(lds, fsl, fc, lane, px, py, pz, eps, fe0);
        f32x4 oc; float gr[3];
        fd_forward(lds, fsl, lane, px, py, pz, eps, a.bound, fe0, oc, gr);
        if (b < B) {
            *reinterpret_cast<f32x4 *>(out16 + (size_t)b * 16 + 4 * g) = oc;
            if (g == 0) { grad[3 * (size_t)b] = gr[0]; grad[3 * (size_t)b + 1] = gr[1]; grad[3 * (size_t)b + 2] = gr[2]; }
        }
        wave_sync();
    }
}

// extra weight fragments of the backward: W2^T (ga = W2^T d2, fp32) and the feature rows of W1^T (dinp = W1^T d1) split into bf16 hi + lo
// for v_mfma_f32_16x16x32_bf16: K = the 64 hidden units in two blocks of 32; within block b lane group kk supplies slots i = 0..7 =
// units 16 (2b + (i >> 2)) + 4 kk + (i & 3) -- exactly the sixteen d1 values lane (n, kk) holds (accumulator tiles 2b, 2b+1), no data movement
__device__ __forceinline__ void fill_lds_bwd(float *lds, const RenderArgs &a)
{
    for (int e = threadIdx.x; e < 16 * 64; e += blockDim.x) {       // fragment (t, s): lane (m, kk) = W2[o = 4 kk + s][unit = 16 t + m]
        const int l = e & 63, fs = e >> 6, t = fs >> 2, s = fs & 3, m = l & 15, kk = l >> 4;
        lds[OFF_W2T + e] = a.W2[(4 * kk + s) * 64 + 16 * t + m];
    }
    uint32_t *lw = reinterpret_cast<uint32_t *>(lds);
    for (int e = threadIdx.x; e < 2 * 2 * 64 * 4; e += blockDim.x) { // dword q of lane l, k block b, output tile tp
        const int q = e & 3, l = (e >> 2) & 63, b = (e >> 8) & 1, tp = e >> 9, m = l & 15, kk = l >> 4;
        const int s1 = 4 * tp + (m & 3);                            // output row m = feature slot 2j + c of lane group m >> 2
        const int col = 3 + 2 * (4 * (s1 >> 1) + (m >> 2)) + (s1 & 1);
        uint32_t hi2 = 0, lo2 = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = 2 * q + h, unit = 16 * (2 * b + (i >> 2)) + 4 * kk + (i & 3);
            const float w = a.W1[unit * 35 + col];
            const uint32_t hb = bf16_rne_bits(w), lb = bf16_rne_bits(w - __uint_as_float(hb << 16));
            hi2 |= hb << (16 * h); lo2 |= lb << (16 * h);
        }
        lw[OFF_W1TH + e] = hi2; lw[OFF_W1TL + e] = lo2;
    }
    fill_lds_fast<OFF_BW1H, OFF_BW1L, OFF_BW1C>(lds, a);
}

// SAVED: the features of the seven stencil points come from the forward launch (ac_render_out.feat7, [B / 16][14][64 lanes][4] in this kernel's lane order) as 14
// coalesced 16-byte loads per lane instead of being gathered from the table again (index arithmetic + ~100 scattered 8-byte loads per lane and tile)
// GX: the sample positions carry a gradient themselves (the curvature term's perturbed points, models/instant_nsr.py:276-288): g_x [B][3] receives the share
// that enters through the MLP's own xyz inputs (include_input, :632-633) -- W1[:, 0:3]^T d1 summed over the seven evaluations, the offset coordinate of an
// offset evaluation through its clamp (:690-702).  The share through the encodings is ac_hash_stencil_input_backward's.
template <bool SAVED, bool GX = false>
__global__ __launch_bounds__(SAVED ? TW_S * 64 : TBLOCK) void sdf_stencil_bwd_kernel(const RenderArgs a, const float *__restrict__ x, const float *__restrict__ g_out,
                                                                 const float *__restrict__ g_grad, uint32_t B, float eps,
                                                                 float *__restrict__ gfeat, float *__restrict__ partials,
                                                                 const float *__restrict__ feat7, float *__restrict__ g_x = nullptr)
{
    static_assert(!(SAVED && GX), "the position gradient is built for the stand-alone operator");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    fill_lds_sdf(lds, a);
    fill_lds_bwd(lds, a);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    constexpr int KW = SAVED ? TW_S : TW, SFE = SAVED ? FE_SLAB : 0;   // waves per workgroup | floats the slab does not hold (SAVED: no feature region)
    float *slab = lds + OFF_TW + wave * (SAVED ? TRAIN_SLAB_S : TRAIN_SLAB);
    float *fsl = slab, *T2 = slab + TS_T2 - SFE, *TA = slab + TS_TA - SFE, *TD0 = slab + TS_TD - SFE, *TI0 = slab + TS_TI - SFE;
    // PIPE: two buffers of (d1, inp) transposes -- evaluation `step` writes buffer step & 1 while the weight-gradient products of the evaluation before it read
    // the other one.  The very first step's "previous evaluation" is a buffer of zeros (products of 0: the accumulators keep their +0).
    constexpr bool PIPE = SAVED && AC_SDFBWD_PIPE && AC_SDFBWD_RANK1;
    unsigned char *const BF = reinterpret_cast<unsigned char *>(TD0);      // PIPE: the two bf16 buffers take the place of the fp32 transposes
    if constexpr (PIPE) {
        uint32_t *bw = reinterpret_cast<uint32_t *>(BF);
        for (int e = lane; e < 2 * BF_BYTES / 4; e += 64) {                 // zeros; the bias column (input row 35) = bf16 1.0 in the hi part, for all 16 samples
            const int o = (4 * e) % BF_BYTES;
            bw[e] = (o >= BF_INH + 35 * BF_ROW && o < BF_INH + 35 * BF_ROW + 32) ? 0x3f803f80u : 0u;
        }
    } else {
        for (int e = lane; e < 48 * TLD; e += 64) TI0[e] = (e >= 35 * TLD && e < 36 * TLD) ? 1.0f : 0.0f;       // bias column, zero padding rows
    }
    __syncthreads();
    const FieldCtx fc = make_ctx(a);
    const float bound = a.bound;
    f32x4 gW1[4][3], gW2[4];
    float gb2[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
#if AC_SDFBWD_RANK1
    float a6[4][4];                                   // running sum over samples and offset evaluations of s * a[unit 16t + 4g + r] (-> dW2 row 0)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) a6[t][r] = 0.0f;
    const W2Row0 w2r0 = load_w2_row0(lds, lane);
#endif
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        gW2[t] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
        for (int c = 0; c < 3; ++c) gW1[t][c] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
    }
    float wxyz[GX ? 4 : 1][4][3];                               // GX: W1[unit 16t + 4g + r][0..2], this lane's 16 units
    if constexpr (GX) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) wxyz[t][r][c] = a.W1[(16 * t + 4 * g + r) * 35 + c];
    }
    uint32_t step = 0;                                          // evaluations this wave has been through (PIPE: selects the buffer)
    // the weight-gradient products of ONE evaluation from its transposes: dW2 += d2 a^T (16 MFMA, centre evaluations only: the offset evaluations' share is
    // the rank-1 running sum a6) and dW1 += d1 inp^T (48 MFMA)
    auto w2_grads = [&]() {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float a2 = T2[n * TLD + 4 * s + g];                         // A: d2[o = lane & 15][sample 4s + kk]
#pragma unroll
            for (int c = 0; c < 4; ++c)
                gW2[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, TA[(16 * c + n) * TLD + 4 * s + g], gW2[c], 0, 0, 0);
        }
    };
    auto w1_grads = [&](const float *__restrict__ TDr, const float *__restrict__ TIr) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float bi[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) bi[c] = TIr[(16 * c + n) * TLD + 4 * s + g];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float a1 = TDr[(16 * t + n) * TLD + 4 * s + g];
#pragma unroll
                for (int c = 0; c < 3; ++c) gW1[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bi[c], gW1[t][c], 0, 0, 0);
            }
        }
    };
    const uint32_t ntiles = (B + 15) / 16;
    const float hs = 0.5f / eps;
    // The inputs of tile i + 1 are requested while tile i is computed (round 4): with one wave per SIMD nothing else covers their ~2 us, and a third of
    // the 512 registers a lone wave may use are free.  TileIn = what a lane reads per tile: its sample's position and upstream gradients and, with SAVED,
    // the 14 x 16 bytes of stencil features the forward kept ([tile][14][lane][4], see render_rays_kernel: each load 1 KB contiguous per wave).
    struct TileIn { float p[3], gg[3]; f32x4 go; };
    auto request = [&](uint32_t tile, TileIn &in) {
        const uint32_t b = tile * 16 + n, bb = b < B ? b : B - 1;
#pragma unroll
        for (int k = 0; k < 3; ++k) { in.p[k] = x[3 * (size_t)bb + k]; in.gg[k] = g_grad[3 * (size_t)bb + k]; }
        in.go = *reinterpret_cast<const f32x4 *>(g_out + (size_t)bb * 16 + 4 * g);
    };
    // SAVED: the eight features of evaluation e of a tile = floats 8 e .. 8 e + 7 of this lane = two 16-byte loads ([tile][14][lane][4]), requested one
    // evaluation ahead
    struct Feat2 { f32x4 a, b; };
    auto request_feat = [&](uint32_t tile, int e, Feat2 &f) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(feat7) + ((size_t)tile * 14 + 2 * (size_t)e) * 64 + lane;
        f.a = src[0]; f.b = src[64];
    };
    const uint32_t tile0 = blockIdx.x * KW + wave, tstride = gridDim.x * KW;
    TileIn nxt;
    Feat2 fnx{};
    if (tile0 < ntiles) { request(tile0, nxt); if constexpr (SAVED) request_feat(tile0, 0, fnx); }
    for (uint32_t tile = tile0; tile < ntiles; tile += tstride) {
        const uint32_t b = tile * 16 + n;
        const bool live = b < B;
        const float px = nxt.p[0], py = nxt.p[1], pz = nxt.p[2];
        f32x4 go = nxt.go;
        float gg[3] = { nxt.gg[0], nxt.gg[1], nxt.gg[2] };
        if (!live) { go = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f }; gg[0] = gg[1] = gg[2] = 0.0f; }
        float fe0[4][2];
        if constexpr (SAVED) {
#pragma unroll
            for (int q_ = 0; q_ < 4; ++q_) { fe0[q_ >> 1][q_ & 1] = fnx.a[q_]; fe0[2 + (q_ >> 1)][q_ & 1] = fnx.b[q_]; }
        } else {
            encode_stencil(lds, fsl, fc, lane, px, py, pz, eps, fe0);
        }
#if AC_SDFBWD_PREFETCH
        if (tile + tstride < ntiles) request(tile + tstride, nxt);
#endif
        const float pc0 = sel4(g, px, py, pz, 0.0f);
        Acc4 h10;                                               // layer 1 of the centre evaluation (set at e == 0)
#pragma unroll
        for (int t = 0; t < 4; ++t) h10.a[t] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
        float gxa[3] = { 0.0f, 0.0f, 0.0f };                    // GX: d loss / d (px, py, pz) through the xyz inputs, summed over the evaluations
#pragma unroll 1
        for (int e = 0; e < 7; ++e) {
            const int k = e > 0 ? (e - 1) >> 1 : 3;
            float *const TD = TD0, *const TI = TI0;                                                      // (!PIPE) this evaluation's fp32 transposes
            if constexpr (PIPE) { if (e == 1) w2_grads(); }                    // (the evaluation before this one was the tile's centre)
            float fe[4][2];
            if constexpr (SAVED) {
#pragma unroll
                for (int q_ = 0; q_ < 4; ++q_) { fe[q_ >> 1][q_ & 1] = fnx.a[q_]; fe[2 + (q_ >> 1)][q_ & 1] = fnx.b[q_]; }
                // the next evaluation's features (the next tile's centre after the last one) leave now: a whole evaluation of arithmetic covers them
                if (e < 6) request_feat(tile, e + 1, fnx);
                else if (tile + tstride < ntiles) request_feat(tile + tstride, 0, fnx);
            } else if (e == 0) {
#pragma unroll
                for (int q_ = 0; q_ < 8; ++q_) fe[q_ >> 1][q_ & 1] = fe0[q_ >> 1][q_ & 1];
            } else {
#pragma unroll
                for (int q_ = 0; q_ < 8; ++q_) fe[q_ >> 1][q_ & 1] = fsl[((e - 1) * 8 + q_) * 64 + lane];
            }
            const float pk = k == 0 ? px : (k == 1 ? py : pz);
            const float poff = clampf(pk + (((e - 1) & 1) ? -eps : eps), -bound, bound);
            const float bx = (e > 0 && g == k) ? poff : pc0;
            // upstream gradient of this evaluation's 16 outputs, in the forward's register layout (o = 4g + r)
            f32x4 d2;
            float s_all = 0.0f;                                                 // the offset evaluations' only upstream value, in every lane of the sample
            if (e == 0) d2 = go;
            else {
                const float gk = k == 0 ? gg[0] : (k == 1 ? gg[1] : gg[2]);
                s_all = ((e - 1) & 1) ? -(gk * hs) : gk * hs;                   // d gradient_k / d sdf(x +- eps e_k) = +-0.5 / eps
                d2 = f32x4{ g == 0 ? s_all : 0.0f, 0.0f, 0.0f, 0.0f };
            }
            // recompute layer 1, softplus and its derivative: the centre in fp32, the six offset evaluations as split-bf16 corrections of it
            // (sdf_l1_delta; the arithmetic of the renderer's "fast" precision: 12 short MFMA instead of 36 fp32 ones)
            Acc4 h1;
            if (e == 0) { h1 = sdf_l1(lds, lane, bx, fe); h10 = h1; }
            else h1 = sdf_l1_delta<OFF_BW1H, OFF_BW1L, OFF_BW1C>(lds, lane, h10, fe, fe0, k, poff - pk);
            Acc4 av, dv;
            if constexpr (PIPE) {
                // The PREVIOUS evaluation's weight-gradient products dW1 += d1 inp^T, issued inside this evaluation's 16 softplus values + derivatives (the
                // largest stretch of vector work of an evaluation: no LDS stores, no branches), the order pinned by scheduling fences.
                // On the bf16 matrix pipe, K = 32 = [hi parts of the tile's 16 samples | lo parts]: per 16 x 16 output tile
                //   gW1 += [d1 hi | d1 lo] x [inp hi | inp hi]   (hi hi + lo hi)      gW1 += [d1 hi | d1 lo] x [inp lo | 0]   (hi lo)
                // -- 24 instructions of 16 clocks that CO-EXECUTE with vector instructions, instead of 48 fp32 ones of 32 clocks that do not (the fp32 matrix
                // instructions run on the vector pipe's multipliers, SQ_VALU_MFMA_COEXEC_CYCLES = 0: interleaving THOSE with the softplus work was built
                // and measured first and bought nothing -- backward 2.244 -> 2.300 ms, profiles/r06_experiments.txt).  The dropped lo x lo term is 2^-16 of
                // a product; the sums stay fp32.  Lane (m, kk) reads 8 consecutive samples of row m: kk = 0, 1 from the hi part, kk = 2, 3 from the lo
                // part (d1) / the hi part again (inp, first product) / a row of zeros (inp, second product).
                const unsigned char *const Br = BF + ((step & 1u) ? 0 : BF_BYTES);
                const float *__restrict__ spg = lds + OFF_SPQ;
                const int kk_ = lane >> 4, m_ = lane & 15;
                const unsigned char *const pa = Br + (kk_ >= 2 ? BF_D1L : BF_D1H) + m_ * BF_ROW + (kk_ & 1) * 16;
                const unsigned char *const pb1 = Br + BF_INH + m_ * BF_ROW + (kk_ & 1) * 16;
                const unsigned char *const pb2 = kk_ >= 2 ? Br + BF_INL + 40 * BF_ROW : Br + BF_INL + m_ * BF_ROW + (kk_ & 1) * 16;    // (input rows 36 .. 47 are zero padding)
                const int pb2s = kk_ >= 2 ? 0 : 16 * BF_ROW;
                bf16x8 opA[4], opB1[3], opB2[3];
                auto rdA = [&](int t_) { opA[t_] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(pa + 16 * t_ * BF_ROW)); };
                auto rdB = [&](int c_) {
                    opB1[c_] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(pb1 + 16 * c_ * BF_ROW));
                    opB2[c_] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(pb2 + c_ * pb2s));
                };
                rdA(0); rdA(1); rdB(0); rdB(1); rdB(2);
                float xs[16], vv[16], qq[16], dq[16];
                float4 cc[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) xs[i] = h1.a[i >> 2][i & 3];
                auto P1 = [&](int i) {                                       // row request of value i
                    const float a4 = __builtin_fminf(__builtin_fabsf(xs[i] * 400.0f), 128.0f);
                    const uint32_t idx = (uint32_t)a4;
                    vv[i] = __builtin_amdgcn_fractf(a4);
                    cc[i] = *reinterpret_cast<const float4 *>(spg + idx * 4);
                };
                auto P2 = [&](int i) {                                       // the cubic and its derivative
                    float q = cc[i].w;
                    q = fma_(q, vv[i], cc[i].z); q = fma_(q, vv[i], cc[i].y); q = fma_(q, vv[i], cc[i].x);
                    float d = 3.0f * cc[i].w;
                    d = fma_(d, vv[i], 2.0f * cc[i].z); d = fma_(d, vv[i], cc[i].y);
                    qq[i] = q; dq[i] = d;
                };
                auto P3 = [&](int i) {                                       // value and derivative of softplus_100
                    const bool pos = xs[i] > 0.0f;
                    av.a[i >> 2][i & 3] = fma_(0.5f, __builtin_fabsf(xs[i]), fma_(0.5f, xs[i], qq[i]));
                    dv.a[i >> 2][i & 3] = (pos ? 1.0f : 0.0f) + (pos ? 400.0f : -400.0f) * dq[i];
                };
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kq = 0; kq < 24; ++kq) {
                    const int t_ = kq / 6, c_ = (kq % 6) >> 1, second = kq & 1;
                    gW1[t_][c_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(opA[t_], second ? opB2[c_] : opB1[c_], gW1[t_][c_], 0, 0, 0);
                    if (kq == 0) rdA(2);
                    if (kq == 2) rdA(3);
                    // two slices of the softplus work per group: P1(0) P1(1) P1(2) | P2(i) P3(i) P1(i + 3) for i = 0 .. 12 | P2(13) P3(13) .. P2(15) P3(15)
#pragma unroll
                    for (int h_ = 0; h_ < 2; ++h_) {
                        const int ks = 2 * kq + h_;
                        if (ks < 3) P1(ks);
                        else if (ks < 42) { const int i = (ks - 3) / 3, ph = (ks - 3) % 3; if (ph == 0) P2(i); else if (ph == 1) P3(i); else P1(i + 3); }
                        else { const int i = 13 + (ks - 42) / 2, ph = (ks - 42) % 2; if (ph == 0) P2(i); else P3(i); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) softplus100_vg4(lds + OFF_SPQ, h1.a[t], av.a[t], dv.a[t]);
            }
            Acc4 d1;
#if AC_SDFBWD_RANK1
            // The six offset evaluations feed the finite-difference gradient through their sdf alone: d2 = (s, 0, ..., 0).  Then
            // ga = W2^T d2 = s * W2[0, :] (16 multiplies instead of 16 MFMA), and their share of dW2 / db2 is row 0 only:
            // dW2[0, u] += sum over samples of s * a[u], kept as a per-lane running sum (reduced over the samples once, at the end).
            const bool rank1 = e > 0;
            if (rank1) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        d1.a[t][r] = (s_all * w2r0.w[t][r]) * dv.a[t][r];
                        a6[t][r] = fma_(s_all, av.a[t][r], a6[t][r]);
                    }
            } else
#endif
            {
                // ga = W2^T d2, d1 = ga * softplus'
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    f32x4 ga = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
                    for (int s = 0; s < 4; ++s) ga = __builtin_amdgcn_mfma_f32_16x16x4f32(lds[OFF_W2T + (t * 4 + s) * 64 + lane], d2[s], ga, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) d1.a[t][r] = ga[r] * dv.a[t][r];
                }
            }
            if constexpr (GX) {
                const float raw = pk + (((e - 1) & 1) ? -eps : eps);
                const float pass = (e == 0 || (raw >= -bound && raw <= bound)) ? 1.0f : 0.0f;       // d clamp / d x (inclusive, like torch.clamp's backward)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float sx = 0.0f;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sx = fma_(d1.a[t][r], wxyz[t][r][c], sx);
                    sx += __shfl_xor(sx, 16); sx += __shfl_xor(sx, 32);                                 // over the four lane groups of the sample
                    gxa[c] += (c == k) ? sx * pass : sx;
                }
            }
            // dinp = W1^T d1: this lane's own features (j = 2t' + (r >> 1), c = r & 1), on the bf16 matrix pipe with both factors split
            // hi + lo (3 products, fp32 accumulate: 2^-16 relative, far below the tolerance of a gradient): 12 short MFMA instead of 32 fp32 ones
            {
                u32x4 bh[2], bl[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float dd[8] = { d1.a[2 * b][0], d1.a[2 * b][1], d1.a[2 * b][2], d1.a[2 * b][3],
                                          d1.a[2 * b + 1][0], d1.a[2 * b + 1][1], d1.a[2 * b + 1][2], d1.a[2 * b + 1][3] };
                    split8_bf16(dd, bh[b], bl[b]);
                }
#pragma unroll
                for (int tp = 0; tp < 2; ++tp) {
                    f32x4 gi = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const bf16x8 Ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(lds + OFF_W1TH + ((tp * 2 + b) * 64 + lane) * 4));
                        const bf16x8 Al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(lds + OFF_W1TL + ((tp * 2 + b) * 64 + lane) * 4));
                        const bf16x8 Bh = __builtin_bit_cast(bf16x8, bh[b]), Bl = __builtin_bit_cast(bf16x8, bl[b]);
                        gi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bh, gi, 0, 0, 0);
                        gi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, Bh, gi, 0, 0, 0);
                        gi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bl, gi, 0, 0, 0);
                    }
                    if (live) {
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const int j = 2 * tp + m;
                            *reinterpret_cast<float2 *>(gfeat + (((size_t)e * 16 + 4 * j + g) * B + b) * 2) = make_float2(gi[2 * m], gi[2 * m + 1]);
                        }
                    }
                }
            }
            // transposes for the weight gradients (K = the 16 samples of the tile)
#if AC_SDFBWD_RANK1
            if (!rank1)
#endif
            {
#pragma unroll
                for (int r = 0; r < 4; ++r) T2[(4 * g + r) * TLD + n] = d2[r];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) TA[(16 * t + 4 * g + r) * TLD + n] = av.a[t][r];
            }
            if constexpr (PIPE) {
                // hi = the value ROUNDED TO NEAREST bf16 (v_cvt_pk_bf16_f32), lo = (value - hi) rounded likewise, 2-byte stores at [row][sample n].  Not
                // split8_bf16's truncation: a weight gradient is a sum of ~3.7 M such products per entry, and truncation errors all point towards zero -- they
                // add up coherently.  dW1 vs the fp64 oracle, of max (tests/test_oracle_backward.py; fp32 products before: 1.9e-4; bound 3e-4):
                //   hi and lo truncated 3.1e-4 | hi truncated (a free d16_hi store), lo rounded 2.2e-4 (the dropped lo x lo term keeps the product's sign, and the
                //   three separately back-propagated loss terms add up to the joint pass only to 2.1e-4) | both rounded 1.2e-4  <- shipped (+0.03 ms per patch)
                unsigned char *const Bw = BF + ((step & 1u) ? BF_BYTES : 0);
                auto put = [&](int part_hi, int part_lo, int row, float v) {
                    const __bf16 hb = (__bf16)v;
                    const __bf16 lb = (__bf16)(v - (float)hb);
                    *reinterpret_cast<__bf16 *>(Bw + part_hi + row * BF_ROW + 2 * n) = hb;
                    *reinterpret_cast<__bf16 *>(Bw + part_lo + row * BF_ROW + 2 * n) = lb;
                };
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) put(BF_D1H, BF_D1L, 16 * t + 4 * g + r, d1.a[t][r]);
                if (g < 3) put(BF_INH, BF_INL, g, bx);
#pragma unroll
                for (int s1 = 0; s1 < 8; ++s1) put(BF_INH, BF_INL, 3 + 2 * (4 * (s1 >> 1) + g) + (s1 & 1), fe[s1 >> 1][s1 & 1]);
            } else {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) TD[(16 * t + 4 * g + r) * TLD + n] = d1.a[t][r];
            if (g < 3) TI[g * TLD + n] = bx;
#pragma unroll
            for (int s1 = 0; s1 < 8; ++s1) TI[(3 + 2 * (4 * (s1 >> 1) + g) + (s1 & 1)) * TLD + n] = fe[s1 >> 1][s1 & 1];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) gb2[r] += d2[r];
            wave_sync();
            ++step;
            if constexpr (!PIPE) {
#if AC_SDFBWD_RANK1
                if (!rank1)
#endif
                    w2_grads();
#ifdef AC_ABL_NODW1           // timing ablation: the weight-gradient products of the six offset evaluations are skipped
                if (e == 0)
#endif
                w1_grads(TD, TI);
                wave_sync();
            }
        }
        if constexpr (GX) {
            if (live && g < 3) g_x[3 * (size_t)b + g] = g == 0 ? gxa[0] : (g == 1 ? gxa[1] : gxa[2]);
        }
#if !AC_SDFBWD_PREFETCH
        if (tile + tstride < ntiles) request(tile + tstride, nxt);
#endif
    }
    if constexpr (PIPE) {                                       // the last evaluation's products (an offset evaluation; a wave without a tile: the zero buffer)
        const unsigned char *const Br = BF + ((step & 1u) ? 0 : BF_BYTES);
        const int kk_ = lane >> 4, m_ = lane & 15;
#pragma unroll
        for (int t_ = 0; t_ < 4; ++t_) {
            const bf16x8 A = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(Br + (kk_ >= 2 ? BF_D1L : BF_D1H) + (16 * t_ + m_) * BF_ROW + (kk_ & 1) * 16));
#pragma unroll
            for (int c_ = 0; c_ < 3; ++c_) {
                const bf16x8 B1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(Br + BF_INH + (16 * c_ + m_) * BF_ROW + (kk_ & 1) * 16));
                const bf16x8 B2 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(Br + BF_INL + (kk_ >= 2 ? 40 : 16 * c_ + m_) * BF_ROW + (kk_ >= 2 ? 0 : (kk_ & 1) * 16)));
                gW1[t_][c_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B1, gW1[t_][c_], 0, 0, 0);
                gW1[t_][c_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B2, gW1[t_][c_], 0, 0, 0);
            }
        }
    }
    // per-wave partial sums: dW1 [64][36] | dW2 [16][64] | db2 [16]
    float *part = partials + (size_t)(blockIdx.x * KW + wave) * NPART;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int unit = 16 * t + 4 * g + r, kcol = 16 * c + n;
                if (kcol < 36) part[unit * 36 + kcol] = gW1[t][c][r];
            }
#if AC_SDFBWD_RANK1
    // row 0 of dW2 also receives the offset evaluations' running sums: unit u = 16c + n of lane (n, g = 0) is held, after a row reduction over
    // the 16 samples, by lane 15 of lane group n >> 2 in register a6[c][n & 3]
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float tot[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) tot[r] = __shfl(row_scan<false>(a6[c][r]), 16 * (n >> 2) + 15);
        const float add = (n & 3) == 0 ? tot[0] : ((n & 3) == 1 ? tot[1] : ((n & 3) == 2 ? tot[2] : tot[3]));
        if (g == 0) gW2[c][0] += add;
    }
#endif
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[64 * 36 + (4 * g + r) * 64 + 16 * c + n] = gW2[c][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float tot = row_scan<false>(gb2[r]);                   // inclusive scan over the 16 samples of the row: lane n == 15 holds the total
        if (n == 15) part[64 * 36 + 16 * 64 + 4 * g + r] = tot;
    }
}

// out[i] = sum over waves of partials[w][i]: RED_OUT outputs x RED_SL wave slices per workgroup, fixed summation order (deterministic).  Round 4: 16 x 64
// instead of 64 x 16 -- four times the workgroups and a quarter of the dependent trip count per thread (21 -> ~8 us for 1024 partial sets)
constexpr uint32_t RED_OUT = 16, RED_SL = 64;
__global__ __launch_bounds__(1024) void sdf_partials_reduce_kernel(const float *__restrict__ partials, uint32_t nwaves, float *__restrict__ out)
{
    __shared__ double red[RED_SL][RED_OUT];  // (double: the order of the partials must not show up in the last bits of a sum of ~1000 of them)
    const uint32_t o = threadIdx.x % RED_OUT, sl = threadIdx.x / RED_OUT;
    const uint32_t i = blockIdx.x * RED_OUT + o;
    double s = 0.0;
    if (i < (uint32_t)NPART)
        for (uint32_t w = sl; w < nwaves; w += RED_SL) s += (double)partials[(size_t)w * NPART + i];
    red[sl][o] = s;
    __syncthreads();
    if (sl == 0 && i < (uint32_t)NPART) {
        double t = 0.0;
#pragma unroll
        for (uint32_t k = 0; k < RED_SL; ++k) t += red[k][o];
        out[i] = (float)t;
    }
}

// ======================================================================================================================
// colour MLP of the render core (forward_color, models/instant_nsr.py:644-663, use_viewdirs = False):
//   rgb = sigmoid(Wc3 relu(Wc2 relu(Wc1 [x(3), normal(3), feat(15)])))            (no biases, weight-normed matrices)
// forward = the renderer's colour tile; backward recomputes it and runs all six products on the matrix pipe:
//   dh2 = Wc3^T do3 (4 MFMA), dh1 = Wc2^T dh2 (64), dinp = Wc1^T dh1 (32; lane (n, g) receives d sdf_out[4g..4g+3] and d normal_g:
//   exactly the layouts sdf_stencil_bwd_kernel and the caller read), dWc3 += do3 h2^T (16), dWc2 += dh2 h1^T (64),
//   dWc1 += dh1 inp^T (32) with K = the 16 samples of the tile through LDS transposes.
#ifndef AC_COLORBWD_BF16
#define AC_COLORBWD_BF16 1    // round 4: layer 3 of the forward recomputation and the two data-gradient products of color_bwd_kernel (dh1 = Wc2^T dh2, dinp = Wc1^T dh1)
#endif                        // as three-term bf16 splits instead of fp32 MFMA.  Layers 1 and 2 of the recomputation stay fp32: they decide the ReLU masks, and a
                              // recomputation that is only 2^-16 accurate flips enough of them to show (dWc1 3e-3 of max against the fp64 oracle, measured);
                              // the weight gradients (K = the tile's samples, through LDS) stay fp32 as well
#if AC_COLORBWD_BF16
// layer 3 of the forward: its two bf16 hi and two lo fragments (k blocks 0, 1; order of color_fast_weight) take the place of the fp32 ones
constexpr int OFF_C3H = OFF_C3F, OFF_C3LO = OFF_C3F + 2 * 64 * 4;
constexpr int OFF_C3T = OFF_WAVE;                  // [4 tiles][64]                       fp32 A fragments of Wc3^T (4 MFMA: not worth splitting)
constexpr int OFF_C2T = OFF_C3T + 4 * 64;          // [4 tiles][2 k blocks][64][4] hi | lo   Wc2^T, v_mfma_f32_16x16x32_bf16 order
constexpr int OFF_C2TL = OFF_C2T + 8 * 64 * 4;
constexpr int OFF_C1T = OFF_C2TL + 8 * 64 * 4;     // [2 tiles][2 k blocks][64][4] hi | lo   Wc1^T (rows: sdf_out[16] | normal, coordinate)
constexpr int OFF_C1TL = OFF_C1T + 4 * 64 * 4;
constexpr int OFF_CW = OFF_C1TL + 4 * 64 * 4;      // per-wave slabs
#else
constexpr int OFF_C3T = OFF_WAVE;                  // [4 tiles][64]              A fragments of Wc3^T
constexpr int OFF_C2T = OFF_C3T + 4 * 64;          // [4 tiles][16 ksteps][64]   Wc2^T
constexpr int OFF_C1T = OFF_C2T + 64 * 64;         // [2 tiles][16 ksteps][64]   Wc1^T (rows: sdf_out[16] | normal, coordinate)
constexpr int OFF_CW = OFF_C1T + 32 * 64;          // per-wave slabs
#endif
constexpr int CS_H1 = 0, CS_H2 = 64 * TLD, CS_D1 = 128 * TLD, CS_D2 = 192 * TLD, CS_O3 = 256 * TLD, CS_IN = 272 * TLD;
constexpr int COLOR_SLAB = ((CS_IN + 32 * TLD + 3) / 4) * 4;
constexpr int CBWD_LDS_FLOATS = OFF_CW + TW * COLOR_SLAB;
static_assert(CBWD_LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
constexpr int NPART_C = 64 * 32 + 64 * 64 + 16 * 64;     // dWc1 [64][32] (columns 0..20 = x, n, feat) | dWc2 [64][64] | dWc3 [16][64] (rows 0..2)

__global__ __launch_bounds__(TBLOCK) void color_fwd_kernel(const RenderArgs a, const float *__restrict__ x, const float *__restrict__ nrm,
                                                           const float *__restrict__ sdf16, uint32_t B, float *__restrict__ rgb_out)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    fill_lds(lds, a);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    const uint32_t ntiles = (B + 15) / 16;
    for (uint32_t tile = blockIdx.x * TW + wave; tile < ntiles; tile += gridDim.x * TW) {
        const uint32_t b = tile * 16 + n, bb = b < B ? b : B - 1;
        const f32x4 so = *reinterpret_cast<const f32x4 *>(sdf16 + (size_t)bb * 16 + 4 * g);
        float rgb[3];
        color_tile(lds, lane, x[3 * (size_t)bb], x[3 * (size_t)bb + 1], x[3 * (size_t)bb + 2], nrm[3 * (size_t)bb], nrm[3 * (size_t)bb + 1],
                   nrm[3 * (size_t)bb + 2], so, rgb);
        if (b < B && g == 0) { rgb_out[3 * (size_t)b] = rgb[0]; rgb_out[3 * (size_t)b + 1] = rgb[1]; rgb_out[3 * (size_t)b + 2] = rgb[2]; }
    }
}

// ---- field evaluation on PACKED samples: what stands between the occupancy-grid marcher and the packed compositor (run_cuda) -------------------
// One tile = 16 samples of whatever rays (the marcher lays a ray's samples out consecutively): the same stencil gather, the same seven SDF MLP
// passes, the same colour tile and the same NeuS alpha arithmetic as the final pass of render_rays_kernel (render_fused.hip) -- a sample gets the
// bits here that it would get there for the same point, direction and section length.
struct SampleArgs {
    const float *xyzs, *dirs, *deltas;          // [M,3] [M,3] [M * dstride] (march_rays_train: dstride 1; march_rays: dstride 2, column 0 = the step)
    uint32_t dstride, M;
    float *alpha, *rgb, *normal;                // [M] [M,3] [M,3]
    float *sdf, *gradient;                      // optional [M] [M,3] (the raw finite-difference gradient: eikonal term)
};

__global__ __launch_bounds__(FBLOCK) void field_samples_kernel(const RenderArgs a, const SampleArgs s)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    fill_lds(lds, a);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    float *fsl = lds + OFF_WAVE + wave * FE_SLAB;
    const FieldCtx fc = make_ctx(a);
    const float inv_s = a.inv_s_dev ? *a.inv_s_dev : a.inv_s;
    const float bound = a.bound, eps = a.eps;
    const uint32_t ntiles = (s.M + 15) / 16;
    for (uint32_t tile = blockIdx.x * FW + wave; tile < ntiles; tile += gridDim.x * FW) {
        const uint32_t b = tile * 16 + n, bb = b < s.M ? b : s.M - 1;
        const float px = clampf(s.xyzs[3 * (size_t)bb], -bound, bound), py = clampf(s.xyzs[3 * (size_t)bb + 1], -bound, bound),
                    pz = clampf(s.xyzs[3 * (size_t)bb + 2], -bound, bound);                                          // new_pts.clamp(-bound, bound)
        const float dx = s.dirs[3 * (size_t)bb], dy = s.dirs[3 * (size_t)bb + 1], dz = s.dirs[3 * (size_t)bb + 2];
        const float delta = s.deltas[(size_t)bb * s.dstride];
        float fe0[4][2];
        encode_stencil(lds, fsl, fc, lane, px, py, pz, eps, fe0);
        f32x4 oc; float gr[3];
        fd_forward(lds, fsl, lane, px, py, pz, eps, bound, fe0, oc, gr);
        const float gx = gr[0], gy = gr[1], gz = gr[2];
        const float gn = __builtin_sqrtf((gx * gx + gy * gy) + gz * gz);
        const float nx = gx / (1e-5f + gn), ny = gy / (1e-5f + gn), nz = gz / (1e-5f + gn);
        float rgb[3];
        if (a.Wsh) {                                                 // use_viewdirs: the layer-1 bias of THIS sample's direction (wave-uniform branch)
            wave_sync();                                             // (every lane is done with the feature slab)
            sample_sh_bias(fsl, a.Wsh, dx, dy, dz, lane);
            color_tile(lds, lane, px, py, pz, nx, ny, nz, oc, rgb, fsl + 4 * lane, 256);
        } else color_tile(lds, lane, px, py, pz, nx, ny, nz, oc, rgb);
        // NeuS alpha, instant_nsr.py:219-243 with the marcher's step as the section length
        const float sdf0 = oc[0];
        const float tc = (dx * nx + dy * ny) + dz * nz;
        const float a1 = dv_softplus100(lds + OFF_SPQ, -tc * 0.5f + 0.5f) * a.one_m_car;
        const float a2 = dv_softplus100(lds + OFF_SPQ, -tc) * a.car;
        const float iter_cos = -(a1 + a2);
        const float half = iter_cos * delta * 0.5f;
        const float pc = dv_sigmoid((sdf0 - half) * inv_s), nc = dv_sigmoid((sdf0 + half) * inv_s);
        const float alpha = clampf((pc - nc + 1e-5f) / (pc + 1e-5f), 0.0f, 1.0f);
        if (b < s.M && g == 0) {
            s.alpha[b] = alpha;
            s.rgb[3 * (size_t)b] = rgb[0]; s.rgb[3 * (size_t)b + 1] = rgb[1]; s.rgb[3 * (size_t)b + 2] = rgb[2];
            s.normal[3 * (size_t)b] = nx; s.normal[3 * (size_t)b + 1] = ny; s.normal[3 * (size_t)b + 2] = nz;
            if (s.sdf) s.sdf[b] = sdf0;
            if (s.gradient) { s.gradient[3 * (size_t)b] = gx; s.gradient[3 * (size_t)b + 1] = gy; s.gradient[3 * (size_t)b + 2] = gz; }
        }
        wave_sync();
    }
}

// ---- the occupancy-grid INFERENCE render as one launch (round 4): march + field + composite per ray, no host round trips -----------------------------
// What NeRFRenderer.run_cuda's eval() loop computes in rounds of compact_rays / march_rays / ac_field_samples / composite_rays (one 4-byte D2H per round),
// computed as if it were ONE round with n_step = 1024: lane = ray.  Per iteration every alive lane marches to its NEXT occupied sample (the body of
// march_rays_kernel, raymarching.hip), the samples of the wave's alive rays are packed into tiles of 16 (ballot ranks through an LDS stage), the tiles go through
// the renderer's stencil gather / MLP / colour / alpha code (the body of field_samples_kernel), and every lane composites its own sample in order (the body of
// composite_rays_kernel: T = 1 - weights_sum, early stop at T < 1e-2).  Bit-identical to the three stand-alone operators run with n_step = 1024.
constexpr int OC_STAGE = 10 * 64;                  // per-wave stage: 64 sample slots x 8 floats (in: x y z dt . . . dl1 | out: alpha r g b nx ny nz, dl1 kept) + slot map [64] + lane map [64]
constexpr int OCC_LDS_FLOATS = FWD_LDS_FLOATS + FW * OC_STAGE;
static_assert(OCC_LDS_FLOATS * 4 + 256 <= 160 * 1024, "LDS budget (+ the training form's static words)");
struct OccArgs {
    const float *rays_o, *rays_d, *grid;
    uint32_t N, H;
    float mean_density;
    float *weights_sum, *depth, *image, *normal_map;     // [N] [N] [N,3] [N,3]: accumulators as composite_rays leaves them (background / depth normalisation: the caller)
    uint32_t *n_samples;                                   // optional [1]: total samples evaluated (atomic, one add per wave)
    uint32_t glog;                                         // a wave marches 2^glog rays at a time (lanes 0 .. 2^glog - 1); the 64 sample slots of an iteration (4 tiles) are
                                                           // shared out among the rays still alive: 64 / alive each -- the last, long rays of a group get whole tiles
    uint32_t max_steps;                                    // a ray stops after this many samples (run_cuda's max_steps; the loop of rounds stops at the first round that
                                                           // brings its step count to >= max_steps, i.e. after max_steps .. max_steps + 7 samples); 0 = no cap
    uint32_t edge_tab;                                     // 1: H + 1 floats of LDS behind the stages hold the voxel faces (rm_skip_target_tab)
    const uint32_t *run_if;                                // NULL, or a device word: the launch does nothing unless it is non-zero (the barrier-free answer to a phased
                                                           // launch whose grid barrier timed out: queued behind it unconditionally, a few microseconds when not needed)
};

__global__ __launch_bounds__(FBLOCK) void occupancy_render_kernel(const RenderArgs a, const OccArgs oc)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (oc.run_if && __hip_atomic_load(oc.run_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;      // (uniform over the grid: written before this launch started)
    fill_lds(lds, a);
    const float *etab = nullptr;
    if (oc.edge_tab) {
        RayCtx c0{}; c0.H = oc.H; c0.bound = a.bound;
        for (uint32_t m = threadIdx.x; m <= oc.H; m += FBLOCK) lds[OCC_LDS_FLOATS + m] = rm_edge(c0, m);
        etab = lds + OCC_LDS_FLOATS;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    float *fsl = lds + OFF_WAVE + wave * FE_SLAB;
    float *stage = lds + FWD_LDS_FLOATS + wave * OC_STAGE;
    uint32_t *slotmap = reinterpret_cast<uint32_t *>(stage + 8 * 64), *lanemap = slotmap + 64;
    const FieldCtx fc = make_ctx(a);
    const float inv_s = a.inv_s_dev ? *a.inv_s_dev : a.inv_s;
    const float bound = a.bound, eps = a.eps;
    const uint32_t gsz = 1u << oc.glog;
    const uint32_t ngroups = (oc.N + gsz - 1) >> oc.glog;
    uint32_t evaluated = 0;
    // groups are dealt to the workgroups first, to the waves of a workgroup second: a small batch spreads over the compute units instead of filling few of them
    for (uint32_t grp = (uint32_t)wave * gridDim.x + blockIdx.x; grp < ngroups; grp += gridDim.x * FW) {
        const uint32_t ray = (grp << oc.glog) + (uint32_t)lane;
        const bool mine = (uint32_t)lane < gsz && ray < oc.N;
        bool alive = mine;
        const uint32_t rr = mine ? ray : oc.N - 1;
        RayCtx c; rm_setup(c, oc.rays_o + 3 * (size_t)rr, oc.rays_d + 3 * (size_t)rr, oc.grid, oc.mean_density, bound, oc.H);
        float near, far;
        cube_near_far(c.ox, c.oy, c.oz, c.dx, c.dy, c.dz, bound, near, far);      // near_far_from_bound(type='cube'), instant_nsr.py:58-77 (what run_cuda passes to march_rays)
        float t = near, last_t = near, tc = near;                                  // marcher's t | its last_t | the compositor's t (rays_t)
        float skip_tt = RM_NO_SKIP;                                                 // the walk's pending skip target (rm_march_batch)
        float ws = 0.0f, dep = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f, mx = 0.0f, my = 0.0f, mz = 0.0f;
        uint32_t taken = 0;                                                         // samples this ray has marched so far
        while (__ballot(alive) != 0ull) {
            // ---- march: this lane's next (up to K) occupied samples into its own slots (march_rays_kernel's loop body).  K = 64 / (rays of the group still
            // alive): a group starts with few samples per ray and iteration and ends with whole tiles for its last, longest rays ----
            const unsigned long long am = __ballot(alive);
            const uint32_t na = (uint32_t)__builtin_popcountll(am), K = 64u / na;
            const uint32_t arank = (uint32_t)__builtin_popcountll(am & ((1ull << lane) - 1ull)), mybase = arank * K;
            if (alive) lanemap[arank] = (uint32_t)lane;
            uint32_t mycnt = 0;
            if (alive) {
                float *sp = stage + 8 * mybase;
                uint32_t room = K;
                if (oc.max_steps && oc.max_steps - taken < room) room = oc.max_steps - taken;     // (taken < max_steps while the ray is alive)
                const uint32_t room0 = room;
                auto emit = [&](float x, float y, float z, float dt, float t_after, uint32_t) {
                    sp[0] = x; sp[1] = y; sp[2] = z; sp[3] = dt; sp[7] = t_after - last_t;
                    last_t = t_after;
                    sp += 8;
                };
                bool more = true;                                                   // grid look-ups RM_BATCH at a time (rm_march_batch): the reference's walk, fewer round trips
                uint32_t kpos = 0;
                while (room > 0 && more) more = rm_march_batch<RM_BATCH>(c, t, skip_tt, far, room, kpos, emit, etab);
                mycnt = room0 - room; taken += mycnt;
                if (!more) alive = false;                                           // t >= far: composite_rays would meet dl[0] == 0 here
                if (oc.max_steps && taken >= oc.max_steps) alive = false;           // the samples staged this iteration are still composited below
            }
            wave_sync();
            // slot s = lane: it belongs to the (s / K)-th alive ray and is valid if that ray produced more than s % K samples this iteration
            const uint32_t orank = (uint32_t)lane / K, owner = orank < na ? lanemap[orank] : 0u;
            const uint32_t owner_cnt = (uint32_t)__shfl((int)mycnt, (int)owner);
            const bool valid = orank < na && ((uint32_t)lane - orank * K) < owner_cnt;
            const unsigned long long vm = __ballot(valid);
            if (vm == 0ull) break;
            const uint32_t cnt = (uint32_t)__builtin_popcountll(vm);
            if (valid) slotmap[__builtin_popcountll(vm & ((1ull << lane) - 1ull))] = (uint32_t)lane;
            wave_sync();
            evaluated += cnt;
            // ---- field on the packed samples, tiles of 16 (field_samples_kernel's body) ----
            for (uint32_t q0 = 0; q0 < cnt; q0 += 16) {
                const uint32_t ci = q0 + (uint32_t)n;
                const uint32_t slot = slotmap[ci < cnt ? ci : cnt - 1];
                const float *sp = stage + 8 * slot;
                const float px = clampf(sp[0], -bound, bound), py = clampf(sp[1], -bound, bound), pz = clampf(sp[2], -bound, bound), delta = sp[3];
                const int src = (int)lanemap[slot / K];
                const float dx = __shfl(c.dx, src), dy = __shfl(c.dy, src), dz = __shfl(c.dz, src);
#ifdef AC_OCC_NOFIELD      // timing ablation: no field evaluation (what the marching and the bookkeeping cost alone)
                const float nx = dx, ny = dy, nz = dz, alpha = 0.02f * delta / (delta + 1e-3f) + 0.0f * (px + py + pz + inv_s);
                float rgb[3] = { 0.5f, 0.5f, 0.5f };
#else
                float fe0[4][2];
                encode_stencil(lds, fsl, fc, lane, px, py, pz, eps, fe0);
                f32x4 o16; float gr[3];
                fd_forward(lds, fsl, lane, px, py, pz, eps, bound, fe0, o16, gr);
                const float gx = gr[0], gy = gr[1], gz = gr[2];
                const float gn = __builtin_sqrtf((gx * gx + gy * gy) + gz * gz);
                const float nx = gx / (1e-5f + gn), ny = gy / (1e-5f + gn), nz = gz / (1e-5f + gn);
                float rgb[3];
                if (a.Wsh) {
                    wave_sync();
                    sample_sh_bias(fsl, a.Wsh, dx, dy, dz, lane);
                    color_tile(lds, lane, px, py, pz, nx, ny, nz, o16, rgb, fsl + 4 * lane, 256);
                } else color_tile(lds, lane, px, py, pz, nx, ny, nz, o16, rgb);
                const float tcos = (dx * nx + dy * ny) + dz * nz;
                const float a1 = dv_softplus100(lds + OFF_SPQ, -tcos * 0.5f + 0.5f) * a.one_m_car;
                const float a2 = dv_softplus100(lds + OFF_SPQ, -tcos) * a.car;
                const float half = -(a1 + a2) * delta * 0.5f;
                const float pc = dv_sigmoid((o16[0] - half) * inv_s), nc = dv_sigmoid((o16[0] + half) * inv_s);
                const float alpha = clampf((pc - nc + 1e-5f) / (pc + 1e-5f), 0.0f, 1.0f);
#endif
                wave_sync();                                                         // every lane has read its inputs: the slots become outputs
                if (g == 0 && ci < cnt) {
                    float *so = stage + 8 * slot;
                    so[0] = alpha; so[1] = rgb[0]; so[2] = rgb[1]; so[3] = rgb[2]; so[4] = nx; so[5] = ny; so[6] = nz;
                }
                wave_sync();
            }
            // ---- composite: every lane its own samples, in order (composite_rays_kernel's loop body) ----
            for (uint32_t k = 0; k < mycnt; ++k) {
                const float *so = stage + 8 * (mybase + k);
                const float alpha = so[0], T = 1 - ws, w = alpha * T;
                ws += w;
                tc += so[7];
                dep += w * tc;
                cr += w * so[1]; cg += w * so[2]; cb += w * so[3];
                mx += w * so[4]; my += w * so[5]; mz += w * so[6];
                if ((double)T < 1e-2) { alive = false; break; }
            }
            wave_sync();
        }
        if (mine) {
            oc.weights_sum[ray] = ws; oc.depth[ray] = dep;
            oc.image[3 * (size_t)ray] = cr; oc.image[3 * (size_t)ray + 1] = cg; oc.image[3 * (size_t)ray + 2] = cb;
            oc.normal_map[3 * (size_t)ray] = mx; oc.normal_map[3 * (size_t)ray + 1] = my; oc.normal_map[3 * (size_t)ray + 2] = mz;
        }
    }
    if (oc.n_samples && lane == 0 && evaluated) atomicAdd(oc.n_samples, evaluated);
}

// ---- the occupancy-grid TRAINING render without autograd as one launch (round 5) -------------------------------------------------------------------
// What NeRFRenderer.run_cuda's train() branch computes under torch.no_grad() -- stylize.py's render_val of a cuda_ray network, which never leaves train mode --
// as march_rays_train (count, scan, write) / ac_field_samples / two composite_rays_train / a dozen torch kernels for the eikonal term and the background.
// One persistent workgroup per compute unit at most (so that every workgroup is resident), four phases separated by grid barriers:
//   A  lane = ray: the walk (march_count_kernel's), counting the occupied steps and recording their positions in the ray's recurrence (RayRecorder); per
//      256 rays a total (integer atomics: order-free);
//   B  lane = ray: the ray's offset in the packed layout = counter[0] + totals of the chunks before + counts before it in its chunk, the reference's budget
//      rule (a ray whose samples would end at or beyond M is left out: raymarching.cu:133), and the samples written by REPLAYING the recurrence at the
//      recorded positions (no second walk);
//   C  tile = 16 consecutive packed samples, dealt to ALL waves (a tile through the field code is ~45 us of latency: a wave that kept its own rays' tiles to
//      itself -- the first form of this kernel, r05_experiments.txt section 8c -- ran three or four in a row while most of the device idled): the body of
//      field_samples_kernel, plus the eikonal term's partial sums;
//   D  lane = ray: composite_train_fwd_kernel's loop for image and normal map on the same weights, background.
// weights_sum / image / normal_map: the bits of the chain of operators (tests/test_gpu_run_cuda.py).  gradient_error: the same terms summed in double
// in a fixed order (per lane, per wave, per workgroup; the last workgroup to leave adds the partials) instead of torch's fp32 tree: equal to ~1e-6 relative.
constexpr uint32_t OT_CHUNK_LOG = 8;                     // 2^8 rays per chunk total (a multiple of the wave's 64)
// bound of a grid barrier's spin, in ticks of the 100 MHz wall clock: two seconds unless AC_OCC_BARRIER_MS or ac_set_occupancy_barrier_ms says otherwise
static std::atomic<uint32_t> g_barrier_ms{0};            // 0 = the default
static uint32_t ot_default_ms()
{
    static const uint32_t d = []() { const char *e = getenv("AC_OCC_BARRIER_MS"); const long ms = e ? atol(e) : 0; return (uint32_t)(ms > 0 ? ms : 2000); }();
    return d;
}
static unsigned long long ot_spin_ticks()
{
    const uint32_t ms = g_barrier_ms.load(std::memory_order_relaxed);
    return (unsigned long long)(ms ? ms : ot_default_ms()) * 100000ull;
}
struct OccTrainArgs {
    const float *rays_o, *rays_d, *grid;
    uint32_t N, H, M_write, M_comp, perturb;             // M_write: capacity of the packed layout (march_write's budget, > 0); M_comp: the compositor's
    float mean_density;
    int32_t *counter;                                    // optional [2]: += samples of all rays, += N (march_rays_train's step counter)
    float *weights_sum, *image, *normal_map, *gradient_error;
    const float *bg; uint32_t bg_mode; float bg_value;   // image += (1 - weights_sum) * bg:  0 none, 1 bg_value, 2 bg[3], 3 bg[N][3]
    uint32_t *sync;                                      // [16]; 0 - 3 zero on entry and on exit: arrivals, departures, barrier failure, samples written; 8: failed launches (sticky)
    int32_t *chunk_tot, *counts, *offs, *ovf;            // [chunks] zero on entry and on exit | [N] | [N] offset of a written ray, else -1 | [N]
    uint32_t *wmask, *rec;                               // [N] | [N][RM_REC_WORDS]: the samples' positions (RayRecorder)
    double *partials;                                    // [gridDim.x][2]
    int32_t *p_ray; float *p_in, *p_out;                 // packed samples: ray [M] | x y z dt [M][4] | alpha r g b nx ny nz - [M][8]
    unsigned long long spin_ticks;                       // bound of a grid barrier's spin (ot_spin_ticks)
};

__device__ __forceinline__ int32_t ot_load(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// every workgroup has arrived `phase` + 1 times (false: timed out after `ticks` of the 100 MHz wall clock -- a foreign kernel held compute units for that long:
// the launch itself cannot be too large, ac::launch_resident sized it -- the caller gives up instead of hanging; the host re-renders, see the launch sites)
__device__ __forceinline__ bool ot_barrier(uint32_t *sync, uint32_t phase, uint32_t *flag, unsigned long long ticks)
{
    // every wave first waits until ITS OWN stores have been written to its XCD's L2 (s_waitcnt vmcnt(0): ADVICE round 5 -- outside tgsplit mode the
    // workgroup-scope release inside __syncthreads() need not wait for them), then ONE thread writes that L2's dirty lines back to where the other XCDs see them
    // (agent-scope release) -- once per workgroup.  Measured: that fence by lane 0 of every wave instead costs the 65 536-ray inference launch 1.08 -> 1.39 ms
    // and the training form 0.335 -> 0.449 (eight L2 write-backs + invalidations per workgroup and barrier); by every thread 0.537 (round 5).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __threadfence();
    if (threadIdx.x == 0) {
        atomicAdd(&sync[0], 1u);
        const uint32_t want = (phase + 1u) * gridDim.x;
        const unsigned long long t0 = wall_clock64();
        bool all = false;
        while (!(all = __hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) && wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(10);
        *flag = all ? 1u : 0u;
        if (!*flag) atomicExch(&sync[2], 1u);
        __threadfence();                                 // acquire: the compute unit's L1 (shared by the workgroup's waves) and the L2's copies of other XCDs' lines are dropped
    }
    __syncthreads();
    return *flag != 0u;
}

__global__ __launch_bounds__(FBLOCK) void occupancy_train_kernel(const RenderArgs a, const OccTrainArgs oc)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ uint32_t bar_flag, last;
    __shared__ double red[2 * FW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    const float bound = a.bound, eps = a.eps;
    const int32_t base = oc.counter ? oc.counter[0] : 0;                  // (read before the first barrier: the last workgroup to leave updates it)
    const uint32_t nchunks = (oc.N + (1u << OT_CHUNK_LOG) - 1) >> OT_CHUNK_LOG;
    const uint32_t wid = (uint32_t)wave * gridDim.x + blockIdx.x, nwaves = gridDim.x * FW;   // work is dealt to the workgroups first, to a workgroup's waves second:
                                                                                             // a 4096-ray batch is one walking wave on each of 64 compute units, not eight on eight
    bool ok = true;
    // ---- A: counts and records ----
    const float *etab = nullptr;                                            // the voxel faces (rm_skip_target_tab) where the weights will be: H + 1 floats
    if (oc.H < (uint32_t)RM_EDGE_MAX) {
        RayCtx c0{}; c0.H = oc.H; c0.bound = bound;
        for (uint32_t m = threadIdx.x; m <= oc.H; m += FBLOCK) lds[m] = rm_edge(c0, m);
        etab = lds;
    }
    __syncthreads();
    for (uint32_t r0 = wid * 64u; r0 < oc.N; r0 += nwaves * 64u) {         // (a wave's 64 consecutive rays lie in one chunk)
        const uint32_t ray = r0 + (uint32_t)lane;
        int32_t cnt = 0;
        if (ray < oc.N) {
            RayCtx c; rm_setup(c, oc.rays_o + 3 * (size_t)ray, oc.rays_d + 3 * (size_t)ray, oc.grid, oc.mean_density, bound, oc.H);
            float near, far; rm_near_far(c, near, far);
            float t = ray_t0(c, near, ray, oc.perturb), skip_tt = RM_NO_SKIP;
            uint32_t room = RM_MAX_STEPS, kpos = 0;
            RayRecorder rr; rr.begin(oc.rec + (size_t)ray * RM_REC_WORDS);
            while (room > 0 && rm_march_batch<RM_BATCH, true>(c, t, skip_tt, far, room, kpos, [&](float, float, float, float, float, uint32_t k) { rr.add(k); }, etab)) {}
            rr.end();
            cnt = (int32_t)(RM_MAX_STEPS - room);
            oc.counts[ray] = cnt; oc.ovf[ray] = rr.ovf ? 1 : 0; oc.wmask[ray] = rr.wmask;
        }
        int32_t tot = cnt;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) tot += __shfl_xor(tot, d);
        if (lane == 0 && tot) atomicAdd(&oc.chunk_tot[r0 >> OT_CHUNK_LOG], tot);
    }
    __syncthreads();                                                        // (the face table is done with: the weights take its place)
    fill_lds(lds, a);
    ok = ot_barrier(oc.sync, 0, &bar_flag, oc.spin_ticks);
    // ---- B: offsets, the budget rule, the packed samples ----
    for (uint32_t r0 = wid * 64u; r0 < oc.N && ok; r0 += nwaves * 64u) {
        const uint32_t ray = r0 + (uint32_t)lane;
        const bool mine = ray < oc.N;
        int32_t pre = 0;
        {
            const uint32_t c0 = r0 >> OT_CHUNK_LOG, s0 = c0 << OT_CHUNK_LOG;
            for (uint32_t cix = (uint32_t)lane; cix < c0; cix += 64) pre += oc.chunk_tot[cix];
            for (uint32_t i = s0 + (uint32_t)lane; i < r0; i += 64) pre += oc.counts[i];
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) pre += __shfl_xor(pre, d);
        }
        const int32_t cnt = mine ? oc.counts[ray] : 0;
        int32_t inc = cnt;                                                  // inclusive scan over the wave's rays
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int32_t v = __shfl_up(inc, d); if (lane >= d) inc += v; }
        const uint32_t offset = (uint32_t)(base + pre + (inc - cnt)), end = offset + (uint32_t)cnt;
        const bool written = mine && cnt > 0 && end < oc.M_write;           // march_write_kernel: `point_index + num_steps >= M -> return`
        if (mine) oc.offs[ray] = written ? (int32_t)offset : -1;
        uint32_t wend = written ? end : 0u;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { const uint32_t v = (uint32_t)__shfl_xor((int)wend, d); wend = v > wend ? v : wend; }
        if (lane == 0 && wend) atomicMax(&oc.sync[3], wend);
        if (written) {
            RayCtx c; rm_setup(c, oc.rays_o + 3 * (size_t)ray, oc.rays_d + 3 * (size_t)ray, oc.grid, oc.mean_density, bound, oc.H);
            float near, far; rm_near_far(c, near, far);
            const float t0 = ray_t0(c, near, ray, oc.perturb);
            float *pi = oc.p_in + 4 * (size_t)offset; int32_t *pr = oc.p_ray + offset;
            auto put = [&](float x, float y, float z, float dt) { pi[0] = x; pi[1] = y; pi[2] = z; pi[3] = dt; pi += 4; *pr++ = (int32_t)ray; };
            if (!oc.ovf[ray]) rm_replay(c, t0, oc.rec + (size_t)ray * RM_REC_WORDS, oc.wmask[ray], (uint32_t)cnt, put);
            else {
                float t = t0, skip_tt = RM_NO_SKIP; uint32_t room = (uint32_t)cnt, kpos = 0;
                while (room > 0 && rm_march_batch<RM_BATCH>(c, t, skip_tt, far, room, kpos, [&](float x, float y, float z, float dt, float, uint32_t) { put(x, y, z, dt); })) {}
            }
        }
    }
    ok = ok && ot_barrier(oc.sync, 1, &bar_flag, oc.spin_ticks);
    // ---- C: the field on the packed samples, tiles dealt to all waves ----
    double e_num = 0.0, e_den = 0.0;                                       // this lane's share of sum(relax * gerr), sum(relax)
    {
        float *fsl = lds + OFF_WAVE + wave * FE_SLAB;
        const FieldCtx fc = make_ctx(a);
        const float inv_s = a.inv_s_dev ? *a.inv_s_dev : a.inv_s;
        const uint32_t W = ok ? __hip_atomic_load(&oc.sync[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const uint32_t ntiles = (W + 15u) / 16u;
        for (uint32_t tile = wid; tile < ntiles; tile += nwaves) {
            const uint32_t b = tile * 16 + (uint32_t)n, bb = b < W ? b : W - 1;
            const float4 in = *reinterpret_cast<const float4 *>(oc.p_in + 4 * (size_t)bb);
            const int32_t ray = oc.p_ray[bb];
            const float sx = in.x, sy = in.y, sz = in.z, delta = in.w;
            const float px = clampf(sx, -bound, bound), py = clampf(sy, -bound, bound), pz = clampf(sz, -bound, bound);
            const float dx = oc.rays_d[3 * (size_t)ray], dy = oc.rays_d[3 * (size_t)ray + 1], dz = oc.rays_d[3 * (size_t)ray + 2];
            float fe0[4][2];
            encode_stencil(lds, fsl, fc, lane, px, py, pz, eps, fe0);
            f32x4 o16; float gr[3];
            fd_forward(lds, fsl, lane, px, py, pz, eps, bound, fe0, o16, gr);
            const float gx = gr[0], gy = gr[1], gz = gr[2];
            const float gn = __builtin_sqrtf((gx * gx + gy * gy) + gz * gz);
            const float nx = gx / (1e-5f + gn), ny = gy / (1e-5f + gn), nz = gz / (1e-5f + gn);
            float rgb[3];
            if (a.Wsh) {
                wave_sync();
                sample_sh_bias(fsl, a.Wsh, dx, dy, dz, lane);
                color_tile(lds, lane, px, py, pz, nx, ny, nz, o16, rgb, fsl + 4 * lane, 256);
            } else color_tile(lds, lane, px, py, pz, nx, ny, nz, o16, rgb);
            const float tcos = (dx * nx + dy * ny) + dz * nz;
            const float a1 = dv_softplus100(lds + OFF_SPQ, -tcos * 0.5f + 0.5f) * a.one_m_car;
            const float a2 = dv_softplus100(lds + OFF_SPQ, -tcos) * a.car;
            const float half = -(a1 + a2) * delta * 0.5f;
            const float pc = dv_sigmoid((o16[0] - half) * inv_s), nc = dv_sigmoid((o16[0] + half) * inv_s);
            const float alpha = clampf((pc - nc + 1e-5f) / (pc + 1e-5f), 0.0f, 1.0f);
            if (g == 0 && b < W) {
                // eikonal term over the packed samples (instant_nsr.py:266-272): relax = |x| < 1.2 on the marcher's point, (|gradient| - 1)^2
                if (__builtin_sqrtf((sx * sx + sy * sy) + sz * sz) < 1.2f) { const float d1 = gn - 1.0f; e_num += (double)(d1 * d1); e_den += 1.0; }
                float *po = oc.p_out + 8 * (size_t)b;
                *reinterpret_cast<float4 *>(po) = make_float4(alpha, rgb[0], rgb[1], rgb[2]);
                *reinterpret_cast<float4 *>(po + 4) = make_float4(nx, ny, nz, 0.0f);
            }
            wave_sync();
        }
    }
    ok = ok && ot_barrier(oc.sync, 2, &bar_flag, oc.spin_ticks);
    // ---- D: the packed compositor per ray (composite_train_fwd_kernel's loop), image and normal map on the same weights; background ----
    for (uint32_t r0 = wid * 64u; r0 < oc.N && ok; r0 += nwaves * 64u) {
        const uint32_t ray = r0 + (uint32_t)lane;
        if (ray >= oc.N) continue;
        const int32_t off = oc.offs[ray], cnt = oc.counts[ray];
        const bool composited = off >= 0 && (uint32_t)off + (uint32_t)cnt < oc.M_comp;      // `offset + num_steps >= M -> zeros`
        float T = 1.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f, mx = 0.0f, my = 0.0f, mz = 0.0f;
        if (composited) {
            const float *po = oc.p_out + 8 * (size_t)off;
            for (int32_t k = 0; k < cnt; ++k, po += 8) {
                if (T < 1e-4f) break;
                const float4 u = *reinterpret_cast<const float4 *>(po), v = *reinterpret_cast<const float4 *>(po + 4);
                const float alpha = u.x, w = alpha * T;
                cr += w * u.y; cg += w * u.z; cb += w * u.w;
                mx += w * v.x; my += w * v.y; mz += w * v.z;
                T *= 1.0f - alpha;
            }
        }
        const float ws = composited ? 1.0f - T : 0.0f;
        if (oc.bg_mode) {                                                   // image + (1 - weights_sum) * bg, torch's three operations in torch's order
            const float om = 1.0f - ws;
            const size_t bo = oc.bg_mode == 3 ? 3 * (size_t)ray : 0;
            const float b0 = oc.bg_mode == 1 ? oc.bg_value : oc.bg[bo], b1 = oc.bg_mode == 1 ? oc.bg_value : oc.bg[bo + 1], b2 = oc.bg_mode == 1 ? oc.bg_value : oc.bg[bo + 2];
            cr = cr + om * b0; cg = cg + om * b1; cb = cb + om * b2;
        }
        oc.weights_sum[ray] = ws;
        oc.image[3 * (size_t)ray] = cr; oc.image[3 * (size_t)ray + 1] = cg; oc.image[3 * (size_t)ray + 2] = cb;
        oc.normal_map[3 * (size_t)ray] = mx; oc.normal_map[3 * (size_t)ray + 1] = my; oc.normal_map[3 * (size_t)ray + 2] = mz;
    }
    // ---- eikonal partials: lane -> wave -> workgroup (fixed order); the last workgroup to leave adds the workgroups' and re-arms the scratch ----
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { e_num += __shfl_xor(e_num, d); e_den += __shfl_xor(e_den, d); }
    if (lane == 0) { red[2 * wave] = e_num; red[2 * wave + 1] = e_den; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sn = 0.0, sd = 0.0;
        for (int w = 0; w < FW; ++w) { sn += red[2 * w]; sd += red[2 * w + 1]; }
        oc.partials[2 * blockIdx.x] = sn; oc.partials[2 * blockIdx.x + 1] = sd;
        __threadfence();
        last = atomicAdd(&oc.sync[1], 1u) == gridDim.x - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (wave == 0) {
        double sn = 0.0, sd = 0.0;
        for (uint32_t b = (uint32_t)lane; b < gridDim.x; b += 64) {
            sn += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long *>(oc.partials) + 2 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            sd += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long *>(oc.partials) + 2 * b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { sn += __shfl_xor(sn, d); sd += __shfl_xor(sd, d); }
        int32_t tot = 0;
        for (uint32_t cix = (uint32_t)lane; cix < nchunks; cix += 64) { tot += ot_load(oc.chunk_tot + cix); oc.chunk_tot[cix] = 0; }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) tot += __shfl_xor(tot, d);
        if (lane == 0) {
            const bool failed = __hip_atomic_load(&oc.sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            oc.gradient_error[0] = failed ? __builtin_nanf("") : (float)sn / ((float)sd + 1e-5f);
            if (oc.counter) { oc.counter[0] = base + tot; oc.counter[1] += (int32_t)oc.N; }
            if (failed) oc.sync[8] += 1u;                                   // (sticky: launches of this scratch whose barrier timed out -- never reset)
            oc.sync[0] = 0u; oc.sync[1] = 0u; oc.sync[2] = 0u; oc.sync[3] = 0u;
        }
    }
}

// ---- the occupancy-grid INFERENCE render in phases (round 5): rounds of march | field | composite inside ONE launch, grid barriers between them --------------
// occupancy_render_kernel above gives every wave 8 - 16 rays and lets it march (a quarter of its lanes), evaluate (its own tiles, one after the other: ~45 us of
// latency each) and composite them.  What the training kernel taught -- walk with every lane, deal the tiles to ALL waves -- applies here too, except that a ray's
// march depends on its composite (it stops at T < 1e-2): so the reference's loop of rounds comes back, inside the launch, without its host read-backs:
//   M  lane = alive ray (64 per wave, waves dealt over the device): up to n_step samples into the ray's slots, the slot ids appended to the round's tile list
//      (one integer atomic per wave: the order of the list does not reach any result);
//   F  the field on tiles of 16 listed samples, dealt to all waves (field_samples_kernel's body);
//   C  lane = alive ray: composite_rays_kernel's loop over the ray's samples of this round; rays that go on are appended to the next round's list.
// A ray's samples, their order and every operation on them are those of occupancy_render_kernel (and of the three stand-alone operators run as one round):
// the same bits for any n_step (tests/test_gpu_run_cuda.py); n_step only decides how many samples past a ray's last one are evaluated in vain.
struct OccPhArgs {
    const float *rays_o, *rays_d, *grid;
    uint32_t N, H, max_steps, nlog;                        // n_step = 1 << nlog samples per ray and round
    float mean_density;
    float *weights_sum, *depth, *image, *normal_map;       // accumulators, as composite_rays leaves them
    uint32_t *n_samples;                                   // optional [1]
    uint32_t *sync;                                        // [16]; 0 - 7 zero on entry and on exit: 0 arrivals, 1 departures, 2 failure, 4 - 5 rays alive, 6 - 7 samples listed; 8: failed launches (sticky)
    int32_t *alive;                                        // [2][N]
    float *st;                                             // [N][4] per ray: marcher's t, its last_t, the compositor's t, samples taken (bits)
    uint32_t *cnt;                                         // [N] per alive entry: samples of this round | bit 31: the walk ended
    uint32_t *list;                                        // [N << nlog] slot ids
    float *s_in, *s_out;                                   // [N << nlog][8]: x y z dt dl1 - - - | alpha r g b nx ny nz -
    unsigned long long spin_ticks;                         // bound of a grid barrier's spin (ot_spin_ticks)
};

__global__ __launch_bounds__(FBLOCK) void occupancy_phased_kernel(const RenderArgs a, const OccPhArgs oc)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ uint32_t bar_flag, last;
    fill_lds(lds, a);
    const float *etab = nullptr;
    if ((FWD_LDS_FLOATS + (size_t)oc.H + 1) * sizeof(float) + 64 <= 160 * 1024) {
        RayCtx c0{}; c0.H = oc.H; c0.bound = a.bound;
        for (uint32_t m = threadIdx.x; m <= oc.H; m += FBLOCK) lds[FWD_LDS_FLOATS + m] = rm_edge(c0, m);
        etab = lds + FWD_LDS_FLOATS;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    float *fsl = lds + OFF_WAVE + wave * FE_SLAB;
    const FieldCtx fc = make_ctx(a);
    const float inv_s = a.inv_s_dev ? *a.inv_s_dev : a.inv_s;
    const float bound = a.bound, eps = a.eps;
    const uint32_t wid = (uint32_t)wave * gridDim.x + blockIdx.x, nwaves = gridDim.x * FW;
    uint32_t n_al = oc.N, round = 0, phase = 0, evaluated = 0;
    bool ok = true;
    for (;;) {
        const uint32_t cur = round & 1u, nxt = cur ^ 1u;
        // samples per ray in this round: 2^oc.nlog while most rays are alive, up to 64 once the slots allow it (like the reference's n_step = N / n_alive: the
        // stragglers -- rays along a limb -- finish in a round or two instead of one round per 16 samples, and a round costs three barriers whatever its size)
        uint32_t nlog = oc.nlog;
        while (nlog < 6u && ((uint64_t)n_al << (nlog + 1u)) <= ((uint64_t)oc.N << oc.nlog)) ++nlog;
        const uint32_t nstep = 1u << nlog;
        const int32_t *L = oc.alive + (size_t)cur * oc.N;
        // ---- M: march ----
        if (blockIdx.x == 0 && threadIdx.x == 0) oc.sync[4 + nxt] = 0u;                 // (the next round's ray counter: idle until this round's phase C)
        for (uint32_t a0 = wid * 64u; a0 < n_al; a0 += nwaves * 64u) {
            const uint32_t e = a0 + (uint32_t)lane;
            const bool mine = e < n_al;
            const uint32_t ray = mine ? (round ? (uint32_t)L[e] : e) : 0u;
            RayCtx c; rm_setup(c, oc.rays_o + 3 * (size_t)ray, oc.rays_d + 3 * (size_t)ray, oc.grid, oc.mean_density, bound, oc.H);
            float near, far;
            cube_near_far(c.ox, c.oy, c.oz, c.dx, c.dy, c.dz, bound, near, far);
            float t = near, last_t = near, skip_tt = RM_NO_SKIP;
            uint32_t taken = 0;
            float4 *stp = reinterpret_cast<float4 *>(oc.st) + ray;
            if (mine) {
                if (round) { const float4 v = *stp; t = v.x; last_t = v.y; taken = __float_as_uint(v.w); }
                else {
                    oc.weights_sum[ray] = 0.0f; oc.depth[ray] = 0.0f;
                    for (int k = 0; k < 3; ++k) { oc.image[3 * (size_t)ray + k] = 0.0f; oc.normal_map[3 * (size_t)ray + k] = 0.0f; }
                }
            }
            uint32_t room = nstep, mycnt = 0;
            bool ended = false;
            if (mine) {
                if (oc.max_steps && oc.max_steps - taken < room) room = oc.max_steps - taken;
                const uint32_t room0 = room;
                float *sp = oc.s_in + 8 * ((size_t)e << nlog);
                auto emit = [&](float x, float y, float z, float dt, float t_after, uint32_t) {
                    *reinterpret_cast<float4 *>(sp) = make_float4(x, y, z, dt); sp[4] = t_after - last_t;
                    last_t = t_after; sp += 8;
                };
                bool more = true;
                uint32_t kpos = 0;
                while (room > 0 && more) more = rm_march_batch<RM_BATCH>(c, t, skip_tt, far, room, kpos, emit, etab);
                mycnt = room0 - room; taken += mycnt;
                ended = !more || (oc.max_steps && taken >= oc.max_steps);
                const float tc0 = round ? (*stp).z : near;
                *stp = make_float4(t, last_t, tc0, __uint_as_float(taken));
                oc.cnt[e] = mycnt | (ended ? 0x80000000u : 0u);
            }
            // the round's tile list: a wave reserves room for its samples with one atomic
            uint32_t inc = mycnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t v = (uint32_t)__shfl_up((int)inc, d); if (lane >= d) inc += v; }
            const uint32_t tot = (uint32_t)__shfl((int)inc, 63);
            uint32_t base = 0;
            if (lane == 0 && tot) base = atomicAdd(&oc.sync[6 + cur], tot);
            base = (uint32_t)__shfl((int)base, 0) + (inc - mycnt);
            for (uint32_t k = 0; k < mycnt; ++k) oc.list[base + k] = (e << nlog) + k;
        }
        ok = ok && ot_barrier(oc.sync, phase++, &bar_flag, oc.spin_ticks);
        // ---- F: field on the listed samples ----
        if (blockIdx.x == 0 && threadIdx.x == 0) oc.sync[6 + nxt] = 0u;                 // (the next round's list counter)
        const uint32_t nl = ok ? __hip_atomic_load(&oc.sync[6 + cur], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        for (uint32_t tile = wid; tile * 16u < nl; tile += nwaves) {
            const uint32_t q = tile * 16u + (uint32_t)n, slot = oc.list[q < nl ? q : nl - 1];
            const uint32_t e = slot >> nlog, ray = round ? (uint32_t)L[e] : e;
            const float4 in = *reinterpret_cast<const float4 *>(oc.s_in + 8 * (size_t)slot);
            const float px = clampf(in.x, -bound, bound), py = clampf(in.y, -bound, bound), pz = clampf(in.z, -bound, bound), delta = in.w;
            const float dx = oc.rays_d[3 * (size_t)ray], dy = oc.rays_d[3 * (size_t)ray + 1], dz = oc.rays_d[3 * (size_t)ray + 2];
#ifdef AC_OCC_NOFIELD      // timing ablation: no field evaluation (what the walk, the barriers and the bookkeeping cost alone)
            const float nx = dx, ny = dy, nz = dz, alpha = 0.02f * delta / (delta + 1e-3f) + 0.0f * (px + py + pz + inv_s);
            float rgb[3] = { 0.5f, 0.5f, 0.5f };
#else
            float fe0[4][2];
            encode_stencil(lds, fsl, fc, lane, px, py, pz, eps, fe0);
            f32x4 o16; float gr[3];
            fd_forward(lds, fsl, lane, px, py, pz, eps, bound, fe0, o16, gr);
            const float gx = gr[0], gy = gr[1], gz = gr[2];
            const float gn = __builtin_sqrtf((gx * gx + gy * gy) + gz * gz);
            const float nx = gx / (1e-5f + gn), ny = gy / (1e-5f + gn), nz = gz / (1e-5f + gn);
            float rgb[3];
            if (a.Wsh) {
                wave_sync();
                sample_sh_bias(fsl, a.Wsh, dx, dy, dz, lane);
                color_tile(lds, lane, px, py, pz, nx, ny, nz, o16, rgb, fsl + 4 * lane, 256);
            } else color_tile(lds, lane, px, py, pz, nx, ny, nz, o16, rgb);
            const float tcos = (dx * nx + dy * ny) + dz * nz;
            const float a1 = dv_softplus100(lds + OFF_SPQ, -tcos * 0.5f + 0.5f) * a.one_m_car;
            const float a2 = dv_softplus100(lds + OFF_SPQ, -tcos) * a.car;
            const float half = -(a1 + a2) * delta * 0.5f;
            const float pc = dv_sigmoid((o16[0] - half) * inv_s), nc = dv_sigmoid((o16[0] + half) * inv_s);
            const float alpha = clampf((pc - nc + 1e-5f) / (pc + 1e-5f), 0.0f, 1.0f);
#endif
            if (g == 0 && q < nl) {
                float *po = oc.s_out + 8 * (size_t)slot;
                *reinterpret_cast<float4 *>(po) = make_float4(alpha, rgb[0], rgb[1], rgb[2]);
                *reinterpret_cast<float4 *>(po + 4) = make_float4(nx, ny, nz, 0.0f);
            }
            wave_sync();
        }
        if (wid == 0 && lane == 0) evaluated += nl;
        ok = ok && ot_barrier(oc.sync, phase++, &bar_flag, oc.spin_ticks);
        // ---- C: composite, and who goes on ----
        for (uint32_t a0 = wid * 64u; a0 < n_al && ok; a0 += nwaves * 64u) {
            const uint32_t e = a0 + (uint32_t)lane;
            const bool mine = e < n_al;
            bool goes = false;
            uint32_t ray = 0;
            if (mine) {
                ray = round ? (uint32_t)L[e] : e;
                const uint32_t cw = oc.cnt[e], mycnt = cw & 0x7fffffffu;
                bool alive = !(cw >> 31);
                float4 *stp = reinterpret_cast<float4 *>(oc.st) + ray;
                float tc = (*stp).z;
                float ws = oc.weights_sum[ray], dep = oc.depth[ray];
                float cr = oc.image[3 * (size_t)ray], cg = oc.image[3 * (size_t)ray + 1], cb = oc.image[3 * (size_t)ray + 2];
                float mx = oc.normal_map[3 * (size_t)ray], my = oc.normal_map[3 * (size_t)ray + 1], mz = oc.normal_map[3 * (size_t)ray + 2];
                const float *pi = oc.s_in + 8 * ((size_t)e << nlog), *po = oc.s_out + 8 * ((size_t)e << nlog);
                for (uint32_t k = 0; k < mycnt; ++k, pi += 8, po += 8) {            // composite_rays_kernel's loop body
                    const float4 u = *reinterpret_cast<const float4 *>(po), v = *reinterpret_cast<const float4 *>(po + 4);
                    const float alpha = u.x, T = 1 - ws, w = alpha * T;
                    ws += w;
                    tc += pi[4];
                    dep += w * tc;
                    cr += w * u.y; cg += w * u.z; cb += w * u.w;
                    mx += w * v.x; my += w * v.y; mz += w * v.z;
                    if ((double)T < 1e-2) { alive = false; break; }
                }
                (*stp).z = tc;
                oc.weights_sum[ray] = ws; oc.depth[ray] = dep;
                oc.image[3 * (size_t)ray] = cr; oc.image[3 * (size_t)ray + 1] = cg; oc.image[3 * (size_t)ray + 2] = cb;
                oc.normal_map[3 * (size_t)ray] = mx; oc.normal_map[3 * (size_t)ray + 1] = my; oc.normal_map[3 * (size_t)ray + 2] = mz;
                goes = alive;
            }
            const unsigned long long gm = __ballot(goes);
            if (gm) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&oc.sync[4 + nxt], (uint32_t)__builtin_popcountll(gm));
                base = (uint32_t)__shfl((int)base, 0);
                if (goes) oc.alive[(size_t)nxt * oc.N + base + (uint32_t)__builtin_popcountll(gm & ((1ull << lane) - 1ull))] = (int32_t)ray;
            }
        }
        ok = ok && ot_barrier(oc.sync, phase++, &bar_flag, oc.spin_ticks);
        n_al = ok ? __hip_atomic_load(&oc.sync[4 + nxt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        if (n_al == 0u) break;
        ++round;
    }
    if (oc.n_samples && wid == 0 && lane == 0 && evaluated) atomicAdd(oc.n_samples, evaluated);
    // the last workgroup to leave re-arms the scratch
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence(); last = atomicAdd(&oc.sync[1], 1u) == gridDim.x - 1u ? 1u : 0u; }
    __syncthreads();
    if (last && threadIdx.x < 8) {
        // a barrier timed out: a NaN pixel, the sticky count (word 8) and THIS launch's verdict (word 9, rewritten by every launch): the barrier-free launch the
        // host has queued behind this one (ac_render_rays_occupancy_phased) runs if and only if it is set, and overwrites every output
        if (threadIdx.x == 2) {
            const bool failed = oc.sync[2] != 0u;
            if (failed) { oc.weights_sum[0] = __builtin_nanf(""); oc.sync[8] += 1u; if (oc.n_samples) *oc.n_samples = 0u; }
            __hip_atomic_store(&oc.sync[9], failed ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __threadfence();
        oc.sync[threadIdx.x] = 0u;
    }
}

__device__ __forceinline__ void fill_lds_color_bwd(float *lds, const RenderArgs &a)
{
    for (int e = threadIdx.x; e < 4 * 64; e += blockDim.x) {        // fragment to: lane (m, kk) = Wc3[o = kk][unit = 16 to + m]
        const int l = e & 63, to = e >> 6, m = l & 15, kk = l >> 4;
        lds[OFF_C3T + e] = kk < 3 ? a.Wc3[kk * 64 + 16 * to + m] : 0.0f;
    }
#if AC_COLORBWD_BF16
    // K order of both products = the register layout the previous product leaves its result in (slot i of lane group kk in k block s = unit
    // 16 (2s + (i >> 2)) + 4 kk + (i & 3)): no data moves between lanes.  Weights split hi + lo by round-to-nearest like fill_lds_color_fast.
    uint32_t *lw = reinterpret_cast<uint32_t *>(lds);
    for (int e = threadIdx.x; e < (8 + 4) * 64 * 4; e += blockDim.x) {
        const int q = e & 3, l = (e >> 2) & 63, f = e >> 8, m = l & 15, kk = l >> 4;
        const bool c2 = f < 8;
        const int t = c2 ? f >> 1 : (f - 8) >> 1, sblk = f & 1;
        int col = -1;                                                 // Wc1^T rows: tile 0 row m = sdf_out[m] (feat m - 1; the sdf itself is no input), tile 1 row 4 g = normal component g
        if (!c2) { if (t == 0) col = m == 0 ? -1 : 6 + (m - 1); else if ((m & 3) == 0 && (m >> 2) < 3) col = 3 + (m >> 2); }
        uint32_t hi2 = 0, lo2 = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = 2 * q + h, unit = 16 * (2 * sblk + (i >> 2)) + 4 * kk + (i & 3);
            const float w = c2 ? a.Wc2[unit * 64 + 16 * t + m] : (col < 0 ? 0.0f : a.Wc1[unit * 21 + col]);
            const uint32_t hb = bf16_rne_bits(w), lb = bf16_rne_bits(w - __uint_as_float(hb << 16));
            hi2 |= hb << (16 * h); lo2 |= lb << (16 * h);
        }
        if (c2) { lw[OFF_C2T + e] = hi2; lw[OFF_C2TL + e] = lo2; }
        else { lw[OFF_C1T + (e - 8 * 256)] = hi2; lw[OFF_C1TL + (e - 8 * 256)] = lo2; }
    }
    for (int e = threadIdx.x; e < 2 * 64 * 4; e += blockDim.x) {      // layer 3 of the forward, k blocks 0 and 1 (rows 3 .. 15 zero)
        const int q = e & 3, l = (e >> 2) & 63, sblk = e >> 8;
        uint32_t hi2 = 0, lo2 = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float w = color_fast_weight(a, 12 + sblk, l, 2 * q + h);
            const uint32_t hb = bf16_rne_bits(w), lb = bf16_rne_bits(w - __uint_as_float(hb << 16));
            hi2 |= hb << (16 * h); lo2 |= lb << (16 * h);
        }
        lw[OFF_C3H + e] = hi2; lw[OFF_C3LO + e] = lo2;
    }
    return;
#endif
#if !AC_COLORBWD_BF16
    for (int e = threadIdx.x; e < 64 * 64; e += blockDim.x) {       // fragment (t, ks = 4 to + r): lane (m, kk) = Wc2[i = 16 to + 4 kk + r][j = 16 t + m]
        const int l = e & 63, fs = e >> 6, t = fs >> 4, ks = fs & 15, to = ks >> 2, r = ks & 3, m = l & 15, kk = l >> 4;
        lds[OFF_C2T + e] = a.Wc2[(16 * to + 4 * kk + r) * 64 + 16 * t + m];
    }
    for (int e = threadIdx.x; e < 32 * 64; e += blockDim.x) {       // fragment (tp, ks = 4 t + r): lane (m, kk) = Wc1[u = 16 t + 4 kk + r][col(tp, m)]
        const int l = e & 63, fs = e >> 6, tp = fs >> 4, ks = fs & 15, t = ks >> 2, r = ks & 3, m = l & 15, kk = l >> 4;
        int col = -1;
        if (tp == 0) col = m == 0 ? -1 : 6 + (m - 1);                // row m = sdf_out[m]: feat m-1 (the sdf itself is not an input)
        else if ((m & 3) == 0 && (m >> 2) < 3) col = 3 + (m >> 2);   // row 4 g: normal component g
        lds[OFF_C1T + e] = col < 0 ? 0.0f : a.Wc1[(16 * t + 4 * kk + r) * 21 + col];
    }
#endif
}

#if AC_COLORBWD_BF16
// acc += A (bf16 hi + lo fragment pair at hi_off / lo_off, fragment f) x B (split activations): three of the four partial products
__device__ __forceinline__ f32x4 cb_mma(const float *__restrict__ lds, int hi_off, int lo_off, int f, int lane, const u32x4 &bh, const u32x4 &bl, f32x4 acc)
{
    const bf16x8 Ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(lds + hi_off + (f * 64 + lane) * 4));
    const bf16x8 Al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(lds + lo_off + (f * 64 + lane) * 4));
    const bf16x8 Bh = __builtin_bit_cast(bf16x8, bh), Bl = __builtin_bit_cast(bf16x8, bl);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, Bh, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bl, acc, 0, 0, 0);
}
// 8 fp32 values -> bf16 hi and lo parts, both ROUNDED to nearest (v_cvt_pk_bf16_f32, two values per instruction): |x - hi - lo| <= 2^-17 |x| and the
// dropped lo x lo term has either sign.  The renderer's split8_bf16 truncates both parts (2^-14, biased towards zero): good enough for an image, but
// through the five chained products of this kernel the bias showed as 6.5e-4 of max in dWc1 against the fp64 oracle (contract 3e-4); rounded: where the
// fp32 products were.  Same packing (value 2q in the low half of dword q), same instruction count.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8_bf16_rn(const float (&d)[8], u32x4 &bh, u32x4 &bl)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x2_t v = { d[2 * q], d[2 * q + 1] };
        const uint32_t hb = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
        const f32x2_t r = { d[2 * q] - __uint_as_float(hb << 16), d[2 * q + 1] - __uint_as_float(hb & 0xffff0000u) };
        bh[q] = hb;
        bl[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2_t));
    }
}
__device__ __forceinline__ void split_tiles(const f32x4 (&v)[4], u32x4 (&bh)[2], u32x4 (&bl)[2])
{
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
        const float in[8] = { v[2 * sb][0], v[2 * sb][1], v[2 * sb][2], v[2 * sb][3], v[2 * sb + 1][0], v[2 * sb + 1][1], v[2 * sb + 1][2], v[2 * sb + 1][3] };
        split8_bf16_rn(in, bh[sb], bl[sb]);
    }
}
#endif

// use_viewdirs (sh_bias != NULL): sample b belongs to ray b / T (T a multiple of 16: a tile never straddles two rays); layer 1 of the recomputed forward
// starts from the ray's bias sh_bias[ray][64] (ac_sh_bias: the forward's own bits, so the ReLU masks are the forward's), and the gradient of that bias --
// the tile's sum over its 16 samples of d h1 -- goes to g_sh_tiles[tile][64]: d Wc1_sh = sum over rays of (sum of the ray's tiles) (x) sh(d_ray) is formed by
// the caller (a [N, 64]^T x [N, 16] product: the direction is constant along a ray, there is nothing per sample to multiply).
__global__ __launch_bounds__(TBLOCK) void color_bwd_kernel(const RenderArgs a, const float *__restrict__ x, const float *__restrict__ nrm,
                                                           const float *__restrict__ sdf16, const float *__restrict__ g_rgb, uint32_t B,
                                                           float *__restrict__ g_nrm, float *__restrict__ g_sdf16, float *__restrict__ partials,
                                                           const float *__restrict__ sh_bias, uint32_t T, float *__restrict__ g_sh_tiles)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    fill_lds(lds, a);
    __syncthreads();                                  // (the bf16 fragments of layer 3 overwrite the fp32 ones fill_lds has just written)
    fill_lds_color_bwd(lds, a);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    float *slab = lds + OFF_CW + wave * COLOR_SLAB;
    float *TH1 = slab + CS_H1, *TH2 = slab + CS_H2, *TD1 = slab + CS_D1, *TD2 = slab + CS_D2, *TO3 = slab + CS_O3, *TIN = slab + CS_IN;
    for (int e = lane; e < 16 * TLD; e += 64) TO3[e] = 0.0f;          // rows 3..15 stay zero
    for (int e = lane; e < 32 * TLD; e += 64) TIN[e] = 0.0f;          // rows 21..31 stay zero
    __syncthreads();
    f32x4 gW1[4][2], gW2[4][4], gW3[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        gW3[t] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
        for (int c = 0; c < 2; ++c) gW1[t][c] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
        for (int c = 0; c < 4; ++c) gW2[t][c] = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
    }
    const uint32_t ntiles = (B + 15) / 16;
    // like sdf_stencil_bwd_kernel: the next tile's inputs are requested while this one is computed (one wave per SIMD: nothing else covers the loads)
    struct TileIn { float p[3], nr[3], up[3]; f32x4 so; };
    auto request = [&](uint32_t tile, TileIn &in) {
        const uint32_t b = tile * 16 + n, bb = b < B ? b : B - 1;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            in.p[k] = x[3 * (size_t)bb + k]; in.nr[k] = nrm[3 * (size_t)bb + k];
            in.up[k] = (b < B && g == 0) ? g_rgb[3 * (size_t)bb + k] : 0.0f;
        }
        in.so = *reinterpret_cast<const f32x4 *>(sdf16 + (size_t)bb * 16 + 4 * g);
    };
    const uint32_t tile0 = blockIdx.x * TW + wave, tstride = gridDim.x * TW;
    TileIn nxt;
    if (tile0 < ntiles) request(tile0, nxt);
    for (uint32_t tile = tile0; tile < ntiles; tile += tstride) {
        const uint32_t b = tile * 16 + n;
        const bool live = b < B;
        const float px = nxt.p[0], py = nxt.p[1], pz = nxt.p[2];
        const float nx = nxt.nr[0], ny = nxt.nr[1], nz = nxt.nr[2];
        const float ups[3] = { nxt.up[0], nxt.up[1], nxt.up[2] };
        const f32x4 so = nxt.so;
        if (tile + tstride < ntiles) request(tile + tstride, nxt);
        const float bxyz = sel4(g, px, py, pz, 0.0f), bn = sel4(g, nx, ny, nz, 0.0f);
        // forward recompute, activations kept: layers 1 and 2 with the instruction sequence of color_tile (fp32: the ReLU masks are the forward's own)
        f32x4 h1[4], h2[4];
        const float *const shb = sh_bias ? sh_bias + (size_t)((tile * 16u) / T) * 64 + 4 * g : nullptr;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const float bv = s < 4 ? so[s] : (s == 4 ? bxyz : bn);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(lds[OFF_C1F + (t * 6 + s) * 64 + lane], bv, acc, 0, 0, 0);
                if (s == 4 && shb) {                             // the view-direction bias: the forward's position in the chain (color_tile)
                    const f32x4 bq = *reinterpret_cast<const f32x4 *>(shb + 16 * t);
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
        f32x4 o3 = { 0.0f, 0.0f, 0.0f, 0.0f };
#if AC_COLORBWD_BF16
        {
            u32x4 ch[2], cl[2];
            split_tiles(h2, ch, cl);
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) o3 = cb_mma(lds, OFF_C3H, OFF_C3LO, sb, lane, ch[sb], cl[sb], o3);
        }
#else
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
            o3 = __builtin_amdgcn_mfma_f32_16x16x4f32(lds[OFF_C3F + kk * 64 + lane], h2[kk >> 2][kk & 3], o3, 0, 0, 0);
#endif
        // d o3 = d rgb * rgb (1 - rgb), held by the lanes g == 0 (o = r); broadcast to lane group kk = o as the B operand of Wc3^T
        float d3[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const float rgb = dv_sigmoid(o3[o]);
            const float up = ups[o];
            d3[o] = up * (rgb * (1.0f - rgb));
        }
        const float s0 = __shfl(d3[0], n), s1 = __shfl(d3[1], n), s2 = __shfl(d3[2], n);
        const float b3 = g == 0 ? s0 : (g == 1 ? s1 : (g == 2 ? s2 : 0.0f));
        // dh2 = Wc3^T d3 (.) [h2 > 0]; dh1 = Wc2^T dh2 (.) [h1 > 0]
        f32x4 dh2[4], dh1[4];
#pragma unroll
        for (int to = 0; to < 4; ++to) {
            f32x4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(lds[OFF_C3T + to * 64 + lane], b3, acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = h2[to][r] > 0.0f ? acc[r] : 0.0f;
            dh2[to] = acc;
        }
#if AC_COLORBWD_BF16
        u32x4 gbh[2], gbl[2];
        split_tiles(dh2, gbh, gbl);
#endif
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
#if AC_COLORBWD_BF16
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) acc = cb_mma(lds, OFF_C2T, OFF_C2TL, 2 * t + sb, lane, gbh[sb], gbl[sb], acc);
#else
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(lds[OFF_C2T + (t * 16 + ks) * 64 + lane], dh2[ks >> 2][ks & 3], acc, 0, 0, 0);
#endif
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = h1[t][r] > 0.0f ? acc[r] : 0.0f;
            dh1[t] = acc;
        }
        if (g_sh_tiles) {                                            // d bias of this tile: row sums over the 16 samples (dead samples of a ragged last tile: d h1 = 0, their upstream is 0)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 sm;
#pragma unroll
                for (int r = 0; r < 4; ++r) sm[r] = row_scan<false>(dh1[t][r]);
                if (n == 15) *reinterpret_cast<f32x4 *>(g_sh_tiles + (size_t)tile * 64 + 16 * t + 4 * g) = sm;
            }
        }
#if AC_COLORBWD_BF16
        split_tiles(dh1, gbh, gbl);
#endif
        // dinp = Wc1^T dh1: tile 0 -> d sdf_out[4g + r], tile 1 reg 0 -> d normal_g
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
            f32x4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
#if AC_COLORBWD_BF16
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) acc = cb_mma(lds, OFF_C1T, OFF_C1TL, 2 * tp + sb, lane, gbh[sb], gbl[sb], acc);
#else
#pragma unroll
            for (int ks = 0; ks < 16; ++ks)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(lds[OFF_C1T + (tp * 16 + ks) * 64 + lane], dh1[ks >> 2][ks & 3], acc, 0, 0, 0);
#endif
            if (live) {
                if (tp == 0) *reinterpret_cast<f32x4 *>(g_sdf16 + (size_t)b * 16 + 4 * g) = acc;
                else if (g < 3) g_nrm[3 * (size_t)b + g] = acc[0];
            }
        }
        // weight gradients (K = the tile's 16 samples)
        if (g == 0) { TO3[0 * TLD + n] = d3[0]; TO3[1 * TLD + n] = d3[1]; TO3[2 * TLD + n] = d3[2]; }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int u = (16 * t + 4 * g + r) * TLD + n;
                TH1[u] = h1[t][r]; TH2[u] = h2[t][r]; TD1[u] = dh1[t][r]; TD2[u] = dh2[t][r];
            }
        // inputs in the column order of Wc1: x (0..2), normal (3..5), feat (6..20) = sdf_out[1..15]
        if (g < 3) { TIN[g * TLD + n] = bxyz; TIN[(3 + g) * TLD + n] = bn; }
#pragma unroll
        for (int s = 0; s < 4; ++s) { const int o = 4 * g + s; if (o > 0) TIN[(6 + o - 1) * TLD + n] = so[s]; }
        wave_sync();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float a3 = TO3[n * TLD + 4 * s + g];
            float bh2[4], bh1[4], bin[2];
#pragma unroll
            for (int c = 0; c < 4; ++c) { bh2[c] = TH2[(16 * c + n) * TLD + 4 * s + g]; bh1[c] = TH1[(16 * c + n) * TLD + 4 * s + g]; }
#pragma unroll
            for (int c = 0; c < 2; ++c) bin[c] = TIN[(16 * c + n) * TLD + 4 * s + g];
#pragma unroll
            for (int c = 0; c < 4; ++c) gW3[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, bh2[c], gW3[c], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float a2 = TD2[(16 * t + n) * TLD + 4 * s + g], a1 = TD1[(16 * t + n) * TLD + 4 * s + g];
#pragma unroll
                for (int c = 0; c < 4; ++c) gW2[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, bh1[c], gW2[t][c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 2; ++c) gW1[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bin[c], gW1[t][c], 0, 0, 0);
            }
        }
        wave_sync();
    }
    float *part = partials + (size_t)(blockIdx.x * TW + wave) * NPART_C;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * t + 4 * g + r;
#pragma unroll
            for (int c = 0; c < 2; ++c) part[row * 32 + 16 * c + n] = gW1[t][c][r];
#pragma unroll
            for (int c = 0; c < 4; ++c) part[64 * 32 + row * 64 + 16 * c + n] = gW2[t][c][r];
        }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[64 * 32 + 64 * 64 + (4 * g + r) * 64 + 16 * c + n] = gW3[c][r];
}

// ======================================================================================================================
// NeuS alpha + compositing of the render core (models/instant_nsr.py:219-263,290-299) for training: one wave per ray, the
// forward is the renderer's own arithmetic (16-sample tile scans with sequential carry), the backward recomputes it.
//   forward : (z, sdf, normal, colour, ray) -> image, weights_sum, depth, normal_map, weights, alpha
//   backward: (d image, d weights_sum, d depth, d normal_map) -> d sdf, d normal, d colour, per-ray partial of d inv_s
struct CompArgs {
    const float *rays_o, *rays_d, *z, *sdf, *nrm, *col, *bg;
    int n_rays, T0, T;
    float bound, inv_s, car, one_m_car;
    const float *inv_s_dev;      // non-NULL: inv_s lives in device memory (the trainable variance), read once per kernel
    // posed space (the backward of run(render_can=False), instant_nsr.py:147-153,246-249): mesh-guided range where finite, alpha * mask; NULL = off
    const float *near_m, *far_m;
    const uint8_t *mask;
};

struct CompSample { float alpha, om, u, pc, nc, half, delta, tc, zn; };

__device__ __forceinline__ CompSample comp_sample(const float *__restrict__ spg, const CompArgs &a, const float *__restrict__ zr, int i, float sdf0,
                                                  float nx, float ny, float nz, float dx, float dy, float dz, float near, float span, float sample_dist)
{
    CompSample s;
    const float zi = zr[i];
    s.delta = (i < a.T - 1) ? zr[i + 1] - zi : sample_dist;
    s.tc = (dx * nx + dy * ny) + dz * nz;
    const float a1 = dv_softplus100(spg, -s.tc * 0.5f + 0.5f) * a.one_m_car;
    const float a2 = dv_softplus100(spg, -s.tc) * a.car;
    const float iter_cos = -(a1 + a2);
    s.half = iter_cos * s.delta * 0.5f;
    s.pc = dv_sigmoid((sdf0 - s.half) * a.inv_s); s.nc = dv_sigmoid((sdf0 + s.half) * a.inv_s);
    s.u = (s.pc - s.nc + 1e-5f) / (s.pc + 1e-5f);
    s.alpha = clampf(s.u, 0.0f, 1.0f);
    s.om = 1.0f - s.alpha + 1e-7f;
    s.zn = clampf((zi - near) / span, 0.0f, 1.0f);
    return s;
}

__global__ __launch_bounds__(256) void composite_fwd_kernel(const CompArgs a_in, float *__restrict__ image, float *__restrict__ wsum,
                                                            float *__restrict__ depth, float *__restrict__ nmap, float *__restrict__ weights,
                                                            float *__restrict__ alpha_out)
{
    CompArgs a = a_in;
    if (a.inv_s_dev) a.inv_s = *a.inv_s_dev;
    __shared__ float spg[SPQ_FLOATS];
    __shared__ float zsh[4][128];
    for (int e = threadIdx.x; e < SPQ_FLOATS; e += blockDim.x) spg[e] = AC_SP_G[e >> 2][e & 3];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15;
    float *zr = zsh[wave];
    for (int ray = blockIdx.x * 4 + wave; ray < a.n_rays; ray += gridDim.x * 4) {
        const float ox = a.rays_o[3 * ray], oy = a.rays_o[3 * ray + 1], oz = a.rays_o[3 * ray + 2];
        const float dx = a.rays_d[3 * ray], dy = a.rays_d[3 * ray + 1], dz = a.rays_d[3 * ray + 2];
        float near, far;
        cube_near_far(ox, oy, oz, dx, dy, dz, a.bound, near, far);
        if (a.near_m) {
            const float nm = a.near_m[ray], fm = a.far_m[ray];
            if (!is_inf(nm)) near = nm;
            if (!is_inf(fm)) far = fm;
        }
        const float span = far - near, sample_dist = span / (float)a.T0;
        for (int i = lane; i < a.T; i += 64) zr[i] = a.z[(size_t)ray * a.T + i];
        wave_sync();
        float cT = 1.0f, s_w = 0.0f, s_r = 0.0f, s_g = 0.0f, s_b = 0.0f, s_nx = 0.0f, s_ny = 0.0f, s_nz = 0.0f, s_d = 0.0f;
        for (int c = 0; c < a.T / 16; ++c) {
            const int i = 16 * c + n;
            const size_t si = (size_t)ray * a.T + i;
            const float nx = a.nrm[3 * si], ny = a.nrm[3 * si + 1], nz = a.nrm[3 * si + 2];
            CompSample s = comp_sample(spg, a, zr, i, a.sdf[si], nx, ny, nz, dx, dy, dz, near, span, sample_dist);
            if (a.mask && !a.mask[si]) { s.alpha = 0.0f; s.om = 1.0f + 1e-7f; }                 // alpha * 0; the factor 1 - 0 + 1e-7 stays in the product
            const float loc = row_scan<true>(s.om);
            const float sh = dpp_shr<1>(1.0f, loc);
            float Tex;
            if (n == 0) Tex = (c == 0) ? 1.0f : cT;
            else Tex = (c == 0) ? sh : cT * sh;
            const float tot = lane_bcast(loc, 15);
            cT = (c == 0) ? tot : cT * tot;
            const float wgt = s.alpha * Tex;
            const float r = a.col[3 * si], g = a.col[3 * si + 1], b = a.col[3 * si + 2];
#define AC_ACC(S, V) { const float t_ = lane_bcast(row_scan<false>(V), 15); S = (c == 0) ? t_ : S + t_; }
            AC_ACC(s_w, wgt)
            AC_ACC(s_r, r * wgt) AC_ACC(s_nx, nx * wgt)
            AC_ACC(s_g, g * wgt) AC_ACC(s_ny, ny * wgt)
            AC_ACC(s_b, b * wgt) AC_ACC(s_nz, nz * wgt)
            AC_ACC(s_d, wgt * s.zn)
#undef AC_ACC
            if (lane < 16) { weights[si] = wgt; alpha_out[si] = s.alpha; }
        }
        if (lane == 0) {
            const float b0 = a.bg ? a.bg[3 * ray] : 1.0f, b1 = a.bg ? a.bg[3 * ray + 1] : 1.0f, b2 = a.bg ? a.bg[3 * ray + 2] : 1.0f;
            image[3 * ray] = s_r + (1.0f - s_w) * b0; image[3 * ray + 1] = s_g + (1.0f - s_w) * b1; image[3 * ray + 2] = s_b + (1.0f - s_w) * b2;
            nmap[3 * ray] = s_nx; nmap[3 * ray + 1] = s_ny; nmap[3 * ray + 2] = s_nz;
            wsum[ray] = s_w; depth[ray] = s_d;
        }
        wave_sync();
    }
}

__global__ __launch_bounds__(256) void composite_bwd_kernel(const CompArgs a_in, const float *__restrict__ g_image, const float *__restrict__ g_wsum,
                                                            const float *__restrict__ g_depth, const float *__restrict__ g_nmap,
                                                            float *__restrict__ g_sdf, float *__restrict__ g_nrm, float *__restrict__ g_col,
                                                            float *__restrict__ g_invs_ray)
{
    CompArgs a = a_in;
    if (a.inv_s_dev) a.inv_s = *a.inv_s_dev;
    __shared__ float spg[SPQ_FLOATS];
    __shared__ float zsh[4][128], tex[4][128], wq[4][128], psum[4][128];
    for (int e = threadIdx.x; e < SPQ_FLOATS; e += blockDim.x) spg[e] = AC_SP_G[e >> 2][e & 3];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15;
    float *zr = zsh[wave];
    for (int ray = blockIdx.x * 4 + wave; ray < a.n_rays; ray += gridDim.x * 4) {
        const float ox = a.rays_o[3 * ray], oy = a.rays_o[3 * ray + 1], oz = a.rays_o[3 * ray + 2];
        const float dx = a.rays_d[3 * ray], dy = a.rays_d[3 * ray + 1], dz = a.rays_d[3 * ray + 2];
        float near, far;
        cube_near_far(ox, oy, oz, dx, dy, dz, a.bound, near, far);
        if (a.near_m) {
            const float nm = a.near_m[ray], fm = a.far_m[ray];
            if (!is_inf(nm)) near = nm;
            if (!is_inf(fm)) far = fm;
        }
        const float span = far - near, sample_dist = span / (float)a.T0;
        for (int i = lane; i < a.T; i += 64) zr[i] = a.z[(size_t)ray * a.T + i];
        wave_sync();
        const float gi0 = g_image ? g_image[3 * ray] : 0.0f, gi1 = g_image ? g_image[3 * ray + 1] : 0.0f, gi2 = g_image ? g_image[3 * ray + 2] : 0.0f;
        const float gws = g_wsum ? g_wsum[ray] : 0.0f, gdp = g_depth ? g_depth[ray] : 0.0f;          // NULL upstream = zero gradient
        const float gn0 = g_nmap ? g_nmap[3 * ray] : 0.0f, gn1 = g_nmap ? g_nmap[3 * ray + 1] : 0.0f, gn2 = g_nmap ? g_nmap[3 * ray + 2] : 0.0f;
        const float b0 = a.bg ? a.bg[3 * ray] : 1.0f, b1 = a.bg ? a.bg[3 * ray + 1] : 1.0f, b2 = a.bg ? a.bg[3 * ray + 2] : 1.0f;
        // pass A: transmittance, weights, d loss / d w_i, prefix sums of dw_i w_i
        float cT = 1.0f, run = 0.0f;
        for (int c = 0; c < a.T / 16; ++c) {
            const int i = 16 * c + n;
            const size_t si = (size_t)ray * a.T + i;
            const float nx = a.nrm[3 * si], ny = a.nrm[3 * si + 1], nz = a.nrm[3 * si + 2];
            CompSample s = comp_sample(spg, a, zr, i, a.sdf[si], nx, ny, nz, dx, dy, dz, near, span, sample_dist);
            if (a.mask && !a.mask[si]) { s.alpha = 0.0f; s.om = 1.0f + 1e-7f; }                 // alpha * 0; the factor 1 - 0 + 1e-7 stays in the product
            const float loc = row_scan<true>(s.om);
            const float sh = dpp_shr<1>(1.0f, loc);
            float Tex;
            if (n == 0) Tex = (c == 0) ? 1.0f : cT;
            else Tex = (c == 0) ? sh : cT * sh;
            const float tot = lane_bcast(loc, 15);
            cT = (c == 0) ? tot : cT * tot;
            const float wgt = s.alpha * Tex;
            const float r = a.col[3 * si], g = a.col[3 * si + 1], b = a.col[3 * si + 2];
            const float dw = ((gi0 * (r - b0) + gi1 * (g - b1)) + gi2 * (b - b2)) + gws + gdp * s.zn + ((gn0 * nx + gn1 * ny) + gn2 * nz);
            const float incl = row_scan<false>(dw * wgt) + run;
            run = lane_bcast(incl, 15);
            if (lane < 16) { tex[wave][i] = Tex; wq[wave][i] = dw; psum[wave][i] = incl; }
        }
        const float total = run;
        wave_sync();
        // pass B: every lane one sample (two chunks of 64)
        float ds_acc = 0.0f;
        for (int i = lane; i < a.T; i += 64) {
            const size_t si = (size_t)ray * a.T + i;
            const float nx = a.nrm[3 * si], ny = a.nrm[3 * si + 1], nz = a.nrm[3 * si + 2];
            const float sdf0 = a.sdf[si];
            CompSample s = comp_sample(spg, a, zr, i, sdf0, nx, ny, nz, dx, dy, dz, near, span, sample_dist);
            const bool masked = a.mask && !a.mask[si];
            if (masked) { s.alpha = 0.0f; s.om = 1.0f + 1e-7f; }
            const float Tex = tex[wave][i], dw = wq[wave][i];
            const float wgt = s.alpha * Tex;
            const float suffix = total - psum[wave][i];                         // sum over j > i of dw_j w_j
            const float dalpha = masked ? 0.0f : dw * Tex - suffix / s.om;      // d (alpha * mask) / d alpha = mask
            const float du = (s.u >= 0.0f && s.u <= 1.0f) ? dalpha : 0.0f;      // torch.clip passes the gradient on the closed interval
            const float den = s.pc + 1e-5f;
            const float dpc = du * s.nc / (den * den), dnc = -du / den;
            const float dap = dpc * s.pc * (1.0f - s.pc), dan = dnc * s.nc * (1.0f - s.nc);
            const float dsdf = (dap + dan) * a.inv_s;
            const float dhalf = (dan - dap) * a.inv_s;
            ds_acc += dap * (sdf0 - s.half) + dan * (sdf0 + s.half);
            const float dic = dhalf * s.delta * 0.5f;
            float v1, d1, v2, d2;
            softplus100_vg(spg, -s.tc * 0.5f + 0.5f, v1, d1);
            softplus100_vg(spg, -s.tc, v2, d2);
            const float dtc = dic * (0.5f * d1 * a.one_m_car + d2 * a.car);
            g_sdf[si] = dsdf;
            g_nrm[3 * si] = dtc * dx + wgt * gn0; g_nrm[3 * si + 1] = dtc * dy + wgt * gn1; g_nrm[3 * si + 2] = dtc * dz + wgt * gn2;
            g_col[3 * si] = wgt * gi0; g_col[3 * si + 1] = wgt * gi1; g_col[3 * si + 2] = wgt * gi2;
        }
        // per-ray partial of d inv_s: wave reduction
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) ds_acc += __shfl_xor(ds_acc, d);
        if (lane == 0) g_invs_ray[ray] = ds_acc;
        wave_sync();
    }
}

// generic: out[i] = sum over waves of partials[w][i], i < n_out
__global__ __launch_bounds__(1024) void partials_reduce_kernel(const float *__restrict__ partials, uint32_t nwaves, uint32_t n_out, float *__restrict__ out)
{
    __shared__ double red[RED_SL][RED_OUT]; // (double: the order of the partials must not show up in the last bits of a sum of ~1000 of them)
    const uint32_t o = threadIdx.x % RED_OUT, sl = threadIdx.x / RED_OUT;
    const uint32_t i = blockIdx.x * RED_OUT + o;
    double s = 0.0;
    if (i < n_out)
        for (uint32_t w = sl; w < nwaves; w += RED_SL) s += (double)partials[(size_t)w * n_out + i];
    red[sl][o] = s;
    __syncthreads();
    if (sl == 0 && i < n_out) {
        double t = 0.0;
#pragma unroll
        for (uint32_t k = 0; k < RED_SL; ++k) t += red[k][o];
        out[i] = (float)t;
    }
}

// ---- glue of the whole-core backward (ac_render_core_backward) -------------------------------------------------------------------
// normal = gradient / (1e-5 + |gradient|)  (instant_nsr.py:215), the renderer's own arithmetic: the colour / compositing backward
// kernels recompute their forward from exactly the normals the forward launch used
__global__ __launch_bounds__(256) void core_normals_kernel(const float *__restrict__ grad, uint32_t B, float *__restrict__ nrm)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float gx = grad[3 * (size_t)b], gy = grad[3 * (size_t)b + 1], gz = grad[3 * (size_t)b + 2];
    const float gn = __builtin_sqrtf((gx * gx + gy * gy) + gz * gz);
    nrm[3 * (size_t)b] = gx / (1e-5f + gn); nrm[3 * (size_t)b + 1] = gy / (1e-5f + gn); nrm[3 * (size_t)b + 2] = gz / (1e-5f + gn);
}

// joins the gradients that reach the SDF query: d sdf_out = (colour backward) with column 0 += (compositing backward: d sdf);
// d gradient = backward of n = g / (1e-5 + |g|) applied to d normal (compositing + colour), plus the eikonal term
// d/dg [ sum relax (|g| - 1)^2 / (sum relax + 1e-5) ] = relax 2 (|g| - 1) / den * g / |g|   (instant_nsr.py:266-272; |g| = 0 -> 0 like
// torch.linalg.norm's backward)
__global__ __launch_bounds__(256) void core_mid_kernel(const float *__restrict__ grad, const float *__restrict__ pts, const float *__restrict__ g_sdf,
                                                       const float *__restrict__ g_nrm_a, const float *__restrict__ g_nrm_b, const float *__restrict__ g_eik,
                                                       const float *__restrict__ eik_den, uint32_t B, float *__restrict__ g_s16, float *__restrict__ g_grad,
                                                       uint32_t group_samples, uint32_t den_stride)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const size_t b3 = 3 * (size_t)b;
    const float gx = grad[b3], gy = grad[b3 + 1], gz = grad[b3 + 2];
    const float r = __builtin_sqrtf((gx * gx + gy * gy) + gz * gz), c = 1e-5f + r;
    const float ux = g_nrm_a[b3] + g_nrm_b[b3], uy = g_nrm_a[b3 + 1] + g_nrm_b[b3 + 1], uz = g_nrm_a[b3 + 2] + g_nrm_b[b3 + 2];
    const float dot = (gx * ux + gy * uy) + gz * uz;
    float k = r > 0.0f ? -dot / (r * c * c) : 0.0f;                   // d(1 / (1e-5 + r)) / dg = -g / (r c^2)
    if (g_eik) {
        const float px = pts[b3], py = pts[b3 + 1], pz = pts[b3 + 2];
        const float relax = __builtin_sqrtf((px * px + py * py) + pz * pz) < 1.2f ? 1.0f : 0.0f;
        const uint32_t grp = group_samples ? b / group_samples : 0u;      // (several patches in one launch: ac_core_upstream.eik_group_rays)
        if (r > 0.0f) k += g_eik[grp] * relax * 2.0f * (r - 1.0f) / (eik_den[(size_t)grp * den_stride] * r);
    }
    g_grad[b3] = ux / c + k * gx; g_grad[b3 + 1] = uy / c + k * gy; g_grad[b3 + 2] = uz / c + k * gz;
    g_s16[(size_t)b * 16] += g_sdf[b];
}

uint32_t train_grid(uint32_t B)
{
    const uint32_t ntiles = (B + 15) / 16;
    uint32_t blocks = (ntiles + TW - 1) / TW;
#ifdef AC_TRAIN_GRID
    const uint32_t cap = AC_TRAIN_GRID;
#else
    const uint32_t cap = ac::cu_count();               // persistent, ONE workgroup per CU: their ~145 KB of LDS admit no second one, and every workgroup
#endif                                                 // first lays the weights out in LDS (512 workgroups: backward 3.32 ms per SDS step, 256: 3.20)
    if (blocks > cap) blocks = cap;
    return blocks ? blocks : 1;
}

int prep_args(RenderArgs &a, const ac_field *field, float bound, float eps)
{
    if (int rc = fill_args(a, field, bound)) return rc;
    a.eps = eps;
    for (int j = 0; j < 4; ++j) {
        a.jfine[j] = 0;
        for (int g = 0; g < 4; ++g) {
            const double cells = (double)eps / (double)a.two_bound * (double)a.lvl[4 * j + g].scale;
            if (!(cells * 1.001 + 1e-3 < 1.0)) a.jfine[j] = 1;
        }
    }
    return AC_OK;
}

// the colour kernels use the weight fragments only: no hash-level validation (the table of `field` is never touched)
int prep_color_args(RenderArgs &a, const ac_field *f)
{
    if (!f || !f->W1 || !f->b1 || !f->W2 || !f->b2 || !f->Wc1 || !f->Wc2 || !f->Wc3) { ac::set_error("ac_field: NULL parameter pointer"); return AC_ERR_BAD_ARG; }
    a.table = f->table; a.table_bytes = 0;
    a.W1 = f->W1; a.b1 = f->b1; a.W2 = f->W2; a.b2 = f->b2; a.Wc1 = f->Wc1; a.Wc2 = f->Wc2; a.Wc3 = f->Wc3;
    a.bound = 1.0f; a.two_bound = 2.0f;
    return AC_OK;
}

}  // namespace

AC_API int ac_sdf_stencil_forward(const ac_field *field, const float *x, uint32_t B, float bound, float eps, float *out16, float *grad,
                                  ac_stream_t stream)
{
    if (B == 0) return AC_OK;
    if (!x || !out16 || !grad || !(eps > 0.0f)) { ac::set_error("sdf_stencil_forward: NULL buffer or eps <= 0"); return AC_ERR_BAD_ARG; }
    RenderArgs a{};
    if (int rc = prep_args(a, field, bound, eps)) return rc;
    const size_t lds_bytes = FWD_LDS_FLOATS * sizeof(float);
    static uint64_t seen = 0;
    ac::allow_dynamic_lds(seen, reinterpret_cast<const void *>(sdf_stencil_fwd_kernel), lds_bytes);
    uint32_t blocks = ((B + 15) / 16 + FW - 1) / FW;              // persistent: the 140 KB of LDS allow one workgroup per CU, and every
    const uint32_t cus = ac::cu_count();                          // workgroup first lays the weights out in LDS
    if (blocks > cus) blocks = cus;
    hipLaunchKernelGGL(sdf_stencil_fwd_kernel, dim3(blocks), dim3(FBLOCK), lds_bytes, (hipStream_t)stream, a, x, B, eps, out16, grad);
    return ac::check_launch("sdf_stencil_forward");
}

AC_API int ac_field_samples(const ac_field *field, const float *xyzs, const float *dirs, const float *deltas, uint32_t delta_stride, uint32_t M,
                            float bound, float eps, float inv_s, const float *inv_s_dev, float cos_anneal_ratio, float *alpha, float *rgb, float *normal,
                            float *sdf, float *gradient, ac_stream_t stream)
{
    if (M == 0) return AC_OK;
    if (!xyzs || !dirs || !deltas || !alpha || !rgb || !normal || delta_stride == 0 || !(eps > 0.0f)) {
        ac::set_error("field_samples: NULL buffer, delta_stride == 0 or eps <= 0"); return AC_ERR_BAD_ARG;
    }
    RenderArgs a{};
    if (int rc = prep_args(a, field, bound, eps)) return rc;
    a.inv_s = inv_s; a.inv_s_dev = inv_s_dev; a.car = cos_anneal_ratio; a.one_m_car = (float)(1.0 - (double)cos_anneal_ratio);
    SampleArgs sa{ xyzs, dirs, deltas, delta_stride, M, alpha, rgb, normal, sdf, gradient };
    const size_t lds_bytes = FWD_LDS_FLOATS * sizeof(float);
    static uint64_t seen = 0;
    ac::allow_dynamic_lds(seen, reinterpret_cast<const void *>(field_samples_kernel), lds_bytes);
    uint32_t blocks = ((M + 15) / 16 + FW - 1) / FW;              // persistent, one workgroup per CU (140 KB of LDS), like the SDF query
    const uint32_t cus = ac::cu_count();
    if (blocks > cus) blocks = cus;
    hipLaunchKernelGGL(field_samples_kernel, dim3(blocks), dim3(FBLOCK), lds_bytes, (hipStream_t)stream, a, sa);
    return ac::check_launch("field_samples");
}

static int render_rays_occupancy_impl(const ac_field *field, const float *rays_o, const float *rays_d, uint32_t N, const float *grid, uint32_t H,
                                      float mean_density, float bound, float eps, float inv_s, const float *inv_s_dev, float cos_anneal_ratio,
                                      float *weights_sum, float *depth, float *image, float *normal_map, uint32_t *n_samples, uint32_t max_steps,
                                      ac_stream_t stream, const uint32_t *run_if)
{
    if (N == 0) return AC_OK;
    if (!rays_o || !rays_d || !grid || !weights_sum || !depth || !image || !normal_map || H < 2 || !(eps > 0.0f)) {
        ac::set_error("render_rays_occupancy: NULL buffer, H < 2 or eps <= 0"); return AC_ERR_BAD_ARG;
    }
    RenderArgs a{};
    if (int rc = prep_args(a, field, bound, eps)) return rc;
    a.inv_s = inv_s; a.inv_s_dev = inv_s_dev; a.car = cos_anneal_ratio; a.one_m_car = (float)(1.0 - (double)cos_anneal_ratio);
    const uint32_t cus = ac::cu_count();
    // rays per wave (2^glog): 16 for whole views, 8 for small batches (more waves in flight: the march is a chain of ~200 dependent grid look-ups per ray,
    // 0.26 ms end to end, and only concurrency hides it).  Measured on the 256 x 256 bench view (profiles/r04_experiments.txt section 11): 65 536 rays in one
    // launch 2.67 / 2.37 / 2.92 / 3.26 ms for 8 / 16 / 32 / 64 rays per wave; in 4096-ray launches 12.1 / 15.6 / 20.0 / 23.4 ms.  AC_OCC_GLOG = 2 .. 6 overrides.
    static const int env_glog = []() { const char *e = getenv("AC_OCC_GLOG"); return (e && e[0] >= '2' && e[0] <= '6' && !e[1]) ? e[0] - '0' : -1; }();
    const uint32_t glog = env_glog >= 0 ? (uint32_t)env_glog : (N >= 32768u ? 4u : 3u);
    const uint32_t gsz = 1u << glog;
    // the voxel faces as a table behind the stages when H + 1 floats still fit the compute unit's LDS (H = 128: 516 of the 1.9 KB left)
    const bool tab = (OCC_LDS_FLOATS + (size_t)H + 1) * sizeof(float) + 64 <= 160 * 1024;
    OccArgs oc{ rays_o, rays_d, grid, N, H, mean_density, weights_sum, depth, image, normal_map, n_samples, glog, max_steps, tab ? 1u : 0u, run_if };
    const size_t lds_bytes = (OCC_LDS_FLOATS + (tab ? (size_t)H + 1 : 0)) * sizeof(float);
    static uint64_t seen = 0;
    ac::allow_dynamic_lds(seen, reinterpret_cast<const void *>(occupancy_render_kernel), 160 * 1024 - 64);      // (a ceiling, set once: lds_bytes depends on H)
    uint32_t blocks = (N + gsz - 1) / gsz;                         // one group per workgroup first (see the kernel's loop), one persistent workgroup per CU at most
    if (blocks > cus) blocks = cus;
    hipLaunchKernelGGL(occupancy_render_kernel, dim3(blocks), dim3(FBLOCK), lds_bytes, (hipStream_t)stream, a, oc);
    return ac::check_launch("render_rays_occupancy");
}

AC_API int ac_render_rays_occupancy(const ac_field *field, const float *rays_o, const float *rays_d, uint32_t N, const float *grid, uint32_t H,
                                    float mean_density, float bound, float eps, float inv_s, const float *inv_s_dev, float cos_anneal_ratio,
                                    float *weights_sum, float *depth, float *image, float *normal_map, uint32_t *n_samples, uint32_t max_steps,
                                    ac_stream_t stream)
{
    return render_rays_occupancy_impl(field, rays_o, rays_d, N, grid, H, mean_density, bound, eps, inv_s, inv_s_dev, cos_anneal_ratio, weights_sum, depth, image,
                                      normal_map, n_samples, max_steps, stream, nullptr);
}

// scratch of ac_render_rays_occupancy_train: [16] sync words (word 8: launches whose grid barrier timed out, sticky) | [chunks] totals | counts, offsets, overflow flags, word masks [N] each | records [N][32] | [CUs][2] doubles |
// packed samples: ray [M], x y z dt [M][4], alpha r g b nx ny nz - [M][8].  ZERO-FILLED by the caller once (the sync words and the totals; every call leaves
// them zero again), reusable for calls with the SAME N and capacity on the same stream (the layout depends on both).
struct OccTrainLayout { size_t tot, cnt, offs, ovf, wmask, rec, part, p_ray, p_in, p_out, total; };
static OccTrainLayout occ_train_layout(uint32_t N, uint32_t M)
{
    OccTrainLayout l{};
    const size_t chunks = ((size_t)N + (1u << OT_CHUNK_LOG) - 1) >> OT_CHUNK_LOG;
    size_t o = 16 * sizeof(uint32_t);
    l.tot = o; o += chunks * sizeof(int32_t);
    l.cnt = o; o += (size_t)N * sizeof(int32_t);
    l.offs = o; o += (size_t)N * sizeof(int32_t);
    l.ovf = o; o += (size_t)N * sizeof(int32_t);
    l.wmask = o; o += (size_t)N * sizeof(uint32_t);
    l.rec = o; o += (size_t)N * RM_REC_WORDS * sizeof(uint32_t);
    o = (o + 15) & ~(size_t)15;
    l.part = o; o += (size_t)ac::cu_count() * 2 * sizeof(double);
    l.p_ray = o; o += (size_t)M * sizeof(int32_t);
    o = (o + 15) & ~(size_t)15;
    l.p_in = o; o += (size_t)M * 4 * sizeof(float);
    l.p_out = o; o += (size_t)M * 8 * sizeof(float);
    l.total = o;
    return l;
}
AC_API size_t ac_render_rays_occupancy_train_scratch(uint32_t N, uint32_t capacity) { return occ_train_layout(N, capacity).total; }

AC_API int ac_render_rays_occupancy_train(const ac_field *field, const float *rays_o, const float *rays_d, uint32_t N, const float *grid, uint32_t H,
                                          float mean_density, float bound, float eps, float inv_s, const float *inv_s_dev, float cos_anneal_ratio,
                                          uint32_t perturb, uint32_t capacity, uint32_t composite_capacity, int32_t *counter, const float *bg,
                                          uint32_t bg_mode, float bg_value, float *weights_sum, float *image, float *normal_map, float *gradient_error,
                                          void *scratch, size_t scratch_bytes, ac_stream_t stream)
{
    if (!gradient_error) { ac::set_error("render_rays_occupancy_train: NULL gradient_error"); return AC_ERR_BAD_ARG; }
    if (N == 0) { hipMemsetAsync(gradient_error, 0, sizeof(float), (hipStream_t)stream); return AC_OK; }
    if (!rays_o || !rays_d || !grid || !weights_sum || !image || !normal_map || !scratch || H < 2 || !(eps > 0.0f) || bg_mode > 3u || (bg_mode >= 2u && !bg)) {
        ac::set_error("render_rays_occupancy_train: NULL buffer, H < 2, eps <= 0 or bad background mode"); return AC_ERR_BAD_ARG;
    }
    if (capacity == 0u || composite_capacity == 0u) {
        ac::set_error("render_rays_occupancy_train: capacity / composite_capacity must be > 0 (the packed layout lives in the scratch: a budgeted call)"); return AC_ERR_BAD_ARG;
    }
    const OccTrainLayout l = occ_train_layout(N, capacity);
    if (scratch_bytes < l.total) { ac::set_error("render_rays_occupancy_train: scratch of %zu bytes needed, %zu given", l.total, scratch_bytes); return AC_ERR_BAD_ARG; }
    RenderArgs a{};
    if (int rc = prep_args(a, field, bound, eps)) return rc;
    a.inv_s = inv_s; a.inv_s_dev = inv_s_dev; a.car = cos_anneal_ratio; a.one_m_car = (float)(1.0 - (double)cos_anneal_ratio);
    char *sc = static_cast<char *>(scratch);
    OccTrainArgs oc{ rays_o, rays_d, grid, N, H, capacity, composite_capacity, perturb, mean_density, counter, weights_sum, image, normal_map, gradient_error,
                     bg, bg_mode, bg_value, reinterpret_cast<uint32_t *>(sc), reinterpret_cast<int32_t *>(sc + l.tot), reinterpret_cast<int32_t *>(sc + l.cnt),
                     reinterpret_cast<int32_t *>(sc + l.offs), reinterpret_cast<int32_t *>(sc + l.ovf), reinterpret_cast<uint32_t *>(sc + l.wmask),
                     reinterpret_cast<uint32_t *>(sc + l.rec),
                     reinterpret_cast<double *>(sc + l.part), reinterpret_cast<int32_t *>(sc + l.p_ray), reinterpret_cast<float *>(sc + l.p_in),
                     reinterpret_cast<float *>(sc + l.p_out), ot_spin_ticks() };
    const size_t lds_bytes = FWD_LDS_FLOATS * sizeof(float);
    static uint64_t seen = 0;
    ac::allow_dynamic_lds(seen, reinterpret_cast<const void *>(occupancy_train_kernel), lds_bytes);
    // one persistent workgroup per compute unit AT MOST: the kernel's grid barriers need every workgroup resident (FBLOCK threads at two waves per SIMD and
    // the LDS image make it one per CU); fewer when the batch has less than one 64-ray wave of walking per workgroup
    const uint32_t cus = ac::cu_count();
    uint32_t blocks = (N + 63u) / 64u;
    if (blocks > cus) blocks = cus;
    if (blocks < cus && capacity / 16u > blocks * FW) blocks = cus;         // (the tiles of phase C want every wave)
    void *params[2] = { &a, &oc };
    return ac::launch_resident("render_rays_occupancy_train", reinterpret_cast<const void *>(occupancy_train_kernel), blocks, FBLOCK, lds_bytes, (hipStream_t)stream, params);
}

AC_API uint32_t ac_set_occupancy_barrier_ms(uint32_t ms)
{
    const uint32_t prev = g_barrier_ms.exchange(ms, std::memory_order_relaxed);
    return prev ? prev : ot_default_ms();
}

// scratch of ac_render_rays_occupancy_phased: [16] sync words | alive lists [2][N] | per-ray state [N][4] | per-entry counts [N] | tile list [N << nlog] |
// sample slots in / out [N << nlog][8] each.  ZERO-FILLED by the caller once (the sync words; every call leaves them zero again); any call with the same or a
// smaller N (and the same n_step) may reuse it on the same stream.
struct OccPhLayout { size_t alive, st, cnt, list, s_in, s_out, total; };
static uint32_t occ_phased_nlog()
{
    static const int env = []() { const char *e = getenv("AC_OCC_NLOG"); return (e && e[0] >= '1' && e[0] <= '6' && !e[1]) ? e[0] - '0' : -1; }();
    return env >= 0 ? (uint32_t)env : 4u;
}
static OccPhLayout occ_phased_layout(uint32_t N, uint32_t nlog)
{
    OccPhLayout l{};
    size_t o = 16 * sizeof(uint32_t);
    l.alive = o; o += 2 * (size_t)N * sizeof(int32_t);
    o = (o + 15) & ~(size_t)15;
    l.st = o; o += (size_t)N * 4 * sizeof(float);
    l.cnt = o; o += (size_t)N * sizeof(uint32_t);
    l.list = o; o += ((size_t)N << nlog) * sizeof(uint32_t);
    o = (o + 15) & ~(size_t)15;
    l.s_in = o; o += ((size_t)N << nlog) * 8 * sizeof(float);
    l.s_out = o; o += ((size_t)N << nlog) * 8 * sizeof(float);
    l.total = o;
    return l;
}
AC_API size_t ac_render_rays_occupancy_phased_scratch(uint32_t N) { return occ_phased_layout(N, occ_phased_nlog()).total; }

AC_API int ac_render_rays_occupancy_phased(const ac_field *field, const float *rays_o, const float *rays_d, uint32_t N, const float *grid, uint32_t H,
                                           float mean_density, float bound, float eps, float inv_s, const float *inv_s_dev, float cos_anneal_ratio,
                                           float *weights_sum, float *depth, float *image, float *normal_map, uint32_t *n_samples, uint32_t max_steps,
                                           void *scratch, size_t scratch_bytes, ac_stream_t stream)
{
    if (N == 0) return AC_OK;
    if (!rays_o || !rays_d || !grid || !weights_sum || !depth || !image || !normal_map || !scratch || H < 2 || !(eps > 0.0f)) {
        ac::set_error("render_rays_occupancy_phased: NULL buffer, H < 2 or eps <= 0"); return AC_ERR_BAD_ARG;
    }
    const uint32_t nlog = occ_phased_nlog();
    if (((uint64_t)N << nlog) >= (1ull << 31)) { ac::set_error("render_rays_occupancy_phased: too many rays for 32-bit slot ids"); return AC_ERR_BAD_ARG; }
    const OccPhLayout l = occ_phased_layout(N, nlog);
    if (scratch_bytes < l.total) { ac::set_error("render_rays_occupancy_phased: scratch of %zu bytes needed, %zu given", l.total, scratch_bytes); return AC_ERR_BAD_ARG; }
    RenderArgs a{};
    if (int rc = prep_args(a, field, bound, eps)) return rc;
    a.inv_s = inv_s; a.inv_s_dev = inv_s_dev; a.car = cos_anneal_ratio; a.one_m_car = (float)(1.0 - (double)cos_anneal_ratio);
    char *sc = static_cast<char *>(scratch);
    OccPhArgs oc{ rays_o, rays_d, grid, N, H, max_steps, nlog, mean_density, weights_sum, depth, image, normal_map, n_samples,
                  reinterpret_cast<uint32_t *>(sc), reinterpret_cast<int32_t *>(sc + l.alive), reinterpret_cast<float *>(sc + l.st),
                  reinterpret_cast<uint32_t *>(sc + l.cnt), reinterpret_cast<uint32_t *>(sc + l.list), reinterpret_cast<float *>(sc + l.s_in),
                  reinterpret_cast<float *>(sc + l.s_out), ot_spin_ticks() };
    const bool tab = (FWD_LDS_FLOATS + (size_t)H + 1) * sizeof(float) + 64 <= 160 * 1024;
    const size_t lds_bytes = (FWD_LDS_FLOATS + (tab ? (size_t)H + 1 : 0)) * sizeof(float);
    static uint64_t seen = 0;
    ac::allow_dynamic_lds(seen, reinterpret_cast<const void *>(occupancy_phased_kernel), 160 * 1024 - 64);
    // one persistent workgroup per compute unit AT MOST (grid barriers: every workgroup must be resident)
    const uint32_t cus = ac::cu_count();
    uint32_t blocks = (N + 63u) / 64u;
    if (blocks > cus) blocks = cus;
    if (blocks < cus && N / 4u > blocks * FW) blocks = cus;              // (the tiles of phase F want every wave)
    void *params[2] = { &a, &oc };
    if (int rc = ac::launch_resident("render_rays_occupancy_phased", reinterpret_cast<const void *>(occupancy_phased_kernel), blocks, FBLOCK, lds_bytes, (hipStream_t)stream, params)) return rc;
    // never a partial result (the reference's loop, raymarching/raymarching.py:136-188, cannot produce one): the barrier-free kernel is queued behind the
    // phased one and does nothing unless that launch's verdict word says a grid barrier timed out -- then it renders every ray again (the same bits: tests)
    return render_rays_occupancy_impl(field, rays_o, rays_d, N, grid, H, mean_density, bound, eps, inv_s, inv_s_dev, cos_anneal_ratio, weights_sum, depth, image,
                                      normal_map, n_samples, max_steps, stream, reinterpret_cast<const uint32_t *>(sc) + 9);
}

// ---- the shading glue of run_cuda's TRAINING form under autograd (round 6) ---------------------------------------------------------------
// Between the fused SDF query (ac_sdf_stencil_*) and the packed compositor (composite_rays_train) the chain ran ~40 torch kernels forward and backward per batch
// (normalisation of the finite-difference gradient, the cos-annealed NeuS alpha, the eikonal term): 2.5 ms per 4096-ray batch against 0.5 ms for the no-grad launch.
// One elementwise kernel each way, the arithmetic of ac_field_samples (forward) and of composite_bwd_kernel / core_mid_kernel (backward):
//   normal = g / (1e-5 + |g|);  alpha = clip((pc - nc + 1e-5) / (pc + 1e-5), 0, 1), pc / nc = sigmoid((sdf -+ half) inv_s), half = iter_cos dt / 2,
//   iter_cos = -(softplus(-tc / 2 + 1 / 2) (1 - car) + softplus(-tc) car), tc = d . normal;  eik = (relax (|g| - 1)^2, relax), relax = [|x| < 1.2][row < n_valid]
struct PackedShade {
    const float *sdf16, *gradient, *xyzs, *dirs, *deltas;
    uint32_t delta_stride, M;
    const int32_t *n_valid;             // device scalar: rows >= *n_valid are alignment padding (no eikonal share)
    float inv_s; const float *inv_s_dev; float car, one_m_car;
};
__global__ __launch_bounds__(256) void packed_shading_fwd_kernel(const PackedShade a, float *__restrict__ alpha, float *__restrict__ normal, float *__restrict__ eik)
{
    __shared__ float spg[SPQ_FLOATS];
    for (int e = threadIdx.x; e < SPQ_FLOATS; e += blockDim.x) spg[e] = AC_SP_G[e >> 2][e & 3];
    __syncthreads();
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.M) return;
    const float inv_s = a.inv_s_dev ? *a.inv_s_dev : a.inv_s;
    const size_t b3 = 3 * (size_t)b;
    const float gx = a.gradient[b3], gy = a.gradient[b3 + 1], gz = a.gradient[b3 + 2];
    const float r = __builtin_sqrtf((gx * gx + gy * gy) + gz * gz), c = 1e-5f + r;
    const float nx = gx / c, ny = gy / c, nz = gz / c;
    const float tc = (a.dirs[b3] * nx + a.dirs[b3 + 1] * ny) + a.dirs[b3 + 2] * nz;
    const float a1 = dv_softplus100(spg, -tc * 0.5f + 0.5f) * a.one_m_car, a2 = dv_softplus100(spg, -tc) * a.car;
    const float half = -(a1 + a2) * a.deltas[(size_t)b * a.delta_stride] * 0.5f;
    const float sdf = a.sdf16[(size_t)b * 16];
    const float pc = dv_sigmoid((sdf - half) * inv_s), nc = dv_sigmoid((sdf + half) * inv_s);
    alpha[b] = clampf((pc - nc + 1e-5f) / (pc + 1e-5f), 0.0f, 1.0f);
    normal[b3] = nx; normal[b3 + 1] = ny; normal[b3 + 2] = nz;
    const float px = a.xyzs[b3], py = a.xyzs[b3 + 1], pz = a.xyzs[b3 + 2];
    const float relax = (__builtin_sqrtf((px * px + py * py) + pz * pz) < 1.2f && (int32_t)b < *a.n_valid) ? 1.0f : 0.0f;
    eik[2 * (size_t)b] = relax * ((r - 1.0f) * (r - 1.0f)); eik[2 * (size_t)b + 1] = relax;
}
__global__ __launch_bounds__(256) void packed_shading_bwd_kernel(const PackedShade a, const float *__restrict__ g_alpha, const float *__restrict__ g_normal,
                                                                 const float *__restrict__ g_eik, float *__restrict__ g_sdf16, float *__restrict__ g_gradient,
                                                                 float *__restrict__ g_inv_s_rows)
{
    __shared__ float spg[SPQ_FLOATS];
    for (int e = threadIdx.x; e < SPQ_FLOATS; e += blockDim.x) spg[e] = AC_SP_G[e >> 2][e & 3];
    __syncthreads();
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.M) return;
    const float inv_s = a.inv_s_dev ? *a.inv_s_dev : a.inv_s;
    const size_t b3 = 3 * (size_t)b;
    const float gx = a.gradient[b3], gy = a.gradient[b3 + 1], gz = a.gradient[b3 + 2];
    const float r = __builtin_sqrtf((gx * gx + gy * gy) + gz * gz), c = 1e-5f + r;
    const float nx = gx / c, ny = gy / c, nz = gz / c;
    const float dx = a.dirs[b3], dy = a.dirs[b3 + 1], dz = a.dirs[b3 + 2];
    const float tc = (dx * nx + dy * ny) + dz * nz;
    float v1, d1, v2, d2;
    softplus100_vg(spg, -tc * 0.5f + 0.5f, v1, d1);
    softplus100_vg(spg, -tc, v2, d2);
    const float delta = a.deltas[(size_t)b * a.delta_stride];
    const float half = -(v1 * a.one_m_car + v2 * a.car) * delta * 0.5f;
    const float sdf = a.sdf16[(size_t)b * 16];
    const float pc = dv_sigmoid((sdf - half) * inv_s), nc = dv_sigmoid((sdf + half) * inv_s);
    const float den = pc + 1e-5f, u = (pc - nc + 1e-5f) / den;
    const float du = (u >= 0.0f && u <= 1.0f) ? (g_alpha ? g_alpha[b] : 0.0f) : 0.0f;      // torch.clip passes the gradient on the closed interval
    const float dpc = du * nc / (den * den), dnc = -du / den;
    const float dap = dpc * pc * (1.0f - pc), dan = dnc * nc * (1.0f - nc);
    g_sdf16[(size_t)b * 16] = (dap + dan) * inv_s;
    g_inv_s_rows[b] = dap * (sdf - half) + dan * (sdf + half);
    const float dic = (dan - dap) * inv_s * delta * 0.5f;
    const float dtc = dic * (0.5f * d1 * a.one_m_car + d2 * a.car);
    const float ux = (g_normal ? g_normal[b3] : 0.0f) + dtc * dx, uy = (g_normal ? g_normal[b3 + 1] : 0.0f) + dtc * dy, uz = (g_normal ? g_normal[b3 + 2] : 0.0f) + dtc * dz;
    float k = 0.0f;
    if (r > 0.0f) {
        k = -((gx * ux + gy * uy) + gz * uz) / (r * c * c);                 // d (1 / (1e-5 + r)) / dg = -g / (r c^2)
        const float px = a.xyzs[b3], py = a.xyzs[b3 + 1], pz = a.xyzs[b3 + 2];
        const float relax = (__builtin_sqrtf((px * px + py * py) + pz * pz) < 1.2f && (int32_t)b < *a.n_valid) ? 1.0f : 0.0f;
        if (g_eik) k += g_eik[2 * (size_t)b] * relax * 2.0f * (r - 1.0f) / r;
    }
    g_gradient[b3] = ux / c + k * gx; g_gradient[b3 + 1] = uy / c + k * gy; g_gradient[b3 + 2] = uz / c + k * gz;
}
static int packed_shade_args(PackedShade &a, const char *who, const float *sdf16, const float *gradient, const float *xyzs, const float *dirs, const float *deltas,
                             uint32_t delta_stride, uint32_t M, const int32_t *n_valid, float inv_s, const float *inv_s_dev, float car)
{
    if (!sdf16 || !gradient || !xyzs || !dirs || !deltas || !n_valid || delta_stride < 1 || delta_stride > 2) { ac::set_error("%s: NULL buffer or delta_stride not 1 / 2", who); return AC_ERR_BAD_ARG; }
    a = PackedShade{ sdf16, gradient, xyzs, dirs, deltas, delta_stride, M, n_valid, inv_s, inv_s_dev, car, (float)(1.0 - (double)car) };
    return AC_OK;
}
AC_API int ac_packed_shading_forward(const float *sdf16, const float *gradient, const float *xyzs, const float *dirs, const float *deltas, uint32_t delta_stride,
                                     uint32_t M, const int32_t *n_valid, float inv_s, const float *inv_s_dev, float cos_anneal_ratio, float *alpha, float *normal,
                                     float *eik, ac_stream_t stream)
{
    if (M == 0) return AC_OK;
    PackedShade a;
    if (int rc = packed_shade_args(a, "packed_shading_forward", sdf16, gradient, xyzs, dirs, deltas, delta_stride, M, n_valid, inv_s, inv_s_dev, cos_anneal_ratio)) return rc;
    if (!alpha || !normal || !eik) { ac::set_error("packed_shading_forward: NULL output"); return AC_ERR_BAD_ARG; }
    hipLaunchKernelGGL(packed_shading_fwd_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, alpha, normal, eik);
    return ac::check_launch("packed_shading_forward");
}
AC_API int ac_packed_shading_backward(const float *sdf16, const float *gradient, const float *xyzs, const float *dirs, const float *deltas, uint32_t delta_stride,
                                      uint32_t M, const int32_t *n_valid, float inv_s, const float *inv_s_dev, float cos_anneal_ratio, const float *g_alpha,
                                      const float *g_normal, const float *g_eik, float *g_sdf16, float *g_gradient, float *g_inv_s_rows, ac_stream_t stream)
{
    if (M == 0) return AC_OK;
    PackedShade a;
    if (int rc = packed_shade_args(a, "packed_shading_backward", sdf16, gradient, xyzs, dirs, deltas, delta_stride, M, n_valid, inv_s, inv_s_dev, cos_anneal_ratio)) return rc;
    if (!g_sdf16 || !g_gradient || !g_inv_s_rows) { ac::set_error("packed_shading_backward: NULL output"); return AC_ERR_BAD_ARG; }
    hipLaunchKernelGGL(packed_shading_bwd_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, g_alpha, g_normal, g_eik, g_sdf16, g_gradient, g_inv_s_rows);
    return ac::check_launch("packed_shading_backward");
}

AC_API size_t ac_sdf_stencil_backward_scratch(uint32_t B)
{
    return (size_t)train_grid(B) * (TW_S > TW ? TW_S : TW) * NPART * sizeof(float);
}

static int sdf_stencil_backward_impl(const ac_field *field, const float *x, const float *g_out16, const float *g_grad, uint32_t B, float bound,
                                     float eps, float *gfeat, float *gparams, void *scratch, size_t scratch_bytes, ac_stream_t stream, const float *feat7,
                                     float *g_x = nullptr)
{
    if (!gparams) { ac::set_error("sdf_stencil_backward: NULL gparams"); return AC_ERR_BAD_ARG; }
    if (g_x && feat7) { ac::set_error("sdf_stencil_backward: the position gradient is not built for the saved-feature form"); return AC_ERR_BAD_ARG; }
    if (B == 0) { hipMemsetAsync(gparams, 0, NPART * sizeof(float), (hipStream_t)stream); return AC_OK; }
    if (!x || !g_out16 || !g_grad || !gfeat || !scratch || !(eps > 0.0f)) { ac::set_error("sdf_stencil_backward: NULL buffer or eps <= 0"); return AC_ERR_BAD_ARG; }
    const size_t need = ac_sdf_stencil_backward_scratch(B);
    if (scratch_bytes < need) { ac::set_error("sdf_stencil_backward: scratch of %zu bytes needed, %zu given", need, scratch_bytes); return AC_ERR_BAD_ARG; }
    RenderArgs a{};
    if (int rc = prep_args(a, field, bound, eps)) return rc;
    const size_t lds_bytes = BWD_LDS_FLOATS * sizeof(float), lds_bytes_s = BWD_LDS_FLOATS_S * sizeof(float);
    static uint64_t seen = 0, seen_s = 0;
    ac::allow_dynamic_lds(seen, reinterpret_cast<const void *>(sdf_stencil_bwd_kernel<false>), lds_bytes);
    ac::allow_dynamic_lds(seen_s, reinterpret_cast<const void *>(sdf_stencil_bwd_kernel<true>), lds_bytes_s);
    uint32_t blocks = train_grid(B);
    if (feat7) {
        const uint32_t need_blocks = ((B + 15) / 16 + TW_S - 1) / TW_S;           // (persistent: one workgroup of TW_S waves per compute unit)
        if (blocks > need_blocks) blocks = need_blocks ? need_blocks : 1;
        hipLaunchKernelGGL(sdf_stencil_bwd_kernel<true>, dim3(blocks), dim3(TW_S * 64), lds_bytes_s, (hipStream_t)stream, a, x, g_out16, g_grad, B, eps, gfeat,
                           static_cast<float *>(scratch), feat7, (float *)nullptr);
    } else if (g_x) {
        static uint64_t seen_x = 0;
        ac::allow_dynamic_lds(seen_x, reinterpret_cast<const void *>(sdf_stencil_bwd_kernel<false, true>), lds_bytes);
        hipLaunchKernelGGL((sdf_stencil_bwd_kernel<false, true>), dim3(blocks), dim3(TBLOCK), lds_bytes, (hipStream_t)stream, a, x, g_out16, g_grad, B, eps, gfeat,
                           static_cast<float *>(scratch), feat7, g_x);
    } else
        hipLaunchKernelGGL(sdf_stencil_bwd_kernel<false>, dim3(blocks), dim3(TBLOCK), lds_bytes, (hipStream_t)stream, a, x, g_out16, g_grad, B, eps, gfeat,
                           static_cast<float *>(scratch), feat7, (float *)nullptr);
    hipLaunchKernelGGL(sdf_partials_reduce_kernel, dim3((NPART + RED_OUT - 1) / RED_OUT), dim3(1024), 0, (hipStream_t)stream, static_cast<const float *>(scratch),
                       blocks * (feat7 ? TW_S : TW), gparams);
    return ac::check_launch("sdf_stencil_backward");
}

AC_API int ac_sdf_stencil_backward(const ac_field *field, const float *x, const float *g_out16, const float *g_grad, uint32_t B, float bound,
                                   float eps, float *gfeat, float *gparams, void *scratch, size_t scratch_bytes, ac_stream_t stream)
{
    return sdf_stencil_backward_impl(field, x, g_out16, g_grad, B, bound, eps, gfeat, gparams, scratch, scratch_bytes, stream, nullptr);
}

AC_API int ac_sdf_stencil_backward_inputs(const ac_field *field, const float *x, const float *g_out16, const float *g_grad, uint32_t B, float bound,
                                          float eps, float *gfeat, float *gparams, float *g_x, void *scratch, size_t scratch_bytes, ac_stream_t stream)
{
    if (!g_x && B) { ac::set_error("sdf_stencil_backward_inputs: NULL g_x"); return AC_ERR_BAD_ARG; }
    return sdf_stencil_backward_impl(field, x, g_out16, g_grad, B, bound, eps, gfeat, gparams, scratch, scratch_bytes, stream, nullptr, g_x);
}

AC_API int ac_color_forward(const ac_field *field, const float *x, const float *normal, const float *sdf16, uint32_t B, float *rgb, ac_stream_t stream)
{
    if (B == 0) return AC_OK;
    if (!x || !normal || !sdf16 || !rgb) { ac::set_error("color_forward: NULL buffer"); return AC_ERR_BAD_ARG; }
    if (field && field->Wc1_sh) { ac::set_error("color_forward: a field with view directions goes through ac_field_color_dirs / the fused renderer (the operator has no direction input)"); return AC_ERR_BAD_ARG; }
    RenderArgs a{};
    if (int rc = prep_color_args(a, field)) return rc;
    const size_t lds_bytes = OFF_WAVE * sizeof(float);
    uint32_t blocks = ((B + 15) / 16 + TW - 1) / TW;
    const uint32_t resident = 3 * ac::cu_count();                 // 41 KB of LDS per workgroup: three per CU, persistent
    if (blocks > resident) blocks = resident;
    hipLaunchKernelGGL(color_fwd_kernel, dim3(blocks), dim3(TBLOCK), lds_bytes, (hipStream_t)stream, a, x, normal, sdf16, B, rgb);
    return ac::check_launch("color_forward");
}

AC_API size_t ac_color_backward_scratch(uint32_t B)
{
    return (size_t)train_grid(B) * TW * NPART_C * sizeof(float);
}

// ---- use_viewdirs: per-ray layer-1 bias of the colour network, bias[r][u] = sum_j Wc1_sh[u][j] sh_j(rays_d[r]) -- the bits ac_render_rays forms in its prologue
__global__ __launch_bounds__(256) void sh_bias_kernel(const float *__restrict__ Wsh, const float *__restrict__ rays_d, uint32_t N, float *__restrict__ bias,
                                                      float *__restrict__ sh_out)
{
    __shared__ float slab[4][80];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t ray = blockIdx.x * 4 + wave; ray < N; ray += gridDim.x * 4) {
        ray_sh_bias<true>(slab[wave], Wsh, rays_d[3 * (size_t)ray], rays_d[3 * (size_t)ray + 1], rays_d[3 * (size_t)ray + 2], lane);
        bias[(size_t)ray * 64 + lane] = slab[wave][lane];
        if (sh_out && lane < 16) sh_out[(size_t)ray * 16 + lane] = slab[wave][64 + lane];
    }
}

AC_API int ac_sh_bias(const ac_field *field, const float *rays_d, uint32_t N, float *bias, float *sh, ac_stream_t stream)
{
    if (N == 0) return AC_OK;
    if (!field || !field->Wc1_sh) { ac::set_error("sh_bias: the field has no view-direction weights (ac_field.Wc1_sh)"); return AC_ERR_BAD_ARG; }
    if (!rays_d || !bias) { ac::set_error("sh_bias: NULL buffer"); return AC_ERR_BAD_ARG; }
    uint32_t blocks = (N + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sh_bias_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, field->Wc1_sh, rays_d, N, bias, sh);
    return ac::check_launch("sh_bias");
}

static int color_backward_impl(const ac_field *field, const float *x, const float *normal, const float *sdf16, const float *g_rgb, uint32_t B,
                               float *g_normal, float *g_sdf16, float *gparams, void *scratch, size_t scratch_bytes, ac_stream_t stream,
                               const float *sh_bias, uint32_t T, float *g_sh_tiles);

AC_API int ac_color_backward(const ac_field *field, const float *x, const float *normal, const float *sdf16, const float *g_rgb, uint32_t B,
                             float *g_normal, float *g_sdf16, float *gparams, void *scratch, size_t scratch_bytes, ac_stream_t stream)
{
    if (field && field->Wc1_sh) { ac::set_error("color_backward: a field with view directions goes through ac_render_core_backward (the operator has no direction input)"); return AC_ERR_BAD_ARG; }
    return color_backward_impl(field, x, normal, sdf16, g_rgb, B, g_normal, g_sdf16, gparams, scratch, scratch_bytes, stream, nullptr, 16, nullptr);
}

static int color_backward_impl(const ac_field *field, const float *x, const float *normal, const float *sdf16, const float *g_rgb, uint32_t B,
                               float *g_normal, float *g_sdf16, float *gparams, void *scratch, size_t scratch_bytes, ac_stream_t stream,
                               const float *sh_bias, uint32_t T, float *g_sh_tiles)
{
    if (!gparams) { ac::set_error("color_backward: NULL gparams"); return AC_ERR_BAD_ARG; }
    if (B == 0) { hipMemsetAsync(gparams, 0, NPART_C * sizeof(float), (hipStream_t)stream); return AC_OK; }
    if (!x || !normal || !sdf16 || !g_rgb || !g_normal || !g_sdf16 || !scratch) { ac::set_error("color_backward: NULL buffer"); return AC_ERR_BAD_ARG; }
    const size_t need = ac_color_backward_scratch(B);
    if (scratch_bytes < need) { ac::set_error("color_backward: scratch of %zu bytes needed, %zu given", need, scratch_bytes); return AC_ERR_BAD_ARG; }
    RenderArgs a{};
    if (int rc = prep_color_args(a, field)) return rc;
    const size_t lds_bytes = CBWD_LDS_FLOATS * sizeof(float);
    static uint64_t seen = 0;
    ac::allow_dynamic_lds(seen, reinterpret_cast<const void *>(color_bwd_kernel), lds_bytes);
    const uint32_t blocks = train_grid(B);
    hipLaunchKernelGGL(color_bwd_kernel, dim3(blocks), dim3(TBLOCK), lds_bytes, (hipStream_t)stream, a, x, normal, sdf16, g_rgb, B, g_normal, g_sdf16,
                       static_cast<float *>(scratch), sh_bias, T ? T : 16u, g_sh_tiles);
    hipLaunchKernelGGL(partials_reduce_kernel, dim3((NPART_C + RED_OUT - 1) / RED_OUT), dim3(1024), 0, (hipStream_t)stream, static_cast<const float *>(scratch),
                       blocks * TW, (uint32_t)NPART_C, gparams);
    return ac::check_launch("color_backward");
}

static int comp_args(CompArgs &a, const char *who, const float *rays_o, const float *rays_d, const float *z, const float *sdf, const float *nrm,
                     const float *col, const float *bg, int32_t n_rays, int32_t T0, int32_t T, float bound, float inv_s, float car)
{
    if (!rays_o || !rays_d || !z || !sdf || !nrm || !col) { ac::set_error("%s: NULL buffer", who); return AC_ERR_BAD_ARG; }
    if (T0 <= 0 || T % 16 || T < T0 || T > 128) { ac::set_error("%s: T0=%d T=%d unsupported (T a multiple of 16, <= 128)", who, T0, T); return AC_ERR_BAD_ARG; }
    a.rays_o = rays_o; a.rays_d = rays_d; a.z = z; a.sdf = sdf; a.nrm = nrm; a.col = col; a.bg = bg;
    a.n_rays = n_rays; a.T0 = T0; a.T = T; a.bound = bound; a.inv_s = inv_s; a.car = car; a.one_m_car = (float)(1.0 - (double)car);
    a.inv_s_dev = nullptr;
    a.near_m = a.far_m = nullptr; a.mask = nullptr;
    return AC_OK;
}

AC_API int ac_composite_forward(const float *rays_o, const float *rays_d, const float *z_vals, const float *sdf, const float *normal, const float *color,
                                const float *bg, int32_t n_rays, int32_t num_steps, int32_t T, float bound, float inv_s, float cos_anneal_ratio,
                                float *image, float *weights_sum, float *depth, float *normal_map, float *weights, float *alpha, ac_stream_t stream)
{
    if (n_rays <= 0) return AC_OK;
    CompArgs a{};
    if (int rc = comp_args(a, "composite_forward", rays_o, rays_d, z_vals, sdf, normal, color, bg, n_rays, num_steps, T, bound, inv_s, cos_anneal_ratio)) return rc;
    if (!image || !weights_sum || !depth || !normal_map || !weights || !alpha) { ac::set_error("composite_forward: NULL output"); return AC_ERR_BAD_ARG; }
    int blocks = (n_rays + 3) / 4; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(composite_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, image, weights_sum, depth, normal_map, weights, alpha);
    return ac::check_launch("composite_forward");
}

AC_API int ac_composite_backward(const float *rays_o, const float *rays_d, const float *z_vals, const float *sdf, const float *normal, const float *color,
                                 const float *bg, int32_t n_rays, int32_t num_steps, int32_t T, float bound, float inv_s, float cos_anneal_ratio,
                                 const float *g_image, const float *g_weights_sum, const float *g_depth, const float *g_normal_map,
                                 float *g_sdf, float *g_normal, float *g_color, float *g_inv_s_per_ray, ac_stream_t stream)
{
    if (n_rays <= 0) return AC_OK;
    CompArgs a{};
    if (int rc = comp_args(a, "composite_backward", rays_o, rays_d, z_vals, sdf, normal, color, bg, n_rays, num_steps, T, bound, inv_s, cos_anneal_ratio)) return rc;
    if (!g_image || !g_weights_sum || !g_depth || !g_normal_map || !g_sdf || !g_normal || !g_color || !g_inv_s_per_ray) {
        ac::set_error("composite_backward: NULL buffer"); return AC_ERR_BAD_ARG;
    }
    int blocks = (n_rays + 3) / 4; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(composite_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, g_image, g_weights_sum, g_depth, g_normal_map, g_sdf,
                       g_normal, g_color, g_inv_s_per_ray);
    return ac::check_launch("composite_backward");
}

// ---- the whole render core backward -----------------------------------------------------------------------------------------------
namespace {
struct CoreLayout { size_t nrm, g_sdf, g_nrm_a, g_col, g_nrm_b, g_s16, g_grad, gfeat, part_sdf, part_col, hash, total, hash_bytes; };
CoreLayout core_layout(const ac_field *field, uint32_t B)
{
    CoreLayout l{};
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    l.nrm = take((size_t)B * 12); l.g_sdf = take((size_t)B * 4); l.g_nrm_a = take((size_t)B * 12); l.g_col = take((size_t)B * 12);
    l.g_nrm_b = take((size_t)B * 12); l.g_s16 = take((size_t)B * 64); l.g_grad = take((size_t)B * 12);
    l.gfeat = take((size_t)B * 7 * 16 * 2 * 4);
    l.part_sdf = take(ac_sdf_stencil_backward_scratch(B)); l.part_col = take(ac_color_backward_scratch(B));
    l.hash_bytes = field ? ac_hash_stencil_backward_scratch(field->offsets, 16, field->S, field->H, 16, B) : 0;
    l.hash = take(l.hash_bytes);
    l.total = o;
    return l;
}
}  // namespace

// hash_stencil.hip: ac_hash_stencil_backward with the accumulation split at a level and a side stream ordered behind the first part
int hash_stencil_backward_split(const float *grad, const float *x, const int32_t *offsets_host, float *grad_embeddings, uint32_t B,
                                uint32_t C, uint32_t L, float S, uint32_t H, float eps, float bound, void *scratch, size_t scratch_bytes,
                                ac_stream_t stream, uint32_t split_level, ac_stream_t side_stream);

AC_API size_t ac_render_core_backward_scratch(const ac_field *field, int32_t n_rays, int32_t T)
{
    if (!field || n_rays <= 0 || T <= 0) return 0;
    return core_layout(field, (uint32_t)n_rays * (uint32_t)T).total;
}

AC_API int ac_render_core_backward(const ac_field *field, const ac_render_opts *op, const float *rays_o, const float *rays_d, const float *bg,
                                   const ac_core_saved *sv, const ac_core_upstream *up, const ac_core_grads *gr, void *scratch, size_t scratch_bytes,
                                   ac_stream_t stream)
{
    if (!field || !op || !sv || !up || !gr) { ac::set_error("render_core_backward: NULL argument struct"); return AC_ERR_BAD_ARG; }
    if (op->n_rays <= 0) return AC_OK;
    const int N = op->n_rays, T0 = op->num_steps, T = T0 + op->upsample_steps;
    if (!rays_o || !rays_d || !sv->z_vals || !sv->pts || !sv->sdf || !sv->sdf_out16 || !sv->gradient || !sv->color || (up->g_eik && !sv->eik_den) ||
        !gr->g_table || !gr->g_sdf_params || !gr->g_color_params || !gr->g_inv_s_per_ray) {
        ac::set_error("render_core_backward: NULL buffer"); return AC_ERR_BAD_ARG;
    }
    if (!(op->fd_eps > 0.0f)) { ac::set_error("render_core_backward: fd_eps must be positive"); return AC_ERR_BAD_ARG; }
    const uint32_t B = (uint32_t)N * (uint32_t)T;
    const CoreLayout l = core_layout(field, B);
    if (!scratch || scratch_bytes < l.total) { ac::set_error("render_core_backward: scratch of %zu bytes needed, %zu given", l.total, scratch_bytes); return AC_ERR_BAD_ARG; }
    char *sb = static_cast<char *>(scratch);
    float *nrm = reinterpret_cast<float *>(sb + l.nrm), *g_sdf = reinterpret_cast<float *>(sb + l.g_sdf), *g_nrm_a = reinterpret_cast<float *>(sb + l.g_nrm_a);
    float *g_col = reinterpret_cast<float *>(sb + l.g_col), *g_nrm_b = reinterpret_cast<float *>(sb + l.g_nrm_b), *g_s16 = reinterpret_cast<float *>(sb + l.g_s16);
    float *g_grad = reinterpret_cast<float *>(sb + l.g_grad), *gfeat = reinterpret_cast<float *>(sb + l.gfeat);
    hipStream_t st = (hipStream_t)stream;
    const uint32_t eb = (B + 255) / 256;
    hipLaunchKernelGGL(core_normals_kernel, dim3(eb), dim3(256), 0, st, sv->gradient, B, nrm);
    {   // NeuS alpha + compositing (instant_nsr.py:219-263,290-299)
        CompArgs a{};
        if (int rc = comp_args(a, "render_core_backward", rays_o, rays_d, sv->z_vals, sv->sdf, nrm, sv->color, bg, N, T0, T, op->bound, op->inv_s, op->cos_anneal_ratio)) return rc;
        a.inv_s_dev = op->inv_s_dev;
        if ((op->near_m != nullptr) != (op->far_m != nullptr)) { ac::set_error("render_core_backward: near_m and far_m go together"); return AC_ERR_BAD_ARG; }
        a.near_m = op->near_m; a.far_m = op->far_m; a.mask = sv->mask;          // posed space: the range and the alpha mask of the forward
        int blocks = (N + 3) / 4; if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(composite_bwd_kernel, dim3(blocks), dim3(256), 0, st, a, up->g_image, up->g_weights_sum, up->g_depth, up->g_normal_map, g_sdf,
                           g_nrm_a, g_col, gr->g_inv_s_per_ray);
    }
    if ((field->Wc1_sh != nullptr) != (sv->sh_bias != nullptr) || (field->Wc1_sh != nullptr) != (gr->g_sh_tiles != nullptr)) {
        ac::set_error("render_core_backward: a field with view directions (ac_field.Wc1_sh) needs ac_core_saved.sh_bias and ac_core_grads.g_sh_tiles, one without takes neither");
        return AC_ERR_BAD_ARG;
    }
    if (int rc = color_backward_impl(field, sv->pts, nrm, sv->sdf_out16, g_col, B, g_nrm_b, g_s16, gr->g_color_params, sb + l.part_col,
                                     ac_color_backward_scratch(B), stream, sv->sh_bias, (uint32_t)T, gr->g_sh_tiles)) return rc;
    if (up->eik_group_rays < 0 || (up->eik_group_rays > 0 && up->eik_den_stride < 1)) { ac::set_error("render_core_backward: eik_group_rays < 0 or eik_den_stride < 1"); return AC_ERR_BAD_ARG; }
    hipLaunchKernelGGL(core_mid_kernel, dim3(eb), dim3(256), 0, st, sv->gradient, sv->pts, g_sdf, g_nrm_a, g_nrm_b, up->g_eik, sv->eik_den, B, g_s16, g_grad,
                       (uint32_t)up->eik_group_rays * (uint32_t)T, (uint32_t)(up->eik_group_rays > 0 ? up->eik_den_stride : 0));
    if (int rc = sdf_stencil_backward_impl(field, sv->pts, g_s16, g_grad, B, op->bound, op->fd_eps, gfeat, gr->g_sdf_params, sb + l.part_sdf,
                                           ac_sdf_stencil_backward_scratch(B), stream, sv->feat7)) return rc;
    if (int rc = hash_stencil_backward_split(gfeat, sv->pts, field->offsets, gr->g_table, B, 2, 16, field->S, field->H, op->fd_eps, op->bound,
                                             l.hash_bytes ? sb + l.hash : nullptr, l.hash_bytes, stream,
                                             gr->side_stream ? (uint32_t)(gr->split_level > 0 ? gr->split_level : 0) : 0u, gr->side_stream)) return rc;
    return ac::check_launch("render_core_backward");
}
