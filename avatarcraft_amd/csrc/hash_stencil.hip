// avatarcraft_amd/csrc/hash_stencil.hip -- the hash-grid encoder on the 7-point finite-difference stencil (training path).
//
// One SDF query of the render core is 7 encoder calls in the reference: the sample x (forward_sdf, models/instant_nsr.py:627-642)
// and x +- eps e_k clamped to the bound (finite_difference_normals_approximator, :687-704), each through
// HashEncoder.forward / _hash_encode.backward (encoder/hashencoder/hashgrid.py:11-73,126-142).  This operator evaluates the seven
// points of every sample in one launch.  The backward (the table-gradient scatter, profiles/r01_sds.txt) removes work in three
// steps: (1) on the levels where eps spans less than one cell all seven points lie in the same or a neighbouring cell, so their
// 7 x 8 corners collapse onto at most 32 distinct table entries (normally 8) in registers; (2) runs of lanes in the same cell
// (a wave holds 64 depth-sorted samples of a ray) are summed with a segmented shuffle scan; (3) the surviving records do not
// touch the table with atomics: they are queued per destination bucket and summed per bucket in LDS (BinSink below).  The same
// scatter serves the reference's one-point operator (hash_bwd_binned_kernel).
//
// Point order p = 0..6: x, +x, -x, +y, -y, +z, -z.  Layouts: x [B,3] fp32 world space (already clamped to the bound, as the
// render core passes it); features / their gradient [7, L, B, 2] (level-major like the reference's [L,B,C] kernel output).
// Arithmetic per point is the one of hash_fwd_kernel (fma(x, scale, 0.5), floor, trilinear weights as products in d order);
// the normalisation to [0,1] is (clamp(x +- eps) + bound) / (2 bound) with a true division, like the fused renderer.
#include <mutex>
#include "ac_common.hpp"
#include "ac_devmath.hpp"

using namespace acdev;

namespace {

#ifndef AC_RUN_EARLY_OUT
#define AC_RUN_EARLY_OUT 1     // the segmented run scan stops as soon as every lane of the wave has met the head of its run (run_reduce)
#endif
#ifndef AC_FILL_PREFETCH
#define AC_FILL_PREFETCH 1     // hash_stencil_bwd_binned_kernel requests the next group's inputs before it scatters the current one
#endif
#ifndef AC_ACC_SPLIT
#define AC_ACC_SPLIT 0         // 1: the two channels of the bucket sums in separate halves of the LDS slice (fewer bank conflicts of the LDS atomics) -- measured: nothing
#endif
#ifndef AC_ACC_W
#define AC_ACC_W 8             // bucket_accumulate_kernel: table elements per thread whose read-modify-write is issued together
#endif
#ifndef AC_ABL_FLUSH
#define AC_ABL_FLUSH 0
#endif
#ifdef AC_ABL_NOATOMIC      // timing ablation (tools/ablate_stencil.sh): keep the address math, drop the atomic itself
#define AC_ATOMIC_ADD(P, V) { if ((V) == 123456.789f) *(P) = (V); }
#else
#define AC_ATOMIC_ADD(P, V) unsafeAtomicAdd((P), (V))
#endif

__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

#ifdef AC_PROFILE_FILL
__device__ unsigned long long g_fill_prof[AC_MAX_LEVELS * 6];      // [level][combine, records, reservation, write-out, whole wave, flushes] s_memtime ticks summed over waves
#endif

struct LevelC { float scale; uint32_t stride1, size, hashed, mask; };

__device__ __forceinline__ uint32_t gindex(const LevelC &L, uint32_t x, uint32_t y, uint32_t z)
{
    uint32_t index;
    if (L.hashed) index = x ^ (y * 2654435761u) ^ (z * 805459861u);
    else index = x + (y + z * L.stride1) * L.stride1;
    if (L.mask) index &= L.mask;
    else if (index >= L.size) index %= L.size;
    return index;
}

// the same index from PRE-MULTIPLIED axis terms: ty = y * my, tz = z * mz with (my, mz) = the two hash primes on a hashed level, (stride, stride^2) on a dense one
// (x + (y + z s) s = x + y s + z s^2 in 32-bit arithmetic).  A cell's corners and its neighbour planes differ from it by +-1 / +2 along an axis, i.e. by multiples of
// my / mz: their terms are ADDS on the cell's -- two 32-bit multiplications (quarter rate) per point and level instead of two per corner (round 6: the scatter's fill
// kernel spent a quarter of its vector-pipe time in v_mul_lo_u32, each inside a predicated per-corner block the compiler cannot hoist it out of).  Same bits.
__device__ __forceinline__ uint32_t gindex_t(const LevelC &L, uint32_t tx, uint32_t ty, uint32_t tz)
{
    uint32_t index = L.hashed ? (tx ^ ty ^ tz) : (tx + ty + tz);
    if (L.mask) index &= L.mask;
    else if (index >= L.size) index %= L.size;
    return index;
}
__device__ __forceinline__ uint32_t level_my(const LevelC &L) { return L.hashed ? 2654435761u : L.stride1; }
__device__ __forceinline__ uint32_t level_mz(const LevelC &L) { return L.hashed ? 805459861u : L.stride1 * L.stride1; }

struct Loc { uint32_t pg; float fr; bool oob; };

// world coordinate -> cell + fraction on one axis
// inv_tb: RN(1 / two_bound) for a divisor unit_div is verified for (ac::verified_reciprocal), else 0 = IEEE division -- the same bits either way
__device__ __forceinline__ Loc locate(float xw, float bound, float two_bound, float scale, float inv_tb = 0.0f)
{
    const float u = inv_tb != 0.0f ? unit_div(xw + bound, two_bound, inv_tb) : (xw + bound) / two_bound;
    Loc r;
    r.oob = (u < 0.0f) | (u > 1.0f);
    const float p = fma_(u, scale, 0.5f);
    const float fl = __builtin_floorf(p);
    r.pg = (uint32_t)fl;
    r.fr = p - (float)r.pg;
    return r;
}

__device__ __forceinline__ LevelC level_of(const ac::LevelTable &lt, uint32_t level)
{
    LevelC L;
    L.scale = lt.scale[level]; L.stride1 = lt.stride1[level]; L.size = lt.size[level]; L.hashed = lt.hashed[level]; L.mask = lt.pow2mask[level];
    return L;
}

__device__ __forceinline__ float offset_coord(float xc, int sign, float eps, float bound)
{
    return clampf(xc + (sign ? -eps : eps), -bound, bound);
}

// ---- forward: out[p][level][b][0..1] ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hash_stencil_fwd_kernel(const float *__restrict__ x, const float *__restrict__ grid,
                                                               float *__restrict__ out, uint32_t B, ac::LevelTable lt, float eps,
                                                               float bound, float two_bound)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y, Lc = lt.L;
    const LevelC L = level_of(lt, level);
    const float2 *g = reinterpret_cast<const float2 *>(grid) + lt.offset[level];
    const float xc[3] = { x[3 * (size_t)b], x[3 * (size_t)b + 1], x[3 * (size_t)b + 2] };
    Loc c[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) c[d] = locate(xc[d], bound, two_bound, L.scale);
#pragma unroll
    for (int p = 0; p < 7; ++p) {
        Loc q[3] = { c[0], c[1], c[2] };
        if (p > 0) { const int k = (p - 1) >> 1; q[k] = locate(offset_coord(xc[k], (p - 1) & 1, eps, bound), bound, two_bound, L.scale); }
        float a0 = 0.0f, a1 = 0.0f;
        if (!(q[0].oob | q[1].oob | q[2].oob)) {
#pragma unroll
            for (uint32_t idx = 0; idx < 8; ++idx) {
                float w = 1.0f; uint32_t pl[3];
#pragma unroll
                for (uint32_t d = 0; d < 3; ++d) {
                    if ((idx & (1u << d)) == 0) { w *= 1.0f - q[d].fr; pl[d] = q[d].pg; }
                    else { w *= q[d].fr; pl[d] = q[d].pg + 1u; }
                }
                const float2 f = g[gindex(L, pl[0], pl[1], pl[2])];
                a0 = fma_(w, f.x, a0); a1 = fma_(w, f.y, a1);
            }
        }
        reinterpret_cast<float2 *>(out)[((size_t)p * Lc + level) * B + b] = make_float2(a0, a1);
    }
}

// ---- backward to the INPUT: d loss / d x through the seven encodings ------------------------------------------------------------------
// What the reference does with dy_dx (kernel_grid's derivative branch, hashencoder.cu:177-220, and kernel_input_backward, :311-337) for a sample whose
// position itself carries a gradient -- the curvature term's perturbed points (models/instant_nsr.py:276-288: they are a function of the normal).  Nothing
// is stored: the eight corners of every stencil point are gathered again and the derivative of the trilinear weights is formed in registers.
//   gx_part[grp][b][d] = sum over the levels of group grp (4 levels each), points p, channels c of
//        gfeat[p][level][b][c] * scale * sum_{corners of the two other axes} w_other * (T[corner + e_d] - T[corner])[c] / (2 bound) * pass_p,d
// pass = the derivative of the offset point's clamp (1 inside [-bound, bound], torch.clamp's inclusive rule; finite_difference_normals_approximator :690-702).
__global__ __launch_bounds__(256) void hash_stencil_input_bwd_kernel(const float *__restrict__ gfeat, const float *__restrict__ x, const float *__restrict__ grid,
                                                                     float *__restrict__ gx_part, uint32_t B, ac::LevelTable lt, float eps, float bound,
                                                                     float two_bound)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t Lc = lt.L;
    const float xc[3] = { x[3 * (size_t)b], x[3 * (size_t)b + 1], x[3 * (size_t)b + 2] };
    float acc[3] = { 0.0f, 0.0f, 0.0f };
    for (uint32_t level = blockIdx.y * 4; level < blockIdx.y * 4 + 4 && level < Lc; ++level) {
        const LevelC L = level_of(lt, level);
        const float2 *g = reinterpret_cast<const float2 *>(grid) + lt.offset[level];
        Loc c[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) c[d] = locate(xc[d], bound, two_bound, L.scale);
#pragma unroll 1
        for (int p = 0; p < 7; ++p) {
            const float2 gf = reinterpret_cast<const float2 *>(gfeat)[((size_t)p * Lc + level) * B + b];
            if (gf.x == 0.0f && gf.y == 0.0f) continue;                  // (the centre of a gradient-only use, masked samples)
            Loc q[3] = { c[0], c[1], c[2] };
            const int k = p > 0 ? (p - 1) >> 1 : 3;
            float pass = 1.0f;
            if (p > 0) {
                const float raw = xc[k] + (((p - 1) & 1) ? -eps : eps);
                pass = (raw >= -bound && raw <= bound) ? 1.0f : 0.0f;
                q[k] = locate(clampf(raw, -bound, bound), bound, two_bound, L.scale);
            }
            if (q[0].oob | q[1].oob | q[2].oob) continue;                // (hashencoder.cu:95-119: an out-of-range input has zero output and zero dy_dx)
            float2 f[8];
#pragma unroll
            for (uint32_t idx = 0; idx < 8; ++idx)
                f[idx] = g[gindex(L, q[0].pg + (idx & 1u), q[1].pg + ((idx >> 1) & 1u), q[2].pg + ((idx >> 2) & 1u))];
#pragma unroll
            for (uint32_t d = 0; d < 3; ++d) {
                const uint32_t da = (d == 0) ? 1u : 0u, db = (d == 2) ? 1u : 2u;
                float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
                for (uint32_t jm = 0; jm < 4; ++jm) {
                    const float w = ((jm & 1u) ? q[da].fr : 1.0f - q[da].fr) * ((jm >> 1) ? q[db].fr : 1.0f - q[db].fr);
                    const uint32_t left = ((jm & 1u) << da) | ((jm >> 1) << db), right = left | (1u << d);
                    s0 = fma_(w, f[right].x - f[left].x, s0); s1 = fma_(w, f[right].y - f[left].y, s1);
                }
                const float dd = L.scale * fma_(s0, gf.x, s1 * gf.y) / two_bound;
                acc[d] += ((int)d == k) ? dd * pass : dd;
            }
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) gx_part[((size_t)blockIdx.y * B + b) * 3 + d] = acc[d];
}

// ---- backward ----------------------------------------------------------------------------------------------------------------
// A wave holds 64 consecutive samples of one ray (B index = ray * T + sample, sorted by depth), and the importance sampling packs
// most of them into a few cells: neighbouring lanes that sit in the same cell address the same table entries.  Their
// contributions are summed inside the wave first (segmented inclusive scan over runs of equal cell, 6 shuffle steps) and only
// the last lane of each run issues the atomics.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_f(float v)
{
    // lanes without a valid source (first lanes of a row for row_shr, rows outside ROW_MASK for row_bcast) receive 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf));
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf); }

// one step of the segmented scan: every lane that has not met the head of its run yet (f == 0) adds the value CTRL points at.
// v += dpp(v) * m with m in {0, 1} is ONE v_fmac_f32_dpp per value (the compiler's DPP combiner does not fold a DPP move into
// v_fmac, hence the inline assembly).  The s_nop covers the "VALU write -> DPP read" (2) and "VALU writes EXEC -> DPP" (5) wait states for the first value of a step
// (the hazard recogniser does not look into inline assembly); within a step and between steps N - 1 >= 15 instructions lie
// between the write of a register and its DPP read.
#define AC_FMAC_DPP(MODS) asm volatile("v_fmac_f32_dpp %0, %0, %1 " MODS : "+v"(v[i]) : "v"(m))
template <int N, int CTRL, int ROW_MASK>
__device__ __forceinline__ void run_step(float (&v)[N], int &f)
{
    const float m = f ? 0.0f : 1.0f;
    asm volatile("s_nop 4");
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if constexpr (CTRL == 0x111) AC_FMAC_DPP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0");
        else if constexpr (CTRL == 0x112) AC_FMAC_DPP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0");
        else if constexpr (CTRL == 0x114) AC_FMAC_DPP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0");
        else if constexpr (CTRL == 0x118) AC_FMAC_DPP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0");
        else if constexpr (CTRL == 0x142) AC_FMAC_DPP("row_bcast:15 row_mask:0xa bank_mask:0xf");
        else AC_FMAC_DPP("row_bcast:31 row_mask:0xc bank_mask:0xf");
    }
    f |= dpp_i<CTRL, ROW_MASK>(f);
}
#undef AC_FMAC_DPP

// segmented inclusive sums over runs of lanes (head = first lane of a run): DPP only, no LDS traffic -- Kogge-Stone inside every
// 16-lane row (row_shr 1, 2, 4, 8), then the open tail of the previous row(s) is carried over with row_bcast15 / row_bcast31.
// Returns true in the last lane of every run (the one that holds the run's total).
template <int N>
__device__ __forceinline__ bool run_reduce(float (&v)[N], bool head, int lane, bool norun = false)
{
#ifdef AC_ABL_NORUN         // timing ablation: every lane is its own run
    return true;
#endif
    // norun (wave-uniform): a lane of this wave carries an Inf / NaN upstream gradient.  The scan below adds `neighbour * {0, 1}`, and
    // Inf * 0 = NaN would leak into every lane of the row: such a group scatters lane by lane, so that a non-finite gradient reaches
    // exactly the entries the reference's atomicAdd would give it (hashencoder.cu:302-305)
    if (norun) return true;
    int f = head ? 1 : 0;
#ifdef AC_RUN_SHUFFLE       // the first implementation: 6 Kogge-Stone steps over the whole wave with ds_bpermute
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int tf = __shfl_up(f, d);
        const bool take = (lane >= d) && !f;
#pragma unroll
        for (int i = 0; i < N; ++i) { const float t = __shfl_up(v[i], d); v[i] += take ? t : 0.0f; }
        if (lane >= d) f |= tf;
    }
#else
    // every value has to be final BEFORE the first DPP read (the scheduler may otherwise sink the instruction that produces v[i + 1]
    // between two of the inline-assembly statements, inside the hazard window): empty volatile statements pin them
    static_assert(N % 8 == 0, "run_reduce works on multiples of 8 values");
#pragma unroll
    for (int c = 0; c < N; c += 8)
        asm volatile("" : "+v"(v[c]), "+v"(v[c + 1]), "+v"(v[c + 2]), "+v"(v[c + 3]), "+v"(v[c + 4]), "+v"(v[c + 5]), "+v"(v[c + 6]), "+v"(v[c + 7]));
    // A step only moves values into lanes that have not met the head of their run yet (f == 0): once no such lane is left in the wave, the
    // remaining steps would add `neighbour * 0` everywhere and are skipped (wave-uniform branch; round 4: the scan is more than half of the
    // fill's vector instructions, and outside the densely sampled shell of the surface runs are a few lanes long on all but the coarsest levels).
    // The skipped additions of +-0 can only turn a -0 into +0: a value that the != 0 predicate drops or the fixed-point sum reads as 0 either way.
#if AC_RUN_EARLY_OUT
#define AC_RUN_DONE() (__builtin_amdgcn_ballot_w64(f == 0) == 0ull)
#else
#define AC_RUN_DONE() false
#endif
    do {
        if (AC_RUN_DONE()) break;
        run_step<N, 0x111, 0xf>(v, f);
        if (AC_RUN_DONE()) break;
        run_step<N, 0x112, 0xf>(v, f);
        if (AC_RUN_DONE()) break;
        run_step<N, 0x114, 0xf>(v, f);
        if (AC_RUN_DONE()) break;
        run_step<N, 0x118, 0xf>(v, f);
        if (AC_RUN_DONE()) break;
        run_step<N, 0x142, 0xa>(v, f);          // row_bcast15: rows 1 and 3 take lane 15 of rows 0 and 2
        if (AC_RUN_DONE()) break;
        run_step<N, 0x143, 0xc>(v, f);          // row_bcast31: rows 2 and 3 take lane 31
    } while (false);
#undef AC_RUN_DONE
#endif
    const int next_head = __shfl_down(head ? 1 : 0, 1);
    return lane == 63 || next_head != 0;
}

__device__ __forceinline__ bool run_head(const Loc (&q)[3], bool ok, int lane)
{
    const uint32_t p0 = __shfl_up(q[0].pg, 1), p1 = __shfl_up(q[1].pg, 1), p2 = __shfl_up(q[2].pg, 1);
    const int pok = __shfl_up(ok ? 1 : 0, 1);
    return lane == 0 || p0 != q[0].pg || p1 != q[1].pg || p2 != q[2].pg || pok != (ok ? 1 : 0);    // runs are homogeneous in `ok`
}

// ---- where the combined gradients go -------------------------------------------------------------------------------------------
// DirectSink: two hardware float atomics per table entry (the device does ~20.9 G of those per second, whatever the scope, table
// size or address spread: tools/atomics_bench.hip) -- used for the small dense levels (with private copies).
// BinSink: the hashed levels.  A record (entry, v0, v1) is appended to a per-wave LDS buffer; a flush bins the buffer by
// destination bucket (64 buckets of 8192 entries per level), reserves the slots of every bucket with ONE global atomic per
// (flush, bucket) and streams the records into per-bucket queues in HBM; bucket_accumulate_kernel then gives every bucket to
// one workgroup that sums its queue in LDS (ds_add_f32, ~1000x the global atomic rate) and adds the 64 KB slice to the table
// without atomics.  Global atomics per record: 2 -> ~1/30.
struct DirectSink {
    static constexpr bool packs = false;
    float2 *gg;
    __device__ __forceinline__ void tick(int) const {}
    // eight table entries at once: pred[k] says whether this lane contributes (v[2k], v[2k+1]) to entry index_of(k)
    template <class F> __device__ __forceinline__ void add8(const bool (&pred)[8], F index_of, const float (&v)[16]) const
    {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (pred[k]) { float *t = reinterpret_cast<float *>(gg + index_of(k)); AC_ATOMIC_ADD(t, v[2 * k]); AC_ATOMIC_ADD(t + 1, v[2 * k + 1]); }
    }
};

constexpr int NBUCKET = 64;
#ifndef AC_RCAP
#define AC_RCAP 1536
#endif
constexpr int RCAP = AC_RCAP;              // records per wave buffer; add8 reserves room for 8 x 64 records
#ifndef AC_FILL_PACK
#define AC_FILL_PACK 1      // coarse levels: the run tails' contributions transposed through LDS so that the records are formed by DENSE lanes (BinSink::add_packed)
#endif
#ifndef AC_FILL_PACK_MAX
#define AC_FILL_PACK_MAX 32  // most run tails per wave for which the packed path is taken (8 per chunk; above ~40 the sparse walk is cheaper)
#endif
constexpr int PACK_TAILS = 8;                          // run tails staged per chunk: 8 tails x 8 entries = the wave's 64 lanes
constexpr int PACK_STRIDE = 20;                        // staged words per tail: 16 values (8 entries x 2 channels) + the cell's three axis terms (+ 1 pad: 16-byte rows)
constexpr int STAGE_WORDS = AC_FILL_PACK ? PACK_TAILS * PACK_STRIDE : 0;
constexpr int WAVE_WORDS = 3 * RCAP + 2 * NBUCKET + STAGE_WORDS;     // LDS words per wave: ridx, rv0, rv1 [RCAP], hist, base [64], the packing stage
static_assert(RCAP % 2 == 0 && RCAP >= 1024, "wave buffer: room for two batches of 8 x 64 records");
#ifndef AC_REC16
#define AC_REC16 0          // 1: queue records padded to 16 bytes (one aligned dwordx4 store / load per record instead of three dword accesses; +33 % queue bytes)
#endif
#ifndef AC_REC8
#define AC_REC8 1           // 1 (round 4): 8-byte queue records -- 13 bits of entry index inside its bucket | v0 rounded to 16 explicit mantissa bits (25 bits) | v1
#endif                      //    rounded to 17 (26 bits).  PRECISION CONTRACT of the table gradient: every (entry, v0, v1) contribution is rounded
                            //    to nearest at 2^-17 / 2^-18 of its own magnitude before the order-independent fixed-point sum -- below the fp32 round-off of a sum
                            //    of a few records, checked against the fp64 oracle backward at 3e-4 of max (observed 1.2e-4, unchanged).  The queue traffic is what
                            //    the two scatter passes are short of: 16-byte records cost the step +0.21 ms, 8-byte ones gain (profiles/r04_experiments.txt 8b).
#if AC_REC16
struct __attribute__((aligned(16))) Rec { uint32_t idx; float v0, v1; uint32_t pad; };
#elif AC_REC8
struct __attribute__((aligned(8))) Rec { uint32_t lo, hi; };
// values are rounded when they are RECORDED (add8), so that the level's max |v| -- the fixed-point scale -- is taken over what is actually summed
__device__ __forceinline__ float rec_round(float v, int drop)        // round to nearest at bit `drop` (quiet NaN and Inf survive; FLT_MAX may round to Inf)
{
    return __uint_as_float((__float_as_uint(v) + (1u << (drop - 1))) & ~((1u << drop) - 1u));
}
__device__ __forceinline__ Rec rec_pack(uint32_t idx13, float v0, float v1)
{
    const uint32_t a = __float_as_uint(v0) >> 7, b = __float_as_uint(v1) >> 6;      // 25 and 26 bits (the dropped bits are zero: rec_round)
    Rec r; r.lo = idx13 | (a << 13); r.hi = (a >> 19) | (b << 6);
    return r;
}
__device__ __forceinline__ void rec_unpack(const Rec &r, uint32_t &idx13, float &v0, float &v1)
{
    idx13 = r.lo & 0x1fffu;
    v0 = __uint_as_float(((r.lo >> 13) | ((r.hi & 0x3fu) << 19)) << 7);
    v1 = __uint_as_float((r.hi >> 6) << 6);
}
#else
struct Rec { uint32_t idx; float v0, v1; };
#endif

struct BinSink {
    static constexpr bool packs = true;
    uint32_t *ridx; float *rv0, *rv1;      // this wave's LDS record buffer [RCAP]; ridx = entry | rank inside its bucket << 19
    uint32_t *hist, *base;                 // this wave's LDS [NBUCKET] each (hist zero between flushes)
    uint32_t cnt;                          // wave-uniform
    uint32_t sh;                           // bucket = index >> sh (2^sh entries per bucket, <= NBUCKET buckets per level)
    uint32_t mx;                           // per lane: bits of the largest |v| this lane has recorded (published once, by finish())
    uint32_t *qcount;                      // global [NBUCKET] of this level
    uint32_t *vmax;                        // global: bits of max |v| over this level's records (positive floats order like uints)
    Rec *queue;                            // global [NBUCKET][cap] of this level
    uint32_t cap;
    float2 *gg;                            // overflow path: direct atomics
    int lane;
    float *stage;                          // this wave's LDS [PACK_TAILS][PACK_STRIDE] (AC_FILL_PACK)
#ifdef AC_PROFILE_FILL                     // s_memtime per phase: 0 stencil combine + run scan, 1 records -> LDS, 2 flush: slot reservation, 3 flush: write-out
    unsigned long long ft, facc[4], nflush;
    __device__ __forceinline__ void tick(int slot) { const unsigned long long t2 = __builtin_amdgcn_s_memtime(); facc[slot] += t2 - ft; ft = t2; }
#else
    __device__ __forceinline__ void tick(int) {}
#endif
    // the bucket histogram is taken while the record is still in registers: the flush is then ONE pass over the buffer (three passes over LDS cost 1.2 of
    // the kernel's 1.85 ms, profiles/r01_sds.txt); that pass also rounds the values to the record's precision and tracks the running max |v| (round 6)
    // eight table entries at once (after reserving room for 8 x 64 records): the bucket ranks of all eight are requested first
    // (LDS atomics with return), then the records are written -- one LDS round trip per eight slots instead of eight
    template <class F> __device__ __forceinline__ void add8(const bool (&pred)[8], F index_of, const float (&v)[16])
    {
        if (cnt + 512u > (uint32_t)RCAP) flush();
        unsigned long long m[8];
        uint32_t index[8], rank[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            m[k] = __ballot(pred[k]);
            index[k] = 0u; rank[k] = 0u;
            if (m[k] != 0ull && pred[k]) {                                         // wave-uniform skip: most neighbour slots of a coarse level are empty
                index[k] = index_of(k);
                rank[k] = atomicAdd(&hist[index[k] >> sh], 1u);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (m[k] == 0ull) continue;
            if (pred[k]) {
                const uint32_t pos = cnt + (uint32_t)__builtin_popcountll(m[k] & ((1ull << lane) - 1ull));
                // (the contribution is stored as it is: its rounding to the record's precision and the running max |v| happen in flush(), on the DENSE record
                // stream -- here, after the run combining, a tenth to a half of the lanes are live and every instruction costs a full issue slot all the same)
                ridx[pos] = index[k] | (rank[k] << 19); rv0[pos] = v[2 * k]; rv1[pos] = v[2 * k + 1];
            }
            cnt += (uint32_t)__builtin_popcountll(m[k]);
        }
    }
    // one table entry per lane (the packed path: every lane may carry one)
    __device__ __forceinline__ void add1(bool pred, uint32_t index, float v0, float v1)
    {
        if (cnt + 64u > (uint32_t)RCAP) flush();
        const unsigned long long m = __ballot(pred);
        if (m == 0ull) return;
        if (pred) {
            const uint32_t rank = atomicAdd(&hist[index >> sh], 1u);
            const uint32_t pos = cnt + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
            ridx[pos] = index | (rank << 19); rv0[pos] = v0; rv1[pos] = v1;
        }
        cnt += (uint32_t)__builtin_popcountll(m);
    }
    // the level's largest |v| (the fixed-point scale of bucket_accumulate_kernel): one wave reduction and one global atomic per WAVE,
    // after its last flush (it used to be per flush: 0.26 of the fill's 1.8 ms, profiles/r01_sds.txt "without the max pass")
    __device__ __forceinline__ void finish()
    {
        flush();
        uint32_t wmx = mx;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)wmx, d); wmx = wmx > o ? wmx : o; }
        if (lane == 0 && wmx) atomicMax(vmax, wmx);
    }
    __device__ __forceinline__ void flush()
    {
#if AC_ABL_FLUSH == 1       // timing ablation: drop the records
        cnt = 0; return;
#endif
        tick(1);
        wave_sync_lds();
        {
            const uint32_t c = hist[lane];
            base[lane] = c ? atomicAdd(&qcount[lane], c) : 0u;
            hist[lane] = 0u;
        }
        wave_sync_lds();
        tick(2);
        // write-out, WU records per lane and trip: the LDS reads of a trip are issued together (one LDS latency per trip, not per record)
        constexpr uint32_t WU = 4;
        for (uint32_t i0 = lane; i0 < cnt; i0 += 64 * WU) {
            uint32_t packed[WU]; float v0[WU], v1[WU]; uint32_t bs[WU];
#pragma unroll
            for (uint32_t u = 0; u < WU; ++u) {
                const uint32_t i = i0 + 64 * u, ic = i < cnt ? i : i0;
                packed[u] = ridx[ic]; v0[u] = rv0[ic]; v1[u] = rv1[ic];
#if AC_REC8
                v0[u] = rec_round(v0[u], 7); v1[u] = rec_round(v1[u], 6);             // the record's precision (rec_pack drops nothing after this)
#endif
                if (i < cnt) {
                    const uint32_t a0 = __float_as_uint(v0[u]) & 0x7fffffffu, a1 = __float_as_uint(v1[u]) & 0x7fffffffu;
                    mx = mx > a0 ? mx : a0; mx = mx > a1 ? mx : a1;
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < WU; ++u) bs[u] = base[(packed[u] & 0x7ffffu) >> sh];
#pragma unroll
            for (uint32_t u = 0; u < WU; ++u) {
                if (i0 + 64 * u >= cnt) continue;
                const uint32_t idx = packed[u] & 0x7ffffu, bucket = idx >> sh;
                const uint32_t slot = bs[u] + (packed[u] >> 19);
                if (slot < cap) {
#if AC_REC8
                    const Rec r = rec_pack(idx - (bucket << sh), v0[u], v1[u]);
#else
                    Rec r; r.idx = idx - (bucket << sh); r.v0 = v0[u]; r.v1 = v1[u];
#endif
#if AC_REC16
                    r.pad = 0u;
#endif
#if AC_ABL_FLUSH == 2       // timing ablation: bin the records but do not write them
                    if (v0[u] == 123456.789f)
#endif
                    queue[(size_t)bucket * cap + slot] = r;
                } else {                    // queue full (never with the default sizing): fall back to atomics
                    float *t = reinterpret_cast<float *>(gg + idx); AC_ATOMIC_ADD(t, v0[u]); AC_ATOMIC_ADD(t + 1, v1[u]);
                }
            }
        }
        wave_sync_lds();
        cnt = 0;
        tick(3);
#ifdef AC_PROFILE_FILL
        ++nflush;
#endif
    }
};

// entries per bucket = 2^bucket_shift: the smallest power of two that covers the level with NBUCKET buckets
__host__ __device__ __forceinline__ uint32_t bucket_shift(uint32_t size)
{
    uint32_t sh = 0;
    while (((size + (1u << sh) - 1u) >> sh) > (uint32_t)NBUCKET) ++sh;
    return sh;
}
// Largest level the binned scatter takes (binned_levels): an entry's index inside its bucket must fit the 13 bits rec_pack gives it (8-byte records) and the
// 19 bits ridx gives the entry in the wave buffer (BinSink::add8, every record layout); a bucket's slice must fit bucket_accumulate_kernel's LDS image.
constexpr uint32_t BIN_IDX_BITS = 13, BIN_MAX_LEVEL = 1u << 19;
static_assert((uint32_t)NBUCKET << BIN_IDX_BITS >= BIN_MAX_LEVEL, "rec_pack: the in-bucket index of the largest binned level needs more than 13 bits");


// one point's 8 corners, combined over the run of lanes in the same cell
__device__ __forceinline__ bool nonfinite(float v) { return (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u; }

template <class Sink>
__device__ __forceinline__ void scatter8_runs(Sink &sink, const LevelC &L, const Loc (&q)[3], float g0, float g1, int lane, bool norun = false)
{
    const bool ok = !(q[0].oob | q[1].oob | q[2].oob);
    float v[16];
#pragma unroll
    for (uint32_t idx = 0; idx < 8; ++idx) {
        float w = 1.0f;
#pragma unroll
        for (uint32_t d = 0; d < 3; ++d) w *= ((idx >> d) & 1u) ? q[d].fr : 1.0f - q[d].fr;
        v[2 * idx] = ok ? w * g0 : 0.0f; v[2 * idx + 1] = ok ? w * g1 : 0.0f;
    }
#ifdef AC_FINE_NORUN        // experiment: no run combining on the per-point path (few same-cell neighbours on the fine levels)
    const bool tail = ok;
#else
    const bool tail = run_reduce<16>(v, run_head(q, ok, lane), lane, norun) && ok;
#endif
    sink.tick(0);
    bool pred[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) pred[k] = tail && (v[2 * k] != 0.0f || v[2 * k + 1] != 0.0f);
    const uint32_t my = level_my(L), mz = level_mz(L), ty0 = q[1].pg * my, tz0 = q[2].pg * mz;
    sink.add8(pred, [&](int k) { return gindex_t(L, q[0].pg + ((uint32_t)k & 1u), ty0 + ((((uint32_t)k >> 1) & 1u) ? my : 0u), tz0 + ((((uint32_t)k >> 2) & 1u) ? mz : 0u)); }, v);
    sink.tick(1);
}

// the seven stencil points of one sample per lane (64 consecutive samples per wave) on one level.
// fine: eps can reach a non-neighbouring cell on this level -> the seven points scatter independently
template <class Sink>
__device__ __forceinline__ void stencil_scatter(Sink &sink, const LevelC &L, bool fine, const float (&xc)[3], const float2 (&gp)[7], float eps,
                                                float bound, float two_bound, int lane, float inv_tb = 0.0f)
{
    Loc c[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) c[d] = locate(xc[d], bound, two_bound, L.scale, inv_tb);
    const bool cen_ok = !(c[0].oob | c[1].oob | c[2].oob);
    bool bad = false;
#pragma unroll
    for (int p = 0; p < 7; ++p) bad |= nonfinite(gp[p].x) | nonfinite(gp[p].y);
    const bool norun = __any(bad);                       // wave-uniform: Inf / NaN somewhere in this group of 64 samples (see run_reduce)

    if (fine || norun || !__all(cen_ok)) {               // wave-uniform: every lane takes the same path (shuffles inside)
#pragma unroll
        for (int p = 0; p < 7; ++p) {
            Loc q[3] = { c[0], c[1], c[2] };
            if (p > 0) { const int k = (p - 1) >> 1; q[k] = locate(offset_coord(xc[k], (p - 1) & 1, eps, bound), bound, two_bound, L.scale, inv_tb); }
            scatter8_runs(sink, L, q, gp[p].x, gp[p].y, lane, norun);
        }
        return;
    }
    // combine the seven points: v[2*idx+c] = the 8 corners of the centre cell; v[16 + ((k*2+side)*4+jm)*2 + c] = the 4 corners one
    // plane below (side 0) / above (side 1) the centre cell along axis k (jm = corner bits of the two other axes, in axis order)
    float v[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = 0.0f;
    const float wd[3][2] = { { 1.0f - c[0].fr, c[0].fr }, { 1.0f - c[1].fr, c[1].fr }, { 1.0f - c[2].fr, c[2].fr } };
#pragma unroll
    for (uint32_t idx = 0; idx < 8; ++idx) {
        float w = 1.0f;
#pragma unroll
        for (uint32_t d = 0; d < 3; ++d) w *= wd[d][(idx >> d) & 1u];
        v[2 * idx] = w * gp[0].x; v[2 * idx + 1] = w * gp[0].y;
    }
    // products of the two off-axis weights of a face corner, shared by the two offset points of an axis: pw[k][jm], jm = corner bits of the other two axes
    float pw[3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int da = (k == 0) ? 1 : 0, db = (k == 2) ? 1 : 2;
#pragma unroll
        for (uint32_t jm = 0; jm < 4; ++jm) pw[k][jm] = wd[da][jm & 1u] * wd[db][jm >> 1];
    }
    // Corner bit b (along axis k) of an offset point lies on plane delta + b of the centre cell, delta = its cell minus the centre's: 0, or +1 for
    // x + eps / -1 for x - eps on a coarse level.  Planes 0 and 1 are the centre cell's own corners, -1 and 2 the neighbour planes.  The two cases
    // are weighted with {0, 1} masks folded into the corner weight -- 4 fma per corner instead of 8 selects + 8 adds (round 3: the combine was 36 %
    // of the fill's time, tools/fill_profile.py).
#pragma unroll
    for (int p = 1; p < 7; ++p) {
        const int k = (p - 1) >> 1, sign = (p - 1) & 1;                 // sign 0: + eps, 1: - eps (offset_coord)
        const Loc q = locate(offset_coord(xc[k], sign, eps, bound), bound, two_bound, L.scale, inv_tb);
        const float live = q.oob ? 0.0f : 1.0f;
        const int delta = (int)q.pg - (int)c[k].pg;
        const float m0 = (delta == 0) ? live : 0.0f, m1 = (delta == (sign ? -1 : 1)) ? live : 0.0f;
        const float wk[2] = { 1.0f - q.fr, q.fr };
#pragma unroll
        for (uint32_t jm = 0; jm < 4; ++jm) {
            const uint32_t lo = jm & 1u, hi = jm >> 1;
            const uint32_t c0 = (k == 0) ? ((lo << 1) | (hi << 2)) : (k == 1) ? (lo | (hi << 2)) : (lo | (hi << 1));      // centre corner with bit k = 0
            const uint32_t c1 = c0 | (1u << k);
            const uint32_t e0 = 16 + ((k * 2 + 0) * 4 + jm) * 2, e1 = 16 + ((k * 2 + 1) * 4 + jm) * 2;
#pragma unroll
            for (uint32_t b = 0; b < 2; ++b) {
                const float w = pw[k][jm] * wk[b], wa = w * m0, wb = w * m1;
                const uint32_t ta = 2 * (b ? c1 : c0);                                      // same cell as the centre: plane b
                const uint32_t tb = sign ? (b ? 2 * c0 : e0) : (b ? e1 : 2 * c1);           // neighbouring cell: plane b - 1 / b + 1
                v[ta] = __builtin_fmaf(wa, gp[p].x, v[ta]); v[ta + 1] = __builtin_fmaf(wa, gp[p].y, v[ta + 1]);
                v[tb] = __builtin_fmaf(wb, gp[p].x, v[tb]); v[tb + 1] = __builtin_fmaf(wb, gp[p].y, v[tb + 1]);
            }
        }
    }
    const bool tail = run_reduce<64>(v, run_head(c, true, lane), lane);
    sink.tick(0);
    // the centre cell's pre-multiplied terms: every one of the 32 entries below is these plus / minus multiples of (1, cmy, cmz) (gindex_t)
    const uint32_t cmy = level_my(L), cmz = level_mz(L), cty = c[1].pg * cmy, ctz = c[2].pg * cmz;
#if AC_FILL_PACK
    if constexpr (Sink::packs) {
        // After the run combining only the TAIL lanes carry anything -- a handful per wave on the coarse levels -- and the four add8 calls below walk 32 entries with
        // a tenth of the lanes live (half of the fill's time: tools/fill_profile.py, round 6).  With few tails the work is transposed through LDS instead: a chunk of
        // 8 tails stages, per group of 8 entries, its 16 values (+ once: the cell's axis terms), and lane j of the wave forms the record of (tail j / 8, entry j % 8)
        // -- one dense pass per group instead of eight sparse ones.  The same (entry, v0, v1) contributions reach the queues (their order does not enter the sums).
        const unsigned long long T = __ballot(tail);
        const int nt = __builtin_popcountll(T);
        if (nt <= AC_FILL_PACK_MAX) {
            const int myrank = __builtin_popcountll(T & ((1ull << lane) - 1ull));
            const int t = lane >> 3, k = lane & 7;
            float *const st = sink.stage;
            for (int base = 0; base < nt; base += PACK_TAILS) {
                const bool mine = tail && myrank >= base && myrank < base + PACK_TAILS;
                float *const row = st + (myrank - base) * PACK_STRIDE;
                const bool valid = base + t < nt;
                uint32_t q0 = 0u, qy = 0u, qz = 0u;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (mine) {
#pragma unroll
                        for (int w = 0; w < 4; ++w)
                            *reinterpret_cast<float4 *>(row + 4 * w) = make_float4(v[16 * g + 4 * w], v[16 * g + 4 * w + 1], v[16 * g + 4 * w + 2], v[16 * g + 4 * w + 3]);
                        if (g == 0) { row[16] = __uint_as_float(c[0].pg); row[17] = __uint_as_float(cty); row[18] = __uint_as_float(ctz); }
                    }
                    wave_sync_lds();
                    const float2 val = *reinterpret_cast<const float2 *>(st + t * PACK_STRIDE + 2 * k);
                    if (g == 0) { q0 = __float_as_uint(st[t * PACK_STRIDE + 16]); qy = __float_as_uint(st[t * PACK_STRIDE + 17]); qz = __float_as_uint(st[t * PACK_STRIDE + 18]); }
                    uint32_t tx, ty, tz;
                    if (g == 0) {                                   // the centre cell's corner k
                        tx = q0 + ((uint32_t)k & 1u); ty = qy + ((k & 2) ? cmy : 0u); tz = qz + ((k & 4) ? cmz : 0u);
                    } else {                                        // slot k = side * 4 + jm of the two planes next to the cell along axis g - 1
                        const bool lo = (k & 1) != 0, hi = (k & 2) != 0, side = (k & 4) != 0;
                        if (g == 1) { tx = side ? q0 + 2u : q0 - 1u; ty = qy + (lo ? cmy : 0u); tz = qz + (hi ? cmz : 0u); }
                        else if (g == 2) { tx = q0 + (lo ? 1u : 0u); ty = side ? qy + 2u * cmy : qy - cmy; tz = qz + (hi ? cmz : 0u); }
                        else { tx = q0 + (lo ? 1u : 0u); ty = qy + (hi ? cmy : 0u); tz = side ? qz + 2u * cmz : qz - cmz; }
                    }
                    sink.add1(valid && (val.x != 0.0f || val.y != 0.0f), gindex_t(L, tx, ty, tz), val.x, val.y);
                    wave_sync_lds();                                // (the next group's values overwrite the rows)
                }
            }
            sink.tick(1);
            return;
        }
    }
#endif
    {
        bool pred[8]; float vv[16];
#pragma unroll
        for (int k = 0; k < 8; ++k) { vv[2 * k] = v[2 * k]; vv[2 * k + 1] = v[2 * k + 1]; pred[k] = tail && (v[2 * k] != 0.0f || v[2 * k + 1] != 0.0f); }
        sink.add8(pred, [&](int k) { return gindex_t(L, c[0].pg + ((uint32_t)k & 1u), cty + ((((uint32_t)k >> 1) & 1u) ? cmy : 0u), ctz + ((((uint32_t)k >> 2) & 1u) ? cmz : 0u)); }, vv);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {                            // the two planes next to the centre cell along axis k: slot n = side * 4 + jm
        bool pred[8]; float vv[16];
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const uint32_t e = 16 + (k * 8 + n) * 2;
            vv[2 * n] = v[e]; vv[2 * n + 1] = v[e + 1]; pred[n] = tail && (v[e] != 0.0f || v[e + 1] != 0.0f);
        }
        sink.add8(pred, [&](int n) {
            const uint32_t jm = (uint32_t)n & 3u, lo = jm & 1u, hi = jm >> 1;
            const uint32_t m3[3] = { 1u, cmy, cmz };
            uint32_t t[3];
            t[0] = c[0].pg + (k == 0 ? 0u : lo);
            t[1] = cty + ((k == 1 ? 0u : (k == 0 ? lo : hi)) ? cmy : 0u);
            t[2] = ctz + ((k == 2 ? 0u : hi) ? cmz : 0u);
            const uint32_t base = k == 0 ? c[0].pg : (k == 1 ? cty : ctz);
            t[k] = (n >> 2) ? base + 2u * m3[k] : base - m3[k];                  // plane + 2 / - 1 along axis k (32-bit wrap-around like the product's)
            return gindex_t(L, t[0], t[1], t[2]);
        }, vv);
    }
    sink.tick(1);
}

// direct atomics: one sample per thread, level = blockIdx.y in [0, n_levels)
__global__ __launch_bounds__(256) void hash_stencil_bwd_kernel(const float *__restrict__ grad, const float *__restrict__ x,
                                                               float *__restrict__ grad_grid, uint32_t B, ac::LevelTable lt, float eps,
                                                               float bound, float two_bound, uint32_t fine_mask, float *__restrict__ priv,
                                                               uint32_t n_priv, uint32_t priv_entries, uint32_t n_copies, uint32_t direct_mask)
{
    const uint32_t level = blockIdx.y, Lc = lt.L;
    if (!((direct_mask >> level) & 1u)) return;          // this level goes through the binned path
    const uint32_t b0 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = b0 < B;
    const uint32_t b = valid ? b0 : B - 1;
    const int lane = threadIdx.x & 63;
    const LevelC L = level_of(lt, level);
    // the small dense levels take bursts of same-address atomics from neighbouring rays: spread them over n_copies private
    // copies (one per workgroup, round robin), summed into the table by priv_reduce_kernel
    DirectSink sink;
    sink.gg = (level < n_priv) ? reinterpret_cast<float2 *>(priv) + (size_t)(blockIdx.x % n_copies) * priv_entries + lt.offset[level]
                               : reinterpret_cast<float2 *>(grad_grid) + lt.offset[level];
    const float xc[3] = { x[3 * (size_t)b], x[3 * (size_t)b + 1], x[3 * (size_t)b + 2] };
    float2 gp[7];
#pragma unroll
    for (int p = 0; p < 7; ++p) {
        gp[p] = reinterpret_cast<const float2 *>(grad)[((size_t)p * Lc + level) * B + b];
        if (!valid) gp[p] = make_float2(0.0f, 0.0f);
    }
    stencil_scatter(sink, L, ((fine_mask >> level) & 1u) != 0, xc, gp, eps, bound, two_bound, lane);
}

// binned path: blockIdx.y indexes the binned levels; every wave walks over groups of 64 samples and flushes its record buffer
// when the next batch might not fit
__global__ __launch_bounds__(256) void hash_stencil_bwd_binned_kernel(const float *__restrict__ grad, const float *__restrict__ x,
                                                                      float *__restrict__ grad_grid, uint32_t B, ac::LevelTable lt, float eps,
                                                                      float bound, float two_bound, uint32_t fine_mask, uint32_t binned_mask,
                                                                      uint32_t *__restrict__ qcount, Rec *__restrict__ queues, uint32_t cap, float inv_tb)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    // the finest levels cost 2.5x the coarse ones (every stencil point scatters on its own): dispatch them FIRST, so that the cheap
    // levels fill the tail of the launch.  ylev = which binned level this workgroup serves (queue / counter index), highest first.
    const uint32_t ylev = gridDim.y - 1u - blockIdx.y;
    // the ylev-th set bit of binned_mask
    uint32_t level = 0, seen = 0;
    for (uint32_t l = 0; l < lt.L; ++l) if ((binned_mask >> l) & 1u) { if (seen == ylev) level = l; ++seen; }
    const uint32_t Lc = lt.L;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const LevelC L = level_of(lt, level);
    uint32_t *wbase = smem + wave * WAVE_WORDS;
    BinSink sink;
    sink.ridx = wbase; sink.rv0 = reinterpret_cast<float *>(wbase + RCAP); sink.rv1 = reinterpret_cast<float *>(wbase + 2 * RCAP);
    sink.hist = wbase + 3 * RCAP; sink.base = sink.hist + NBUCKET;
    sink.stage = reinterpret_cast<float *>(sink.base + NBUCKET);
    sink.cnt = 0; sink.lane = lane;
    sink.sh = bucket_shift(lt.size[level]); sink.mx = 0u;
    sink.qcount = qcount + (size_t)ylev * NBUCKET;
    sink.vmax = qcount + (size_t)gridDim.y * NBUCKET + ylev;
    sink.queue = queues + (size_t)ylev * NBUCKET * cap;
    sink.cap = cap;
    sink.gg = reinterpret_cast<float2 *>(grad_grid) + lt.offset[level];
    sink.hist[lane] = 0u;
#ifdef AC_PROFILE_FILL
    sink.facc[0] = sink.facc[1] = sink.facc[2] = sink.facc[3] = 0ull; sink.nflush = 0ull; sink.ft = __builtin_amdgcn_s_memtime();
    const unsigned long long prof_t0 = sink.ft;
#endif
    const bool fine = ((fine_mask >> level) & 1u) != 0;
    const uint32_t ngroups = (B + 63) / 64;
    // the next group's position and seven feature gradients are requested before this group is scattered (round 4)
    struct GroupIn { float xc[3]; float2 gp[7]; };
    auto request = [&](uint32_t grp, GroupIn &in) {
        const uint32_t b0 = grp * 64 + lane;
        const bool valid = b0 < B;
        const uint32_t b = valid ? b0 : B - 1;
#pragma unroll
        for (int k = 0; k < 3; ++k) in.xc[k] = x[3 * (size_t)b + k];
#pragma unroll
        for (int p = 0; p < 7; ++p) {
            in.gp[p] = reinterpret_cast<const float2 *>(grad)[((size_t)p * Lc + level) * B + b];
            if (!valid) in.gp[p] = make_float2(0.0f, 0.0f);
        }
    };
    const uint32_t grp0 = blockIdx.x * 4 + wave, gstride = gridDim.x * 4;
    GroupIn nxt;
    if (grp0 < ngroups) request(grp0, nxt);
    for (uint32_t grp = grp0; grp < ngroups; grp += gstride) {
        const float xc[3] = { nxt.xc[0], nxt.xc[1], nxt.xc[2] };
        float2 gp[7];
#pragma unroll
        for (int p = 0; p < 7; ++p) gp[p] = nxt.gp[p];
#if AC_FILL_PREFETCH
        if (grp + gstride < ngroups) request(grp + gstride, nxt);
#endif
        stencil_scatter(sink, L, fine, xc, gp, eps, bound, two_bound, lane, inv_tb);
#if !AC_FILL_PREFETCH
        if (grp + gstride < ngroups) request(grp + gstride, nxt);
#endif
    }
    sink.finish();
#ifdef AC_PROFILE_FILL
    if (lane == 0) {
        unsigned long long *o = g_fill_prof + (size_t)level * 6;
        atomicAdd(o, sink.facc[0]); atomicAdd(o + 1, sink.facc[1]); atomicAdd(o + 2, sink.facc[2]); atomicAdd(o + 3, sink.facc[3]);
        atomicAdd(o + 4, __builtin_amdgcn_s_memtime() - prof_t0); atomicAdd(o + 5, sink.nflush);
    }
#endif
}

// the reference's one-point backward (kernel_grid_backward, hashencoder.cu:223-308; D = 3, C = 2) through the binned scatter:
// inputs [B,3] in [0,1], grad [L,B,2]; one sample per lane, runs of lanes in the same cell are combined like in the stencil path
__global__ __launch_bounds__(256) void hash_bwd_binned_kernel(const float *__restrict__ grad, const float *__restrict__ inputs,
                                                              float *__restrict__ grad_grid, uint32_t B, ac::LevelTable lt, uint32_t binned_mask,
                                                              uint32_t *__restrict__ qcount, Rec *__restrict__ queues, uint32_t cap)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t ylev = gridDim.y - 1u - blockIdx.y;            // finest level first (see hash_stencil_bwd_binned_kernel)
    uint32_t level = 0, seen = 0;
    for (uint32_t l = 0; l < lt.L; ++l) if ((binned_mask >> l) & 1u) { if (seen == ylev) level = l; ++seen; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const LevelC L = level_of(lt, level);
    uint32_t *wbase = smem + wave * WAVE_WORDS;
    BinSink sink;
    sink.ridx = wbase; sink.rv0 = reinterpret_cast<float *>(wbase + RCAP); sink.rv1 = reinterpret_cast<float *>(wbase + 2 * RCAP);
    sink.hist = wbase + 3 * RCAP; sink.base = sink.hist + NBUCKET;
    sink.stage = reinterpret_cast<float *>(sink.base + NBUCKET);
    sink.cnt = 0; sink.lane = lane;
    sink.sh = bucket_shift(lt.size[level]); sink.mx = 0u;
    sink.qcount = qcount + (size_t)ylev * NBUCKET;
    sink.vmax = qcount + (size_t)gridDim.y * NBUCKET + ylev;
    sink.queue = queues + (size_t)ylev * NBUCKET * cap;
    sink.cap = cap;
    sink.gg = reinterpret_cast<float2 *>(grad_grid) + lt.offset[level];
    sink.hist[lane] = 0u;
    const uint32_t ngroups = (B + 63) / 64;
    for (uint32_t grp = blockIdx.x * 4 + wave; grp < ngroups; grp += gridDim.x * 4) {
        const uint32_t b0 = grp * 64 + lane;
        const bool valid = b0 < B;
        const uint32_t b = valid ? b0 : B - 1;
        Loc q[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float u = inputs[(size_t)b * 3 + d];
            q[d].oob = (u < 0.0f) | (u > 1.0f);
            const float p = fma_(u, L.scale, 0.5f);
            q[d].pg = (uint32_t)__builtin_floorf(p);
            q[d].fr = p - (float)q[d].pg;
        }
        float2 g = reinterpret_cast<const float2 *>(grad)[(size_t)level * B + b];
        if (!valid) g = make_float2(0.0f, 0.0f);
        scatter8_runs(sink, L, q, g.x, g.y, lane, __any(nonfinite(g.x) | nonfinite(g.y)) != 0);
    }
    sink.finish();
}

// one workgroup per (bucket, binned level): sum the bucket's queue in LDS, then add the slice to the table (no atomics: the
// workgroup owns these entries, and every producer of records has finished -- kernel boundary).
// LDS float atomics (ds_add_f32) run at 0.33 lane-operations per clock and CU on gfx950, integer ones (ds_add_u32 / ds_add_u64) at
// > 3 (tools/lds_atomics_bench.hip), so the sums are taken in 64-bit fixed point: v * 2^k with 2^k chosen from the level's largest
// |v| (collected by the queue fill) and the length n of this queue, |v| 2^k < 2^(62 - ceil(log2(n + 1))), so that no entry can
// overflow.  Contributions keep >= 24 significant bits down to 2^-18 of the level's maximum and vanish below 2^-42 of it; in
// exchange the result does not depend on the order of the records (deterministic), which float atomics never gave.
// One launch covers the binned levels of rank y0 .. y0 + gridDim.y - 1 (rank = position among the set bits of binned_mask, n_binned of them).
__global__ __launch_bounds__(1024) void bucket_accumulate_kernel(float *__restrict__ grad_grid, ac::LevelTable lt, uint32_t binned_mask,
                                                                 const uint32_t *__restrict__ qcount, const Rec *__restrict__ queues, uint32_t cap,
                                                                 uint32_t y0, uint32_t n_binned)
{
    extern __shared__ __attribute__((aligned(16))) unsigned long long acc[];        // [entries per bucket][2]
    const uint32_t ylev = y0 + gridDim.y - 1u - blockIdx.y;       // longest queues (finest levels) first
    uint32_t level = 0, seen = 0;
    for (uint32_t l = 0; l < lt.L; ++l) if ((binned_mask >> l) & 1u) { if (seen == ylev) level = l; ++seen; }
    const uint32_t per = 1u << bucket_shift(lt.size[level]), bucket = blockIdx.x;
    const uint32_t first = bucket * per;
    const uint32_t mine = first >= lt.size[level] ? 0u : (lt.size[level] - first < per ? lt.size[level] - first : per);     // the last bucket may be short
    uint32_t n = qcount[(size_t)ylev * NBUCKET + bucket];
    n = n < cap ? n : cap;
    const uint32_t mbits = qcount[(size_t)n_binned * NBUCKET + ylev];
    if (n == 0 || mbits == 0 || mine == 0) return;                                   // wave-uniform: nothing queued for this bucket
    const Rec *q = queues + ((size_t)ylev * NBUCKET + bucket) * cap;
    float *dst = grad_grid + ((size_t)lt.offset[level] + (size_t)first) * 2;
    if ((mbits >> 23) == 255u) {
        // an Inf or NaN record on this level (the largest |v| carries exponent 255): no fixed-point scale exists.  Sum this bucket in float
        // instead (slow LDS float atomics, but only ever on a diverged step), so that Inf / NaN reach the table exactly as the
        // reference's atomicAdd would deliver them (hashencoder.cu:302-305) instead of being quantised away.
        float *accf = reinterpret_cast<float *>(acc);
        for (uint32_t e = threadIdx.x; e < per * 2; e += blockDim.x) accf[e] = 0.0f;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
#if AC_REC8
            uint32_t ri; float rv0, rv1; rec_unpack(q[i], ri, rv0, rv1);
#else
            const Rec r = q[i]; const uint32_t ri = r.idx; const float rv0 = r.v0, rv1 = r.v1;
#endif
            atomicAdd(&accf[2 * ri], rv0); atomicAdd(&accf[2 * ri + 1], rv1);
        }
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < mine * 2; e += blockDim.x)
            if (accf[e] != 0.0f) dst[e] += accf[e];                                  // NaN != 0 is true: NaN is written
        return;
    }
    for (uint32_t e = threadIdx.x; e < per * 2; e += blockDim.x) acc[e] = 0ull;
    __syncthreads();
    // fixed-point sums of the bucket's entries: channel c of entry i at AC_ACC_AT(i, c).  Split (round 4): the two channels in separate halves of the slice,
    // so that the 64 lanes of one LDS atomic spread over 32 bank pairs instead of 16 bank quads
#if AC_ACC_SPLIT
#define AC_ACC_AT(I, CH) ((I) + (CH) * per)
#else
#define AC_ACC_AT(I, CH) (2u * (I) + (CH))
#endif
    const int emax = (int)(mbits >> 23) - 126;                                       // |v| < 2^emax for every record of the level
    const int head = 32 - __builtin_clz(n);                                          // ceil(log2(n + 1))
    const int k = 62 - head - emax;
#ifndef AC_ACC_U
#define AC_ACC_U 4
#endif
    constexpr int U = AC_ACC_U;                          // queue records in flight per thread
    for (uint32_t i0 = threadIdx.x; i0 < n; i0 += blockDim.x * U) {
        Rec raw[U];
        struct { uint32_t idx; float v0, v1; } r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const uint32_t i = i0 + u * blockDim.x; raw[u] = q[i < n ? i : i0]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t i = i0 + u * blockDim.x;
#if AC_REC8
            rec_unpack(raw[u], r[u].idx, r[u].v0, r[u].v1);
#else
            r[u].idx = raw[u].idx; r[u].v0 = raw[u].v0; r[u].v1 = raw[u].v1;
#endif
            if (i >= n) { r[u].v0 = 0.0f; r[u].v1 = 0.0f; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#ifdef AC_ABL_NOLDSATOMIC
            if (r[u].v0 != 0.0f || r[u].v1 != 0.0f) { acc[AC_ACC_AT(r[u].idx, 0u)] = 1ull; acc[AC_ACC_AT(r[u].idx, 1u)] = 1ull; }
#else
            if (r[u].v0 != 0.0f) atomicAdd(&acc[AC_ACC_AT(r[u].idx, 0u)], (unsigned long long)__double2ll_rn(ldexp((double)r[u].v0, k)));
            if (r[u].v1 != 0.0f) atomicAdd(&acc[AC_ACC_AT(r[u].idx, 1u)], (unsigned long long)__double2ll_rn(ldexp((double)r[u].v1, k)));
#endif
        }
    }
    __syncthreads();
    // slice += sums, AC_ACC_W elements per thread at a time with their table loads issued together: one element per trip made every workgroup pay 16
    // dependent global round trips here (~1 / 4 of the kernel: r04_experiments.txt 8d)
    constexpr uint32_t W = AC_ACC_W;
    for (uint32_t e0 = threadIdx.x; e0 < mine * 2; e0 += blockDim.x * W) {
        long long v[W]; float old[W];
#pragma unroll
        for (uint32_t u = 0; u < W; ++u) { const uint32_t e = e0 + u * blockDim.x; v[u] = e < mine * 2 ? (long long)acc[AC_ACC_AT(e >> 1, e & 1u)] : 0ll; }
#pragma unroll
        for (uint32_t u = 0; u < W; ++u) old[u] = v[u] != 0 ? dst[e0 + u * blockDim.x] : 0.0f;
#pragma unroll
        for (uint32_t u = 0; u < W; ++u)
            if (v[u] != 0) dst[e0 + u * blockDim.x] = old[u] + (float)ldexp((double)v[u], -k);
    }
}

__global__ __launch_bounds__(256) void priv_reduce_kernel(const float *__restrict__ priv, uint32_t n_floats, uint32_t n_copies,
                                                          float *__restrict__ grad_grid)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_floats) return;
    float s = 0.0f;
    for (uint32_t c = 0; c < n_copies; ++c) s += priv[(size_t)c * n_floats + i];
    if (s != 0.0f) grad_grid[i] += s;
}

int check(const char *who, uint32_t C, uint32_t L, const int32_t *offsets_host, float eps, float bound)
{
    if (C != 2) { ac::set_error("%s: level_dim must be 2, got %u", who, C); return AC_ERR_BAD_ARG; }
    if (L == 0 || L > AC_MAX_LEVELS) { ac::set_error("%s: L=%u out of range (1..%d)", who, L, AC_MAX_LEVELS); return AC_ERR_BAD_ARG; }
    if (!offsets_host) { ac::set_error("%s: offsets_host is NULL", who); return AC_ERR_BAD_ARG; }
    if (!(eps > 0.0f) || !(bound > 0.0f)) { ac::set_error("%s: eps and bound must be positive", who); return AC_ERR_BAD_ARG; }
    return AC_OK;
}

}  // namespace

AC_API int ac_hash_stencil_forward(const float *x, const float *embeddings, const int32_t *offsets_host, float *outputs, uint32_t B,
                                   uint32_t C, uint32_t L, float S, uint32_t H, float eps, float bound, ac_stream_t stream)
{
    if (int rc = check("hash_stencil_forward", C, L, offsets_host, eps, bound)) return rc;
    if (B == 0) return AC_OK;
    if (!x || !embeddings || !outputs) { ac::set_error("hash_stencil_forward: NULL buffer"); return AC_ERR_BAD_ARG; }
    ac::LevelTable lt; ac::make_level_table(lt, L, 3, S, H, offsets_host);
    hipLaunchKernelGGL(hash_stencil_fwd_kernel, dim3((B + 255) / 256, L), dim3(256), 0, (hipStream_t)stream, x, embeddings, outputs, B, lt, eps,
                       bound, (float)(2.0 * (double)bound));
    return ac::check_launch("hash_stencil_forward");
}

AC_API int ac_hash_stencil_input_backward(const float *gfeat, const float *x, const float *embeddings, const int32_t *offsets_host, float *gx_part,
                                          uint32_t B, uint32_t C, uint32_t L, float S, uint32_t H, float eps, float bound, ac_stream_t stream)
{
    if (int rc = check("hash_stencil_input_backward", C, L, offsets_host, eps, bound)) return rc;
    if (B == 0) return AC_OK;
    if (!gfeat || !x || !embeddings || !gx_part) { ac::set_error("hash_stencil_input_backward: NULL buffer"); return AC_ERR_BAD_ARG; }
    ac::LevelTable lt; ac::make_level_table(lt, L, 3, S, H, offsets_host);
    hipLaunchKernelGGL(hash_stencil_input_bwd_kernel, dim3((B + 255) / 256, (L + 3) / 4), dim3(256), 0, (hipStream_t)stream, gfeat, x, embeddings, gx_part, B,
                       lt, eps, bound, (float)(2.0 * (double)bound));
    return ac::check_launch("hash_stencil_input_backward");
}

// entries of the leading dense levels that are worth privatising (none if the caller gives no scratch)
static uint32_t priv_levels(const ac::LevelTable &lt, uint32_t L, uint32_t &entries)
{
    uint32_t n = 0; entries = 0;
    while (n < L && !lt.hashed[n] && lt.size[n] <= (1u << 18) && lt.offset[n] == entries) { entries += lt.size[n]; ++n; }
    return n;
}

// levels that go through the binned path: 64 buckets of <= 8192 entries (64 KB of LDS), i.e. every level of up to 2^19 entries
static uint32_t binned_levels(const ac::LevelTable &lt, uint32_t L)
{
    uint32_t m = 0;
    for (uint32_t l = 0; l < L; ++l)
        if (lt.size[l] >= (uint32_t)NBUCKET && lt.size[l] <= BIN_MAX_LEVEL && bucket_shift(lt.size[l]) <= BIN_IDX_BITS) m |= 1u << l;
    return m;
}
static uint32_t queue_cap(uint32_t B, uint32_t per_sample = 56u) { return (uint32_t)(((uint64_t)B * per_sample * 3u / 2u) / NBUCKET) + 4096u; }   // 1.5 x the average

struct StencilScratch { size_t priv_off, qcount_off, queue_off, total; uint32_t entries, n_priv, binned_mask, n_binned, cap; };
static StencilScratch stencil_layout(const ac::LevelTable &lt, uint32_t L, uint32_t n_copies, uint32_t B, uint32_t per_sample = 56u)
{
    StencilScratch sc{};
    sc.n_priv = (n_copies >= 2 && B == 0) ? priv_levels(lt, L, sc.entries) : 0;       // with queues (B > 0) every level is binned
    size_t off = 0;
    sc.priv_off = off; off += ((size_t)sc.entries * 8 * (sc.n_priv ? n_copies : 0) + 255) & ~(size_t)255;
    sc.binned_mask = B ? binned_levels(lt, L) : 0;
    sc.n_binned = (uint32_t)__builtin_popcount(sc.binned_mask);
    sc.cap = queue_cap(B, per_sample);
    sc.qcount_off = off; off += ((size_t)sc.n_binned * (NBUCKET + 1) * 4 + 255) & ~(size_t)255;       // slot counters + the level's max |v|
    sc.queue_off = off; off += (size_t)sc.n_binned * NBUCKET * sc.cap * sizeof(Rec);
    sc.total = off;
    return sc;
}

AC_API size_t ac_hash_stencil_backward_scratch(const int32_t *offsets_host, uint32_t L, float S, uint32_t H, uint32_t n_copies, uint32_t B)
{
    if (!offsets_host || L == 0 || L > AC_MAX_LEVELS) return 0;
    ac::LevelTable lt; ac::make_level_table(lt, L, 3, S, H, offsets_host);
    return stencil_layout(lt, L, n_copies, B).total;
}

// split_level / side_stream (ac_core_grads, data-parallel training): the accumulation runs in two launches, levels >= split_level first; an event
// recorded after that launch is waited for by side_stream, so that work the caller enqueues there (the all-reduce of that part of the table
// gradient) starts while the second launch -- and whatever follows on `stream` -- still runs.  side_stream == NULL: one launch, no event.
// side waits for everything enqueued on st so far (one event per device, recorded and waited for under one lock: host threads cannot interleave)
static bool order_side_stream(hipStream_t st, hipStream_t side)
{
    static std::mutex mu;
    static hipEvent_t ev[64];
    static bool have[64];
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    dev &= 63;
    if (!have[dev]) { if (hipEventCreateWithFlags(&ev[dev], hipEventDisableTiming) != hipSuccess) return false; have[dev] = true; }
    return hipEventRecord(ev[dev], st) == hipSuccess && hipStreamWaitEvent(side, ev[dev], 0) == hipSuccess;
}

int hash_stencil_backward_split(const float *grad, const float *x, const int32_t *offsets_host, float *grad_embeddings, uint32_t B,
                                uint32_t C, uint32_t L, float S, uint32_t H, float eps, float bound, void *scratch, size_t scratch_bytes,
                                ac_stream_t stream, uint32_t split_level, ac_stream_t side_stream);

AC_API int ac_hash_stencil_backward(const float *grad, const float *x, const int32_t *offsets_host, float *grad_embeddings, uint32_t B,
                                    uint32_t C, uint32_t L, float S, uint32_t H, float eps, float bound, void *scratch, size_t scratch_bytes,
                                    ac_stream_t stream)
{
    return hash_stencil_backward_split(grad, x, offsets_host, grad_embeddings, B, C, L, S, H, eps, bound, scratch, scratch_bytes, stream, 0, nullptr);
}

int hash_stencil_backward_split(const float *grad, const float *x, const int32_t *offsets_host, float *grad_embeddings, uint32_t B,
                                uint32_t C, uint32_t L, float S, uint32_t H, float eps, float bound, void *scratch, size_t scratch_bytes,
                                ac_stream_t stream, uint32_t split_level, ac_stream_t side_stream)
{
    // whatever happens below, a caller that passed a side stream will enqueue work on it that reads the table gradient: on every early return the side
    // stream is ordered behind what `stream` holds so far (the accumulations of earlier patches), like the normal path orders it behind the split launch
    auto leave = [&](int rc) { if (side_stream) (void)order_side_stream((hipStream_t)stream, (hipStream_t)side_stream); return rc; };
    if (int rc = check("hash_stencil_backward", C, L, offsets_host, eps, bound)) return leave(rc);
    if (B == 0) return leave(AC_OK);
    if (!grad || !x || !grad_embeddings) { ac::set_error("hash_stencil_backward: NULL buffer"); return leave(AC_ERR_BAD_ARG); }
    ac::LevelTable lt; ac::make_level_table(lt, L, 3, S, H, offsets_host);
    const float two_bound = (float)(2.0 * (double)bound);
    hipStream_t st = (hipStream_t)stream;
    uint32_t fine_mask = 0;
    for (uint32_t l = 0; l < L; ++l) {           // same rule as the fused renderer's jfine (render_fused.hip)
        const double cells = (double)eps / (double)two_bound * (double)lt.scale[l];
        if (!(cells * 1.001 + 1e-3 < 1.0)) fine_mask |= 1u << l;
    }
    // scratch: the layout for the largest copy count (<= 64) that fits; without scratch everything goes through direct atomics
    StencilScratch sc{};
    uint32_t n_copies = 1;
    if (scratch) {
        for (uint32_t k = 64; k >= 1; k >>= 1) {
            sc = stencil_layout(lt, L, k, B);
            if (sc.total <= scratch_bytes) { n_copies = k; break; }
            if (k == 1) { sc = StencilScratch{}; }
        }
    }
    char *sb = static_cast<char *>(scratch);
    float *priv = sc.n_priv ? reinterpret_cast<float *>(sb + sc.priv_off) : nullptr;
    if (sc.n_priv) hipMemsetAsync(priv, 0, (size_t)sc.entries * 8 * n_copies, st);
    uint32_t *qcount = sc.n_binned ? reinterpret_cast<uint32_t *>(sb + sc.qcount_off) : nullptr;
    Rec *queues = sc.n_binned ? reinterpret_cast<Rec *>(sb + sc.queue_off) : nullptr;
    const uint32_t all = L >= 32 ? 0xffffffffu : ((1u << L) - 1u);
    const uint32_t direct_mask = all & ~sc.binned_mask;
    if (direct_mask)
        hipLaunchKernelGGL(hash_stencil_bwd_kernel, dim3((B + 255) / 256, L), dim3(256), 0, st, grad, x, grad_embeddings, B, lt, eps, bound, two_bound,
                           fine_mask, priv, sc.n_priv, sc.entries, n_copies, direct_mask);
    if (sc.n_binned) {
        hipMemsetAsync(qcount, 0, (size_t)sc.n_binned * (NBUCKET + 1) * 4, st);
        static uint64_t seen1 = 0, seen2 = 0;
        const size_t lds1 = (size_t)4 * WAVE_WORDS * 4, lds2 = (size_t)(1u << 19) / NBUCKET * 16;
        ac::allow_dynamic_lds(seen1, reinterpret_cast<const void *>(hash_stencil_bwd_binned_kernel), lds1);
        ac::allow_dynamic_lds(seen2, reinterpret_cast<const void *>(bucket_accumulate_kernel), lds2);
        uint32_t gx = ((B + 63) / 64 + 3) / 4;
#ifndef AC_FILL_GX
#define AC_FILL_GX 96      // workgroups per level.  Round 2: 256 -> 128 was 3.66 -> 3.39 ms of backward per step (96: 3.51, 64: 3.53, 32: 3.77; profiles/r02_experiments.txt);
                           // round 6, after the fill's index arithmetic shrank: 96 = 2.14 ms against 128: 2.18, 112: 2.15, 80: 2.20, 64: 2.21, 160: 2.29 (16 levels x 96 x 4 waves =
                           // three full rounds of the device's 2048 wave slots; the sums are order-independent, the bits do not depend on this)
#endif
        if (gx > AC_FILL_GX) gx = AC_FILL_GX;            // persistent waves: full record buffers per flush, few partial last ones
        hipLaunchKernelGGL(hash_stencil_bwd_binned_kernel, dim3(gx, sc.n_binned), dim3(256), lds1, st, grad, x, grad_embeddings, B, lt, eps, bound,
                           two_bound, fine_mask, sc.binned_mask, qcount, queues, sc.cap, ac::verified_reciprocal(two_bound));
        // binned levels >= split_level have ranks n_lo .. n_binned - 1
        const uint32_t n_lo = (uint32_t)__builtin_popcount(sc.binned_mask & (split_level >= 32 ? 0xffffffffu : ((1u << split_level) - 1u)));
        const bool every_hi_binned = ((all & ~sc.binned_mask) >> (split_level >= 32 ? 31 : split_level)) == 0 && split_level < 32;
        if (side_stream && every_hi_binned && n_lo > 0 && n_lo < sc.n_binned && !sc.n_priv) {
            hipLaunchKernelGGL(bucket_accumulate_kernel, dim3(NBUCKET, sc.n_binned - n_lo), dim3(1024), lds2, st, grad_embeddings, lt, sc.binned_mask, qcount,
                               queues, sc.cap, n_lo, sc.n_binned);
            if (!order_side_stream(st, (hipStream_t)side_stream)) {
                ac::set_error("hash_stencil_backward: cannot order the side stream behind the first level group"); return AC_ERR_BAD_ARG;
            }
            side_stream = nullptr;                                           // (done: the fallback below is for the cases that could not split)
            hipLaunchKernelGGL(bucket_accumulate_kernel, dim3(NBUCKET, n_lo), dim3(1024), lds2, st, grad_embeddings, lt, sc.binned_mask, qcount,
                               queues, sc.cap, 0u, sc.n_binned);
        } else {
            hipLaunchKernelGGL(bucket_accumulate_kernel, dim3(NBUCKET, sc.n_binned), dim3(1024), lds2, st, grad_embeddings, lt, sc.binned_mask, qcount,
                               queues, sc.cap, 0u, sc.n_binned);
        }
    }
    if (sc.n_priv)
        hipLaunchKernelGGL(priv_reduce_kernel, dim3((sc.entries * 2 + 255) / 256), dim3(256), 0, st, priv, sc.entries * 2, n_copies, grad_embeddings);
    if (side_stream) {                          // no split happened (direct levels, no scratch): the side stream waits for everything
        if (!order_side_stream(st, (hipStream_t)side_stream)) { ac::set_error("hash_stencil_backward: cannot order the side stream"); return AC_ERR_BAD_ARG; }
    }
    return ac::check_launch("hash_stencil_backward");
}

#ifdef AC_PROFILE_FILL
AC_API void ac_debug_fill_prof(unsigned long long *out, int reset)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fill_prof), sizeof(unsigned long long) * AC_MAX_LEVELS * 6);
    if (reset) { static unsigned long long z[AC_MAX_LEVELS * 6] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fill_prof), z, sizeof(z)); }
}
#endif

// ---- the reference's operator (ac_hash_encode_backward) with caller scratch: binned scatter when D = 3, C = 2 and every level fits
AC_API size_t ac_hash_encode_backward_scratch(const int32_t *offsets_host, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t B)
{
    if (!offsets_host || D != 3 || C != 2 || L == 0 || L > AC_MAX_LEVELS || B == 0) return 0;
    ac::LevelTable lt; ac::make_level_table(lt, L, 3, S, H, offsets_host);
    const uint32_t all = L >= 32 ? 0xffffffffu : ((1u << L) - 1u);
    if (binned_levels(lt, L) != all) return 0;
    return stencil_layout(lt, L, 0, B, 8u).total;
}

AC_API int ac_hash_encode_backward_ws(const float *grad, const float *inputs, const float *embeddings, const int32_t *offsets,
                                      const int32_t *offsets_host, float *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                      uint32_t H, int calc_grad_inputs, const float *dy_dx, float *grad_inputs, void *scratch, size_t scratch_bytes,
                                      ac_stream_t stream)
{
    const size_t need = ac_hash_encode_backward_scratch(offsets_host, D, C, L, S, H, B);
    if (!scratch || need == 0 || scratch_bytes < need || calc_grad_inputs)         // not coverable: the direct-atomic operator
        return ac_hash_encode_backward(grad, inputs, embeddings, offsets, offsets_host, grad_embeddings, B, D, C, L, S, H, calc_grad_inputs, dy_dx,
                                       grad_inputs, stream);
    if (!grad || !inputs || !grad_embeddings) { ac::set_error("hash_encode_backward: NULL buffer"); return AC_ERR_BAD_ARG; }
    ac::LevelTable lt; ac::make_level_table(lt, L, 3, S, H, offsets_host);
    const StencilScratch sc = stencil_layout(lt, L, 0, B, 8u);
    hipStream_t st = (hipStream_t)stream;
    char *sb = static_cast<char *>(scratch);
    uint32_t *qcount = reinterpret_cast<uint32_t *>(sb + sc.qcount_off);
    Rec *queues = reinterpret_cast<Rec *>(sb + sc.queue_off);
    hipMemsetAsync(qcount, 0, (size_t)sc.n_binned * (NBUCKET + 1) * 4, st);
    static uint64_t seen1 = 0, seen2 = 0;
    const size_t lds1 = (size_t)4 * WAVE_WORDS * 4, lds2 = (size_t)(1u << 19) / NBUCKET * 16;
    ac::allow_dynamic_lds(seen1, reinterpret_cast<const void *>(hash_bwd_binned_kernel), lds1);
    ac::allow_dynamic_lds(seen2, reinterpret_cast<const void *>(bucket_accumulate_kernel), lds2);
    uint32_t gx = ((B + 63) / 64 + 3) / 4;
    if (gx > AC_FILL_GX) gx = AC_FILL_GX;
    hipLaunchKernelGGL(hash_bwd_binned_kernel, dim3(gx, sc.n_binned), dim3(256), lds1, st, grad, inputs, grad_embeddings, B, lt, sc.binned_mask, qcount,
                       queues, sc.cap);
    hipLaunchKernelGGL(bucket_accumulate_kernel, dim3(NBUCKET, sc.n_binned), dim3(1024), lds2, st, grad_embeddings, lt, sc.binned_mask, qcount, queues,
                       sc.cap, 0u, sc.n_binned);
    return ac::check_launch("hash_encode_backward");
}
