// avatarcraft_amd/csrc/rm_device.hpp -- device-side pieces of the occupancy-grid ray marcher (raymarching/src/raymarching.cu:56-222, 497-599), shared by
// the stand-alone marching operators (raymarching.hip) and the fused occupancy renderer (sdf_train.hip): step sizes, the voxel lookup, the skip to the next
// voxel, the slab test, pcg32.  Anonymous namespace: each translation unit gets its own copy.
#pragma once
#include "ac_common.hpp"
#include "ac_devmath.hpp"

namespace {

constexpr int RM_MAX_STEPS = 1024;
constexpr float RM_SQRT3 = 1.73205080757f;
constexpr float RM_MIN_NEAR = 0.05f;

__device__ __forceinline__ float rm_clamp(float x, float lo, float hi) { return __builtin_fminf(hi, __builtin_fmaxf(lo, x)); }
__device__ __forceinline__ float rm_sign(float x) { return __builtin_copysignf(1.0f, x); }

struct RayCtx {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz, bound, rbound, dt_min, dt_max, dt_gamma, thresh;
    uint32_t H;
    const float *grid;
};

__device__ __forceinline__ void rm_setup(RayCtx &c, const float *o, const float *d, const float *grid, float mean_density,
                                         float bound, uint32_t H)
{
    c.ox = o[0]; c.oy = o[1]; c.oz = o[2]; c.dx = d[0]; c.dy = d[1]; c.dz = d[2];
    c.rdx = 1 / c.dx; c.rdy = 1 / c.dy; c.rdz = 1 / c.dz;
    c.bound = bound; c.rbound = 1 / bound; c.H = H; c.grid = grid;
    c.dt_min = (2 * RM_SQRT3 / RM_MAX_STEPS) * bound;
    c.dt_max = 2 * bound / (float)(H - 1);
    c.dt_gamma = bound > 1 ? (1.f / 256.f) : 0.0f;
    c.thresh = __builtin_fminf(10.0f, mean_density);
}
__device__ __forceinline__ void rm_pos(const RayCtx &c, float t, float &x, float &y, float &z)
{
    x = rm_clamp(c.ox + t * c.dx, -c.bound, c.bound);
    y = rm_clamp(c.oy + t * c.dy, -c.bound, c.bound);
    z = rm_clamp(c.oz + t * c.dz, -c.bound, c.bound);
}
__device__ __forceinline__ void rm_voxel(const RayCtx &c, float x, float y, float z, int &nx, int &ny, int &nz)
{
    // The reference's expression is (float)(0.5 * (double)(x * rbound + 1) * (double)H) (raymarching.cu: `0.5 * (x * rbound + 1) * H` with a double literal).
    // v = x * rbound + 1 is a float in {0} u [2^-24, 2]: 0.5 v is exact in either precision, and its product with H < 2^24 is exact in double (24 + 24 < 53
    // bits), so the reference rounds the exact product to float ONCE -- which is what the fp32 product (0.5f * v) * (float)H does.  Same bits, no fp64
    // conversions (three v_cvt_f64_f32, six v_mul_f64, three v_cvt_f32_f64 per position on a path that is bound by its instruction count).
    const float hm1 = (float)(c.H - 1), hf = (float)c.H;
    nx = (int)rm_clamp((0.5f * (x * c.rbound + 1)) * hf, 0.0f, hm1);
    ny = (int)rm_clamp((0.5f * (y * c.rbound + 1)) * hf, 0.0f, hm1);
    nz = (int)rm_clamp((0.5f * (z * c.rbound + 1)) * hf, 0.0f, hm1);
}
__device__ __forceinline__ float rm_density(const RayCtx &c, float t, float &x, float &y, float &z, int &nx, int &ny, int &nz)
{
    rm_pos(c, t, x, y, z);
    rm_voxel(c, x, y, z, nx, ny, nz);
    return c.grid[(uint32_t)nx * c.H * c.H + (uint32_t)ny * c.H + (uint32_t)nz];
}
// where the walk leaves the voxel (nx, ny, nz) it looked up at t: the marcher steps on until t >= this
__device__ __forceinline__ float rm_skip_target(const RayCtx &c, float t, float x, float y, float z, int nx, int ny, int nz)
{
    const float hm1 = (float)(c.H - 1);
    const float tx = (((nx + 0.5f + 0.5f * rm_sign(c.dx)) / hm1 * 2 - 1) * c.bound - x) * c.rdx;
    const float ty = (((ny + 0.5f + 0.5f * rm_sign(c.dy)) / hm1 * 2 - 1) * c.bound - y) * c.rdy;
    const float tz = (((nz + 0.5f + 0.5f * rm_sign(c.dz)) / hm1 * 2 - 1) * c.bound - z) * c.rdz;
    return t + __builtin_fmaxf(0.0f, __builtin_fminf(tx, __builtin_fminf(ty, tz)));
}
// The same target from a table of the voxel faces: nx + 0.5f + 0.5f * sign is the integer nx or nx + 1 (exactly), so the face coordinate
// ((nx + 0.5 + 0.5 sign) / (H - 1) * 2 - 1) * bound takes H + 1 values per launch -- edge[m], m = 0 .. H, formed by rm_edge with the expression above, the same
// bits -- and the three IEEE divisions per visited empty position (~30 of the walk's ~160 vector instructions per position) become three LDS reads.
__device__ __forceinline__ float rm_edge(const RayCtx &c, uint32_t m)
{
    const float hm1 = (float)(c.H - 1);
    return (((float)m) / hm1 * 2 - 1) * c.bound;
}
__device__ __forceinline__ float rm_skip_target_tab(const RayCtx &c, const float *edge, float t, float x, float y, float z, int nx, int ny, int nz)
{
    const float tx = (edge[nx + (__builtin_signbitf(c.dx) ? 0 : 1)] - x) * c.rdx;
    const float ty = (edge[ny + (__builtin_signbitf(c.dy) ? 0 : 1)] - y) * c.rdy;
    const float tz = (edge[nz + (__builtin_signbitf(c.dz) ? 0 : 1)] - z) * c.rdz;
    return t + __builtin_fmaxf(0.0f, __builtin_fminf(tx, __builtin_fminf(ty, tz)));
}
__device__ __forceinline__ float rm_skip(const RayCtx &c, float t, float x, float y, float z, int nx, int ny, int nz)
{
    const float tt = rm_skip_target(c, t, x, y, z, nx, ny, nz);
    do { t += rm_clamp(t * c.dt_gamma, c.dt_min, c.dt_max); } while (t < tt);
    return t;
}

// ---- the same walk with its grid look-ups issued B at a time (round 5) ------------------------------------------------------------------------------
// The reference's marcher (raymarching.cu:56-222, 497-599) is a chain of DEPENDENT loads: look up the voxel at t; occupied -> a sample, one step on; empty ->
// step on until the voxel is left; look up again.  A ray crosses ~200 voxels of the 128^3 grid: ~200 round trips to L2, 0.15 - 0.25 ms per ray whatever the
// number of rays in flight -- the whole time of a 4096-ray launch.  But every position the walk can ever stand on comes from ONE recurrence,
// t' = t + clamp(t * dt_gamma, dt_min, dt_max), whether the step is a sample's or a skip's: the positions are known before any voxel is.  So a lane looks
// up the next B positions of the recurrence at once (B independent loads: one round trip), and then replays the reference's decisions over them in registers:
// a position below the pending skip target was never visited (its look-up is wasted, not wrong); a visited one is tested against far and the caller's
// step budget exactly where the reference's loop tests them, yields a sample if its voxel is occupied and a new skip target (the reference's expression on
// the reference's operands, rm_skip_target) if not.  The state between two calls is (t, skip_tt).  Same samples, same bits, ~B / 2.5 times fewer round trips in
// empty space (a voxel is 2 - 5 steps wide) and B times fewer inside the body.
//   t        in: a position of the recurrence the walk has not decided yet; out: the next one
//   skip_tt  pending skip target (-inf: none); a NaN target (0 * inf in a direction component) compares false like in the reference's do-while: one step
//   room     samples the caller still takes (the reference's `step < n_step`): at 0 the walk stops AT the next visited position without consuming it
//   emit(x, y, z, dt, t_after, k)   one sample, k = its position's index in the recurrence
// Returns false when the walk has ended (a visited position >= far, or NaN).
//   kpos     index of position t in the ray's recurrence (0 = the walk's first position); emit's last argument is the sample's index
//   edge     optional table of the voxel faces (rm_edge: H + 1 floats, LDS); nullptr: the skip target's divisions are computed
//   KEEP     the look-up's point and voxel stay in registers (6 B of them) for the decisions instead of being formed again: a quarter fewer instructions where
//            registers are free (the counting kernels); the fused inference kernel, at its register limit, recomputes
template <int B, bool KEEP = false, class Emit>
__device__ __forceinline__ bool rm_march_batch(const RayCtx &c, float &t, float &skip_tt, float far, uint32_t &room, uint32_t &kpos, Emit &&emit,
                                               const float *edge = nullptr)
{
    float ts[B + 1], den[B];
    float kx[KEEP ? B : 1], ky[KEEP ? B : 1], kz[KEEP ? B : 1];
    int kn[KEEP ? 3 * B : 1];
    ts[0] = t;
#pragma unroll
    for (int j = 0; j < B; ++j) {
        float x, y, z; int nx, ny, nz;
        den[j] = rm_density(c, ts[j], x, y, z, nx, ny, nz);       // (positions are clamped into the volume: any t addresses a voxel)
        if (KEEP) { kx[j] = x; ky[j] = y; kz[j] = z; kn[3 * j] = nx; kn[3 * j + 1] = ny; kn[3 * j + 2] = nz; }
        ts[j + 1] = ts[j] + rm_clamp(ts[j] * c.dt_gamma, c.dt_min, c.dt_max);
    }
    bool open = true, more = true;
#pragma unroll
    for (int j = 0; j < B; ++j) {
        const float tj = ts[j];
        if (open && !(tj < skip_tt)) {                             // a position the reference's loop stands on
            if (!(tj < far)) { open = false; more = false; t = tj; kpos += (uint32_t)j; }
            else if (room == 0) { open = false; t = tj; kpos += (uint32_t)j; }
            else {
                float x, y, z;
                if (KEEP) { x = kx[j]; y = ky[j]; z = kz[j]; } else rm_pos(c, tj, x, y, z);
                if (den[j] > c.thresh) {
                    emit(x, y, z, rm_clamp(tj * c.dt_gamma, c.dt_min, c.dt_max), ts[j + 1], kpos + (uint32_t)j);
                    --room;
                } else {
                    int nx, ny, nz;
                    if (KEEP) { nx = kn[3 * j]; ny = kn[3 * j + 1]; nz = kn[3 * j + 2]; } else rm_voxel(c, x, y, z, nx, ny, nz);
                    skip_tt = edge ? rm_skip_target_tab(c, edge, tj, x, y, z, nx, ny, nz) : rm_skip_target(c, tj, x, y, z, nx, ny, nz);
                }
            }
        }
    }
    if (open) { t = ts[B]; kpos += (uint32_t)B; }
    return more;
}
#ifndef AC_RM_BATCH
#define AC_RM_BATCH 8
#endif
constexpr int RM_BATCH = AC_RM_BATCH;
constexpr float RM_NO_SKIP = -__builtin_inff();
__device__ __forceinline__ void rm_near_far(const RayCtx &c, float &near, float &far)
{
    float nx = (-c.bound - c.ox) * c.rdx, fx = (c.bound - c.ox) * c.rdx;
    if (nx > fx) { float s = nx; nx = fx; fx = s; }
    float ny = (-c.bound - c.oy) * c.rdy, fy = (c.bound - c.oy) * c.rdy;
    if (ny > fy) { float s = ny; ny = fy; fy = s; }
    float nz = (-c.bound - c.oz) * c.rdz, fz = (c.bound - c.oz) * c.rdz;
    if (nz > fz) { float s = nz; nz = fz; fz = s; }
    near = __builtin_fmaxf(__builtin_fmaxf(nx, __builtin_fmaxf(ny, nz)), RM_MIN_NEAR);
    far = __builtin_fminf(fx, __builtin_fminf(fy, fz));
}

// pcg32(initstate, initseq).next_float()   (raymarching/src/pcg32.h:57-72,107-116)
__device__ __forceinline__ uint32_t pcg_next(uint64_t &state, uint64_t inc)
{
    const uint64_t old = state;
    state = old * 0x5851f42d4c957f2dULL + inc;
    const uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u), rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((~rot + 1u) & 31));
}
__device__ __forceinline__ float pcg_first_float(uint64_t initstate, uint64_t initseq)
{
    uint64_t state = 0u; const uint64_t inc = (initseq << 1u) | 1u;
    pcg_next(state, inc); state += initstate; pcg_next(state, inc);
    return __uint_as_float((pcg_next(state, inc) >> 9) | 0x3f800000u) - 1.0f;
}

// ---- a ray's samples as a bit mask over the positions of its recurrence (round 5) ---------------------------------------------------------------------
// The training marcher walks every ray twice: once to count (the packed layout needs every ray's offset first), once to write.  The walk is the expensive
// part (see above); the samples are a handful of positions out of ~400.  So the counting pass RECORDS them -- bit k of RM_REC_WORDS words per ray (+ a mask of
// the words in use, so that neither pass touches the empty ones) = position k
// of t' = t + clamp(t dt_gamma, dt_min, dt_max) is a sample; far - near <= the cube's diagonal = 1024 dt_min bounds k -- and the writing pass replays the
// recurrence (four instructions per position, no voxel, no look-up) and emits at the set bits: the same positions by construction, the same bits.
// A sample at k >= 1024 (never seen: it needs a ray along the exact diagonal) raises the ray's overflow flag and the writer walks that ray again.
constexpr int RM_REC_WORDS = 32;
constexpr int RM_EDGE_MAX = 1026;            // the face table serves grids up to H = 1024
struct RayRecorder {                       // one lane = one ray.  Only the words that hold a sample are written (and later read): wmask names them
    uint32_t *rec; uint32_t w, bits, wmask; bool ovf;
    __device__ __forceinline__ void begin(uint32_t *r) { rec = r; w = 0; bits = 0; wmask = 0; ovf = false; }
    __device__ __forceinline__ void add(uint32_t k)
    {
        const uint32_t kw = k >> 5;
        if (kw >= (uint32_t)RM_REC_WORDS) { ovf = true; return; }
        if (kw != w) { if (bits) rec[w] = bits; w = kw; bits = 0u; }
        bits |= 1u << (k & 31u);
        wmask |= 1u << kw;
    }
    __device__ __forceinline__ void end() { if (bits) rec[w] = bits; }
};
// emit(x, y, z, dt) for the first `count` recorded samples of the ray whose walk starts at t0
template <class Emit>
__device__ __forceinline__ void rm_replay(const RayCtx &c, float t0, const uint32_t *rec, uint32_t wmask, uint32_t count, Emit &&emit)
{
    float t = t0;
    uint32_t k = 0, done = 0;
    while (wmask && done < count) {
        const uint32_t w = (uint32_t)__builtin_ctz(wmask);
        wmask &= wmask - 1u;
        uint32_t bits = rec[w];
        while (bits && done < count) {
            const uint32_t kk = (w << 5) + (uint32_t)__builtin_ctz(bits);
            bits &= bits - 1u;
            for (; k < kk; ++k) t += rm_clamp(t * c.dt_gamma, c.dt_min, c.dt_max);
            float x, y, z;
            rm_pos(c, t, x, y, z);
            emit(x, y, z, rm_clamp(t * c.dt_gamma, c.dt_min, c.dt_max));
            ++done;
        }
    }
}

// first position of the training walk (kernel_march_rays_train: t0 = near + dt_min * rng.next_float(), the generator seeded per ray)
__device__ __forceinline__ float ray_t0(const RayCtx &c, float near, uint32_t n, uint32_t perturb)
{
    float t0 = near;
    if (perturb) t0 += c.dt_min * pcg_first_float((uint64_t)n, 1);
    return t0;
}

}  // namespace
