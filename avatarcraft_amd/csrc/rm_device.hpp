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
__device__ __forceinline__ float rm_density(const RayCtx &c, float t, float &x, float &y, float &z, int &nx, int &ny, int &nz)
{
    x = rm_clamp(c.ox + t * c.dx, -c.bound, c.bound);
    y = rm_clamp(c.oy + t * c.dy, -c.bound, c.bound);
    z = rm_clamp(c.oz + t * c.dz, -c.bound, c.bound);
    const float hm1 = (float)(c.H - 1);
    nx = (int)rm_clamp((float)(0.5 * (double)(x * c.rbound + 1) * (double)c.H), 0.0f, hm1);
    ny = (int)rm_clamp((float)(0.5 * (double)(y * c.rbound + 1) * (double)c.H), 0.0f, hm1);
    nz = (int)rm_clamp((float)(0.5 * (double)(z * c.rbound + 1) * (double)c.H), 0.0f, hm1);
    return c.grid[(uint32_t)nx * c.H * c.H + (uint32_t)ny * c.H + (uint32_t)nz];
}
__device__ __forceinline__ float rm_skip(const RayCtx &c, float t, float x, float y, float z, int nx, int ny, int nz)
{
    const float hm1 = (float)(c.H - 1);
    const float tx = (((nx + 0.5f + 0.5f * rm_sign(c.dx)) / hm1 * 2 - 1) * c.bound - x) * c.rdx;
    const float ty = (((ny + 0.5f + 0.5f * rm_sign(c.dy)) / hm1 * 2 - 1) * c.bound - y) * c.rdy;
    const float tz = (((nz + 0.5f + 0.5f * rm_sign(c.dz)) / hm1 * 2 - 1) * c.bound - z) * c.rdz;
    const float tt = t + __builtin_fmaxf(0.0f, __builtin_fminf(tx, __builtin_fminf(ty, tz)));
    do { t += rm_clamp(t * c.dt_gamma, c.dt_min, c.dt_max); } while (t < tt);
    return t;
}
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

}  // namespace
