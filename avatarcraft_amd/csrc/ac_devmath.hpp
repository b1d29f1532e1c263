// avatarcraft_amd/csrc/ac_devmath.hpp -- device-side fp32 helpers for the gfx950 kernels.
//
// The hot path promises results that are bit-identical to a scalar CPU evaluation of the same
// arithmetic (DESIGN.md "Numerics contract"): every operation below is an IEEE-754 correctly
// rounded +,-,*,/ or an explicit fma, plus integer bit manipulation.  No ocml transcendental is
// used on the parity-critical path because their rounding is not specified.
//   dv_exp   : Cody-Waite reduction by ln2 (hi/lo split) + degree-6 polynomial (Cephes expf
//              coefficients), scaled by two exact powers of two.
//   dv_log1p : fdlibm log1pf scheme (FreeBSD msun s_log1pf.c constants; "Copyright (C) 1993 by
//              Sun Microsystems, Inc. ... Permission to use, copy, modify, and distribute this
//              software is freely granted, provided that this notice is preserved.").
// Build flags required: -ffp-contract=off (no implicit contraction), default correctly rounded
// fp32 divide/sqrt (do NOT pass -ffast-math / -fno-hip-fp32-correctly-rounded-divide-sqrt).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace acdev {

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float bits2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t f2bits(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ float dv_exp(float x)
{
    if (x > 88.72283f) return __builtin_inff();
    if (!(x >= -87.33654f)) return (x != x) ? x : 0.0f;
    const float magic = 12582912.0f;                    // 1.5 * 2^23: round-to-nearest-integer trick
    float t = fma_(x, 1.44269504f, magic);
    float n = t - magic;
    float r = fma_(n, -0.693359375f, x);
    r = fma_(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fma_(p, r, 1.3981999507e-3f);
    p = fma_(p, r, 8.3334519073e-3f);
    p = fma_(p, r, 4.1665795894e-2f);
    p = fma_(p, r, 1.6666665459e-1f);
    p = fma_(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float e = fma_(p, r2, r) + 1.0f;
    int ni = (int)n;
    int n1 = ni >> 1;
    int n2 = ni - n1;
    float s1 = bits2f((uint32_t)(n1 + 127) << 23);
    float s2 = bits2f((uint32_t)(n2 + 127) << 23);
    return (e * s1) * s2;
}

__device__ __forceinline__ float dv_log1p(float x)
{
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    const float Lg1 = 0.66666662693f, Lg2 = 0.40000972152f, Lg3 = 0.28498786688f, Lg4 = 0.24279078841f;
    if (!(x > -1.0f)) return (x == -1.0f) ? -__builtin_inff() : __builtin_nanf("");
    if (__builtin_fabsf(x) < 5.9604645e-08f) return x;
    if (x == __builtin_inff()) return x;
    float u = 1.0f + x;
    uint32_t iu = f2bits(u);
    iu += 0x3f800000u - 0x3f3504f3u;
    int k = (int)(iu >> 23) - 127;
    float c;
    if (k < 25) {
        c = (k >= 2) ? 1.0f - (u - x) : x - (u - 1.0f);
        c = c / u;
    } else {
        c = 0.0f;
    }
    iu = (iu & 0x007fffffu) + 0x3f3504f3u;
    float f = bits2f(iu) - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float w = z * z;
    float t1 = w * fma_(w, Lg4, Lg2);
    float t2 = z * fma_(w, Lg3, Lg1);
    float R = t2 + t1;
    float hfsq = 0.5f * f * f;
    float dk = (float)k;
    return fma_(s, hfsq + R, fma_(dk, ln2_lo, c)) - hfsq + f + dk * ln2_hi;
}

// torch.nn.Softplus(beta=100, threshold=20) (reference models/instant_nsr.py:231,591):
//   softplus_100(x) = max(x, 0) + G(|100 x|),  G(a) = log1p(exp(-a)) / 100 from a 128-piece cubic table on [0, 32] indexed in a4 = |400 x|
//   (ac_sp_table.hpp; tools/gen_softplus_table.py; row 128 = 0) -- no exponential, no division: 10 VALU + one ds_read_b128
//   (v_mul | v_min |.| | v_cvt_u32 | v_fract | v_lshlrev | 3 v_fma | 2 v_fma adding max(x, 0) as 0.5 x + 0.5 |x|); 14 before round 3.
// spg: the table, [129][4] floats (LDS or global).  Bit-identical to oracle/ac_math.h: orc_softplus100.
__device__ __forceinline__ float dv_softplus100(const float *__restrict__ spg, float x)
{
    const float a4 = __builtin_fminf(__builtin_fabsf(x * 400.0f), 128.0f);
    const uint32_t idx = (uint32_t)a4;
    const float v = __builtin_amdgcn_fractf(a4);
    const float4 c = *reinterpret_cast<const float4 *>(spg + idx * 4);
    float q = c.w;
    q = fma_(q, v, c.z); q = fma_(q, v, c.y); q = fma_(q, v, c.x);
    return fma_(0.5f, __builtin_fabsf(x), fma_(0.5f, x, q));
}

// N values at once: all table rows are requested before any of them is used -- one LDS round trip per batch instead of one per value
// (written value by value, the compiler emits `ds_read_b128; s_waitcnt lgkmcnt(0)` sixteen times per MLP evaluation: ~100 clocks of exposed
// latency each).  Same arithmetic per value: bit-identical to dv_softplus100.
template <int N>
__device__ __forceinline__ void dv_softplus100_n(const float *__restrict__ spg, const float (&x)[N], float (&o)[N])
{
    float v[N];
    float4 c[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float a4 = __builtin_fminf(__builtin_fabsf(x[i] * 400.0f), 128.0f);
        const uint32_t idx = (uint32_t)a4;
        v[i] = __builtin_amdgcn_fractf(a4);
        c[i] = *reinterpret_cast<const float4 *>(spg + idx * 4);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float q = c[i].w;
        q = fma_(q, v[i], c[i].z); q = fma_(q, v[i], c[i].y); q = fma_(q, v[i], c[i].x);
        o[i] = fma_(0.5f, __builtin_fabsf(x[i]), fma_(0.5f, x[i], q));
    }
}

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f splat2(float a) { v2f r = { a, a }; return r; }

// two independent values (kept for the call sites that hold accumulator pairs; evaluated as two scalar chains: the table rows of
// the two halves differ, so packed fma would need register shuffles that cost more than they save)
__device__ __forceinline__ v2f dv_softplus100_x2(const float *__restrict__ spg, v2f x)
{
    v2f o; o.x = dv_softplus100(spg, x.x); o.y = dv_softplus100(spg, x.y);
    return o;
}

// u = a / two_bound, a = p + bound with p clamped to [-bound, bound] (so a in [0, 2 bound], or NaN).  The reference divides ((x + size) / (2 size),
// hashgrid.py:130) and so does the oracle; an IEEE fp32 division is ~10 vector instructions here, one of them (v_rcp_f32) quarter rate, and a tile of the final
// pass forms nine quotients per lane.  With inv = RN(1 / d) the sequence  q = a * inv;  r = fma(-q, d, a);  u = fma(r, inv, q)  (Markstein's correction) returns
// the correctly rounded quotient -- the IEEE division's bits -- for every fp32 a with 1e-30 <= |a| <= 1e30, for +0 and for NaN: verified EXHAUSTIVELY (all 2^32
// bit patterns) for the divisors fill_args accepts (tests/div_check.c, tests/test_div_check.py; the host side: ac::verified_reciprocal, ac_common.hpp).  inv_tb == 0 (any other bound, AC_EXACT_DIV=1): IEEE division.
__device__ __forceinline__ float unit_div(float a, float d, float inv) { const float q = a * inv; return fma_(fma_(-q, d, a), inv, q); }

// torch.sigmoid
__device__ __forceinline__ float dv_sigmoid(float x) { return 1.0f / (1.0f + dv_exp(-x)); }

}  // namespace acdev
