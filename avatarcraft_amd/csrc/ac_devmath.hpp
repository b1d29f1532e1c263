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

// exp(-a), a >= 0, result in (0,1]; 0 above a = 82 (same algorithm as dv_exp, single power-of-two scale)
__device__ __forceinline__ float dv_exp_neg(float a)
{
    const float magic = 12582912.0f;
    float t = fma_(a, -1.44269504f, magic);
    float n = t - magic;
    float r = fma_(n, -0.693359375f, -a);
    r = fma_(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fma_(p, r, 1.3981999507e-3f);
    p = fma_(p, r, 8.3334519073e-3f);
    p = fma_(p, r, 4.1665795894e-2f);
    p = fma_(p, r, 1.6666665459e-1f);
    p = fma_(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float e = fma_(p, r2, r) + 1.0f;
    float u = e * bits2f((uint32_t)((int)n + 127) << 23);
    return (a <= 82.0f) ? u : ((a != a) ? a : 0.0f);
}

// a / 100 with three fp32 ops; equal to the correctly rounded IEEE quotient for every a in {0} U [2^-120, 2^10)
// (exhaustive proof: tools/verify_div100.c).  Every softplus argument lies in that set.
__device__ __forceinline__ float dv_div100(float a)
{
    const float y = 0x1.47ae14p-7f;                 // RN(1/100)
    float q0 = a * y;
    float r = fma_(-q0, 100.0f, a);
    return fma_(r, y, q0);
}

// torch.nn.Softplus(beta=100, threshold=20) (reference models/instant_nsr.py:231,591) in the overflow-free form
//   log1p(exp(t)) = max(t,0) + u*Q(u), u = exp(-|t|); Q = 8 x degree-5 polynomial table (ac_sp_table.hpp).
// spq: the table, [8][8] floats (LDS or global).  Bit-identical to oracle/ac_math.h: orc_softplus100.
__device__ __forceinline__ float dv_softplus100(const float *__restrict__ spq, float x)
{
    const float t = x * 100.0f;
    const float u = dv_exp_neg(__builtin_fabsf(t));
    int idx = (int)(u * 8.0f);
    idx = idx > 7 ? 7 : idx;
    const float v = u - ((float)idx + 0.5f) * 0.125f;
    const float4 c03 = *reinterpret_cast<const float4 *>(spq + idx * 8);
    const float2 c45 = *reinterpret_cast<const float2 *>(spq + idx * 8 + 4);
    float q = c45.y;
    q = fma_(q, v, c45.x); q = fma_(q, v, c03.w); q = fma_(q, v, c03.z); q = fma_(q, v, c03.y); q = fma_(q, v, c03.x);
    const float s = (t > 0.0f ? t : 0.0f) + u * q;
    const float res = dv_div100(s);
    return (t > 20.0f) ? x : ((t != t) ? t : res);
}

// the same softplus on two independent values with packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 are
// IEEE-exact per half, so each half is bit-identical to dv_softplus100)
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat2(float a) { v2f r = { a, a }; return r; }

__device__ __forceinline__ v2f dv_softplus100_x2(const float *__restrict__ spq, v2f x)
{
    const v2f t = x * splat2(100.0f);
    v2f a; a.x = __builtin_fabsf(t.x); a.y = __builtin_fabsf(t.y);
    const v2f magic = splat2(12582912.0f);
    const v2f tt = pk_fma(a, splat2(-1.44269504f), magic);
    const v2f n = tt - magic;
    v2f r = pk_fma(n, splat2(-0.693359375f), -a);
    r = pk_fma(n, splat2(2.12194440e-4f), r);
    v2f p = splat2(1.9875691500e-4f);
    p = pk_fma(p, r, splat2(1.3981999507e-3f));
    p = pk_fma(p, r, splat2(8.3334519073e-3f));
    p = pk_fma(p, r, splat2(4.1665795894e-2f));
    p = pk_fma(p, r, splat2(1.6666665459e-1f));
    p = pk_fma(p, r, splat2(5.0000001201e-1f));
    const v2f r2 = r * r;
    const v2f e = pk_fma(p, r2, r) + splat2(1.0f);
    v2f sc; sc.x = bits2f((uint32_t)((int)n.x + 127) << 23); sc.y = bits2f((uint32_t)((int)n.y + 127) << 23);
    v2f u = e * sc;
    u.x = (a.x <= 82.0f) ? u.x : ((a.x != a.x) ? a.x : 0.0f);
    u.y = (a.y <= 82.0f) ? u.y : ((a.y != a.y) ? a.y : 0.0f);
    const v2f u8 = u * splat2(8.0f);
    int i0 = (int)u8.x, i1 = (int)u8.y;
    i0 = i0 > 7 ? 7 : i0; i1 = i1 > 7 ? 7 : i1;
    v2f ctr; ctr.x = ((float)i0 + 0.5f) * 0.125f; ctr.y = ((float)i1 + 0.5f) * 0.125f;
    const v2f v = u - ctr;
    const float4 a03 = *reinterpret_cast<const float4 *>(spq + i0 * 8), b03 = *reinterpret_cast<const float4 *>(spq + i1 * 8);
    const float2 a45 = *reinterpret_cast<const float2 *>(spq + i0 * 8 + 4), b45 = *reinterpret_cast<const float2 *>(spq + i1 * 8 + 4);
    v2f q = { a45.y, b45.y };
    { v2f c = { a45.x, b45.x }; q = pk_fma(q, v, c); }
    { v2f c = { a03.w, b03.w }; q = pk_fma(q, v, c); }
    { v2f c = { a03.z, b03.z }; q = pk_fma(q, v, c); }
    { v2f c = { a03.y, b03.y }; q = pk_fma(q, v, c); }
    { v2f c = { a03.x, b03.x }; q = pk_fma(q, v, c); }
    v2f tp; tp.x = t.x > 0.0f ? t.x : 0.0f; tp.y = t.y > 0.0f ? t.y : 0.0f;
    const v2f sgm = tp + u * q;
    const v2f y = splat2(0x1.47ae14p-7f);
    const v2f q0 = sgm * y;
    const v2f rr = pk_fma(-q0, splat2(100.0f), sgm);
    const v2f res = pk_fma(rr, y, q0);
    v2f o;
    o.x = (t.x > 20.0f) ? x.x : ((t.x != t.x) ? t.x : res.x);
    o.y = (t.y > 20.0f) ? x.y : ((t.y != t.y) ? t.y : res.y);
    return o;
}

// torch.sigmoid
__device__ __forceinline__ float dv_sigmoid(float x) { return 1.0f / (1.0f + dv_exp(-x)); }

}  // namespace acdev
