/* tests/div_check.c -- exhaustive proof obligation of unit_div (avatarcraft_amd/csrc/ac_devmath.hpp): for a divisor d and inv = RN(1 / d),
 *     q = a * inv;  r = fma(-q, d, a);  u = fma(r, inv, q)
 * equals the IEEE-754 fp32 quotient a / d bit for bit.  Every one of the 2^32 dividends is tried; the program prints how many differ, the magnitude range of
 * those that do, and how many of them lie in 1e-30 <= |a| <= 1e30 (the renderer's dividends are p + bound with p clamped to [-bound, bound]: 0, NaN, or a
 * magnitude between 1e-7 and 4).  Build: gcc -O2 -ffp-contract=off -fopenmp tests/div_check.c -lm     (test infrastructure; tests/test_div_check.py runs it) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float fl(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char **argv)
{
    int rc = 0;
    for (int k = 1; k < argc; ++k) {
        const float d = strtof(argv[k], 0);
        volatile float one = 1.0f;
        const float y = one / d;
        unsigned long long bad = 0, bad_mid = 0, bad_zero = 0;
        float lo = INFINITY, hi = 0.0f;
#pragma omp parallel for schedule(static) reduction(+ : bad, bad_mid, bad_zero) reduction(min : lo) reduction(max : hi)
        for (long long i = 0; i < (1LL << 32); ++i) {
            const float a = fl((uint32_t)i);
            const float ie = a / d;
            const float q = a * y, r = fmaf(-q, d, a), f = fmaf(r, y, q);
            const int same = (bits(ie) == bits(f)) || (ie != ie && f != f);
            if (!same) {
                const float m = fabsf(a);
                ++bad;
                if (m < lo) lo = m;
                if (m > hi) hi = m;
                if (m >= 1e-30f && m <= 1e30f) ++bad_mid;
                if (bits(a) == 0u) ++bad_zero;                 /* +0 (p = -bound exactly) must be exact; -0 cannot arise from p + bound */
            }
        }
        printf("d=%.9g inv=%.9g differ=%llu magnitudes=[%g, %g] in_domain=%llu plus_zero=%llu\n", (double)d, (double)y, bad, (double)lo, (double)hi, bad_mid, bad_zero);
        if (bad_mid || bad_zero) rc = 1;
    }
    return rc;
}
