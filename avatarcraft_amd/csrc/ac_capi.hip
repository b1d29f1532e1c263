// avatarcraft_amd/csrc/ac_capi.hip -- library identification and error reporting of the C ABI.
#include "ac_common.hpp"
#include "ac_devmath.hpp"

namespace ac {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}
}  // namespace ac

AC_API int ac_version(void) { return 10; }
AC_API const char *ac_last_error(void) { return ac::g_err; }

AC_API void ac_hash_level_table(uint32_t L, float S, uint32_t H, float *scale_host, uint32_t *res_host)
{
    for (uint32_t l = 0; l < L; ++l) {
        float sc = ac::exp2_f32((float)l * S) * (float)H - 1.0f;
        scale_host[l] = sc;
        res_host[l] = (uint32_t)ceilf(sc) + 1u;
    }
}

// test utility (include/avatarcraft_hip.h): a foreign workload that HOLDS compute units -- every workgroup spins on the 100 MHz wall clock
__global__ __launch_bounds__(1024) void hold_cus_kernel(unsigned long long ticks, uint32_t *sink)
{
    extern __shared__ uint32_t hold_lds[];
    if (threadIdx.x == 0) hold_lds[0] = blockIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(100);
    if (sink && hold_lds[0] == 0xffffffffu) *sink = 1u;       // (keeps the LDS allocation alive)
}
AC_API int ac_debug_hold_cus(uint32_t blocks, uint32_t lds_bytes, uint32_t millis, ac_stream_t stream)
{
    if (blocks == 0 || lds_bytes > 160 * 1024 - 64) { ac::set_error("ac_debug_hold_cus: blocks == 0 or more LDS than a compute unit has"); return AC_ERR_BAD_ARG; }
    static uint64_t seen = 0;
    ac::allow_dynamic_lds(seen, reinterpret_cast<const void *>(hold_cus_kernel), 160 * 1024 - 64);
    hipLaunchKernelGGL(hold_cus_kernel, dim3(blocks), dim3(1024), lds_bytes < 4 ? 4 : lds_bytes, (hipStream_t)stream, (unsigned long long)millis * 100000ull, (uint32_t *)nullptr);
    return ac::check_launch("ac_debug_hold_cus");
}

// test utility (include/avatarcraft_hip.h): unit_div against the IEEE division ON THE DEVICE, over every fp32 dividend whose bit pattern lies in [lo_bits, hi_bits]
__global__ __launch_bounds__(256) void unit_div_check_kernel(uint32_t lo_bits, uint32_t hi_bits, float d, float inv, unsigned long long *mismatches)
{
    const unsigned long long n = (unsigned long long)hi_bits - lo_bits + 1ull;
    unsigned long long bad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float a = __uint_as_float(lo_bits + (uint32_t)i);
        const float q = a / d, f = acdev::unit_div(a, d, inv);
        if (__float_as_uint(q) != __float_as_uint(f) && !(q != q && f != f)) ++bad;
    }
    if (bad) atomicAdd(mismatches, bad);
}
AC_API int ac_debug_unit_div_check(uint32_t lo_bits, uint32_t hi_bits, float d, unsigned long long *mismatches, ac_stream_t stream)
{
    if (!mismatches || hi_bits < lo_bits || !(d > 0.0f)) { ac::set_error("ac_debug_unit_div_check: NULL counter, empty range or d <= 0"); return AC_ERR_BAD_ARG; }
    volatile float one = 1.0f;
    hipLaunchKernelGGL(unit_div_check_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, lo_bits, hi_bits, d, one / d, mismatches);
    return ac::check_launch("ac_debug_unit_div_check");
}
