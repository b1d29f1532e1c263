// tools/oob_order_probe.hip -- do buffer loads whose lanes are ALL out of range (dropped by the descriptor's bounds check) complete
// in order with the loads issued before them?  `s_waitcnt vmcnt(N)` -- and the compiler's placement of it -- assumes they do.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/_bin/oob_order_probe tools/oob_order_probe.hip && tools/_bin/oob_order_probe
//
// Per lane: load A = a cache-missing gather (random dword of a 1 GiB buffer), load B = a second load issued right behind it, then
// `s_waitcnt vmcnt(1)` -- "at most one load outstanding", i.e. A must have landed if loads complete in order -- and A's register is copied.
// A lane is STALE when the copy still holds the preset instead of the loaded dword.
//   mode 0: B in range (another random dword)          -> control, expect 0 stale lanes
//   mode 1: B out of range in every lane (offset -8)   -> stale lanes = out-of-order completion of B
//   mode 2: B out of range in the odd lanes only
//   mode 3: B in range but with EXEC = 0 for it ... not expressible; instead: B issued by no lane is skipped (s_cbranch_execz), not tested
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <random>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const uint32_t *big, uint32_t big_bytes, const uint32_t *offs, int mode, int nB, uint32_t *stale_count, uint32_t *total)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t p = reinterpret_cast<uint64_t>(big);
    u32x4 rsrc = { (uint32_t)p, (uint32_t)(p >> 32) & 0xffffu, big_bytes, 0x00020000u };
    rsrc[0] = __builtin_amdgcn_readfirstlane(rsrc[0]); rsrc[1] = __builtin_amdgcn_readfirstlane(rsrc[1]);
    rsrc[2] = __builtin_amdgcn_readfirstlane(rsrc[2]); rsrc[3] = __builtin_amdgcn_readfirstlane(rsrc[3]);
    const uint32_t offA = offs[2 * tid], offR = offs[2 * tid + 1];
    uint32_t offB = offR;
    if (mode == 1) offB = 0xfffffff8u;
    if (mode == 2 && (threadIdx.x & 1)) offB = 0xfffffff8u;
    uint32_t a = 0xdeadbeefu, early = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    if (nB == 1) {
        asm volatile("buffer_load_dword %0, %3, %5, 0 offen\n\t"
                     "buffer_load_dword %1, %4, %5, 0 offen\n\t"
                     "s_waitcnt vmcnt(1)\n\t"
                     "v_mov_b32 %2, %0\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "+v"(a), "=&v"(b0), "=&v"(early) : "v"(offA), "v"(offB), "s"(rsrc) : "memory");
    } else {       // four B loads behind A, wait for "at most four outstanding"
        asm volatile("buffer_load_dword %0, %6, %8, 0 offen\n\t"
                     "buffer_load_dword %1, %7, %8, 0 offen\n\t"
                     "buffer_load_dword %2, %7, %8, 0 offen offset:4\n\t"
                     "buffer_load_dword %3, %7, %8, 0 offen offset:8\n\t"
                     "buffer_load_dword %4, %7, %8, 0 offen offset:12\n\t"
                     "s_waitcnt vmcnt(4)\n\t"
                     "v_mov_b32 %5, %0\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "+v"(a), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(early) : "v"(offA), "v"(offB), "s"(rsrc) : "memory");
    }
    const bool stale = early != a;
    if (stale) atomicAdd(stale_count, 1u);
    if (a != big[offA / 4] ) atomicAdd(total + 1, 1u);          // sanity: the load itself is right
    if (threadIdx.x == 0) atomicAdd(total, 64u);
    if ((b0 | b1 | b2 | b3) == 0x12345678u) atomicAdd(total + 2, 1u);       // keep the B results alive
}

int main()
{
    const size_t bytes = size_t(1) << 30;
    uint32_t *big; hipMalloc(&big, bytes);
    std::vector<uint32_t> h(bytes / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u) | 1u;       // never 0xdeadbeef-like zero
    hipMemcpy(big, h.data(), bytes, hipMemcpyHostToDevice);
    const int waves = 256 * 8 * 4, threads = waves * 64;
    std::mt19937_64 rng(1);
    std::vector<uint32_t> offs(2 * threads);
    for (auto &o : offs) o = (uint32_t)((rng() % (bytes / 4 - 8)) * 4);
    uint32_t *doffs, *dcount, *dtotal;
    hipMalloc(&doffs, offs.size() * 4); hipMalloc(&dcount, 4); hipMalloc(&dtotal, 12);
    hipMemcpy(doffs, offs.data(), offs.size() * 4, hipMemcpyHostToDevice);
    const char *names[3] = { "B in range (control)", "B out of range in every lane", "B out of range in odd lanes" };
    for (int nB = 1; nB <= 4; nB += 3)
        for (int mode = 0; mode < 3; ++mode) {
            uint32_t stale_sum = 0, lanes = 0, bad = 0;
            for (int rep = 0; rep < 8; ++rep) {
                hipMemset(dcount, 0, 4); hipMemset(dtotal, 0, 12);
                hipLaunchKernelGGL(probe, dim3(threads / 256), dim3(256), 0, 0, big, (uint32_t)bytes, doffs, mode, nB, dcount, dtotal);
                uint32_t c, t[3]; hipMemcpy(&c, dcount, 4, hipMemcpyDeviceToHost); hipMemcpy(t, dtotal, 12, hipMemcpyDeviceToHost);
                stale_sum += c; lanes += t[0]; bad += t[1];
            }
            printf("%d B load(s) behind A, %-32s: %9u stale lanes of %u (%.4f %%), wrong final values %u\n", nB, names[mode], stale_sum, lanes,
                   100.0 * stale_sum / lanes, bad);
        }
    return 0;
}
