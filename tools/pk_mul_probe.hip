// tools/pk_mul_probe.hip -- are v_pk_mul_f32 results with crossed op_sel right in every lane while buffer loads are in flight?
// (round 3: the face-value stencil variant was non-deterministic exactly when the SLP vectorizer formed these instructions for the
//  xy weight products, and only in lanes 48..63.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/_bin/pk_mul_probe tools/pk_mul_probe.hip && tools/_bin/pk_mul_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <random>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(const uint32_t *big, uint32_t big_bytes, const uint32_t *offs, const float *xy, int mode, uint32_t *bad_by_lane, uint32_t *sink)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    const uint64_t p = reinterpret_cast<uint64_t>(big);
    u32x4 rsrc = { (uint32_t)p, (uint32_t)(p >> 32) & 0xffffu, big_bytes, 0x00020000u };
    for (int k = 0; k < 4; ++k) rsrc[k] = __builtin_amdgcn_readfirstlane(rsrc[k]);
    uint32_t acc = 0;
    for (int it = 0; it < 64; ++it) {
        const float qx = xy[(tid * 64 + it) * 2 % (1 << 22)], qy = xy[((tid * 64 + it) * 2 + 1) % (1 << 22)];
        f32x2 q = { qx, qy }, w = { 1.0f - qx, 1.0f - qy };
        uint32_t o0 = offs[(tid + it * 977u) % (1u << 20)], o1 = offs[(tid + it * 977u + 131u) % (1u << 20)];
        uint32_t l0 = 0, l1 = 0, l2 = 0, l3 = 0;
        f32x2 w00, w10, w01, w11;
        if (mode == 0) {          // products only
            asm volatile("v_pk_mul_f32 %0, %4, %4 op_sel:[1,0] op_sel_hi:[0,1]\n\t"
                         "v_pk_mul_f32 %1, %5, %4 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                         "v_pk_mul_f32 %2, %5, %4 op_sel:[1,0] op_sel_hi:[0,1]\n\t"
                         "v_pk_mul_f32 %3, %5, %5 op_sel:[1,0] op_sel_hi:[0,1]"
                         : "=&v"(w00), "=&v"(w10), "=&v"(w01), "=&v"(w11) : "v"(w), "v"(q));
        } else {                  // the same with four gathers in flight around them (issued before, consumed after)
            asm volatile("buffer_load_dword %4, %10, %12, 0 offen\n\t"
                         "buffer_load_dword %5, %11, %12, 0 offen\n\t"
                         "buffer_load_dword %6, %10, %12, 0 offen offset:64\n\t"
                         "buffer_load_dword %7, %11, %12, 0 offen offset:64\n\t"
                         "v_pk_mul_f32 %0, %8, %8 op_sel:[1,0] op_sel_hi:[0,1]\n\t"
                         "v_pk_mul_f32 %1, %9, %8 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                         "v_pk_mul_f32 %2, %9, %8 op_sel:[1,0] op_sel_hi:[0,1]\n\t"
                         "v_pk_mul_f32 %3, %9, %9 op_sel:[1,0] op_sel_hi:[0,1]\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(w00), "=&v"(w10), "=&v"(w01), "=&v"(w11), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
                         : "v"(w), "v"(q), "v"(o0), "v"(o1), "s"(rsrc) : "memory");
        }
        float dep0 = 0.0f, dep1 = 0.0f, dep2 = 0.0f, dep3 = 0.0f;
        const float cz = 1.0f - qx * 0.5f;
        float czv = cz;
        if (mode >= 2) {          // the sequence of the failing build: every product is read by a plain VALU multiply 2 .. 3 instructions after the packed
                                  // multiply that wrote it (mode 3: the same with four gathers in flight)
            if (mode == 3)
                asm volatile("buffer_load_dword %0, %4, %6, 0 offen\n\tbuffer_load_dword %1, %5, %6, 0 offen\n\t"
                             "buffer_load_dword %2, %4, %6, 0 offen offset:64\n\tbuffer_load_dword %3, %5, %6, 0 offen offset:64"
                             : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3) : "v"(o0), "v"(o1), "s"(rsrc) : "memory");
            asm volatile("v_pk_mul_f32 v[100:101], %4, %4 op_sel:[1,0] op_sel_hi:[0,1]\n\t"
                         "v_pk_mul_f32 v[102:103], %5, %4 op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                         "v_mul_f32 %0, %6, v100\n\t"
                         "v_pk_mul_f32 v[104:105], %5, %4 op_sel:[1,0] op_sel_hi:[0,1]\n\t"
                         "v_mul_f32 %1, %6, v102\n\t"
                         "v_pk_mul_f32 v[106:107], %5, %5 op_sel:[1,0] op_sel_hi:[0,1]\n\t"
                         "v_mul_f32 %2, %6, v104\n\t"
                         "v_mul_f32 %3, %6, v106\n\t"
                         "v_mov_b32 %7, v100\n\tv_mov_b32 %8, v101\n\tv_mov_b32 %9, v102\n\tv_mov_b32 %10, v103\n\t"
                         "v_mov_b32 %11, v104\n\tv_mov_b32 %12, v105\n\tv_mov_b32 %13, v106\n\tv_mov_b32 %14, v107\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(dep0), "=&v"(dep1), "=&v"(dep2), "=&v"(dep3), "+v"(w), "+v"(q), "+v"(czv),
                           "=&v"(w00.x), "=&v"(w00.y), "=&v"(w10.x), "=&v"(w10.y), "=&v"(w01.x), "=&v"(w01.y), "=&v"(w11.x), "=&v"(w11.y)
                         :: "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "memory");
        }
        const float e00 = w.y * w.x, e10 = q.x * w.y, e01 = q.y * w.x, e11 = q.y * q.x;
        if (mode >= 2 && !(dep0 == cz * e00 && dep1 == cz * e10 && dep2 == cz * e01 && dep3 == cz * e11)) atomicAdd(bad_by_lane + 64 + lane, 1u);
        const bool ok = w00.x == e00 && w00.y == e00 && w10.x == e10 && w10.y == e01 && w01.x == e01 && w01.y == e10 && w11.x == e11 && w11.y == e11;
        if (!ok) atomicAdd(bad_by_lane + lane, 1u);
        acc += l0 ^ l1 ^ l2 ^ l3;
    }
    if (acc == 0x12345u) sink[0] = acc;
}

int main()
{
    const size_t bytes = size_t(1) << 30;
    uint32_t *big; (void)hipMalloc(&big, bytes); (void)hipMemset(big, 1, bytes);
    std::mt19937_64 rng(1);
    std::vector<uint32_t> offs(1 << 20);
    for (auto &o : offs) o = (uint32_t)((rng() % (bytes / 4 - 64)) * 4);
    std::vector<float> xy(1 << 22);
    for (auto &v : xy) v = (float)((rng() >> 11) * (1.0 / 9007199254740992.0));
    uint32_t *doffs, *dbad, *dsink; float *dxy;
    (void)hipMalloc(&doffs, offs.size() * 4); (void)hipMalloc(&dbad, 512); (void)hipMalloc(&dsink, 4); (void)hipMalloc(&dxy, xy.size() * 4);
    (void)hipMemcpy(doffs, offs.data(), offs.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dxy, xy.data(), xy.size() * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 4; ++mode) {
        (void)hipMemset(dbad, 0, 512);
        for (int rep = 0; rep < 4; ++rep) hipLaunchKernelGGL(probe, dim3(256 * 16), dim3(512), 0, 0, big, (uint32_t)bytes, doffs, dxy, mode, dbad, dsink);
        uint32_t bad[128]; (void)hipMemcpy(bad, dbad, 512, hipMemcpyDeviceToHost);
        uint32_t tot = 0, hi = 0, dtot = 0, dhi = 0;
        for (int l = 0; l < 64; ++l) { tot += bad[l]; dtot += bad[64 + l]; if (l >= 48) { hi += bad[l]; dhi += bad[64 + l]; } }
        const char *names[4] = { "products only", "products, gathers in flight", "products + dependent VALU reads at distance 2..3", "the same, gathers in flight" };
        printf("mode %d (%s): wrong product sets %u (lanes 48..63: %u), wrong dependent results %u (lanes 48..63: %u) of %u\n", mode, names[mode], tot, hi, dtot, dhi,
               4u * 256 * 16 * 512 * 64);
    }
    return 0;
}
