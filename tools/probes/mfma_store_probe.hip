// What a plane store costs the MFMA stream of a one-wave-per-SIMD kernel: v_mfma_f32_32x32x16_f16 back to back, 3 VALU
// fillers per MFMA, and ONE store of 1 KB per wave every EVERY MFMAs in different instruction forms.  Each wave writes its own
// contiguous region (like the saved planes); prints shader cycles per MFMA and the cycles one store adds.
//   hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mfma_store tools/probes/mfma_store_probe.hip && /tmp/mfma_store
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

// FORM 0: none; 1: global_store_dwordx4 nt (vaddr); 2: global_store_dwordx4 plain; 3: two global_store_dwordx2 nt;
// 4: global_store_dwordx4 nt, SGPR base + 32-bit lane offset (saddr form); 5: four global_store_dword nt
template <int FORM, int EVERY>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, unsigned char* dst, int iters) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(float)((threadIdx.x * 7 + e) % 13 - 6); b[e] = (_Float16)(float)((blockIdx.x + e) % 5 - 2); }
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = (float)threadIdx.x * 0.001f + i;
    u4 data = {(unsigned)threadIdx.x, 2u, 3u, 4u};
    unsigned char* wbase = dst + ((size_t)blockIdx.x * 4 + wave) * ((size_t)iters * (12 / EVERY) * 1024);
    size_t off = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 3; ++q) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[q % 8]) : "v"(f[(q + 3) % 8]));
            if (FORM != 0 && u % EVERY == EVERY - 1) {
                unsigned char* p = wbase + off + lane * 16;
                if (FORM == 1) __builtin_nontemporal_store(data, reinterpret_cast<u4*>(p));
                if (FORM == 2) *reinterpret_cast<volatile u4*>(p) = data;
                if (FORM == 3) {
                    __builtin_nontemporal_store(u2{data[0], data[1]}, reinterpret_cast<u2*>(wbase + off + lane * 8));
                    __builtin_nontemporal_store(u2{data[2], data[3]}, reinterpret_cast<u2*>(wbase + off + 512 + lane * 8));
                }
                if (FORM == 4) {
                    const unsigned char* sb = wbase + off;
                    asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(lane * 16), "v"(data), "s"(sb) : "memory");
                }
                if (FORM == 5) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) __builtin_nontemporal_store(data[c], reinterpret_cast<unsigned*>(wbase + off + c * 256 + lane * 4));
                }
                off += 1024;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int i = 0; i < 8; ++i) s += f[i];
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

static double base_cycles = 0;
template <int FORM, int EVERY>
void run(int iters, unsigned char* dst, const char* what) {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 8);
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        k<FORM, EVERY><<<256, 256>>>(out, cyc, dst, iters);
        (void)hipDeviceSynchronize();
        unsigned long long c = 0;
        (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double v = (double)c / ((double)iters * 12);
        if (rep > 0 && v < best) best = v;
    }
    if (FORM == 0) base_cycles = best;
    printf("%-62s every %2d MFMAs: %.2f cycles per MFMA", what, EVERY, best);
    if (FORM != 0) printf("  (+%.1f cycles per KiB stored; %.1f B/clk per SIMD)", (best - base_cycles) * EVERY, 1024.0 / ((best - base_cycles) * EVERY));
    printf("\n");
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    const int it = 2000;
    unsigned char* dst; (void)hipMalloc(&dst, (size_t)256 * 4 * it * 12 * 1024);
    run<0, 6>(it, dst, "no stores");
    run<1, 6>(it, dst, "global_store_dwordx4 nt (64-bit lane address)");
    run<2, 6>(it, dst, "global_store_dwordx4, default policy");
    run<4, 6>(it, dst, "global_store_dwordx4 nt, SGPR base + lane offset");
    run<3, 6>(it, dst, "2 x global_store_dwordx2 nt");
    run<5, 6>(it, dst, "4 x global_store_dword nt");
    run<1, 3>(it, dst, "global_store_dwordx4 nt (64-bit lane address)");
    run<4, 3>(it, dst, "global_store_dwordx4 nt, SGPR base + lane offset");
    run<1, 12>(it, dst, "global_store_dwordx4 nt (64-bit lane address)");
    return 0;
}
