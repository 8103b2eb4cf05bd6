// Does a filler instruction between two MFMAs on the SAME accumulator cost more than between MFMAs on DIFFERENT
// accumulators?  (MI355X_MICROARCH.md: "+43 cyc for the first extra issue state between two MFMAs on the same
// accumulator ... ~6 cyc/state on different accumulators".)  One wave per SIMD, v_mfma_f32_32x32x16_f16, NFILL
// independent v_fma_f32 fillers after every MFMA, NACC accumulators used round-robin; prints shader cycles per MFMA.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_dep tools/probes/mfma_dep_probe.hip && /tmp/mfma_dep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int NFILL>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(float)((threadIdx.x * 7 + e) % 13 - 6); b[e] = (_Float16)(float)((blockIdx.x + e) % 5 - 2); }
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = (float)threadIdx.x * 0.001f + i;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {      // 12 MFMAs per trip: a multiple of 1, 2, 3, 4, 6
            acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u % NACC], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NFILL; ++q) {
                asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[q % 8]) : "v"(f[(q + 3) % 8]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int t = 0; t < NACC; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int i = 0; i < 8; ++i) s += f[i];
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC, int NFILL>
void run(int iters) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4); hipMalloc(&cyc, 8);
    k<NACC, NFILL><<<256, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    k<NACC, NFILL><<<256, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("accumulators %d  fillers/MFMA %d  %.2f cycles per MFMA\n", NACC, NFILL, (double)c / ((double)iters * 12));
    hipFree(out); hipFree(cyc);
}

int main() {
    const int it = 20000;
    run<1, 0>(it); run<2, 0>(it);
    run<1, 1>(it); run<2, 1>(it); run<3, 1>(it);
    run<1, 2>(it); run<2, 2>(it);
    run<1, 4>(it); run<2, 4>(it); run<3, 4>(it); run<4, 4>(it);
    run<1, 6>(it); run<2, 6>(it);
    return 0;
}
