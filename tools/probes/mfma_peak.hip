// Practical MFMA ceiling of this chip: back-to-back v_mfma_f32_32x32x16_bf16 with NACC independent
// accumulators per wave, WPS waves per SIMD, nothing else in the loop.  Prints TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/probes/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(blockIdx.x + e); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < NACC; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) out[0] = s;
}

template <int NACC>
void run(int wps, int iters) {
    float* out; hipMalloc(&out, 4);
    const int threads = 256 * wps, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, threads>>>(out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, threads>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * (threads / 64) * iters * NACC * 2.0 * 32 * 32 * 16;
    printf("nacc=%d waves/SIMD=%d  %.3f ms  %.1f TFLOP/s\n", NACC, wps, ms, flop / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run<1>(1, 20000); run<2>(1, 10000); run<4>(1, 5000);
        run<1>(2, 20000); run<2>(2, 10000); run<4>(2, 5000);
        run<4>(4, 5000);
    }
    // sustained (hundreds of ms): what the power/clock management leaves of the short-burst rate
    for (int rep = 0; rep < 3; ++rep) run<4>(2, 400000);
    return 0;
}
