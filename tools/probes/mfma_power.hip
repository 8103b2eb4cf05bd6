// Sustained MFMA rate with operands that toggle like real data: 8 random bf16 fragments per operand are
// cycled through back-to-back v_mfma_f32_32x32x16_bf16.  Compare with mfma_peak.hip (constant operands):
// the difference is what the power/clock management takes at realistic switching activity.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_power tools/probes/mfma_power.hip && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <bool RANDOM>
__global__ void k(float* out, int iters, unsigned long long* clk) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    bf16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i)
        for (int e = 0; e < 8; ++e) {
            const unsigned h = hash(threadIdx.x * 977u + blockIdx.x * 131u + i * 17u + e);
            const float va = RANDOM ? ((int)(h & 0xffff) - 32768) * (1.0f / 32768.f) : 1.0f;
            const float vb = RANDOM ? ((int)(h >> 16) - 32768) * (1.0f / 32768.f) : 1.0f;
            a[i][e] = (__bf16)va; b[i][e] = (__bf16)vb;
        }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b[(j + t) & 7], acc[t], 0, 0, 0);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 777 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <bool RANDOM>
void run(int iters) {
    float* out; unsigned long long* clk; hipMalloc(&out, 4); hipMalloc(&clk, 16);
    const int threads = 512, blocks = 256 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<RANDOM><<<blocks, threads>>>(out, iters, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<RANDOM><<<blocks, threads>>>(out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flop = (double)blocks * (threads / 64) * iters * 32 * 2.0 * 32 * 32 * 16;
    printf("%s operands  %.1f ms  %.1f TFLOP/s  shader clock %.3f GHz\n", RANDOM ? "random  " : "constant", ms,
           flop / ms / 1e9, (double)h[0] / ((double)h[1] / 100.0) / 1e3);
}

int main() {
    for (int rep = 0; rep < 2; ++rep) { run<false>(40000); run<true>(40000); }
    return 0;
}
