// Measurement aid of bench.py (NOT part of libplnerf_hip.so): the MFMA rate the chip sustains when the operands toggle like
// real data.  A back-to-back v_mfma_f32_32x32x16_f16 stream (the instruction of the 16-bit modes' kernels) on random half
// fragments, every CU busy, nothing else in the loop -- the power management then holds the shader clock near 1.7 GHz and
// the stream tops out near 1.73 PFLOP/s of the nominal 2.5 (tools/probes/mfma_power.hip, profiles/r01_mfma_power_probe.txt).
// bench.py runs it for ~0.2 s in the SAME process as the timed step and puts the figure next to the nominal peak on the
// driver line: `roofline.sustained_peak`, `roofline.frac_of_sustained`.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/probes/libplnerf_probes.so tools/probes/mfma_sustained.hip
#include <hip/hip_runtime.h>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
__device__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ __launch_bounds__(512) void mfma_stream_kernel(float* out, int iters, int random) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    h16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i)
        for (int e = 0; e < 8; ++e) {
            const unsigned h = hash32(threadIdx.x * 977u + blockIdx.x * 131u + i * 17u + e);
            a[i][e] = (_Float16)(random ? ((int)(h & 0xffff) - 32768) * (1.0f / 32768.f) : 1.0f);
            b[i][e] = (_Float16)(random ? ((int)(h >> 16) - 32768) * (1.0f / 32768.f) : 1.0f);
        }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[(j + t) & 7], acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) out[0] = s;      // (never true: keeps the stream alive)
}
}  // namespace

// TFLOP/s of the stream (second of two launches, HIP events on `stream`); < 0 on a HIP error.  random != 0: toggling operands.
extern "C" double plnerf_probe_mfma_tflops(int iters, int random, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    float* out = nullptr;
    if (hipMalloc(&out, 4) != hipSuccess) return -1.0;
    const int threads = 512, blocks = 256 * 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_stream_kernel, dim3(blocks), dim3(threads), 0, st, out, iters / 8 + 1, random);      // (clock settles)
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL(mfma_stream_kernel, dim3(blocks), dim3(threads), 0, st, out, iters, random);
    (void)hipEventRecord(e1, st);
    double tf = -1.0;
    float ms = 0.f;
    if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f) {
        const double flop = (double)blocks * (threads / 64) * (double)iters * 32 * 2.0 * 32 * 32 * 16;
        tf = flop / ms / 1e9;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(out);
    return tf;
}
