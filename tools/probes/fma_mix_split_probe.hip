#include <hip/hip_runtime.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned lo_pair(const unsigned h, const float v0, const float v1) {
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(v0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(v1));
    return l;
}
__global__ void k(const float* in, unsigned* out) {
    const int i = threadIdx.x;
    const float v0 = in[2 * i], v1 = in[2 * i + 1];
    const f2 v = {v0, v1};
    const h2 h = __builtin_convertvector(v, h2);
    const h2 lr = __builtin_convertvector(v - __builtin_convertvector(h, f2), h2);
    out[3 * i] = __builtin_bit_cast(unsigned, h);
    out[3 * i + 1] = lo_pair(__builtin_bit_cast(unsigned, h), v0, v1);
    out[3 * i + 2] = __builtin_bit_cast(unsigned, lr);
}
int main() {
    const int n = 256;
    float hin[2 * n]; unsigned hout[3 * n];
    unsigned s = 12345;
    for (int i = 0; i < 2 * n; ++i) { s = s * 1664525u + 1013904223u; float f = (float)(int)(s >> 8) / 8388608.0f - 1.0f; hin[i] = f * (i % 7 == 0 ? 70000.0f : (i % 5 == 0 ? 1e-5f : 3.0f)); }
    float* din; unsigned* dout;
    hipMalloc(&din, sizeof(hin)); hipMalloc(&dout, sizeof(hout));
    hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(n), 0, 0, din, dout);
    hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) if (hout[3 * i + 1] != hout[3 * i + 2]) { if (bad < 5) printf("mismatch %d: %08x %08x (h %08x, v %g %g)\n", i, hout[3*i+1], hout[3*i+2], hout[3*i], hin[2*i], hin[2*i+1]); ++bad; }
    printf("mismatches: %d of %d\n", bad, n);
    return 0;
}
