#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned relu_half_word(const h16x8 frag) {
    const u32x4 d = __builtin_bit_cast(u32x4, frag);
    const unsigned one = 0x00010001u;
    unsigned acc = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned b;
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(b) : "v"(d[q]), "v"(one));
        acc |= b << (8 * (q >> 1) + 2 * (q & 1));
    }
    return (acc | ((acc >> 16) << 1)) & 0xFFFFu;
}
__global__ void k(const h16x8* in, unsigned* out) {
    const int i = threadIdx.x;
    const h16x8 f = in[i];
    unsigned ref = 0;
    for (int e = 0; e < 8; ++e) {
        const int q = e >> 1;
        const int pos = 8 * (q >> 1) + 2 * (q & 1) + (e & 1);
        if ((float)f[e] > 0.0f) ref |= 1u << pos;
    }
    out[2 * i] = relu_half_word(f);
    out[2 * i + 1] = ref;
}
int main() {
    const int n = 256;
    _Float16 h[8 * n]; unsigned o[2 * n];
    unsigned s = 7;
    for (int i = 0; i < 8 * n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (s >> 28) < 7 ? (_Float16)0.0f : (_Float16)((float)(s >> 8 & 0xffff) / 1000.0f + 1e-4f); }
    h16x8* din; unsigned* dout;
    (void)hipMalloc(&din, sizeof(h)); (void)hipMalloc(&dout, sizeof(o));
    (void)hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(n), 0, 0, din, dout);
    (void)hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) if (o[2 * i] != o[2 * i + 1]) { if (bad < 5) printf("%d: %04x vs %04x\n", i, o[2 * i], o[2 * i + 1]); ++bad; }
    printf("relu_half_word mismatches: %d of %d\n", bad, n);
}
