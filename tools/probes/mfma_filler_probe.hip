// What one filler of each kind costs beside v_mfma_f32_32x32x16_f16 with ONE wave per SIMD (the register-resident
// forward's regime): per MFMA, NV independent v_fma_f32 + NL ds_read_b128 of a 1 KB weight fragment (lane-contiguous, as
// the ring is read) with the s_waitcnt lgkmcnt the real stream needs; every DMAP-th MFMA also issues one LDS-DMA piece
// (s_mov m0 / s_nop / global_load_lds_dwordx4, L2-resident source) like issue_piece.  Prints shader cycles per MFMA.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_fill tools/probes/mfma_filler_probe.hip && /tmp/mfma_fill
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u4 lds_u4;

template <int NV, int NL, int DMAP, bool USE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, const unsigned char* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = i * 2654435761u;
    __syncthreads();
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_byte*)smem;
    lds_u4* base = (lds_u4*)((__attribute__((address_space(3))) const unsigned char*)smem + lane * 16);
    asm volatile("" : "+v"(base));
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(float)((threadIdx.x * 7 + e) % 13 - 6); b[e] = (_Float16)(float)((blockIdx.x + e) % 5 - 2); }
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = (float)threadIdx.x * 0.001f + i;
    u4 frag[4] = {};
    const unsigned char* gsrc = src + (size_t)wave * 1024;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            if (USE) {      // the MFMA's A operand is a fragment read a few MFMAs ago (as the real stream's are)
                a = __builtin_bit_cast(h8, frag[(u + 2) % 4]);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NL; ++q) {
                frag[(u * NL + q) % 4] = *(const lds_u4*)((__attribute__((address_space(3))) const unsigned char*)base + ((u * NL + q) % 32) * 1024);
            }
#pragma unroll
            for (int q = 0; q < NV; ++q) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[q % 8]) : "v"(f[(q + 3) % 8]));
            if (DMAP > 0 && u % DMAP == DMAP - 1) {
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane * 16), "s"(gsrc + (u % 8) * 4096),
                             "s"(lds_base + 32768 + ((u % 8) * 4 + wave) * 1024) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (DMAP > 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int i = 0; i < 8; ++i) s += f[i];
    for (int i = 0; i < 4; ++i) s += (float)frag[i][0];
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NV, int NL, int DMAP, bool USE>
void run(int iters, const unsigned char* src) {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 8);
    (void)hipFuncSetAttribute((const void*)k<NV, NL, DMAP, USE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) {
        k<NV, NL, DMAP, USE><<<256, 256, 65536>>>(out, cyc, src, iters);
        (void)hipDeviceSynchronize();
    }
    unsigned long long c = 0;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("per MFMA: %d v_fma + %d ds_read_b128%s%s -> %.2f cycles per MFMA\n", NV, NL,
           DMAP ? (DMAP == 6 ? " + 1/6 LDS-DMA piece" : " + 1/3 LDS-DMA piece") : "", USE ? " (fragments consumed)" : "",
           (double)c / ((double)iters * 12));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    unsigned char* src; (void)hipMalloc(&src, 1 << 20); (void)hipMemset(src, 1, 1 << 20);
    const int it = 4000;
    run<0, 0, 0, false>(it, src);
    run<3, 0, 0, false>(it, src);
    run<0, 1, 0, false>(it, src); run<0, 2, 0, false>(it, src);
    run<2, 1, 0, false>(it, src); run<3, 1, 0, false>(it, src); run<4, 1, 0, false>(it, src);
    run<3, 1, 0, true>(it, src);
    run<3, 0, 6, false>(it, src); run<3, 0, 3, false>(it, src);
    run<2, 1, 6, false>(it, src); run<3, 1, 6, false>(it, src); run<3, 1, 6, true>(it, src);
    return 0;
}
