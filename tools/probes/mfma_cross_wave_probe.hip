// Does vector-ALU / LDS-store / global-store work of ANOTHER wave on the same SIMD slow a wave's MFMA stream?  (The
// question behind overlapping the dgrad kernel's epilogue with its K loop on two wave groups: DESIGN.md section 10.)
// One workgroup of 8 waves per CU (two per SIMD: waves w and w + 4 share one).  Waves 0-3 run the K loop's stream --
// per v_mfma_f32_32x32x16_f16, NL ds_read_b128 of a 1 KB fragment, fragments consumed -- and wave 0 reports shader cycles
// per MFMA.  Waves 4-7 do, for the same time:
//   mode 0  nothing (they exit)
//   mode 1  a stream of independent v_fma_f32
//   mode 2  the epilogue's mixture per 16 values: 16 v_med3 + 8 cvt + 8 v_and + 2 ds_write_b128 + 2 global_store_dwordx4 (1 KB each)
//   mode 3  the same MFMA stream (two K loops on a SIMD: the unpipelined kernel's K phase)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_cross tools/probes/mfma_cross_wave_probe.hip && /tmp/mfma_cross
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u4 lds_u4;
typedef __attribute__((address_space(3))) u4 lds_u4w;

template <int NL, int NACC, int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, u4* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 131072 / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = i * 2654435761u;
    __syncthreads();
    lds_u4* base = (lds_u4*)((__attribute__((address_space(3))) const unsigned char*)smem + lane * 16);
    asm volatile("" : "+v"(base));
    float s = 0.f;
    if (wave < 4 || MODE == 3) {
        f32x16 acc[NACC];
        for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        h8 b;
        for (int e = 0; e < 8; ++e) b[e] = (_Float16)(float)((blockIdx.x + e) % 5 - 2);
        u4 frag[8] = {};
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                const h8 a = __builtin_bit_cast(h8, frag[(u + 4) % 8]);
                acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u % NACC], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < NL; ++q)
                    frag[(u * NL + q) % 8] = *(const lds_u4*)((__attribute__((address_space(3))) const unsigned char*)base + ((u * NL + q + wave * 8) % 64) * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_readcyclecounter();
        for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
        for (int i = 0; i < 8; ++i) s += (float)frag[i][0];
        if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
    } else if (MODE == 1) {
        float f[8];
        for (int i = 0; i < 8; ++i) f[i] = (float)threadIdx.x * 0.001f + i;
        for (int i = 0; i < iters * 12 * 8; ++i) {      // ~8 v_fma per MFMA of the other wave: never idle
#pragma unroll
            for (int q = 0; q < 8; ++q) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[q]) : "v"(f[(q + 3) % 8]));
        }
        for (int i = 0; i < 8; ++i) s += f[i];
    } else if (MODE == 2) {
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = (float)threadIdx.x * 0.37f + i;
        lds_u4w* wdst = (lds_u4w*)((__attribute__((address_space(3))) unsigned char*)smem + 65536 + (wave - 4) * 8192 + lane * 16);
        u4* gdst = sink + ((size_t)blockIdx.x * 4 + (wave - 4)) * 4096 + lane;
        for (int i = 0; i < iters * 2; ++i) {           // (one 16-value group costs the ALU ~35 instructions)
            unsigned w[8];
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(-65504.f), "v"(65504.f));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w[j]) : "v"(v[2 * j]), "v"(v[2 * j + 1]));
                asm volatile("v_and_b32 %0, %0, %1" : "+v"(w[j]) : "v"(0xffff0000u | i));
            }
            const u4 x0 = {w[0], w[1], w[2], w[3]}, x1 = {w[4], w[5], w[6], w[7]};
            wdst[(i & 3) * 64] = x0;
            wdst[(i & 3) * 64 + 256] = x1;
            __builtin_nontemporal_store(x0, gdst + (size_t)(i & 31) * 128);
            __builtin_nontemporal_store(x1, gdst + (size_t)(i & 31) * 128 + 64);
        }
        for (int i = 0; i < 16; ++i) s += v[i];
    }
    if (s == 12345.678f) out[0] = s;
}

template <int NL, int NACC, int MODE>
void run(int iters, u4* sink) {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 8);
    (void)hipFuncSetAttribute((const void*)k<NL, NACC, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int rep = 0; rep < 2; ++rep) {
        k<NL, NACC, MODE><<<256, 512, 131072>>>(out, cyc, sink, iters);
        (void)hipDeviceSynchronize();
    }
    unsigned long long c = 0;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    static const char* what[] = {"alone on its SIMD", "beside a v_fma stream", "beside the epilogue's mixture (ALU + LDS stores + dz stores)",
                                 "beside a second MFMA stream"};
    printf("%d ds_read_b128 per MFMA, %d accumulators, %s -> %.2f cycles per MFMA of this wave\n", NL, NACC, what[MODE],
           (double)c / ((double)iters * 12));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    u4* sink; (void)hipMalloc(&sink, (size_t)256 * 4 * 4096 * 16 + (1 << 20));
    const int it = 2000;
    run<0, 6, 0>(it, sink); run<0, 6, 1>(it, sink); run<0, 6, 2>(it, sink); run<0, 6, 3>(it, sink);
    run<1, 6, 0>(it, sink); run<1, 6, 1>(it, sink); run<1, 6, 2>(it, sink); run<1, 6, 3>(it, sink);
    run<1, 3, 0>(it, sink); run<1, 3, 2>(it, sink);
    return 0;
}
