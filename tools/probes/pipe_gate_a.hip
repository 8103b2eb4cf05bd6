// Gate A of the layer-pipelined backward (LABNOTES R5-13's lead): can a producer workgroup hand a layer's
// dz tile (192 rows x 256 halves = 96 KB) to a consumer workgroup on another CU INSIDE one launch -- through flags and the
// XCD's L2 / the Infinity Cache -- cheaply enough that one persistent launch of 128 producer + 128 consumer workgroups beats
// the two launches it would replace (a gradient-chain kernel that writes the tiles to HBM, then a weight-gradient kernel
// that reads them back)?
//
// The probe keeps what decides that and drops the arithmetic's meaning:
//   producer, per tile:  768 MFMAs per workgroup (= one 256 x 256 dgrad layer on 192 rows, 25.2 MFLOP), then the tile out
//   consumer, per tile:  the tile in + a second 96 KB tile from an HBM stream (the saved activation plane), both through
//                        LDS, 768 MFMAs per workgroup (= the layer's weight-gradient job on 192 rows)
// Modes:  separate  launch A: 256 producer workgroups, tiles to a linear HBM buffer with non-temporal stores;
//                   launch B: 256 consumer workgroups read that buffer + the stream      -> time(A) + time(B)
//         pipe      ONE launch, 256 workgroups = 128 producer / consumer pairs on one XCD each, two ring slots per pair,
//                   hand-off by {sc1 write-through stores | plain stores + agent release}, monotonic counters, one polling
//                   lane with s_sleep, EVERY wait bounded (a timeout sets an error word and the kernel runs on: a broken
//                   protocol shows up as an error count, never as a hung GPU)
// Same total work in both modes.  hipcc --offload-arch=gfx950 -O3 -o /tmp/pipe_gate_a tools/probes/pipe_gate_a.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 512;
constexpr int TILE_BYTES = 192 * 256 * 2;           // 96 KB
constexpr int CHUNKS = TILE_BYTES / 16 / THREADS;   // 16-byte chunks per thread and tile: 12
constexpr int MFMA_PER_WAVE = 96;                   // 8 waves x 96 = 768 per workgroup and tile
constexpr int SPIN_LIMIT = 1 << 20;

__device__ __forceinline__ void store_sc1(u32x4* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// the producer's arithmetic: 96 dependent-free MFMAs per wave on toggling operands
__device__ __forceinline__ void mfma_block(f32x16 (&acc)[4], const h16x8 (&a)[4], const h16x8 (&b)[4]) {
#pragma unroll
    for (int i = 0; i < MFMA_PER_WAVE / 16; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[(j + t) & 3], acc[t], 0, 0, 0);
}

struct Args {
    u32x4* ring;            // pipe: [pairs][2 slots][TILE]; separate: [workgroups][tiles][TILE]
    const u32x4* stream;    // [consumers][tiles][TILE]: the consumer's second operand, read once each
    unsigned* produced;     // [pairs] monotonic
    unsigned* consumed;     // [pairs]
    unsigned* errors;
    float* sink;
    int tiles;              // per producer
    int mode;               // 0 = pipe with sc1 stores, 1 = pipe with plain stores + release fence
};

__device__ __forceinline__ bool wait_ge(unsigned* word, unsigned want, unsigned* errors) {
    // one lane polls (relaxed, agent scope = an L1-bypassing load), sleeps between polls, gives up after SPIN_LIMIT
    if (__hip_atomic_load(errors, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;      // (someone timed out already: run on without waiting)
    for (int i = 0; i < SPIN_LIMIT; ++i) {
        if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
        __builtin_amdgcn_s_sleep(8);
    }
    atomicAdd(errors, 1u);
    return false;
}

__device__ __forceinline__ void init_operands(h16x8 (&a)[4], h16x8 (&b)[4], f32x16 (&acc)[4], unsigned seed) {
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            unsigned h = (threadIdx.x * 977u + seed * 131u + i * 17u + e) * 0x9E3779B1u;
            h ^= h >> 15;
            a[i][e] = (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.f));
            b[i][e] = (_Float16)(((int)(h >> 16) - 32768) * (1.0f / 32768.f));
        }
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
}

__device__ void producer(const Args& p, int pair, bool pipe) {
    h16x8 a[4], b[4];
    f32x16 acc[4];
    init_operands(a, b, acc, pair);
    __shared__ int ok;
    const int tid = threadIdx.x;
    for (int t = 0; t < p.tiles; ++t) {
        mfma_block(acc, a, b);
        u32x4* dst = pipe ? p.ring + ((size_t)pair * 2 + (t & 1)) * (TILE_BYTES / 16) : p.ring + ((size_t)pair * p.tiles + t) * (TILE_BYTES / 16);
        if (pipe && t >= 2) {      // slot t & 1 is free once tile t - 2 has been consumed
            if (tid == 0) ok = wait_ge(p.consumed + pair, (unsigned)(t - 1), p.errors);
            __syncthreads();
        }
        u32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = __float_as_uint(acc[q][tid & 15]) ^ (unsigned)t;
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) {
            u32x4* q = dst + c * THREADS + tid;
            if (!pipe) __builtin_nontemporal_store(v, q);
            else if (p.mode == 0) store_sc1(q, v);
            else *q = v;
        }
        if (pipe) {
            if (p.mode == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            } else {
                __syncthreads();
                if (tid == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            }
            if (tid == 0) __hip_atomic_store(p.produced + pair, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) p.sink[0] = s;
}

__device__ void consumer(const Args& p, int pair, bool pipe) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // two operand tiles: 192 KB would not fit -> half tiles, twice
    u32x4* lds = reinterpret_cast<u32x4*>(smem);
    h16x8 a[4], b[4];
    f32x16 acc[4];
    init_operands(a, b, acc, 1000 + pair);
    const int tid = threadIdx.x;
    __shared__ int ok;
    for (int t = 0; t < p.tiles; ++t) {
        const u32x4* src = pipe ? p.ring + ((size_t)pair * 2 + (t & 1)) * (TILE_BYTES / 16) : p.ring + ((size_t)pair * p.tiles + t) * (TILE_BYTES / 16);
        const u32x4* hs = p.stream + ((size_t)pair * p.tiles + t) * (TILE_BYTES / 16);
        u32x4 h[CHUNKS];
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) h[c] = __builtin_nontemporal_load(hs + c * THREADS + tid);      // (does not depend on the producer)
        if (pipe) {
            if (tid == 0) {
                ok = wait_ge(p.produced + pair, (unsigned)(t + 1), p.errors);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
        }
        u32x4 d[CHUNKS];
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) d[c] = src[c * THREADS + tid];
        // both operands through LDS (what the transposing fragment reads of the real kernel need), half a tile at a time
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int c = 0; c < CHUNKS / 2; ++c) {
                lds[c * THREADS + tid] = d[half * (CHUNKS / 2) + c];
                lds[(CHUNKS / 2 + c) * THREADS + tid] = h[half * (CHUNKS / 2) + c];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4 x = lds[((tid * 7 + j * 64) & (CHUNKS * THREADS - 1))];
                a[j] = __builtin_bit_cast(h16x8, x);
            }
            __syncthreads();
        }
        if (pipe && tid == 0) __hip_atomic_store(p.consumed + pair, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mfma_block(acc, a, b);
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) p.sink[1] = s;
}

// pipe: block b -> XCD b % 8 (observed placement); pair = (b / 16) * 8 + b % 8, role = (b / 8) & 1: a pair shares an XCD
__global__ __launch_bounds__(THREADS) void pipe_kernel(Args p) {
    const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
    const int pair = (i >> 1) * 8 + x;
    if (i & 1) consumer(p, pair, true); else producer(p, pair, true);
}
__global__ __launch_bounds__(THREADS) void producer_kernel(Args p) { producer(p, blockIdx.x, false); }
__global__ __launch_bounds__(THREADS) void consumer_kernel(Args p) { consumer(p, blockIdx.x, false); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int tiles_pipe = argc > 1 ? atoi(argv[1]) : 256;      // per producer (128 producers); separate mode: half as many per workgroup (256 workgroups)
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    const size_t lds = (size_t)TILE_BYTES;
    const size_t tile16 = TILE_BYTES / 16;
    u32x4 *ring, *lin, *stream;
    unsigned *produced, *consumed, *errors;
    float* sink;
    const size_t total_tiles = (size_t)128 * tiles_pipe;
    CK(hipMalloc(&ring, (size_t)128 * 2 * TILE_BYTES));
    CK(hipMalloc(&lin, total_tiles * TILE_BYTES));
    CK(hipMalloc(&stream, total_tiles * TILE_BYTES));
    CK(hipMalloc(&produced, 128 * 4)); CK(hipMalloc(&consumed, 128 * 4)); CK(hipMalloc(&errors, 4)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(stream, 1, total_tiles * TILE_BYTES));
    CK(hipMemset(errors, 0, 4));
    CK(hipFuncSetAttribute((const void*)pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)consumer_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    printf("# %zu tiles of 96 KB (%.2f GB handed over, the same again streamed from HBM), 25.2 MFLOP per tile and side\n", total_tiles,
           total_tiles * (double)TILE_BYTES / 1e9);
    for (int rep = 0; rep < reps; ++rep) {
        {   // separate launches: 256 workgroups each, tiles_pipe / 2 tiles per workgroup
            Args a{lin, stream, produced, consumed, errors, sink, tiles_pipe / 2, 0};
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(producer_kernel, dim3(256), dim3(THREADS), 0, 0, a);
            CK(hipEventRecord(e1));
            hipLaunchKernelGGL(consumer_kernel, dim3(256), dim3(THREADS), lds, 0, a);
            CK(hipEventRecord(e2));
            CK(hipEventSynchronize(e2));
            float ma, mb;
            CK(hipEventElapsedTime(&ma, e0, e1)); CK(hipEventElapsedTime(&mb, e1, e2));
            printf("separate   producer launch %.3f ms (%.2f TB/s written)  consumer launch %.3f ms (%.2f TB/s read)  sum %.3f ms\n", ma,
                   total_tiles * (double)TILE_BYTES / ma / 1e9, mb, 2.0 * total_tiles * (double)TILE_BYTES / mb / 1e9, ma + mb);
        }
        for (int mode = 0; mode < 2; ++mode) {
            CK(hipMemset(produced, 0, 128 * 4)); CK(hipMemset(consumed, 0, 128 * 4));
            Args a{ring, stream, produced, consumed, errors, sink, tiles_pipe, mode};
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(pipe_kernel, dim3(256), dim3(THREADS), lds, 0, a);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned err = 0;
            CK(hipMemcpy(&err, errors, 4, hipMemcpyDeviceToHost));
            printf("pipe %-26s one launch %.3f ms   wait timeouts %u\n", mode == 0 ? "(sc1 write-through stores)" : "(plain stores + release)", ms, err);
        }
    }
    return 0;
}
