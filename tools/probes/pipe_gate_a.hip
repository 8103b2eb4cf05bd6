// Gate A of the layer-pipelined backward (LABNOTES R5-13's lead): can a producer workgroup hand a layer's
// dz tile (192 rows x 256 halves = 96 KB) to a consumer workgroup on another CU INSIDE one launch -- through flags and the
// XCD's L2 / the Infinity Cache -- cheaply enough that one persistent launch of 128 producer + 128 consumer workgroups beats
// the two launches it would replace (a gradient-chain kernel that writes the tiles to HBM, then a weight-gradient kernel
// that reads them back)?
//
// The probe keeps what decides that and drops the arithmetic's meaning:
//   producer, per tile:  768 MFMAs per workgroup (= one 256 x 256 dgrad layer on 192 rows, 25.2 MFLOP), then the tile out
//   consumer, per tile:  the tile in + a second 96 KB tile from an HBM stream (the saved activation plane), in three 64-row
//                        stages through a double-buffered LDS image (the real weight-gradient kernel's structure: stage g + 1
//                        is fetched into registers before stage g's MFMAs and stashed after them, one barrier per stage),
//                        768 MFMAs per workgroup and tile (= the layer's weight-gradient job on 192 rows)
// Modes:  separate  launch A: 256 producer workgroups, tiles to a linear HBM buffer with non-temporal stores;
//                   launch B: 256 consumer workgroups read that buffer + the stream      -> time(A) + time(B)
//         pipe      ONE launch, 256 workgroups = 128 producer / consumer pairs on one XCD each, SLOTS ring slots per pair:
//                   sc1 (write-through) payload stores, a tile's flag published one tile late (its stores have had a whole
//                   compute phase to land), sc1 payload loads (no acquire fence), monotonic counters, ONE polling lane with
//                   s_sleep whose answer rides on the stage's own barrier, EVERY wait bounded (a timeout sets an error word
//                   and the kernel runs on: a broken protocol shows up as an error count, never as a hung GPU)
// Same total work in both modes.  hipcc --offload-arch=gfx950 -O3 -o /tmp/pipe_gate_a tools/probes/pipe_gate_a.hip
// (v1 of this probe -- whole tiles, no software pipelining, an acquire fence per tile -- measured the pipe 1.3-1.6x SLOWER
// than the two launches: profiles/r06_pipe_gate_a_v1_unpipelined.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int THREADS = 512;
constexpr int TILE_BYTES = 192 * 256 * 2;           // 96 KB
constexpr int TILE16 = TILE_BYTES / 16;             // 6144 16-byte chunks
constexpr int CHUNKS = TILE16 / THREADS;            // per thread and tile: 12
constexpr int STAGES = 3, SCH = CHUNKS / STAGES;    // 64-row stages: 4 chunks per thread, operand and stage
constexpr int STAGE16 = TILE16 / STAGES;            // 2048 chunks = 32 KB
constexpr int SLOTS = 4;
constexpr int SPIN_LIMIT = 1 << 20;

__device__ __forceinline__ void store_sc1(u32x4* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void load_sc1(u32x4& d, const u32x4* p) { asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void load_nt(u32x4& d, const u32x4* p) { asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int N>
__device__ __forceinline__ void mfma_block(f32x16 (&acc)[4], const h16x8 (&a)[4], const h16x8 (&b)[4]) {
#pragma unroll
    for (int i = 0; i < N / 16; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[(j + t) & 3], acc[t], 0, 0, 0);
}

struct Args {
    u32x4* ring;            // pipe: [pairs][SLOTS][TILE]; separate: [workgroups][tiles][TILE]
    const u32x4* stream;    // [consumers][tiles][TILE]: the consumer's second operand, read once each
    unsigned* produced;     // [pairs] monotonic: tiles published
    unsigned* consumed;     // [pairs] tiles whose slot is free again
    unsigned* errors;
    float* sink;
    int tiles;              // per producer
    int solo;               // pipe diagnostics: 1 = producers only, 2 = consumers only (the other role returns at once; no waits)
};

__device__ __forceinline__ bool wait_ge(unsigned* word, unsigned want, unsigned* errors) {
    if (!word) return true;
    if (__hip_atomic_load(errors, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;      // (someone timed out already: run on)
    for (int i = 0; i < SPIN_LIMIT; ++i) {
        if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
        __builtin_amdgcn_s_sleep(4);
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

template <bool PIPE>
__device__ void producer(const Args& p, int pair) {
    h16x8 a[4], b[4];
    f32x16 acc[4];
    init_operands(a, b, acc, pair);
    const int tid = threadIdx.x;
    for (int t = 0; t < p.tiles; ++t) {
        mfma_block<96>(acc, a, b);
        u32x4* dst = PIPE ? p.ring + ((size_t)pair * SLOTS + (t % SLOTS)) * TILE16 : p.ring + ((size_t)pair * p.tiles + t) * TILE16;
        if (PIPE) {
            // tile t - 1's stores were issued a whole compute phase ago: wait for them, publish it, then make sure slot
            // t % SLOTS is free (tile t - SLOTS consumed) -- one barrier for both
            wait_vm0();
            __syncthreads();
            if (tid == 0) {
                if (t >= 1) __hip_atomic_store(p.produced + pair, (unsigned)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t >= SLOTS) (void)wait_ge(p.solo ? nullptr : p.consumed + pair, (unsigned)(t - SLOTS + 1), p.errors);
            }
            __syncthreads();
        }
        u32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = __float_as_uint(acc[q][tid & 15]) ^ (unsigned)t;
#pragma unroll
        for (int c = 0; c < CHUNKS; ++c) {
            u32x4* q = dst + c * THREADS + tid;
            if (PIPE) store_sc1(q, v); else __builtin_nontemporal_store(v, q);
        }
    }
    if (PIPE) {
        wait_vm0();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(p.produced + pair, (unsigned)p.tiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.678f) p.sink[0] = s;
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// TWO stages in flight (two register sets; loads retire in order, so "the older set has landed" is vmcnt(2 SCH)): with
// half the chip's CUs on this side of the pipe, a consumer has to move twice the bytes per CU that the stand-alone
// launch's does -- one stage ahead left it latency-bound (profiles/r06_pipe_gate_a_v2_one_stage_ahead.txt)
template <bool PIPE>
__device__ void consumer(const Args& p, int pair) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [buffer 2][operand 2][32 KB]
    u32x4* lds = reinterpret_cast<u32x4*>(smem);
    h16x8 a[4], b[4];
    f32x16 acc[4];
    init_operands(a, b, acc, 1000 + pair);
    const int tid = threadIdx.x;
    const int n_stages = p.tiles * STAGES;      // even
    u32x4 d[2][SCH], h[2][SCH];
    auto fetch = [&](u32x4 (&dd)[SCH], u32x4 (&hh)[SCH], int g) {
        const int gg = g < n_stages ? g : n_stages - 1;      // (past the end: re-read the last stage -- static load counts)
        const int t = gg / STAGES, s = gg - t * STAGES;
        const u32x4* src = (PIPE ? p.ring + ((size_t)pair * SLOTS + (t % SLOTS)) * TILE16 : p.ring + ((size_t)pair * p.tiles + t) * TILE16) + s * STAGE16;
        const u32x4* hs = p.stream + ((size_t)pair * p.tiles + t) * TILE16 + s * STAGE16;
#pragma unroll
        for (int c = 0; c < SCH; ++c) {
            if (PIPE) load_sc1(dd[c], src + c * THREADS + tid); else load_nt(dd[c], src + c * THREADS + tid);
            load_nt(hh[c], hs + c * THREADS + tid);
        }
    };
    auto stash = [&](const u32x4 (&dd)[SCH], const u32x4 (&hh)[SCH], int buf) {
#pragma unroll
        for (int c = 0; c < SCH; ++c) {
            lds[(buf * 2 + 0) * STAGE16 + c * THREADS + tid] = dd[c];
            lds[(buf * 2 + 1) * STAGE16 + c * THREADS + tid] = hh[c];
        }
    };
    auto compute = [&](int g) {
        const u32x4* img = lds + (g & 1) * 2 * STAGE16;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x4 f = img[(tid * 5 + (k * 4 + j) * 67) & (2 * STAGE16 - 1)];
                asm volatile("" : "+v"(f));
                if (k < 2) a[j] = __builtin_bit_cast(h16x8, f);
            }
            if (k < 2) mfma_block<16>(acc, a, b);
        }
    };
    // iteration g: stage g + 1 is in flight in set (g + 1) & 1; request stage g + 2 into set g & 1; compute g; wait for g + 1; stash it
    auto iteration = [&](u32x4 (&d_next)[SCH], u32x4 (&h_next)[SCH], u32x4 (&d_far)[SCH], u32x4 (&h_far)[SCH], int g) {
        const int t = g / STAGES, s = g - t * STAGES;
        fetch(d_far, h_far, g + 2);      // (a new tile when s == 1: its flag was polled at s == 0, a barrier ago)
        compute(g);
        if (PIPE && s == 0 && t + 1 < p.tiles && tid == 0) (void)wait_ge(p.solo ? nullptr : p.produced + pair, (unsigned)(t + 2), p.errors);
        wait_vm<2 * SCH>();
        stash(d_next, h_next, (g + 1) & 1);
        if (PIPE && s == 1 && tid == 0)      // tile t's last stage (requested at s == 0) has landed: its slot is free
            __hip_atomic_store(p.consumed + pair, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    };
    if (PIPE) {
        if (tid == 0) (void)wait_ge(p.solo ? nullptr : p.produced + pair, 1u, p.errors);
        __syncthreads();
    }
    fetch(d[0], h[0], 0);
    fetch(d[1], h[1], 1);
    wait_vm<2 * SCH>();
    stash(d[0], h[0], 0);
    __syncthreads();
    for (int g = 0; g < n_stages; g += 2) {
        iteration(d[1], h[1], d[0], h[0], g);
        iteration(d[0], h[0], d[1], h[1], g + 1);
    }
    wait_vm0();
    float sum = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) sum += acc[t][r];
    if (sum == 12345.678f) p.sink[1] = sum;
}

// pipe: block b -> XCD b % 8 (observed placement); pair = (b / 16) * 8 + b % 8, role = (b / 8) & 1: a pair shares an XCD
__global__ __launch_bounds__(THREADS) void pipe_kernel(Args p) {
    const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
    const int pair = (i >> 1) * 8 + x;
    if (i & 1) { if (p.solo != 1) consumer<true>(p, pair); } else { if (p.solo != 2) producer<true>(p, pair); }
}
__global__ __launch_bounds__(THREADS) void producer_kernel(Args p) { producer<false>(p, blockIdx.x); }
__global__ __launch_bounds__(THREADS) void consumer_kernel(Args p) { consumer<false>(p, blockIdx.x); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int tiles_pipe = argc > 1 ? atoi(argv[1]) : 256;      // per producer (128 producers); separate mode: half as many per workgroup (256 workgroups)
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    const size_t lds = (size_t)4 * STAGE16 * 16;
    u32x4 *ring, *lin, *stream;
    unsigned *produced, *consumed, *errors;
    float* sink;
    const size_t total_tiles = (size_t)128 * tiles_pipe;
    CK(hipMalloc(&ring, (size_t)128 * SLOTS * TILE_BYTES));
    CK(hipMalloc(&lin, total_tiles * TILE_BYTES));
    CK(hipMalloc(&stream, total_tiles * TILE_BYTES));
    CK(hipMalloc(&produced, 128 * 4)); CK(hipMalloc(&consumed, 128 * 4)); CK(hipMalloc(&errors, 4)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(stream, 1, total_tiles * TILE_BYTES));
    CK(hipMemset(errors, 0, 4));
    CK(hipFuncSetAttribute((const void*)pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)consumer_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    printf("# %zu tiles of 96 KB (%.2f GB handed over, the same again streamed from HBM), 25.2 MFLOP per tile and side; ring %d slots per pair\n",
           total_tiles, total_tiles * (double)TILE_BYTES / 1e9, SLOTS);
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
        for (int solo = 0; solo < 3; ++solo) {
            CK(hipMemset(produced, 0, 128 * 4)); CK(hipMemset(consumed, 0, 128 * 4));
            Args a{ring, stream, produced, consumed, errors, sink, tiles_pipe, solo};
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(pipe_kernel, dim3(256), dim3(THREADS), lds, 0, a);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned err = 0;
            CK(hipMemcpy(&err, errors, 4, hipMemcpyDeviceToHost));
            printf("pipe %-42s one launch %.3f ms   wait timeouts %u\n", solo == 0 ? "" : (solo == 1 ? "(128 producers alone, no waits)" : "(128 consumers alone, no waits)"), ms, err);
        }
    }
    return 0;
}
