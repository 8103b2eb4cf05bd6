// A generic dense product on the exact-fp32 MFMA, for NeRF shapes the fused kernels are not compiled for.
//
// Reference: run_nerf_helpers.py:76-128 builds ANY netdepth / netwidth / skip list / encoding width (flags
// run_plnerf.py:784-825); the fused MLP kernels serve the trunk every shipped configuration uses (8 x 256, one skip) and
// what maps onto it exactly.  Everything else runs layer by layer through this entry (pl-nerf_amd/generic.py): the
// layers' products, their input gradients and their weight gradients are all instances of
//
//     C[M, N] (ldc)  =  gate(A)[M, K] . B[K, N]  (+ bias[N])  (relu)  (+ C)
//
// with element (i, k) of A at a[i a_rs + k a_cs] and element (k, j) of B at b[k b_rs + j b_cs] (either operand may be a
// transposed view), `gate` (nullable, indexed like A): A(i, k) counts only where gate(i, k) > 0 -- the ReLU's derivative
// folded into the two backward products -- and `ones_col`: B has one more column than memory holds, all ones (the bias
// gradient as the last column of dW = gate(G)^T [X | 1]).  v_mfma_f32_32x32x2_f32, fp32 accumulation, k ascending:
// deterministic.  k_splits > 1: the k range is dealt out over that many workgroups per tile (a weight gradient has a few
// dozen tiles and 10^5..10^6 rows of k: without it 64 workgroups walked them alone, 430 ms per backward at 8 x 512), the
// partial products go to `partials` [k_splits][M][N] and a second launch adds them in order (+ bias, + C, relu).  Unfused and HBM-bound by construction (every layer's activations make a round trip): a correct
// native route for shapes no shipped configuration uses, not a fast one -- the hot path is mlp_rr / mlp_h16 / mlp_wgrad.
#include "common.h"
#include "mlp_frag.h"

namespace {

struct GemmArgs {
    const float* a; long long a_rs, a_cs;
    const float* b; long long b_rs, b_cs;
    const float* bias;      // [N] or nullptr
    const float* gate;      // indexed like a, or nullptr
    int M, N, K;
    int relu, accumulate, ones_col;
    float* c; long long ldc;
    int k_splits;           // > 1: blockIdx.z owns k in [z k_per, (z + 1) k_per) and writes its partial to partials[z][M][N]
    int k_per;
    float* partials;
};

constexpr int GT = 64;      // workgroup tile (rows and columns): four waves, 32 x 32 each
constexpr int GK = 16;      // k per LDS stage
constexpr int GLD = GT + 4; // LDS row stride (floats)

__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs p) {
    __shared__ float As[GK][GLD];      // As[k][i]
    __shared__ float Bs[GK][GLD];      // Bs[k][j]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.y * GT, j0 = blockIdx.x * GT;
    const int wi = (wave & 1) * 32, wj = (wave >> 1) * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    // the thread's four (row, k) / (k, column) slots of a stage, laid out so that the memory-contiguous index runs fastest
    // across the threads
    const bool a_k_fast = p.a_cs == 1, b_j_fast = p.b_cs == 1;
    const int n_mem = p.ones_col ? p.N - 1 : p.N;      // columns of B that exist in memory
    const int k_begin = p.k_splits > 1 ? (int)blockIdx.z * p.k_per : 0;
    const int k_end = p.k_splits > 1 ? min(p.K, k_begin + p.k_per) : p.K;
    for (int k0 = k_begin; k0 < k_end; k0 += GK) {
        float av[4], bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = tid + 256 * q;
            const int ai = a_k_fast ? s >> 4 : s & 63, ak = a_k_fast ? s & 15 : s >> 6;
            const int gi = i0 + ai, gk = k0 + ak;
            float v = 0.0f;
            if (gi < p.M && gk < k_end) {
                const long long off = (long long)gi * p.a_rs + (long long)gk * p.a_cs;
                v = p.a[off];
                if (p.gate && !(p.gate[off] > 0.0f)) v = 0.0f;
            }
            av[q] = v;
            const int bj = b_j_fast ? s & 63 : s >> 4, bk = b_j_fast ? s >> 6 : s & 15;
            const int gj = j0 + bj, gkb = k0 + bk;
            float w = 0.0f;
            if (gj < p.N && gkb < k_end) w = gj < n_mem ? p.b[(long long)gkb * p.b_rs + (long long)gj * p.b_cs] : 1.0f;
            bv[q] = w;
        }
        __syncthreads();      // (the previous stage's fragments have been read)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = tid + 256 * q;
            const int ai = a_k_fast ? s >> 4 : s & 63, ak = a_k_fast ? s & 15 : s >> 6;
            As[ak][ai] = av[q];
            const int bj = b_j_fast ? s & 63 : s >> 4, bk = b_j_fast ? s >> 6 : s & 15;
            Bs[bk][bj] = bv[q];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < GK / 2; ++s) {
            const int kk = 2 * s + (lane >> 5);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[kk][wi + (lane & 31)], Bs[kk][wj + (lane & 31)], acc, 0, 0, 0);
        }
    }
    const int j = j0 + wj + (lane & 31);
    if (j >= p.N) return;
    if (p.k_splits > 1) {      // the partial as it is; bias / accumulate / relu happen once, in the reduction
        float* part = p.partials + (size_t)blockIdx.z * (size_t)p.M * (size_t)p.N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + wi + plnerf::frag_row(r, lane);
            if (i < p.M) part[(size_t)i * p.N + j] = acc[r];
        }
        return;
    }
    const float bj = p.bias ? p.bias[j] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = i0 + wi + plnerf::frag_row(r, lane);
        if (i >= p.M) continue;
        float v = acc[r] + bj;
        float* dst = p.c + (long long)i * p.ldc + j;
        if (p.accumulate) v = v + *dst;
        if (p.relu) v = v > 0.0f ? v : 0.0f;
        *dst = v;
    }
}

__global__ __launch_bounds__(256) void gemm_reduce_kernel(const GemmArgs p) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x, n = (size_t)p.M * (size_t)p.N;
    if (idx >= n) return;
    float v = 0.0f;
    for (int z = 0; z < p.k_splits; ++z) v = v + p.partials[(size_t)z * n + idx];      // in order: deterministic
    const int i = (int)(idx / p.N), j = (int)(idx - (size_t)i * p.N);
    if (p.bias) v = v + p.bias[j];
    float* dst = p.c + (long long)i * p.ldc + j;
    if (p.accumulate) v = v + *dst;
    if (p.relu) v = v > 0.0f ? v : 0.0f;
    *dst = v;
}

}  // namespace

extern "C" int plnerf_gemm_f32(const float* a, int64_t a_row_stride, int64_t a_col_stride, const float* b,
                               int64_t b_row_stride, int64_t b_col_stride, const float* bias, const float* gate, int M, int N,
                               int K, int relu, int accumulate, int ones_col, float* c, int64_t ldc, int k_splits, float* partials,
                               plnerf_stream_t stream) {
    if (M < 0 || N < 0 || K < 0 || ldc < N || !c) return PLNERF_EINVAL;
    if (k_splits > 1 && !partials) return PLNERF_EINVAL;
    if (k_splits > 65535) return PLNERF_ERANGE;
    if (M == 0 || N == 0) return PLNERF_OK;
    if (K > 0 && (!a || (!b && !(ones_col && N == 1)))) return PLNERF_EINVAL;
    if ((long long)((M + GT - 1) / GT) > 65535) return PLNERF_ERANGE;
    GemmArgs p{a, a_row_stride, a_col_stride, b, b_row_stride, b_col_stride, bias, gate, M, N, K, relu ? 1 : 0,
               accumulate ? 1 : 0, ones_col ? 1 : 0, c, ldc, 1, K, nullptr};
    if (k_splits > 1 && K > GK) {
        int per = (K + k_splits - 1) / k_splits;
        per = (per + GK - 1) / GK * GK;                    // whole LDS stages
        p.k_per = per;
        p.k_splits = (K + per - 1) / per;                  // (only ranges that hold a k)
        p.partials = partials;
    }
    hipLaunchKernelGGL(gemm_f32_kernel, dim3((N + GT - 1) / GT, (M + GT - 1) / GT, p.k_splits), dim3(256), 0, (hipStream_t)stream, p);
    PLNERF_CHECK_LAUNCH();
    if (p.k_splits > 1) {
        const size_t n = (size_t)M * (size_t)N;
        hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
        PLNERF_CHECK_LAUNCH();
    }
    return PLNERF_OK;
}
