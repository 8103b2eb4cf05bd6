// The composed view layer of the 16-bit MFMA modes (mlp_layout.h: "the network as the 16-bit MFMA modes evaluate it").
//
// Reference: run_nerf_helpers.py:115-121 -- feature = feature_linear(h); h = cat([feature, input_views]);
// h = relu(views_linears[0](h)) -- no activation between the two linear maps, so the product of the two weight matrices
// serves the forward and the dgrad chain, and autograd's gradients of the two factors follow from G = dz_view^T h7:
//
//   compose_pack_kernel    W_c = W_vf W_f, b_c = W_vf b_f + b_v  (fp64 accumulation, each element rounded to fp32 once)
//                          + fp32 copies of W_vf, W_f, b_f for the backward (which is handed the packed buffer only)
//   compose_grad_kernel    dW_f = W_vf^T G,  db_f = W_vf^T s,  dW_vf = G W_f^T + s b_f^T   (fixed summation order)
//
// Two 8.4 M-MAC products per network and step on the vector ALU: ~3 us each against the 65,536 MACs per ROW they remove
// from the forward, the dgrad chain and the weight-gradient stage.
#include "common.h"
#include "mlp_internal.h"
#include "mlp_layout.h"

using namespace plnerf;
using namespace plnerf::lay;

namespace {

struct ComposePackArgs {
    const float* wv;      // views_linears.0.weight [128][256 + dir_ch]
    const float* bv;      // [128]
    const float* wf;      // feature_linear.weight [256][256]
    const float* bf;      // [256]
    int dir_ch;
    float* cb;
};

// blocks [0, 64): two rows o of W_c each -- thread = (quarter q of the j range, column k), the quarters added in a fixed
// order through LDS (one thread walking all 256 j with its loads eight deep measured 13.5 us per launch: latency of the
// 256 dependent trips, not arithmetic); block 64: b_c; blocks 65..: the copies
constexpr int CP_ROWS = 2;
constexpr int CP_THREADS = 1024, CP_Q = CP_THREADS / W, CP_JQ = W / CP_Q;
constexpr int CP_WC_BLOCKS = HV / CP_ROWS;
constexpr int CP_COPY_FLOATS = HV * W + W * W + W;
constexpr int CP_COPY_BLOCKS = (CP_COPY_FLOATS / 4 + CP_THREADS - 1) / CP_THREADS;
static_assert(CP_COPY_FLOATS % 4 == 0, "copies in float4");

__global__ __launch_bounds__(CP_THREADS) void compose_pack_kernel(ComposePackArgs a) {
    const int tid = threadIdx.x, b = blockIdx.x;
    const int ldv = W + a.dir_ch;
    if (b < CP_WC_BLOCKS) {
        __shared__ float rows[CP_ROWS][W];
        __shared__ double part[CP_Q][CP_ROWS][W];
        const int o0 = b * CP_ROWS, q = tid >> 8, k = tid & (W - 1);
        if (tid < CP_ROWS * W) rows[tid >> 8][k] = a.wv[(size_t)(o0 + (tid >> 8)) * ldv + k];
        __syncthreads();
        double acc[CP_ROWS];
#pragma unroll
        for (int r = 0; r < CP_ROWS; ++r) acc[r] = 0.0;
#pragma unroll
        for (int j0 = 0; j0 < CP_JQ; j0 += 32) {
            float w[32];
#pragma unroll
            for (int t = 0; t < 32; ++t) w[t] = a.wf[(size_t)(q * CP_JQ + j0 + t) * W + k];      // (32 loads in flight)
#pragma unroll
            for (int t = 0; t < 32; ++t)
#pragma unroll
                for (int r = 0; r < CP_ROWS; ++r) acc[r] = fma((double)rows[r][q * CP_JQ + j0 + t], (double)w[t], acc[r]);
        }
#pragma unroll
        for (int r = 0; r < CP_ROWS; ++r) part[q][r][k] = acc[r];
        __syncthreads();
        if (tid < CP_ROWS * W) {
            const int r = tid >> 8;
            double s = part[0][r][k];
#pragma unroll
            for (int qq = 1; qq < CP_Q; ++qq) s += part[qq][r][k];
            a.cb[CB_WC + (o0 + r) * W + k] = (float)s;
        }
    } else if (b == CP_WC_BLOCKS) {
        // b_c[o] = W_vf[o, :] . b_f + b_v[o]: a wavefront per output (coalesced along j), 16 waves x 8 outputs
        const int lane = tid & 63, wave = tid >> 6;
        for (int o = wave; o < HV; o += CP_THREADS / 64) {
            double acc = 0.0;
#pragma unroll
            for (int t = 0; t < W / 64; ++t)
                acc = fma((double)a.wv[(size_t)o * ldv + lane + 64 * t], (double)a.bf[lane + 64 * t], acc);
            acc = wave_sum(acc);
            if (lane == 0) a.cb[CB_BC + o] = (float)(acc + (double)a.bv[o]);
        }
    } else {
        const int q = (b - CP_WC_BLOCKS - 1) * CP_THREADS + tid;      // float4 index into [W_vf | W_f | b_f]
        if (q >= CP_COPY_FLOATS / 4) return;
        const int e = 4 * q;
        float4 v;
        if (e < HV * W) {
            const int o = e >> 8, k = e & 255;                  // (rows of views_linears.0.weight are 256 + dir_ch long: not 16-byte aligned in general)
            const float* src = a.wv + (size_t)o * ldv + k;
            v = make_float4(src[0], src[1], src[2], src[3]);
        } else if (e < HV * W + W * W) {
            v = *reinterpret_cast<const float4*>(a.wf + (e - HV * W));
        } else {
            v = *reinterpret_cast<const float4*>(a.bf + (e - HV * W - W * W));
        }
        *reinterpret_cast<float4*>(a.cb + CB_WVF + e) = v;
    }
}

// ---- the factors' gradients ---------------------------------------------------------------------------------------
// One 32 x 32 output tile per workgroup (256 threads, 4 outputs each), operands staged through LDS in 32-deep steps.
//   tiles [0, 64):   dW_f[k][i]  = sum_o W_vf[o][k] G[o][i]                      (M = 256 k, N = 256 i, K = 128 o)
//   tiles [64, 96):  dW_vf[o][k] = sum_i G[o][i] W_f[k][i] + s[o] b_f[k]         (M = 128 o, N = 256 k, K = 256 i)
//   tile 96:         db_f[k]     = sum_o W_vf[o][k] s[o]
struct ComposeGradArgs {
    const float* cb[2];
    const float* gred[2];
    float* dwf[2];
    float* dbf[2];
    float* dwv[2];
    int dir_ch;
};
constexpr int CG_TILES = 64 + 32 + 1;

__global__ __launch_bounds__(256) void compose_grad_kernel(ComposeGradArgs a) {
    const int net = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const float* cb = a.cb[net];
    const float* G = a.gred[net];
    const float* s = G + HV * W;
    const float* wvf = cb + CB_WVF;
    const float* wf = cb + CB_WF;
    const float* bf = cb + CB_BF;
    __shared__ float As[32][33], Bs[32][33];      // As[kk][m], Bs[kk][n]
    const int tx = tid & 31, ty = tid >> 5;       // outputs (m = ty + 8 q, n = tx), q = 0..3
    if (tile < 64) {
        const int m0 = (tile >> 3) * 32, n0 = (tile & 7) * 32;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < HV; k0 += 32) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int kk = ty + 8 * q;
                As[kk][tx] = wvf[(k0 + kk) * W + m0 + tx];      // A[m = k][kk = o] = W_vf[o][k]
                Bs[kk][tx] = G[(k0 + kk) * W + n0 + tx];        // B[kk = o][n = i]
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                const float bv = Bs[kk][tx];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(As[kk][ty + 8 * q], bv, acc[q]);
            }
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) a.dwf[net][(m0 + ty + 8 * q) * W + n0 + tx] = acc[q];
    } else if (tile < 96) {
        const int t = tile - 64, m0 = (t >> 3) * 32, n0 = (t & 7) * 32;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < W; k0 += 32) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = ty + 8 * q;
                As[tx][r] = G[(m0 + r) * W + k0 + tx];          // A[m = o][kk = i]: rows of G are contiguous in i
                Bs[tx][r] = wf[(n0 + r) * W + k0 + tx];         // B[kk = i][n = k] = W_f[k][i]: rows of W_f contiguous in i
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                const float bv = Bs[kk][tx];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(As[kk][ty + 8 * q], bv, acc[q]);
            }
            __syncthreads();
        }
        const int ldv = W + a.dir_ch;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = m0 + ty + 8 * q, k = n0 + tx;
            a.dwv[net][(size_t)o * ldv + k] = fmaf(s[o], bf[k], acc[q]);
        }
    } else {
        float acc = 0.0f;
        for (int o = 0; o < HV; ++o) acc = fmaf(wvf[o * W + tid], s[o], acc);
        a.dbf[net][tid] = acc;
    }
}

}  // namespace

namespace plnerf {
namespace impl {

int compose_pack(const float* const* params, int dir_ch, float* cb, hipStream_t st) {
    const ComposePackArgs a{params[P_WV], params[P_BV], params[P_WF], params[P_BF], dir_ch, cb};
    hipLaunchKernelGGL(compose_pack_kernel, dim3(CP_WC_BLOCKS + 1 + CP_COPY_BLOCKS), dim3(CP_THREADS), 0, st, a);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

int compose_grads(int n, const ComposeGradJob* jobs, hipStream_t st) {
    if (n < 1 || n > MAX_BWD_JOBS) return PLNERF_EINVAL;
    ComposeGradArgs a{};
    for (int j = 0; j < 2; ++j) {
        const ComposeGradJob& jb = jobs[j < n ? j : 0];
        a.cb[j] = jb.cb; a.gred[j] = jb.gred;
        a.dwf[j] = jb.grads[P_WF]; a.dbf[j] = jb.grads[P_BF]; a.dwv[j] = jb.grads[P_WV];
    }
    a.dir_ch = jobs[0].dir_ch;
    hipLaunchKernelGGL(compose_grad_kernel, dim3(CG_TILES, n), dim3(256), 0, st, a);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

}  // namespace impl
}  // namespace plnerf
