// The PL-NeRF MLP in exact-fp32 mode (PLNERF_PREC_FP32) on v_mfma_f32_32x32x2_f32.
//
// Reference: run_network (run_plnerf.py:78-92) = Embedder (run_nerf_helpers.py:24-54) +
// NeRF.forward (:105-128); the backward is what autograd derives for it at
// loss.backward() (run_plnerf.py:1300).
//
// Forward (mlp_fwd_f32_kernel): one 256-thread workgroup owns a tile of TM = 64 samples
// and walks all 12 layers with the tile's activations resident in LDS:
//   prologue  positional encoding computed into LDS (xyz: 60 sincos per sample; the view
//             direction is encoded from the per-ray vector, never materialised per sample)
//   per layer each of the 4 wavefronts owns a 64-column slab of the output; the K loop
//             reads A fragments (activations) from LDS with 16-byte reads and B fragments
//             (weights, pre-packed in MFMA fragment order, L2 resident) with one coalesced
//             16-byte global load per 4 MFMA steps; accumulators stay in registers until
//             the layer is complete, so the output overwrites the input tile in place
//             (bias + ReLU fused into that write; optional copy to HBM for backward)
//   heads     sigma (256->1) and rgb (128->3) are VALU dot products out of LDS.
// The skip connection (layer 5 = [encoding, h4]) is two K ranges over the two LDS buffers;
// no concatenation is ever materialised.
//
// Backward: mlp_bwd_f32_kernel is the same tile walk in reverse for the dgrad chain
// (dz_l = (dz_{l+1} W_{l+1}) * relu'(h_l)), writing every dz_l to HBM; wgrad_f32_kernel then
// forms dW_l = dz_l^T a_{l-1} as split-K MFMA GEMMs straight out of HBM (both operands are
// K-major in memory, which is exactly the f32 MFMA fragment order, so no LDS staging), and
// wgrad_reduce_kernel sums the split-K partials deterministically into the 24 gradient
// tensors in their original [out][in] layout.
//
// (The weight-gradient stage of every mode lives in mlp_wgrad.hip, Adam in adam.hip.)
//
// Numerics: f32 MFMA is bit-for-bit an fmaf chain; only the summation order differs from
// the reference's CPU GEMM, i.e. fp32 round-off (~1e-7 relative), well inside the 1e-5
// parity bound.
#include "common.h"
#include "mlp_frag.h"
#include "mlp_internal.h"
#include "mlp_layout.h"
#include "mlp_pack_src.h"
#include "pe_sincos.h"

using namespace plnerf;
using namespace plnerf::lay;

namespace {

// sin / cos of x * scale (scale = 2^f): the shared reduction of pe_sincos.h, library sincosf past its range
__device__ __forceinline__ void pe_sincos_any(const float x, const PeTurns t, const float scale, float* s, float* c) {
    if (__builtin_expect(fabsf(x) * scale < PE_FAST_LIMIT, 1)) pe_sincos(t, scale, s, c);
    else sincosf(x * scale, s, c);
}



constexpr int LDA = 260;  // activation tile row stride (floats): == 4 mod 64 -> conflict-free b128 reads
constexpr int LDP = 68;   // xyz-encoding tile row stride
constexpr int LDD = 36;   // direction-encoding tile row stride

// ------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------
__global__ void pack_f32_kernel(ParamPtrs P, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= PACKED_FLOATS) return;
    float v;
    if (idx < FWD_FLOATS) {
        int g = 0, off = 0;
        while (g < N_FWD - 1 && idx >= off + fwd_K[g] * fwd_N[g]) { off += fwd_K[g] * fwd_N[g]; ++g; }
        const int rem = idx - off, KC = fwd_K[g] >> 3;
        const int blk = rem >> 8, w = rem & 255;
        const int jt = blk / KC, kc = blk - jt * KC, lane = w >> 2, t = w & 3;
        v = fwd_src(P, g, kc * 8 + (lane >> 5) * 4 + t, jt * 32 + (lane & 31));
    } else if (idx < HB_END) {
        if (idx < HB_BF) { const int r = idx - HB_BIAS; v = P.p[2 * (r >> 8) + 1][r & 255]; }
        else if (idx < HB_BV) v = P.p[P_BF][idx - HB_BF];
        else if (idx < HB_WA) v = P.p[P_BV][idx - HB_BV];
        else if (idx < HB_BA) v = P.p[P_WA][idx - HB_WA];
        else if (idx < HB_WR) v = (idx == HB_BA) ? P.p[P_BA][0] : 0.0f;
        else if (idx < HB_BR) v = P.p[P_WR][idx - HB_WR];
        else v = (idx - HB_BR) < 3 ? P.p[P_BR][idx - HB_BR] : 0.0f;
    } else {
        int g = 0, off = BWD;
        while (g < N_BWD - 1 && idx >= off + bwd_K[g] * bwd_N[g]) { off += bwd_K[g] * bwd_N[g]; ++g; }
        const int rem = idx - off, KC = bwd_K[g] >> 3;
        const int blk = rem >> 8, w = rem & 255;
        const int jt = blk / KC, kc = blk - jt * KC, lane = w >> 2, t = w & 3;
        v = bwd_src(P, g, kc * 8 + (lane >> 5) * 4 + t, jt * 32 + (lane & 31));
    }
    out[idx] = v;
}

// ------------------------------------------------------------------------------------
// the tile GEMM: acc[i][j] += A[32 i .. +31][kc range] . B[kc range][32 j .. +31]
//   a_lane = &A[lane & 31][4 * (lane >> 5)]           (LDS, row stride lda)
//   b_lane = packed B of the wave's first j tile at this K range, + 4 * lane
// One K chunk = 8 k values = 4 MFMA steps; B for chunk kc+1 is loaded while chunk kc's
// NI*NJ*4 MFMAs (64 cycles each) issue.
// ------------------------------------------------------------------------------------
template <int NI, int NJ>
__device__ __forceinline__ void mma_run(f32x16 (&acc)[NI][NJ], const float* a_lane, const int lda,
                                        const float* __restrict__ b_lane, const int b_jt_stride,
                                        const int kcount) {
    f32x4 bcur[NJ], bnxt[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) bcur[j] = *reinterpret_cast<const f32x4*>(b_lane + j * b_jt_stride);
    for (int kc = 0; kc < kcount; ++kc) {
        if (kc + 1 < kcount) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                bnxt[j] = *reinterpret_cast<const f32x4*>(b_lane + j * b_jt_stride + (kc + 1) * 256);
        }
        f32x4 a[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) a[i] = *reinterpret_cast<const f32x4*>(a_lane + i * 32 * lda + kc * 8);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], bcur[j][t], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bcur[j] = bnxt[j];
    }
}

// ------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------
struct FwdArgs {
    const float* packed;
    const float* pts;
    const float* viewdirs;
    const float* embedded;
    int xyz_ch, dir_ch;   // columns of `embedded`: input_ch | input_ch_views
    int n_rows, spr;
    float* raw_out;
    float* saved;
    float pe_scale;       // the in-kernel encoding's input scale: sin / cos(x * pe_scale * 2^k) (1, or pi: depth variant)
    float act_beta;       // > 0: sigma leaves as softplus(beta)(sigma) (common.h: density_activation)
};

// bias (+relu) epilogue: accumulators -> LDS tile in place (+ saved plane in HBM)
template <int NI, int NJ, bool RELU, bool SAVE>
__device__ __forceinline__ void store_act(f32x16 (&acc)[NI][NJ], const float* __restrict__ bias, float* act,
                                          const int col0, float* __restrict__ plane, const int ld_plane,
                                          const int row0, const int rows_valid, const int lane) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = col0 + j * 32 + (lane & 31);
            const float b = bias[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + frag_row(r, lane);
                float v = acc[i][j][r] + b;
                if (RELU) v = v > 0.0f ? v : 0.0f;
                act[row * LDA + col] = v;
                if (SAVE && row < rows_valid) plane[(size_t)(row0 + row) * ld_plane + col] = v;
            }
        }
}

template <int NI, bool SAVE>
__global__ __launch_bounds__(256) void mlp_fwd_f32_kernel(FwdArgs a) {
    constexpr int TM = 32 * NI;
    constexpr int TPR = 256 / TM;  // threads per row in the VALU heads
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* act = smem;
    float* pe = act + TM * LDA;
    float* dpe = pe + TM * LDP;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row0 = blockIdx.x * TM;
    const int rows_valid = min(TM, a.n_rows - row0);
    const size_t N = (size_t)a.n_rows;

    // ---- prologue: encodings into LDS --------------------------------------------------
    if (a.embedded) {
        const int emb_ch = a.xyz_ch + a.dir_ch;
        for (int e = tid; e < TM * emb_ch; e += 256) {
            const int row = e / emb_ch, c = e - row * emb_ch;
            const int grow = min(row0 + row, a.n_rows - 1);
            const float v = a.embedded[(size_t)grow * emb_ch + c];
            if (c < a.xyz_ch) pe[row * LDP + c] = v;
            else dpe[row * LDD + (c - a.xyz_ch)] = v;
        }
        for (int row = tid; row < TM; row += 256) {
            for (int c = a.xyz_ch; c < PE_K; ++c) pe[row * LDP + c] = 0.0f;
            for (int c = a.dir_ch; c < DPE_K; ++c) dpe[row * LDD + c] = 0.0f;
        }
    } else {
        // thread -> (row, q): q indexes which frequencies this thread encodes
        for (int e = tid; e < TM * 4; e += 256) {
            const int row = e % TM, q = e / TM;
            const int grow = min(row0 + row, a.n_rows - 1);
            const float px = a.pts[3 * (size_t)grow + 0], py = a.pts[3 * (size_t)grow + 1],
                        pz = a.pts[3 * (size_t)grow + 2];
            float* prow = pe + row * LDP;
            // band arguments (x * pe_scale) * 2^k, the reference's association (model/run_nerf_helpers.py:123); pe_scale = 1
            // is run_nerf_helpers.py:45-48
            const float sx = px * a.pe_scale, sy = py * a.pe_scale, sz = pz * a.pe_scale;
            const PeTurns tx = pe_turns(sx), ty = pe_turns(sy), tz = pe_turns(sz);   // one reduction per coordinate
            for (int f = q; f < XYZ_FREQS; f += 4) {
                const float sc = (float)(1 << f);
                float s, c;
                pe_sincos_any(sx, tx, sc, &s, &c); prow[3 + 6 * f + 0] = s; prow[3 + 6 * f + 3] = c;
                pe_sincos_any(sy, ty, sc, &s, &c); prow[3 + 6 * f + 1] = s; prow[3 + 6 * f + 4] = c;
                pe_sincos_any(sz, tz, sc, &s, &c); prow[3 + 6 * f + 2] = s; prow[3 + 6 * f + 5] = c;
            }
            if (q == 0) { prow[0] = px; prow[1] = py; prow[2] = pz; prow[XYZ_CH] = 0.0f; }
            const int ray = grow / a.spr;
            const float dx = a.viewdirs[3 * (size_t)ray + 0], dy = a.viewdirs[3 * (size_t)ray + 1],
                        dz = a.viewdirs[3 * (size_t)ray + 2];
            float* drow = dpe + row * LDD;
            {
                const int f = q;  // DIR_FREQS == 4
                const float sc = (float)(1 << f);
                const float ux = dx * a.pe_scale, uy = dy * a.pe_scale, uz = dz * a.pe_scale;
                float s, c;
                pe_sincos_any(ux, pe_turns(ux), sc, &s, &c); drow[3 + 6 * f + 0] = s; drow[3 + 6 * f + 3] = c;
                pe_sincos_any(uy, pe_turns(uy), sc, &s, &c); drow[3 + 6 * f + 1] = s; drow[3 + 6 * f + 4] = c;
                pe_sincos_any(uz, pe_turns(uz), sc, &s, &c); drow[3 + 6 * f + 2] = s; drow[3 + 6 * f + 5] = c;
            }
            if (q == 1) { drow[0] = dx; drow[1] = dy; drow[2] = dz; }
            if (q == 2) {
#pragma unroll
                for (int c = DIR_CH; c < DPE_K; ++c) drow[c] = 0.0f;
            }
        }
    }
    __syncthreads();
    if (SAVE) {
        float* pe_plane = a.saved + (size_t)SV_PE_OFF * N;
        float* dpe_plane = a.saved + (size_t)SV_DPE_OFF * N;
        for (int e = tid; e < TM * PE_K; e += 256) {
            const int row = e >> 6, c = e & 63;
            if (row < rows_valid) pe_plane[(size_t)(row0 + row) * PE_K + c] = pe[row * LDP + c];
        }
        for (int e = tid; e < TM * DPE_K; e += 256) {
            const int row = e >> 5, c = e & 31;
            if (row < rows_valid) dpe_plane[(size_t)(row0 + row) * DPE_K + c] = dpe[row * LDD + c];
        }
    }

    const float* pk = a.packed;
    const int a_off = (lane & 31), a_k = 4 * (lane >> 5);
    const float* act_lane = act + a_off * LDA + a_k;
    const float* pe_lane = pe + a_off * LDP + a_k;
    const float* dpe_lane = dpe + a_off * LDD + a_k;
    const int jt0 = wave * 2;  // this wave's first 32-column tile (N = 256 layers)
    f32x16 acc[NI][2];

    // plane p of the saved buffer
#define PLANE(p) (a.saved + (size_t)(p) * W * N)
#define BLANE(g, KC) (pk + fwd_off(g) + (size_t)jt0 * (KC) * 256 + lane * 4)

    // L0: encoding -> 256
    zero_acc(acc);
    mma_run<NI, 2>(acc, pe_lane, LDP, BLANE(G_L0, 8), 8 * 256, 8);
    __syncthreads();
    store_act<NI, 2, true, SAVE>(acc, pk + HB_BIAS + 0 * W, act, jt0 * 32, SAVE ? PLANE(0) : nullptr, W, row0,
                                 rows_valid, lane);
    __syncthreads();
    // L1..L4
#pragma unroll 1
    for (int l = 1; l <= 4; ++l) {
        zero_acc(acc);
        mma_run<NI, 2>(acc, act_lane, LDA, pk + fwd_off(G_L1) + (size_t)(l - 1) * W * W + (size_t)jt0 * 32 * 256 + lane * 4,
                       32 * 256, 32);
        __syncthreads();
        store_act<NI, 2, true, SAVE>(acc, pk + HB_BIAS + l * W, act, jt0 * 32, SAVE ? PLANE(l) : nullptr, W, row0,
                                     rows_valid, lane);
        __syncthreads();
    }
    // L5: [encoding | h4] -> 256  (K = 64 + 256, two ranges over the two LDS tiles)
    zero_acc(acc);
    mma_run<NI, 2>(acc, pe_lane, LDP, BLANE(G_L5, 40), 40 * 256, 8);
    mma_run<NI, 2>(acc, act_lane, LDA, BLANE(G_L5, 40) + 8 * 256, 40 * 256, 32);
    __syncthreads();
    store_act<NI, 2, true, SAVE>(acc, pk + HB_BIAS + 5 * W, act, jt0 * 32, SAVE ? PLANE(5) : nullptr, W, row0,
                                 rows_valid, lane);
    __syncthreads();
    // L6, L7
#pragma unroll 1
    for (int l = 6; l <= 7; ++l) {
        zero_acc(acc);
        mma_run<NI, 2>(acc, act_lane, LDA, pk + fwd_off(G_L6) + (size_t)(l - 6) * W * W + (size_t)jt0 * 32 * 256 + lane * 4,
                       32 * 256, 32);
        __syncthreads();
        store_act<NI, 2, true, SAVE>(acc, pk + HB_BIAS + l * W, act, jt0 * 32, SAVE ? PLANE(l) : nullptr, W, row0,
                                     rows_valid, lane);
        __syncthreads();
    }
    // sigma = h7 . w_alpha + b_alpha  (VALU; TPR threads per row, interleaved columns)
    const int hrow = tid / TPR, hq = tid % TPR;
    float sigma = 0.0f;
    {
        const float* wa = pk + HB_WA;
        const float* hr = act + hrow * LDA;
#pragma unroll 8
        for (int k = 0; k < W / TPR; ++k) sigma = fmaf(hr[TPR * k + hq], wa[TPR * k + hq], sigma);
#pragma unroll
        for (int d = 1; d < TPR; d <<= 1) sigma += __shfl_xor(sigma, d);
        sigma += pk[HB_BA];
    }
    // feature = h7 W_f^T + b_f (no activation)
    zero_acc(acc);
    mma_run<NI, 2>(acc, act_lane, LDA, BLANE(G_FEAT, 32), 32 * 256, 32);
    __syncthreads();
    store_act<NI, 2, false, SAVE>(acc, pk + HB_BF, act, jt0 * 32, SAVE ? PLANE(SV_FEAT) : nullptr, W, row0,
                                  rows_valid, lane);
    __syncthreads();
    // view layer: [feature | direction encoding] -> 128, relu; each wave owns 32 columns
    f32x16 accv[NI][1];
    zero_acc(accv);
    {
        const float* bl = pk + fwd_off(G_VIEWS) + (size_t)wave * 36 * 256 + lane * 4;
        mma_run<NI, 1>(accv, act_lane, LDA, bl, 36 * 256, 32);
        mma_run<NI, 1>(accv, dpe_lane, LDD, bl + 32 * 256, 36 * 256, 4);
    }
    __syncthreads();
    store_act<NI, 1, true, SAVE>(accv, pk + HB_BV, act, wave * 32, SAVE ? a.saved + (size_t)SV_HV_OFF * N : nullptr,
                                 HV, row0, rows_valid, lane);
    __syncthreads();
    // rgb = hv W_rgb^T + b_rgb
    {
        const float* wr = pk + HB_WR;
        const float* hr = act + hrow * LDA;
        float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
#pragma unroll 8
        for (int k = 0; k < HV / TPR; ++k) {
            const float h = hr[TPR * k + hq];
            o0 = fmaf(h, wr[0 * HV + TPR * k + hq], o0);
            o1 = fmaf(h, wr[1 * HV + TPR * k + hq], o1);
            o2 = fmaf(h, wr[2 * HV + TPR * k + hq], o2);
        }
#pragma unroll
        for (int d = 1; d < TPR; d <<= 1) {
            o0 += __shfl_xor(o0, d);
            o1 += __shfl_xor(o1, d);
            o2 += __shfl_xor(o2, d);
        }
        if (hq == 0 && hrow < rows_valid) {
            float4 o;
            o.x = o0 + pk[HB_BR + 0];
            o.y = o1 + pk[HB_BR + 1];
            o.z = o2 + pk[HB_BR + 2];
            o.w = density_activation(sigma, a.act_beta);
            reinterpret_cast<float4*>(a.raw_out)[row0 + hrow] = o;
        }
    }
#undef PLANE
#undef BLANE
}

// ------------------------------------------------------------------------------------
// backward: dgrad chain
// ------------------------------------------------------------------------------------
struct BwdArgs {
    const float* packed;
    const float* g_raw;
    int n_rows;
    const float* saved;
    float* dz;
};

// accumulators (dh) -> (optional + g_sigma w_alpha) -> (optional relu mask from the saved
// activation plane) -> LDS tile in place + dz plane in HBM
template <int NI, bool MASK, bool ALPHA>
__device__ __forceinline__ void store_dz(f32x16 (&acc)[NI][2], float* g, const float* gr, const float* __restrict__ wa,
                                         const float* __restrict__ mask_plane, float* __restrict__ out_plane,
                                         const int col0, const int row0, const int rows_valid, const int lane) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = col0 + j * 32 + (lane & 31);
            const float wac = ALPHA ? wa[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + frag_row(r, lane);
                const bool ok = row < rows_valid;
                float v = acc[i][j][r];
                if (ALPHA) v = fmaf(gr[row * 4 + 3], wac, v);
                if (MASK) {
                    const float h = ok ? mask_plane[(size_t)(row0 + row) * W + col] : 0.0f;
                    v = h > 0.0f ? v : 0.0f;
                }
                g[row * LDA + col] = v;
                if (ok) out_plane[(size_t)(row0 + row) * W + col] = v;
            }
        }
}

template <int NI>
__global__ __launch_bounds__(256) void mlp_bwd_f32_kernel(BwdArgs a) {
    constexpr int TM = 32 * NI;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* g = smem;                 // TM x LDA: the current pre-activation gradient tile
    float* gr = g + TM * LDA;        // TM x 4: upstream gradient of (r,g,b,sigma)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row0 = blockIdx.x * TM;
    const int rows_valid = min(TM, a.n_rows - row0);
    const size_t N = (size_t)a.n_rows;
    const float* pk = a.packed;
#define SPLANE(p) (a.saved + (size_t)(p) * W * N)
#define DPLANE(p) (a.dz + (size_t)(p) * W * N)

    for (int row = tid; row < TM; row += 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < rows_valid) v = reinterpret_cast<const float4*>(a.g_raw)[row0 + row];
        reinterpret_cast<float4*>(gr)[row] = v;
    }
    __syncthreads();
    // dz_view = (g_rgb W_rgb) * relu'(hv)
    {
        const float* wr = pk + HB_WR;
        const float* hv_plane = a.saved + (size_t)SV_HV_OFF * N;
        float* dzv_plane = a.dz + (size_t)DZ_V_OFF * N;
        for (int e = tid; e < TM * HV; e += 256) {
            const int row = e >> 7, i = e & 127;
            const bool ok = row < rows_valid;
            float v = gr[row * 4 + 0] * wr[i];
            v = fmaf(gr[row * 4 + 1], wr[HV + i], v);
            v = fmaf(gr[row * 4 + 2], wr[2 * HV + i], v);
            const float h = ok ? hv_plane[(size_t)(row0 + row) * HV + i] : 0.0f;
            v = h > 0.0f ? v : 0.0f;
            g[row * LDA + i] = v;
            if (ok) dzv_plane[(size_t)(row0 + row) * HV + i] = v;
        }
    }
    __syncthreads();
    const float* g_lane = g + (lane & 31) * LDA + 4 * (lane >> 5);
    const int jt0 = wave * 2;
    f32x16 acc[NI][2];
#define DBLANE(d, KC) (pk + bwd_off(d) + (size_t)jt0 * (KC) * 256 + lane * 4)
    // d feature = dz_view . W_view[:, :256]   (K = 128)
    zero_acc(acc);
    mma_run<NI, 2>(acc, g_lane, LDA, DBLANE(D_VIEWS, 16), 16 * 256, 16);
    __syncthreads();
    store_dz<NI, false, false>(acc, g, gr, nullptr, nullptr, DPLANE(DZ_FEAT), jt0 * 32, row0, rows_valid, lane);
    __syncthreads();
    // d h7 = dz_feature . W_f + g_sigma w_alpha, masked by h7
    zero_acc(acc);
    mma_run<NI, 2>(acc, g_lane, LDA, DBLANE(D_FEAT, 32), 32 * 256, 32);
    __syncthreads();
    store_dz<NI, true, true>(acc, g, gr, pk + HB_WA, SPLANE(7), DPLANE(7), jt0 * 32, row0, rows_valid, lane);
    __syncthreads();
    // d h_{l-1} = dz_l . W_l (skip layer: hidden columns only), masked by h_{l-1};  l = 7..1
#pragma unroll 1
    for (int l = 7; l >= 1; --l) {
        zero_acc(acc);
        // packed order after D_FEAT: D_L7, D_L6, D_L5, D_L4, D_L3, D_L2, D_L1
        mma_run<NI, 2>(acc, g_lane, LDA, pk + bwd_off(D_L7) + (size_t)(7 - l) * W * W + (size_t)jt0 * 32 * 256 + lane * 4,
                       32 * 256, 32);
        __syncthreads();
        store_dz<NI, true, false>(acc, g, gr, nullptr, SPLANE(l - 1), DPLANE(l - 1), jt0 * 32, row0, rows_valid, lane);
        __syncthreads();
    }
#undef SPLANE
#undef DPLANE
#undef DBLANE
}

}  // namespace

// ------------------------------------------------------------------------------------
// launchers (called from mlp_api.hip)
// ------------------------------------------------------------------------------------
namespace plnerf {
namespace impl {

size_t f32_packed_bytes() { return (size_t)PACKED_FLOATS * sizeof(float); }

int f32_pack(const float* const* params, int xyz_ch, int dir_ch, void* packed, hipStream_t st) {
    ParamPtrs P;
    P.xyz_ch = xyz_ch; P.dir_ch = dir_ch; P.cb = nullptr;
    for (int i = 0; i < PLNERF_N_PARAM_TENSORS; ++i) {
        if (!params[i]) return PLNERF_EINVAL;
        P.p[i] = params[i];
    }
    const int threads = 256, blocks = (PACKED_FLOATS + threads - 1) / threads;
    hipLaunchKernelGGL(pack_f32_kernel, dim3(blocks), dim3(threads), 0, st, P, (float*)packed);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

int f32_fwd(const void* packed, const float* pts, const float* viewdirs, const float* embedded, int xyz_ch,
            int dir_ch, int n_rows, int samples_per_ray, FwdOpt opt, float* raw_out, void* saved, hipStream_t st) {
    constexpr int NI = 2, TM = 32 * NI;
    FwdArgs a{(const float*)packed, pts, viewdirs, embedded, xyz_ch, dir_ch, n_rows,
              samples_per_ray < 1 ? 1 : samples_per_ray, raw_out, (float*)saved, opt.pe_scale, opt.act_beta};
    const size_t lds = (size_t)TM * (LDA + LDP + LDD) * sizeof(float);
    dim3 grid((n_rows + TM - 1) / TM), block(256);
    if (saved) {
        (void)hipFuncSetAttribute((const void*)mlp_fwd_f32_kernel<NI, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((mlp_fwd_f32_kernel<NI, true>), grid, block, lds, st, a);
    } else {
        (void)hipFuncSetAttribute((const void*)mlp_fwd_f32_kernel<NI, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((mlp_fwd_f32_kernel<NI, false>), grid, block, lds, st, a);
    }
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

int f32_dgrad(const void* packed, const float* g_raw, int n_rows, const float* saved, float* dz, hipStream_t st) {
    constexpr int NI = 2, TM = 32 * NI;
    BwdArgs a{(const float*)packed, g_raw, n_rows, saved, dz};
    const size_t lds = (size_t)(TM * LDA + TM * 4) * sizeof(float);
    (void)hipFuncSetAttribute((const void*)mlp_bwd_f32_kernel<NI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    hipLaunchKernelGGL((mlp_bwd_f32_kernel<NI>), dim3((n_rows + TM - 1) / TM), dim3(256), lds, st, a);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

}  // namespace impl
}  // namespace plnerf
