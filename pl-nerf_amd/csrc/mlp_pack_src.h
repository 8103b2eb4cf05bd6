// Where each element of a (padded) GEMM operand comes from in the 24 parameter tensors.
// Shared by the fp32 and bf16 weight-packing kernels.
#pragma once
#include "common.h"
#include "mlp_layout.h"

namespace plnerf {

// xyz_ch / dir_ch: the network's input_ch / input_ch_views (63 / 27 at the reference's default flags,
// 57 / 3 for the depth-supervised variant); the GEMMs' K ranges are padded to PE_K / DPE_K with zeros.
// cb: the compose block of the packed buffer (mlp_layout.h: CB_*; the 16-bit modes' packing kernels), else unused
struct ParamPtrs { const float* p[PLNERF_N_PARAM_TENSORS]; int xyz_ch, dir_ch; const float* cb; };
struct GradPtrs { float* p[PLNERF_N_PARAM_TENSORS]; int xyz_ch, dir_ch; };

using namespace lay;

// forward operand of GEMM g: value multiplying input channel k for output feature j
// (k indexes the padded input: encodings 63->64 / 27->32, skip layer = [encoding 64 | hidden 256])
__device__ __forceinline__ float fwd_src(const ParamPtrs& P, int g, int k, int j) {
    switch (g) {
        case G_L0: return k < P.xyz_ch ? P.p[0][j * P.xyz_ch + k] : 0.0f;
        case G_L5:
            if (k < PE_K) return k < P.xyz_ch ? P.p[10][j * (W + P.xyz_ch) + k] : 0.0f;
            return P.p[10][j * (W + P.xyz_ch) + P.xyz_ch + (k - PE_K)];
        case G_FEAT: return P.p[P_WF][j * W + k];
        case G_VIEWS:
            if (k < W) return P.p[P_WV][j * (W + P.dir_ch) + k];
            return (k - W) < P.dir_ch ? P.p[P_WV][j * (W + P.dir_ch) + k] : 0.0f;
        default: return P.p[2 * g][j * W + k];  // G_L1..G_L4, G_L6, G_L7: layer index == g
    }
}

// dgrad operand of GEMM g: W[o][col0 + i] (the hidden-state columns only)
__device__ __forceinline__ float bwd_src(const ParamPtrs& P, int g, int o, int i) {
    switch (g) {
        case D_VIEWS: return P.p[P_WV][o * (W + P.dir_ch) + i];
        case D_FEAT: return P.p[P_WF][o * W + i];
        case D_L5: return P.p[10][o * (W + P.xyz_ch) + P.xyz_ch + i];
        case D_L7: return P.p[14][o * W + i];
        case D_L6: return P.p[12][o * W + i];
        case D_L4: return P.p[8][o * W + i];
        case D_L3: return P.p[6][o * W + i];
        case D_L2: return P.p[4][o * W + i];
        default: return P.p[2][o * W + i];  // D_L1
    }
}

// The same for the composed network of the 16-bit modes (FwdGemmC / BwdGemmC): the view layer reads h7 through
// W_c = W_vf W_f, which plnerf_mlp_pack_weights left in the compose block (P.cb) before the packing kernels run.
__device__ __forceinline__ float fwdc_src(const ParamPtrs& P, int g, int k, int j) {
    if (g == GC_VIEWS) {
        if (k < W) return P.cb[CB_WC + j * W + k];
        return (k - W) < P.dir_ch ? P.p[P_WV][j * (W + P.dir_ch) + k] : 0.0f;
    }
    return fwd_src(P, g, k, j);      // GC_L0..GC_L7 = G_L0..G_L7
}
__device__ __forceinline__ float bwdc_src(const ParamPtrs& P, int g, int o, int i) {
    if (g == DC_VIEWS) return P.cb[CB_WC + o * W + i];
    return bwd_src(P, g + 1, o, i);   // DC_L7.. = D_L7.. shifted by the feature GEMM that is not there
}

}  // namespace plnerf
