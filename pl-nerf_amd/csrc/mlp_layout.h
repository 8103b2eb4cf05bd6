// Static geometry of the PL-NeRF MLP (run_nerf_helpers.py:76-128 at the reference's default
// flags) and of the buffers the fp32 MFMA kernels exchange.  Host and device code share this.
//
// Layers as GEMMs (y = x W^T + b; weight tensors are [out][in]):
//   L0  63->256 | L1..L4 256->256 | L5 (63+256)->256, encoding channels first | L6,L7 256->256
//   sigma 256->1 (from h7) | feature 256->256 (from h7, no activation)
//   view layer (256 feature + 27 direction encoding)->128, relu | rgb 128->3
// K is padded to a multiple of 8 (one MFMA chunk): encodings 63->64 and 27->32.
#pragma once
#include <stddef.h>

namespace plnerf {
namespace lay {

constexpr int W = 256;        // trunk width
constexpr int HV = 128;       // view-layer width
constexpr int XYZ_CH = 63;    // 3 + 6*10
constexpr int DIR_CH = 27;    // 3 + 6*4
constexpr int PE_K = 64;      // padded xyz encoding
constexpr int DPE_K = 32;     // padded direction encoding
constexpr int XYZ_FREQS = 10;
constexpr int DIR_FREQS = 4;
constexpr int EMB_CH = XYZ_CH + DIR_CH;  // 90

// state_dict order of the 24 parameter tensors
enum ParamIdx {
    P_W0 = 0, P_B0 = 1,  // pts_linears.i.weight = 2i, .bias = 2i+1 (i = 0..7)
    P_WV = 16, P_BV = 17, P_WF = 18, P_BF = 19, P_WA = 20, P_BA = 21, P_WR = 22, P_BR = 23
};

// ---- packed weights, fp32 mode ------------------------------------------------------
// A GEMM's B operand (logical B[k][j], K x N) is stored as 1 KiB blocks, one per (32-wide
// j tile, 8-deep k chunk), block order [jt][kc]; inside a block lane l = (kh<<5 | jj) owns 4
// consecutive floats t = 0..3 holding B[kc*8 + kh*4 + t][jt*32 + jj].  A wavefront reads one
// block with a single coalesced 16-byte-per-lane load and feeds v_mfma_f32_32x32x2_f32 step t
// with register t; the matching A fragment is one 16-byte LDS read of
// A[row][kc*8 + kh*4 .. +3].
enum FwdGemm { G_L0 = 0, G_L1, G_L2, G_L3, G_L4, G_L5, G_L6, G_L7, G_FEAT, G_VIEWS, N_FWD };
constexpr int fwd_K[N_FWD] = {64, 256, 256, 256, 256, 320, 256, 256, 256, 288};
constexpr int fwd_N[N_FWD] = {256, 256, 256, 256, 256, 256, 256, 256, 256, 128};

// dgrad: dh_in[m][i] = sum_o dz[m][o] W[o][col0+i]  -> B[k=o][j=i]
enum BwdGemm { D_VIEWS = 0, D_FEAT, D_L7, D_L6, D_L5, D_L4, D_L3, D_L2, D_L1, N_BWD };
constexpr int bwd_K[N_BWD] = {128, 256, 256, 256, 256, 256, 256, 256, 256};
constexpr int bwd_N[N_BWD] = {256, 256, 256, 256, 256, 256, 256, 256, 256};

constexpr int fwd_off(int g) {
    int o = 0;
    for (int i = 0; i < g; ++i) o += fwd_K[i] * fwd_N[i];
    return o;
}
constexpr int FWD_FLOATS = fwd_off(N_FWD);  // 593,920

// head block: biases and the two thin output layers, kept in original order
constexpr int HB = FWD_FLOATS;
constexpr int HB_BIAS = HB;                  // b0..b7: 8 x 256
constexpr int HB_BF = HB_BIAS + 8 * W;       // feature bias 256
constexpr int HB_BV = HB_BF + W;             // view-layer bias 128
constexpr int HB_WA = HB_BV + HV;            // sigma weight 256
constexpr int HB_BA = HB_WA + W;             // sigma bias 1 (+3 pad)
constexpr int HB_WR = HB_BA + 4;             // rgb weight 3 x 128
constexpr int HB_BR = HB_WR + 3 * HV;        // rgb bias 3 (+1 pad)
constexpr int HB_END = HB_BR + 4;

constexpr int BWD = HB_END;
constexpr int bwd_off(int g) {
    int o = BWD;
    for (int i = 0; i < g; ++i) o += bwd_K[i] * bwd_N[i];
    return o;
}
constexpr int PACKED_FLOATS = bwd_off(N_BWD);

// ---- the network as the 16-bit MFMA modes evaluate it: feature_linear composed into the view layer ------------------
// run_nerf_helpers.py:115-121 puts NO activation between feature_linear and views_linears[0], so
//     hv_pre = W_vf (W_f h7 + b_f) + W_vd dir + b_v  =  W_c h7 + W_vd dir + b_c,   W_c = W_vf W_f (128 x 256), b_c = W_vf b_f + b_v
// (W_vf / W_vd: the feature / direction columns of views_linears.0.weight).  The 16-bit modes evaluate the right-hand side
// (round 6): the 256 x 256 feature GEMM (65,536 of the row's 593,408 MACs), its dgrad GEMM, its weight-gradient job, the
// `feature` saved plane and the dz_feature plane disappear.  The backward recovers the two factors' gradients exactly from
// G = dz_view^T h7 (128 x 256) and s = column sums of dz_view:  dW_f = W_vf^T G,  db_f = W_vf^T s,  dW_vf = G W_f^T + s b_f^T
// (mlp_compose.hip).  The exact-fp32 mode keeps the reference's two layers op for op (tables above).
enum FwdGemmC { GC_L0 = 0, GC_L1, GC_L2, GC_L3, GC_L4, GC_L5, GC_L6, GC_L7, GC_VIEWS, N_FWDC };
constexpr int fwdc_K[N_FWDC] = {64, 256, 256, 256, 256, 320, 256, 256, 288};
constexpr int fwdc_N[N_FWDC] = {256, 256, 256, 256, 256, 256, 256, 256, 128};
constexpr int fwdc_off(int g) {
    int o = 0;
    for (int i = 0; i < g; ++i) o += fwdc_K[i] * fwdc_N[i];
    return o;
}
constexpr int FWDC_FLOATS = fwdc_off(N_FWDC);  // 528,384
// dgrad: d h7 = dz_view W_c (K = 128), then the trunk as before
enum BwdGemmC { DC_VIEWS = 0, DC_L7, DC_L6, DC_L5, DC_L4, DC_L3, DC_L2, DC_L1, N_BWDC };
constexpr int bwdc_K[N_BWDC] = {128, 256, 256, 256, 256, 256, 256, 256};
constexpr int bwdc_N[N_BWDC] = {256, 256, 256, 256, 256, 256, 256, 256};
constexpr int bwdc_off(int g) {      // (relative to the dgrad section's start)
    int o = 0;
    for (int i = 0; i < g; ++i) o += bwdc_K[i] * bwdc_N[i];
    return o;
}
constexpr int BWDC_FLOATS = bwdc_off(N_BWDC);
// The compose block of the packed buffer (fp32, written by plnerf_mlp_pack_weights before the operand sections are packed
// from it): W_c and b_c for the packing kernels; the factors themselves for the backward's last step, which has the
// packed buffer but not the parameter tensors.
constexpr int CB_WC = 0;                     // [128][256]
constexpr int CB_BC = CB_WC + HV * W;        // [128]
constexpr int CB_WVF = CB_BC + HV;           // [128][256]  views_linears.0.weight[:, :256]
constexpr int CB_WF = CB_WVF + HV * W;       // [256][256]  feature_linear.weight
constexpr int CB_BF = CB_WF + W * W;         // [256]       feature_linear.bias
constexpr int CB_FLOATS = CB_BF + W;         // 131,456
// the reduced G (128 x 256) and s (128) between the weight-gradient reduction and the factors' gradients (workspace)
constexpr int GRED_FLOATS = HV * W + HV;

// ---- forward state saved for backward (floats per row, plane-major: plane p starts at
// plane_off(p) * n_rows) ---------------------------------------------------------------
// planes 0..7 = h0..h7 (post-relu), 8 = feature, then hv [128], pe [64], dpe [32]
constexpr int SV_FEAT = 8;
constexpr int SV_HV_OFF = 9 * W;             // floats-per-row offset of the hv plane
constexpr int SV_PE_OFF = SV_HV_OFF + HV;
constexpr int SV_DPE_OFF = SV_PE_OFF + PE_K;
constexpr int SV_FLOATS = SV_DPE_OFF + DPE_K;       // 2528 plane elements per row (fp32 in fp32 mode, half otherwise)
// relu bit masks written by the 16-bit forward kernels for the dgrad kernel (1 bit per activation instead
// of re-reading the planes): h0..h7 at 32 B/row each, then hv at 16 B/row
constexpr int SV_MASK_BYTES = 8 * (W / 8) + HV / 8; // 272 B per row
constexpr int SAVED_PER_ROW = SV_FLOATS + SV_MASK_BYTES / 4;   // fp32 mode: 2596 floats per row

// ---- backward workspace (fp32 mode; the 16-bit modes' planes: DZC_* below): pre-activation gradients, planes 0..7 = dz0..dz7, 8 = dz_feature,
// then dz_view [128] ----------------------------------------------------------------------
constexpr int DZ_FEAT = 8;
constexpr int DZ_V_OFF = 9 * W;
constexpr int DZ_PER_ROW = DZ_V_OFF + HV;    // 2432

// ---- the same state in the 16-bit MFMA modes: IEEE-half planes ---------------------------------
// The saved activations are O(1) and feed only the weight-gradient contraction over >= 10^5 samples,
// where an unbiased 2^-12 rounding per element averages out: they are stored
// as half [plane][row][width] (same plane order, SV_FLOATS halves per row) followed by the ReLU bit
// masks.  The pre-activation gradients span the fp32 exponent range, so their half planes carry one
// power-of-two scale per launch (max |g_raw| -> [8, 16), 4096x of headroom below the half maximum,
// conversion saturating); the reduction kernel divides it back out.  Both halve the bytes the
// backward pass moves through HBM.
// Rows are padded to a multiple of SV_ROW_PAD in every offset of the half state (plane p starts at
// plane_off(p) * sv_rows(n_rows)), so that the TILED layout below never runs from one plane into the next and every
// wave of the forward kernel that writes it owns a whole tile (rows past n_rows: padding, written, never read).
//
// Layouts of the planes h0..h7 (256 wide), hv (128) and -- since round 5 -- the two encoding planes (64 / 32 wide, in
// the register-resident forward's own column order: sv_enc_channel); the relu bits are always row-major:
//   SV_LAYOUT_ROWS   [row][width]: written by the ping-pong forward (its LDS tile is in this order)
//   SV_LAYOUT_TILED  32-row tiles in the register-resident forward's own order: tile t = row / 32 holds
//                    [slab j = col / 32][fragment f][lane half g][row r = row % 32][8 halves], the 8 halves being
//                    features 32 j + 16 f + 4 g + {0..3} and 32 j + 16 f + 8 + 4 g + {0..3} -- what a lane of the MFMA
//                    accumulator layout holds, so a wave stores a fragment as ONE contiguous KiB straight from its
//                    registers.  4 consecutive features of a row stay contiguous, which is all the weight-gradient
//                    kernel's transposing LDS reads need.
// The register-resident forward's k order of the ENCODING operands (its operand registers hold, in k-step s, lane half g,
// element e, slot t = 8 s + e): xyz (63 channels + 1 pad): t < 30: frequency band 5 g + t / 6, function / axis t % 6 (sin x,
// sin y, sin z, cos x, cos y, cos z, the reference's order); t = 30, 31: x, y | z, pad -- each lane half evaluates whole
// bands.  Direction (27 + 5 pad), t in 0..15: t < 12: band 2 g + t / 6; then x, y, z, channel 27 | channels 28..31.  Both
// are bijections onto 0..63 / 0..31.  Its saved encoding planes are TILED planes whose column 16 s + 8 (e >> 2) + 4 g +
// (e & 3) holds that channel (sv_enc_channel below): what the weight-gradient reduction un-permutes.
__host__ __device__ constexpr int rr_pe_channel(int g, int t) {
    return t < 30 ? 3 + 6 * (5 * g + t / 6) + t % 6 : (t == 30 ? (g ? 2 : 0) : (g ? PE_K - 1 : 1));
}
__host__ __device__ constexpr int rr_dpe_channel(int g, int t) {
    return t < 12 ? 3 + 6 * (2 * g + t / 6) + t % 6 : (g ? 28 + (t - 12) : (t < 15 ? t - 12 : 27));
}
// channel held by column c of a tiled encoding plane (xyz: 64 columns, direction: 32)
__host__ __device__ constexpr int sv_enc_channel(int c, bool dir) {
    const int s = c >> 4, g = (c >> 2) & 1, e = ((c >> 3) & 1) * 4 + (c & 3);
    return dir ? rr_dpe_channel(g, 8 * s + e) : rr_pe_channel(g, 8 * s + e);
}
constexpr int SV_ROW_PAD = 256;     // = the register-resident forward's largest workgroup tile: its plane stores are unconditional
constexpr int SV_LAYOUT_ROWS = 0, SV_LAYOUT_TILED = 1;
__host__ __device__ constexpr size_t sv_rows(size_t n_rows) { return (n_rows + SV_ROW_PAD - 1) / SV_ROW_PAD * SV_ROW_PAD; }
// position (in halves) of element (row, col) of a `width`-wide plane in the tiled layout
__host__ __device__ constexpr size_t sv_tiled_index(size_t row, int col, int width) {
    return (row >> 5) * (size_t)(32 * width) +
           ((size_t)((((col >> 5) * 2 + ((col >> 4) & 1)) * 2 + ((col >> 2) & 1)) * 32) + (row & 31)) * 8 +
           (size_t)(((col >> 3) & 1) * 4 + (col & 3));
}
// The half dz planes (backward workspace): rows padded to the dgrad kernel's tiles; the 256-wide planes dz0..dz7
// in the TILED order above (the dgrad kernel's accumulators leave as contiguous KiB fragments, no LDS ->
// HBM copy pass; the weight-gradient kernels read them as they are), dz_view (128 wide, not an MFMA product) row-major.
#ifndef PLNERF_BWD_TM
#define PLNERF_BWD_TM 192     // rows per workgroup tile of the half dgrad kernel (mlp_h16_body.inc)
#endif
constexpr int DZ_ROW_PAD = PLNERF_BWD_TM;
__host__ __device__ constexpr size_t dz_rows(size_t n_rows) { return (n_rows + DZ_ROW_PAD - 1) / DZ_ROW_PAD * DZ_ROW_PAD; }
// planes of the half state (the composed network has no feature plane and no dz_feature plane): h0..h7, then hv [128], pe [64],
// dpe [32]; dz0..dz7, then dz_view [128]
constexpr int SVC_HV_OFF = 8 * W;
constexpr int SVC_PE_OFF = SVC_HV_OFF + HV;
constexpr int SVC_DPE_OFF = SVC_PE_OFF + PE_K;
constexpr int SVC_FLOATS = SVC_DPE_OFF + DPE_K;                    // 2272 halves per row
constexpr int DZC_V_OFF = 8 * W;
constexpr int DZC_PER_ROW = DZC_V_OFF + HV;                        // 2176 halves per row
constexpr int SVH_BYTES_PER_ROW = SVC_FLOATS * 2 + SV_MASK_BYTES;  // 4816
constexpr int DZH_BYTES_PER_ROW = DZC_PER_ROW * 2;                 // 4352
constexpr int WSH_SCALARS_BYTES = 16;                              // max |g_raw| (fp32 bits) + pad
#ifndef PLNERF_DZH_TARGET_EXP
#define PLNERF_DZH_TARGET_EXP 4
#endif
constexpr float DZH_TARGET_EXP = (float)(PLNERF_DZH_TARGET_EXP);   // scaled max |g_raw| in [2^3, 2^4)
constexpr float H16_MAX = 65504.0f;

// split-K partial sums of the weight gradients (per split)
constexpr int WG_MAIN_JOBS = 8;              // L1..L4, L5 (hidden part), L6, L7, feature
constexpr int PART_MAIN = 0;                                   // 8 x [256][256]
constexpr int PART_VMAIN = PART_MAIN + WG_MAIN_JOBS * W * W;   // [128][256]
constexpr int PART_PE0 = PART_VMAIN + HV * W;                  // L0:  [256][64]
constexpr int PART_PE5 = PART_PE0 + W * PE_K;                  // L5 encoding part: [256][64]
constexpr int PART_VDIR = PART_PE5 + W * PE_K;                 // view layer direction part [128][32]
constexpr int PART_BIAS = PART_VDIR + HV * DPE_K;              // b0..b7, bf: 9 x 256, bv: 128
constexpr int PART_SIGMA = PART_BIAS + 9 * W + HV;             // alpha_linear.weight [256] (tiled half planes: the view job's idle waves sum it, mlp_wgrad.hip)
constexpr int PART_PER_SPLIT = PART_SIGMA + W;
constexpr int MAX_SPLITS = 128;
constexpr int HEAD_PART = 648;               // per-workgroup partial of the sigma/rgb heads
#ifndef PLNERF_MAX_HEAD_WGS
#define PLNERF_MAX_HEAD_WGS 512
#endif
constexpr int MAX_HEAD_WGS = PLNERF_MAX_HEAD_WGS;

}  // namespace lay
}  // namespace plnerf
