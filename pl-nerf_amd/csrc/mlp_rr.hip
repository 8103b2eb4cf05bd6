// Register-resident forward kernel of the 16-bit-operand MFMA MLP (IEEE-half elements: precision modes F16X3 / F16).
//
// Reference: run_network (run_plnerf.py:78-92) = Embedder (run_nerf_helpers.py:24-54) + NeRF.forward (:105-128).
//
// Why a second forward kernel.  The ping-pong kernel (mlp_h16_fwd_pp.inc) keeps the ACTIVATIONS of a 64-row tile in
// LDS and each wave's 32-feature slab of the WEIGHTS in registers: every 64 rows re-fetch all of the network's weight
// fragments from L2 (37 KB per row in split mode, 12 TB/s of L2 reads chip-wide), every layer costs an LDS round
// trip of the activations and two workgroup barriers per half tile, and 160 KB of LDS cap the tile at 64 rows.
// Here the roles are swapped:
//
//   * a wave owns 32 rows for the whole network and keeps their activations in REGISTERS.  With
//     OUT^T[feature][row] = W . X^T the accumulator layout of v_mfma_f32_32x32x16_f16 (lane = row, four consecutive
//     features per register group) IS the next layer's B-operand layout once the k order inside a 16-deep k-step
//     is permuted (bits 2 and 3 of k swapped) -- and the weights' k order is free, so the permutation is folded
//     into the packed weights.  Bias + ReLU + hi/lo split turn 16 accumulator registers into the 16 operand
//     registers of two k-steps, in place: no LDS traffic, no barrier, no other wave involved.
//   * the four waves of a workgroup (one per SIMD, up to 512 registers each) share ONE stream of weight fragments
//     that an LDS-DMA ring (global_load_lds_dwordx4, 4 slots x 32 KB) pulls from L2: 128 rows per fetch instead of
//     64, one barrier per 48 MFMAs whose only job is to hand a landed slot over.
//   * the sigma and rgb heads are folded into the epilogues that hold their inputs in fp32 registers; the saved
//     half planes of the training forward leave straight from the registers, in 32-row tiles of the accumulator
//     layout (SV_LAYOUT_TILED, mlp_layout.h: one contiguous KiB per store instruction); the weight-gradient
//     kernels read that layout as it is.
//
// Per unit (one 32-feature slab x 16 k-steps in split mode) a wave issues 48 MFMAs, 32 ds_read_b128 of weight
// fragments shared by all four waves, and the previous slab's ~45 VALU epilogue; HBM sees pts in, raw (and the saved
// planes) out.
//
// With one wave per SIMD nothing hides an instruction except the MFMA in flight ahead of it, so the stream is kept
// lean on purpose (each item a measured step): lo halves by v_fma_mix, SGPR-based DMA
// addresses, three opaque LDS base registers so that every DS access is base + immediate, no LDS read whose result
// the next MFMA waits for (accumulator set-up and head weights are requested ahead), no branch and no uncounted store
// in the stream (saved rows are padded to whole workgroup tiles; stores share vmcnt with the weight DMA and are
// counted in the hand-over waits), non-temporal plane stores, accumulators in architectural VGPRs (Makefile).
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"
#include "mlp_internal.h"
#include "mlp_layout.h"
#include "mlp_pack_src.h"
#include "pe_sincos.h"

namespace plnerf_rr {
using namespace plnerf;
using namespace plnerf::lay;

typedef _Float16 H16T;
typedef H16T h16x8 __attribute__((ext_vector_type(8)));
typedef H16T h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define RR_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
// LDS-space views.  The ring (128 KB) and the head block behind it lie beyond the 64 KB a DS instruction's immediate
// offset reaches; three base registers (ring slots 0-1, slots 2-3, head block), opaque to the optimiser, keep every
// access at "base + immediate" -- left alone, the compiler rebuilds an address with a v_add_u32 per read.
typedef __attribute__((address_space(3))) const unsigned char lds_cbyte;
typedef __attribute__((address_space(3))) const h16x8 lds_h16x8;
typedef __attribute__((address_space(3))) const float lds_cfloat;
typedef __attribute__((address_space(3))) const f32x2 lds_f32x2;
typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;

#ifdef RR_TRACE
__device__ unsigned long long g_rr_trace[128];     // [0..3] walk begin / end (shader clock, wall clock); [8 + u] start of unit u
#endif

namespace {

#ifndef RR_PFD
#define RR_PFD 2
#endif
#ifndef RR_SCHED
#define RR_SCHED 1      // 1: pin the order of every k-step product; 2: only of the products that carry an epilogue chunk; 0: free
#endif
#ifndef RR_SP_NUM
#define RR_SP_NUM 1
#define RR_SP_DEN 2
#endif
#ifndef RR_ABLATE
#define RR_ABLATE 0    // timing experiments, tools builds only (results wrong): 1 no weight DMA after the prologue, 2 no
                       // weight-fragment LDS reads, 4 no barrier at the sync points, 8 no epilogue arithmetic, 16 no plane stores (training)
#endif
constexpr int HEAD_FLOATS = HB_END - HB;
constexpr size_t HEAD_BYTES = (size_t)HEAD_FLOATS * 4;
constexpr int H_BIAS = HB_BIAS - HB, H_BF = HB_BF - HB, H_BV = HB_BV - HB, H_WA = HB_WA - HB, H_BA = HB_BA - HB,
              H_WR = HB_WR - HB, H_BR = HB_BR - HB;
// largest magnitude the hi + lo pair of halves represents (65504 + 65504 2^-11 ...): activations are clamped to it
constexpr float H16_MAX_SPLIT = 65504.0f;
constexpr int RR_THREADS = 256;     // four waves, one per SIMD
constexpr int RR_ROWS = 128;        // rows per workgroup and row tile (32 per wave)
static_assert(SV_ROW_PAD % (2 * RR_ROWS) == 0, "saved planes are padded to whole workgroup tiles (up to two row tiles per wave)");
constexpr int SLOT_BYTES = 32768;   // one ring slot = one unit's weight fragments
constexpr int NSLOTS = 4;
constexpr int N_LAYERS = 10;        // L0..L7, feature, view layer (= FwdGemm order)

// ---- the k order of the register-resident operand ------------------------------------------------------------
// A 16-deep k-step holds, in lane half g, element e: hidden feature 16 s + 8 (e >> 2) + 4 g + (e & 3)  -- where the
// accumulators of slab s / 2 leave it.  Encoding channels are assigned so that each lane half evaluates whole
// frequency bands: xyz (63 channels + 1 pad): lane half g, slot t = 8 s + e in 0..31: t < 30: band 5 g + t / 6,
// function/axis t % 6 (sin x, sin y, sin z, cos x, cos y, cos z, the reference's order); t = 30, 31: x, y | z, pad.
// Direction (27 + 5 pad): slot t in 0..15: t < 12: band 2 g + t / 6; then x, y, z, channel 27 | channels 28..31 (the five
// channels the reference's encoding does not have: zero weights there; a caller-embedded input may use all 32).
__host__ __device__ constexpr int rr_pe_channel(int g, int t) {
    return t < 30 ? 3 + 6 * (5 * g + t / 6) + t % 6 : (t == 30 ? (g ? 2 : 0) : (g ? PE_K - 1 : 1));
}
__host__ __device__ constexpr int rr_dpe_channel(int g, int t) {      // a bijection onto 0 .. 31, like rr_pe_channel onto 0 .. 63
    return t < 12 ? 3 + 6 * (2 * g + t / 6) + t % 6 : (g ? 28 + (t - 12) : (t < 15 ? t - 12 : 27));
}
// padded input index (the `k` of mlp_pack_src.h's fwd_src) held by k-step s, lane half g, element e of GEMM gm
__host__ __device__ constexpr int rr_k_index(int gm, int s, int g, int e) {
    const int hidden = 8 * (e >> 2) + 4 * g + (e & 3);
    if (gm == G_L0) return rr_pe_channel(g, 8 * s + e);
    if (gm == G_L5) return s < 4 ? rr_pe_channel(g, 8 * s + e) : PE_K + 16 * (s - 4) + hidden;
    if (gm == G_VIEWS) return s < 16 ? 16 * s + hidden : W + rr_dpe_channel(g, 8 * (s - 16) + e);
    return 16 * s + hidden;
}

// ---- packing: same block order as the ping-pong kernel's forward section -- [32-feature slab][k-step][plane]
// [64 lanes x 8 halves] -- with the k order above.  One slab's k-steps are contiguous: a unit is one DMA run.
template <int NS>
__global__ void rr_pack_kernel(ParamPtrs P, unsigned char* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= FWD_FLOATS / 8) return;
    int g = 0, off = 0;
    while (g < N_FWD - 1 && idx >= off + fwd_K[g] * fwd_N[g] / 8) { off += fwd_K[g] * fwd_N[g] / 8; ++g; }
    const int r = idx - off, KS = fwd_K[g] / 16;
    const int blk = r >> 6, lane = r & 63;
    const int ft = blk / KS, ks = blk - ft * KS;
    const int row = ft * 32 + (lane & 31), gl = lane >> 5;
    h16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = fwd_src(P, g, rr_k_index(g, ks, gl, e), row);
        hi[e] = (H16T)v;
        lo[e] = (H16T)(v - (float)hi[e]);
    }
    h16x8* dst = reinterpret_cast<h16x8*>(out) + (size_t)off * NS + ((size_t)blk * NS) * 64 + lane;
    dst[0] = hi;
    if (NS == 2) dst[64] = lo;
}

// ---- compile-time schedule: the network as a sequence of units -------------------------------------------------
// A unit = slabs [j0, j0 + nj) x k-steps [k0, k0 + nk) of one GEMM: one contiguous run of the packed weights
// (<= one ring slot), nj * nk k-step products.  Split mode (2 KB per k-step): L0 four slabs per unit, K = 16 layers
// one slab, the two long K ranges (skip layer 20, view layer 18 k-steps) two units per slab.  Plain mode (1 KB):
// L0 four slabs, K = 16 layers two slabs, long K ranges one slab.
struct Unit {
    int layer, j0, nj, k0, nk;
    bool first, last;       // first / last unit of its slabs' K range
};
constexpr int layer_ks(int l) { return fwd_K[l] / 16; }
constexpr int layer_slabs(int l) { return fwd_N[l] / 32; }
constexpr int units_per_layer(int ns, int l) {
    const int ks = layer_ks(l), sl = layer_slabs(l);
    if (ks == 4) return 2;
    if (ks == 16) return ns == 2 ? sl : sl / 2;
    return ns == 2 ? 2 * sl : sl;
}
constexpr int n_units(int ns) {
    int n = 0;
    for (int l = 0; l < N_LAYERS; ++l) n += units_per_layer(ns, l);
    return n;
}
constexpr Unit unit_desc(int ns, int u) {
    int l = 0;
    while (u >= units_per_layer(ns, l)) { u -= units_per_layer(ns, l); ++l; }
    const int ks = layer_ks(l);
    if (ks == 4) return Unit{l, u * 4, 4, 0, 4, true, true};
    if (ks == 16) { const int nj = ns == 2 ? 1 : 2; return Unit{l, u * nj, nj, 0, 16, true, true}; }
    if (ns == 1) return Unit{l, u, 1, 0, ks, true, true};
    const int h = ks / 2;
    return Unit{l, u >> 1, 1, (u & 1) * h, (u & 1) ? ks - h : h, (u & 1) == 0, (u & 1) == 1};
}
constexpr int unit_pieces(int ns, const Unit& d) { return d.nj * d.nk * ns; }              // 1 KB DMA pieces
constexpr int unit_elem_off(int ns, const Unit& d) {                                         // halves from the section start
    return (fwd_off(d.layer) + (d.j0 * layer_ks(d.layer) + d.k0) * 512) * ns;
}
// Pending epilogues: the slabs completed by unit u - 1 get their epilogues in chunks (eight per slab) dealt out over
// the first `window` products of unit u after product 0: chunk c right BEFORE the MFMAs of product
// 1 + c * window / nch.  A chunk writes into k-step 2 j + (i >> 1) of the NEXT layer's operand (behind the 4 encoding
// k-steps in the skip layer); unit u must not read that k-step earlier.  The window is the widest that respects this.
// (rt row tiles per wave: the chunks of a slab's rt tiles follow each other, c = (slab q * rt + tile) * 8 + chunk)
constexpr bool epi_window_ok(int ns, int rt, int u, int window) {
    const Unit d = unit_desc(ns, u), p = unit_desc(ns, u - 1);
    const int nch = 8 * p.nj * rt;
    if (d.layer != p.layer + 1) return true;
    for (int c = 0; c < nch; ++c) {
        const int tc = 1 + c * window / nch, q = c / (8 * rt), i = (c & 7) >> 1;
        const int kr = 2 * (p.j0 + q) + (i >> 1) + (d.layer == 5 ? 4 : 0);
        if (kr >= d.k0 && kr < d.k0 + d.nk && (kr - d.k0) * d.nj < tc) return false;
    }
    return true;
}
constexpr int epi_window(int ns, int rt, int u) {
    const Unit d = unit_desc(ns, u);
    for (int wdw = d.nj * d.nk - 1; wdw >= 1; --wdw)
        if (epi_window_ok(ns, rt, u, wdw)) return wdw;
    return 0;
}
// Vector-memory STORES a wave issues while it runs unit u (training): the pending epilogues of unit u - 1's slabs put
// out two plane fragments per slab and row tile, and the layer's relu-bit row once its last slab is done (the feature
// layer, 8, has no relu).  They share the vmcnt counter with the weight DMA, in issue order, so the hand-over wait of
// unit u + 1 has to allow for them on top of the DMA pieces that may still fly.
constexpr int stores_in_unit(int ns, int rt, int u) {
    if (u < 1 || u >= n_units(ns)) return 0;
    const Unit p = unit_desc(ns, u - 1);
    if (!p.last) return 0;
    const bool layer_done = p.j0 + p.nj == layer_slabs(p.layer);
    return 2 * p.nj * rt + ((layer_done && p.layer != 8) ? rt : 0);
}
// row tiles (32 rows each) a wave carries through the network: the split mode's two operand planes leave room for one;
// the plain mode takes two, so that every weight fragment fetched from LDS feeds two MFMAs
constexpr int row_tiles(int ns) { return ns == 2 ? 1 : 2; }
constexpr bool schedule_ok(int ns) {
    for (int u = 0; u < n_units(ns); ++u) {
        const Unit d = unit_desc(ns, u);
        if (unit_pieces(ns, d) * 1024 > SLOT_BYTES || d.nj * d.nk < 8) return false;
        if (u > 0 && unit_desc(ns, u - 1).last && epi_window(ns, row_tiles(ns), u) < 1) return false;
    }
    return true;
}
static_assert(schedule_ok(1) && schedule_ok(2), "unit schedule: a unit fits a slot; pending epilogues precede their readers");
constexpr int INIT_AHEAD = 4;
// The accumulators of unit u + 1 are loaded with their biases INIT_AHEAD products before unit u ends (run_step).  Where unit
// u + 1 reuses an accumulator that unit u - 1's pending epilogue still reads during unit u (L0's four-slab units
// followed by L1's slab 0), the epilogue's last chunk on it must come earlier.
constexpr bool init_hoist_ok(int ns) {
    const int rt = row_tiles(ns);
    for (int u = 1; u + 1 < n_units(ns); ++u) {
        const Unit d = unit_desc(ns, u), p = unit_desc(ns, u - 1), n = unit_desc(ns, u + 1);
        if (!p.last || !n.first) continue;
        const int nch = 8 * p.nj * rt, win = epi_window(ns, rt, u), t_init = d.nj * d.nk - INIT_AHEAD;
        for (int sl = n.j0; sl < n.j0 + n.nj; ++sl) {
            if (sl < p.j0 || sl >= p.j0 + p.nj) continue;
            const int c_last = (sl - p.j0 + 1) * 8 * rt - 1;
            if (1 + c_last * win / nch >= t_init) return false;
        }
        for (int sl = n.j0; sl < n.j0 + n.nj; ++sl)          // (and never the running unit's own accumulators)
            if (sl >= d.j0 && sl < d.j0 + d.nj) return false;
    }
    return true;
}
static_assert(init_hoist_ok(1) && init_hoist_ok(2), "hoisted accumulator set-up overwrites a slab that is still in use");

struct FwdArgs {
    const void* packed;     // head block at the start
    const void* wrr;        // the register-resident section of the packed weights
    const float* pts;
    const float* viewdirs;
    int n_rows, spr;
    float* raw_out;
    void* saved;
    unsigned* status;       // range status word of the packed buffer
    const float* embedded;  // caller-supplied encoding [n_rows][in_ch + view_ch] (EMB kernels), else pts / viewdirs
    int in_ch, view_ch;
};

// ---- LDS-DMA: 64 lanes x 16 B from global memory to LDS at a wave-uniform address, not visible to hipcc's wait
// counting (cdna_hip_programming.md section 5.7): completion = the issuing wave's vmcnt, then a barrier.
// (M0 carries the LDS destination; it is compiler-reserved, never allocated to a value, and nothing else in this
// kernel uses it, so it is written in the statement that needs it and not restored.)
// Address = wave-uniform base (an SGPR pair, advanced by scalar adds) + the lane's 16-byte slot (one VGPR for the whole
// kernel): no vector ALU work per piece.
__device__ __forceinline__ void dma_1k(const void* gsrc_uniform, const unsigned lane_off, const unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(gsrc_uniform), "s"(lds_dst)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

template <int NS, int U>
__device__ __forceinline__ void issue_unit(const FwdArgs& a, const unsigned lds_base, const int wave, const int lane) {
    if constexpr (U < n_units(NS)) {
        constexpr Unit d = unit_desc(NS, U);
        constexpr int NP = unit_pieces(NS, d);
        const unsigned char* src = reinterpret_cast<const unsigned char*>(a.wrr) + (size_t)unit_elem_off(NS, d) * 2;
        const unsigned slot = lds_base + (U % NSLOTS) * SLOT_BYTES;
#pragma unroll
        for (int p = 0; p < (NP + 3) / 4; ++p) {
            const int piece = 4 * p + wave;
            if (4 * p + 3 < NP || piece < NP)        // (the view layer's 18 pieces in plain mode: waves 0, 1 take five)
                dma_1k(src + (size_t)piece * 1024, lane * 16, slot + piece * 1024);
        }
    }
}
// pieces every wave has issued for unit u (a wave may have issued one more: waiting with the smaller count over-waits,
// which is safe)
// piece q (of this wave) of unit U
template <int NS, int U>
__device__ __forceinline__ void issue_piece(const FwdArgs& a, const unsigned lds_base, const int wave, const int lane, const int q) {
    if constexpr (U < n_units(NS)) {
        constexpr Unit d = unit_desc(NS, U);
        constexpr int NP = unit_pieces(NS, d);
        const unsigned char* src = reinterpret_cast<const unsigned char*>(a.wrr) + (size_t)unit_elem_off(NS, d) * 2;
        const unsigned slot = lds_base + (U % NSLOTS) * SLOT_BYTES;
        const int piece = 4 * q + wave;
        if (4 * q + 3 < NP || piece < NP) dma_1k(src + (size_t)piece * 1024, lane * 16, slot + piece * 1024);
    }
}
template <int NS>
constexpr int max_pieces_per_wave(int u) { return u < n_units(NS) ? (unit_pieces(NS, unit_desc(NS, u)) + 3) / 4 : 0; }

template <int NS>
constexpr int pieces_per_wave(int u) { return u < n_units(NS) ? unit_pieces(NS, unit_desc(NS, u)) / 4 : 0; }

// ---- the wave's state ----
template <int NS>
struct Wave {
    h16x8 X[16][NS], Y[16][NS];     // activation operands of two consecutive layers (hi | lo planes)
    h16x8 P[4][NS];                 // the encoding in use: xyz (4 k-steps) or direction (the first 2)
    float px, py, pz, dx, dy, dz;   // the row's position and view direction
    f32x16 acc[8];
    float sig, o0, o1, o2;          // head partials
    f32x2 hw[3];                    // head weights (w_alpha | the three w_rgb rows) of the NEXT epilogue chunk, requested one chunk ahead
    float amax;                     // largest activation magnitude split into halves so far (range check)
    unsigned mw[8];                 // relu bit words of the layer in flight (training)
    h16x8 hvf[1];                   // the view layer's half fragment being assembled (training)
    float ud[16];                   // caller-embedded input only: this lane's direction slots, kept for the view layer
};

// two fp32 values -> one dword (two halves) of the hi plane (+ one of the lo plane) of an operand fragment
template <int NS>
__device__ __forceinline__ void put_split(h16x8 (&dst)[NS], const int e0, const float v0, const float v1) {
    const f32x2 v = {v0, v1};
    const h16x2 h = __builtin_convertvector(v, h16x2);
    u32x4 d0 = __builtin_bit_cast(u32x4, dst[0]);
    d0[e0 >> 1] = __builtin_bit_cast(unsigned, h);
    dst[0] = __builtin_bit_cast(h16x8, d0);
    if constexpr (NS == 2) {
        // lo = f16(v - f32(hi)), one v_fma_mix per value: f32(hi) * -1 + v evaluated in fp32 (exact), rounded once to half
        // -- bit-identical to convert / subtract / convert (tools/probes/fma_mix_split_probe.hip), 2 instructions for 5
        unsigned l;
        const unsigned hb = __builtin_bit_cast(unsigned, h);
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hb), "v"(v0));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hb), "v"(v1));
        u32x4 d1 = __builtin_bit_cast(u32x4, dst[1]);
        d1[e0 >> 1] = l;
        dst[1] = __builtin_bit_cast(h16x8, d1);
    }
}

struct PeSinCos { float s, c; };
__device__ __attribute__((noinline)) PeSinCos pe_sincos_far(const float theta) {
    PeSinCos r;
    sincosf(theta, &r.s, &r.c);
    return r;
}

// NB bands starting at band f0 of (x, y, z) -> 6 NB values in the reference's order (sin xyz, cos xyz per band)
template <int NB>
__device__ __forceinline__ void encode_bands(const float x, const float y, const float z, const int f0, float (&v)[6 * NB]) {
    const float co[3] = {x, y, z};
    const float top = ldexpf(1.0f, f0 + NB - 1);
    if (__builtin_expect(fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z)) * top < PE_FAST_LIMIT, 1)) {
        const PeTurns t[3] = {pe_turns(x), pe_turns(y), pe_turns(z)};
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const float sc = ldexpf(1.0f, f0 + k);
#pragma unroll
            for (int d = 0; d < 3; ++d) pe_sincos(t[d], sc, &v[6 * k + d], &v[6 * k + 3 + d]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const float sc = ldexpf(1.0f, f0 + k);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const PeSinCos r = pe_sincos_far(co[d] * sc);
                v[6 * k + d] = r.s; v[6 * k + 3 + d] = r.c;
            }
        }
    }
}

// the 32 xyz-encoding values / 16 direction-encoding values of lane half g, in slot order (rr_pe_channel / rr_dpe_channel)
__device__ __forceinline__ void xyz_values(const float x, const float y, const float z, const int g, float (&v)[32]) {
    float b[30];
    encode_bands<5>(x, y, z, 5 * g, b);
#pragma unroll
    for (int t = 0; t < 30; ++t) v[t] = b[t];
    v[30] = g ? z : x;
    v[31] = g ? 0.0f : y;
}
__device__ __forceinline__ void dir_values(const float x, const float y, const float z, const int g, float (&u)[16]) {
    float b[12];
    encode_bands<2>(x, y, z, 2 * g, b);
#pragma unroll
    for (int t = 0; t < 12; ++t) u[t] = b[t];
    u[12] = g ? 0.0f : x; u[13] = g ? 0.0f : y; u[14] = g ? 0.0f : z; u[15] = 0.0f;
}

struct Lane {
    int lane, g, wave, trow, grow;      // trow: row within the tile; grow: global row (clamped)
    int prow;                           // global row, not clamped: < sv_rows(n_rows), the padded row count of the saved planes
    bool rowv;                          // the row exists
    int wrow0, n_rows;                  // first global row of the wave; rows in the launch
    const unsigned char* lds;           // the workgroup's LDS
    const float* hd;                    // head block in LDS
    lds_cbyte* ring01;                  // LDS: ring slots 0, 1 + 16 lane   (opaque bases, see lds_cbyte)
    lds_cbyte* ring23;                  //      ring slots 2, 3 + 16 lane
    lds_cfloat* hdg;                    //      head block + 4 g floats
    size_t N;
};

template <int NS>
__device__ __forceinline__ void encode_xyz(Wave<NS>& w, const Lane& ln) {
    float v[32];
    xyz_values(w.px, w.py, w.pz, ln.g, v);
#pragma unroll
    for (int t = 0; t < 32; t += 2) put_split<NS>(w.P[t >> 3], t & 7, v[t], v[t + 1]);
}
template <int NS>
__device__ __forceinline__ void encode_dir(Wave<NS>& w, const Lane& ln) {
    float u[16];
    dir_values(w.dx, w.dy, w.dz, ln.g, u);
#pragma unroll
    for (int t = 0; t < 16; t += 2) put_split<NS>(w.P[t >> 3], t & 7, u[t], u[t + 1]);
}

// caller-embedded input: the 32 + 16 slot values of this lane (rr_pe_channel / rr_dpe_channel order) are loaded once per
// tile (kernel prologue) -- v[] goes into the operand registers P for L0 AND the skip layer (they stay live in between,
// there is nothing to re-evaluate them from), the direction channels wait in w.ud until the view layer.
template <int NS>
__device__ __forceinline__ void split_xyz(Wave<NS>& w, const float (&v)[32]) {
#pragma unroll
    for (int t = 0; t < 32; t += 2) put_split<NS>(w.P[t >> 3], t & 7, v[t], v[t + 1]);
}
template <int NS>
__device__ __forceinline__ void split_dir(Wave<NS>& w, const float (&u)[16]) {
#pragma unroll
    for (int t = 0; t < 16; t += 2) put_split<NS>(w.P[t >> 3], t & 7, u[t], u[t + 1]);
}

// a half plane of the saved state / its relu-bit plane (mlp_layout.h)
__device__ __forceinline__ _Float16* plane_ptr(const FwdArgs& a, const int p, const size_t N) {
    return reinterpret_cast<_Float16*>(a.saved) + (size_t)p * W * N;
}
__device__ __forceinline__ unsigned char* mask_ptr(const FwdArgs& a, const int p, const size_t N) {
    return reinterpret_cast<unsigned char*>(a.saved) + (size_t)SV_FLOATS * 2 * N + (size_t)p * (W / 8) * N;
}

// ---- saved half planes (training): SV_LAYOUT_TILED (mlp_layout.h) -- a fragment leaves as the wave's registers hold
// it, one contiguous KiB per store instruction: no transposition, no LDS, two stores per 32-feature slab ----
__device__ __forceinline__ void store_frag_tiled(const h16x8 f, _Float16* tile, const int width, const int j, const int frag,
                                                 const Lane& ln) {
    // Unconditional: the planes' rows are padded to whole workgroup tiles (SV_ROW_PAD, mlp_layout.h), so every wave of
    // every workgroup has a tile -- rows past n_rows land in padding nobody reads.  A store that may or may not be
    // issued could not be counted in the hand-over waits (stores_in_unit), and its branch sat in the MFMA stream.
    // (non-temporal: the backward reads the planes after the whole launch has written 4.2 GB past them; -1 % on the
    // training forward against plain stores, same box)
    if (!(RR_ABLATE & 16))
        __builtin_nontemporal_store(f, reinterpret_cast<h16x8*>(tile + ((size_t)((j * 2 + frag) * 64 + ln.lane)) * 8));
    (void)width;
}

__device__ __forceinline__ unsigned or_halves(const unsigned w) {      // w | (the other lane half's w)
    const u32x2 s = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return s[0] | s[1];
}
// relu bits of a finished slab, from the slab's two hi-plane operand fragments (every half is f16(relu(x)) >= +0, so
// "x > 0" is "half != 0", the ping-pong kernel's test as well).  Dword q of fragment f holds features
// 16 f + 8 (q >> 1) + 2 (q & 1) + {0, 1} (+ 4 g); min(half, 1) is the bit, pair by pair:
//     sum_q  min_u16x2(d_q, {1, 1}) << (8 (q >> 1) + 2 (q & 1))     low halves at their positions, high halves 16 above
// folded (high part one position up) into 16 bits per fragment, instead of a compare, a select and an OR per value.
__device__ __forceinline__ unsigned relu_half_word(const h16x8 frag) {
    const u32x4 d = __builtin_bit_cast(u32x4, frag);
    const unsigned one = 0x00010001u;
    unsigned acc = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        // (the instruction itself: written as __builtin_elementwise_min on a two-short vector, this compiler's lowering
        // -- compares, a byte permute, a multiply -- returned wrong bits; tools/probes/relu_bits_probe.hip checks both)
        unsigned b;
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(b) : "v"(d[q]), "v"(one));
        acc |= b << (8 * (q >> 1) + 2 * (q & 1));
    }
    return (acc | ((acc >> 16) << 1)) & 0xFFFFu;
}
__device__ __forceinline__ unsigned relu_word(const h16x8 f0, const h16x8 f1, const Lane& ln) {
    const unsigned bits = relu_half_word(f0) | (relu_half_word(f1) << 16);
    return or_halves(bits << (4 * ln.g));
}

// ---- epilogue of slab j of layer L in eight chunks (chunk c = accumulator registers 2 c, 2 c + 1 = features
// 32 j + 8 i + 4 g + r0, + 1 with i = c >> 1, r0 = 2 (c & 1); bias included): ReLU, operand split into one dword per
// plane of the next layer's k-step 2 j + (i >> 1), head partials, saved plane + relu bits.  A chunk is ~10 VALU:
// with one wave per SIMD about five other instructions hide behind each MFMA (MI355X_MICROARCH.md), so the side work
// of a unit is dealt out in pieces this small, evenly over its k-step products. ----
template <int NS, bool SAVE, int L>
__device__ __forceinline__ void epi_chunk(Wave<NS>& w, const int j, const int c, const FwdArgs& a, const Lane& ln) {
    const f32x16& acc = w.acc[j];
    const int i = c >> 1, r0 = 2 * (c & 1);
    float v[2];
#pragma unroll
    for (int r = 0; r < 2; ++r)      // ReLU and the half range in one v_med3_f32 (a value beyond it would become inf)
        v[r] = (L == 8) ? __builtin_amdgcn_fmed3f(acc[4 * i + r0 + r], -H16_MAX_SPLIT, H16_MAX_SPLIT)
                        : __builtin_amdgcn_fmed3f(acc[4 * i + r0 + r], 0.0f, H16_MAX_SPLIT);
    // (the clamp above keeps the halves finite; how far the activations really went is remembered for the status word)
    w.amax = fmaxf(w.amax, fmaxf(fabsf(acc[4 * i + r0]), fabsf(acc[4 * i + r0 + 1])));
    asm volatile("" : "+v"(w.amax));
    const int f0 = 32 * j + 8 * i + 4 * ln.g + r0;
    // head weights: this chunk's arrive in w.hw (asked for by the previous chunk of this row tile -- an LDS read whose
    // result is needed at once would stall the wave, and with it the MFMA issue, for the LDS latency); ask for the next
    const int jn = c < 7 ? j : j + 1, cn = (c + 1) & 7;
    const int f0n = 32 * jn + 8 * (cn >> 1) + 2 * (cn & 1);      // (without the lane half's 4 g: that is in hdg)
    if constexpr (L == 7) {          // sigma = w_alpha . relu(h7)
        if (j == 0 && c == 0) w.hw[0] = *reinterpret_cast<lds_f32x2*>(ln.hdg + H_WA + (f0 - 4 * ln.g));
        const f32x2 wa = w.hw[0];
        if (jn < W / 32) w.hw[0] = *reinterpret_cast<lds_f32x2*>(ln.hdg + H_WA + f0n);
        w.sig = fmaf(v[0], wa[0], w.sig);
        w.sig = fmaf(v[1], wa[1], w.sig);
        asm volatile("" : "+v"(w.sig));      // (keeps the partial sums here: LLVM otherwise sinks the whole chain to the
                                             // kernel's end and holds every operand live until then)
    }
    if constexpr (L == 9) {          // rgb = W_rgb . relu(hv); training: the hv plane and its relu bits
        if (j == 0 && c == 0) {
#pragma unroll
            for (int o = 0; o < 3; ++o) w.hw[o] = *reinterpret_cast<lds_f32x2*>(ln.hdg + H_WR + o * HV + (f0 - 4 * ln.g));
        }
        const f32x2 w0 = w.hw[0], w1 = w.hw[1], w2 = w.hw[2];
        if (jn < HV / 32) {
#pragma unroll
            for (int o = 0; o < 3; ++o) w.hw[o] = *reinterpret_cast<lds_f32x2*>(ln.hdg + H_WR + o * HV + f0n);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            w.o0 = fmaf(v[r], w0[r], w.o0);
            w.o1 = fmaf(v[r], w1[r], w.o1);
            w.o2 = fmaf(v[r], w2[r], w.o2);
        }
        asm volatile("" : "+v"(w.o0), "+v"(w.o1), "+v"(w.o2));
        if constexpr (SAVE) {
            // relu bits at the lane-half-0 positions; the whole word moves by 4 g once, in the last chunk
            unsigned bits = c == 0 ? 0u : w.mw[j];
            bits |= (v[0] > 0.0f ? 1u << (8 * i + r0) : 0u) | (v[1] > 0.0f ? 1u << (8 * i + r0 + 1) : 0u);
            put_split<1>(w.hvf, 4 * (i & 1) + r0, v[0], v[1]);
            if (c % 4 == 3)
                store_frag_tiled(w.hvf[0], plane_ptr(a, 0, ln.N) + (size_t)SV_HV_OFF * ln.N + (size_t)ln.wrow0 * HV, HV, j,
                                 i >> 1, ln);
            w.mw[j] = c == 7 ? or_halves(bits << (4 * ln.g)) : bits;
        }
    }
    if constexpr (L <= 8) {
        h16x8 (&dst)[NS] = ((L % 2 == 0) ? w.X : w.Y)[2 * j + (i >> 1)];
        put_split<NS>(dst, 4 * (i & 1) + r0, v[0], v[1]);
        if constexpr (SAVE) {
            // the hi plane IS the saved half plane (f16(x)).  relu bits: x > 0 (a compare and a select per value; the
            // word is assembled at the lane-half-0 positions and moves by 4 g once, in the last chunk)
            if constexpr (L <= 7) {
                if (c == 7) w.mw[j] = relu_word(((L % 2 == 0) ? w.X : w.Y)[2 * j][0], ((L % 2 == 0) ? w.X : w.Y)[2 * j + 1][0], ln);
            }
            if (c % 4 == 3) store_frag_tiled(dst[0], plane_ptr(a, L, ln.N) + (size_t)ln.wrow0 * W, W, j, i >> 1, ln);
        }
    }
}

// the layer's relu-bit words -> the row's 32 (view layer: 16) bytes of the mask plane
template <int NS, int L>
__device__ __forceinline__ void store_mask_row(const Wave<NS>& w, const FwdArgs& a, const Lane& ln) {
    if constexpr (L <= 7) {
        unsigned char* row = mask_ptr(a, L, ln.N) + (size_t)ln.prow * (W / 8);      // (padded row: always inside the plane)
        // (element-wise selects: a vector-valued ?: is lowered through a scratch array indexed by g)
        u32x4 q;
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = ln.g ? w.mw[4 + e] : w.mw[e];
        *reinterpret_cast<u32x4*>(row + 16 * ln.g) = q;
    } else if constexpr (L == 9) {
        unsigned char* row = mask_ptr(a, 8, ln.N) + (size_t)ln.prow * (HV / 8);
        u32x2 q;
#pragma unroll
        for (int e = 0; e < 2; ++e) q[e] = ln.g ? w.mw[2 + e] : w.mw[e];
        *reinterpret_cast<u32x2*>(row + 8 * ln.g) = q;
    }
}

template <int NS, int L>
__device__ __forceinline__ void init_acc(Wave<NS>& w, const int j, const Lane& ln) {
    constexpr int boff = L < 8 ? H_BIAS + L * W : (L == 8 ? H_BF : H_BV);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 b = *reinterpret_cast<lds_f32x4*>(ln.hdg + boff + 32 * j + 8 * i);
#pragma unroll
        for (int r = 0; r < 4; ++r) w.acc[j][4 * i + r] = b[r];
    }
}

// operand fragment of k-step k of layer L
template <int NS, int L>
__device__ __forceinline__ const h16x8 (&b_frag(const Wave<NS>& w, const int k))[NS] {
    if constexpr (L == 0) return w.P[k];
    else if constexpr (L == 5) { if (k < 4) return w.P[k]; return w.X[k - 4]; }
    else if constexpr (L == 9) { if (k < 16) return w.X[k]; return w.P[k - 16]; }
    else if constexpr (L % 2 == 1) return w.X[k];
    else return w.Y[k];
}

// weight fragment (slab sl, k-step kk of the unit in slot `slot`) for this lane
template <int NS>
__device__ __forceinline__ void load_a(h16x8 (&af)[NS], const Lane& ln, const int slot, const int nk,
                                       const int sl, const int kk) {
    lds_cbyte* base = slot < 2 ? ln.ring01 : ln.ring23;
#pragma unroll
    for (int s = 0; s < NS; ++s)
        af[s] = *reinterpret_cast<lds_h16x8*>(base + (slot & 1) * SLOT_BYTES + ((sl * nk + kk) * NS + s) * 1024);
}

constexpr int PFD = RR_PFD;      // weight fragments requested this many k-step products ahead

// Product T of unit U (every index a compile-time constant: nested loops whose bounds depend on an outer loop's
// counter are not reliably unrolled, and a rolled one would index the register arrays dynamically).
template <int NS, int RT, bool SAVE, int U, int T>
__device__ __forceinline__ void run_step(Wave<NS> (&w)[RT], h16x8 (&af)[PFD + 1][NS], const FwdArgs& a, const Lane (&ln)[RT],
                                         const unsigned lds_base) {
    constexpr Unit d = unit_desc(NS, U);
    constexpr int L = d.layer, NSTEP = d.nj * d.nk, SP = 1;
    constexpr bool HAS_NEXT = U + 1 < n_units(NS);
    constexpr bool PEND = U > 0 && unit_desc(NS, U > 0 ? U - 1 : 0).last;
    constexpr Unit pd = unit_desc(NS, U > 0 ? U - 1 : 0);
    constexpr int NCH = PEND ? 8 * pd.nj * RT : 0;
    constexpr int NPC = max_pieces_per_wave<NS>(U + 3);
    constexpr int kk = T / d.nj, sl = T - kk * d.nj;          // (k-step, slab) order: independent accumulators alternate
    if constexpr (T == SP) {
        // my pieces of unit U + 1 have landed; what was issued after them may fly: unit U + 2's pieces and (training) the
        // plane stores of unit U - 1's run, interleaved with them
        wait_vm<pieces_per_wave<NS>(U + 2) + (SAVE ? stores_in_unit(NS, RT, U - 1) : 0)>();
        if (!(RR_ABLATE & 4)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    {   // request the fragment of product T + PFD (this unit's, or the next unit's first ones)
        constexpr int tn = T + PFD;
        if constexpr (RR_ABLATE & 2) {
        } else if constexpr (tn < NSTEP) {
            constexpr int kn = tn / d.nj, sn = tn - kn * d.nj;
            load_a<NS>(af[tn % (PFD + 1)], ln[0], U % NSLOTS, d.nk, sn, kn);
        } else if constexpr (HAS_NEXT) {
            constexpr Unit nd = unit_desc(NS, HAS_NEXT ? U + 1 : U);
            constexpr int t2 = tn - NSTEP, kn = t2 / nd.nj, sn = t2 - kn * nd.nj;
            load_a<NS>(af[tn % (PFD + 1)], ln[0], (U + 1) % NSLOTS, nd.nk, sn, kn);
        }
    }
    if constexpr (PEND && T >= 1 && !(RR_ABLATE & 8)) {
        // the chunks of the pending epilogues that belong to this product: 1 + c * WIN / NCH == T
        //   <=>  ceil((T - 1) NCH / WIN) <= c < ceil(T NCH / WIN);   chunk c = (slab * RT + row tile) * 8 + piece
        constexpr int WIN = epi_window(NS, RT, U);
        constexpr int c0r = ((T - 1) * NCH + WIN - 1) / WIN, c1r = (T * NCH + WIN - 1) / WIN;
        constexpr int c0 = c0r < NCH ? c0r : NCH, c1 = c1r < NCH ? c1r : NCH;
#pragma unroll
        for (int c = c0; c < c1; ++c) {
            const int rt = (c >> 3) % RT, q = c / (8 * RT);
            epi_chunk<NS, SAVE, pd.layer>(w[rt], pd.j0 + q, c & 7, a, ln[rt]);
            if constexpr (SAVE) {
                if ((c & 7) == 7 && pd.j0 + q + 1 == layer_slabs(pd.layer)) store_mask_row<NS, pd.layer>(w[rt], a, ln[rt]);
            }
        }
    }
    if constexpr (T > SP && !(RR_ABLATE & 1)) {
        constexpr int q0 = (T - SP - 1) * NPC / (NSTEP - SP - 1), q1 = (T - SP) * NPC / (NSTEP - SP - 1);
#pragma unroll
        for (int q = q0; q < q1; ++q) issue_piece<NS, U + 3>(a, lds_base, ln[0].wave, ln[0].lane, q);
    }
    if constexpr (HAS_NEXT && T == NSTEP - INIT_AHEAD) {
        // the next unit's accumulators start from the biases: the LDS reads go out a few products early, so that its first
        // MFMA does not wait for them (the registers are free: slab j + 1's last use was a whole layer ago)
        constexpr Unit nd = unit_desc(NS, HAS_NEXT ? U + 1 : U);
        if constexpr (nd.first) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int q = 0; q < nd.nj; ++q) init_acc<NS, nd.layer>(w[rt], nd.j0 + q, ln[rt]);
        }
    }
    const h16x8 (&aw)[NS] = af[T % (PFD + 1)];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {          // one weight fragment, RT row tiles
        const h16x8 (&bf)[NS] = b_frag<NS, L>(w[rt], d.k0 + kk);
        f32x16& acc = w[rt].acc[d.j0 + sl];
        acc = RR_MFMA(aw[0], bf[0], acc, 0, 0, 0);
        if constexpr (NS == 2) {
            acc = RR_MFMA(aw[NS - 1], bf[0], acc, 0, 0, 0);
            acc = RR_MFMA(aw[0], bf[NS - 1], acc, 0, 0, 0);
        }
    }
    // keep this order: the scheduler otherwise hoists every LDS read of the region to its top and spills
    if constexpr (RR_SCHED == 1) __builtin_amdgcn_sched_barrier(0);
}

template <int NS, int RT, bool SAVE, int U, int... Ts>
__device__ __forceinline__ void run_steps(Wave<NS> (&w)[RT], h16x8 (&af)[PFD + 1][NS], const FwdArgs& a, const Lane (&ln)[RT],
                                          const unsigned lds_base,
                                          std::integer_sequence<int, Ts...>) {
    (run_step<NS, RT, SAVE, U, Ts>(w, af, a, ln, lds_base), ...);
}

// Unit U: its k-step products in (k-step, slab) order, each on the wave's RT row tiles.  The weight fragments of the
// first PFD products arrive in `carry` (requested by the previous unit); the last PFD products request the next unit's.
// The sync point after the first product hands the NEXT unit's slot over: every wave waits for its own DMA pieces of
// unit U + 1, the barrier makes all pieces visible (and proves that every wave has left unit U - 1, whose slot unit
// U + 3 may now overwrite).  Side work dealt out over the products: the pending epilogues (the slabs completed by the
// previous unit, eight chunks per slab and row tile), the DMA pieces of unit U + 3.
template <int NS, int RT, bool SAVE, bool EMB, int U>
__device__ __forceinline__ void run_unit(Wave<NS> (&w)[RT], h16x8 (&carry)[PFD][NS], const FwdArgs& a, const Lane (&ln)[RT],
                                         const unsigned lds_base) {
    constexpr Unit d = unit_desc(NS, U);
    constexpr int L = d.layer, NSTEP = d.nj * d.nk;
#ifdef RR_TRACE
    if (blockIdx.x == (RR_TRACE) && threadIdx.x == 0) g_rr_trace[8 + U] = clock64();
#endif
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        if constexpr (d.first && d.j0 == 0 && L == 5 && !EMB) encode_xyz<NS>(w[rt], ln[rt]);      // (EMB: P still holds it)
        if constexpr (d.first && d.j0 == 0 && L == 9) {
            if constexpr (EMB) split_dir<NS>(w[rt], w[rt].ud);
            else encode_dir<NS>(w[rt], ln[rt]);
        }
        if constexpr (d.first && U == 0) {      // (every later unit's accumulators are set up inside the unit before it)
#pragma unroll
            for (int sl = 0; sl < d.nj; ++sl) init_acc<NS, L>(w[rt], d.j0 + sl, ln[rt]);
        }
    }
    h16x8 af[PFD + 1][NS];
#pragma unroll
    for (int p = 0; p < PFD; ++p)
#pragma unroll
        for (int s = 0; s < NS; ++s) af[p][s] = carry[p][s];
    __builtin_amdgcn_sched_barrier(0);
    run_steps<NS, RT, SAVE, U>(w, af, a, ln, lds_base, std::make_integer_sequence<int, NSTEP>{});
#pragma unroll
    for (int p = 0; p < PFD; ++p)
#pragma unroll
        for (int s = 0; s < NS; ++s) carry[p][s] = af[(NSTEP + p) % (PFD + 1)][s];
}

template <int NS, int RT, bool SAVE, bool EMB, int U>
__device__ __forceinline__ void run_units(Wave<NS> (&w)[RT], h16x8 (&carry)[PFD][NS], const FwdArgs& a, const Lane (&ln)[RT],
                                          const unsigned lds_base) {
    if constexpr (U < n_units(NS)) {
        run_unit<NS, RT, SAVE, EMB, U>(w, carry, a, ln, lds_base);
        run_units<NS, RT, SAVE, EMB, U + 1>(w, carry, a, ln, lds_base);
    }
}

// the head block comes in by DMA like the weights: 16 pieces of 1 KB (four per wave; the block itself is 12,320 bytes,
// what follows it in the packed buffer rides along unused)
constexpr int HEAD_PIECES = 16;
static_assert(HEAD_PIECES * 1024 >= (int)HEAD_BYTES && HEAD_PIECES % 4 == 0, "head block DMA");
constexpr size_t rr_lds_bytes() { return (size_t)NSLOTS * SLOT_BYTES + (size_t)HEAD_PIECES * 1024; }

template <int NS, bool SAVE, bool EMB>
__global__ __launch_bounds__(RR_THREADS) void mlp_fwd_rr_kernel(FwdArgs a) {
    constexpr int RT = row_tiles(NS);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_byte*)smem;
    float* hd = reinterpret_cast<float*>(smem + (size_t)NSLOTS * SLOT_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * (RR_ROWS * RT);
    lds_cbyte* ring01 = (lds_cbyte*)smem + lane * 16;
    lds_cbyte* ring23 = ring01 + 2 * SLOT_BYTES;
    lds_cfloat* hdg = (lds_cfloat*)hd + 4 * (lane >> 5);
    asm volatile("" : "+v"(ring01));
    asm volatile("" : "+v"(ring23));
    asm volatile("" : "+v"(hdg));
    Lane ln[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        ln[rt].lane = lane; ln[rt].g = lane >> 5; ln[rt].wave = wave;
        ln[rt].trow = (wave * RT + rt) * 32 + (lane & 31);
        ln[rt].rowv = row0 + ln[rt].trow < a.n_rows;
        ln[rt].grow = min(row0 + ln[rt].trow, a.n_rows - 1);
        ln[rt].prow = row0 + ln[rt].trow;
        ln[rt].lds = smem; ln[rt].hd = hd; ln[rt].N = sv_rows((size_t)a.n_rows);
        ln[rt].ring01 = ring01; ln[rt].ring23 = ring23; ln[rt].hdg = hdg;
        ln[rt].wrow0 = row0 + (wave * RT + rt) * 32; ln[rt].n_rows = a.n_rows;
    }
#ifdef RR_TRACE
    if (blockIdx.x == (RR_TRACE) && tid == 0) { g_rr_trace[0] = clock64(); g_rr_trace[1] = wall_clock64(); }
#endif
    // The rows' six input floats first, as loads the compiler does not track (it would wait for them with vmcnt(0) and
    // drain the DMA queue behind them); then the head block and the first three units' weights, all by DMA.  One
    // counted wait below covers the inputs: they are older than every DMA piece.
    float in6[RT][6];
    float ev[EMB ? RT : 1][32], eu[EMB ? RT : 1][16];      // caller-embedded input: this lane's slot values
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        if constexpr (EMB) {
            // slot (g, t) <- channel rr_pe_channel(g, t) / rr_dpe_channel(g, t) of the row; channels the network does not
            // have (in_ch < 64, view_ch < 32) are read from the row's first element and zeroed after the wait
            const float* row = a.embedded + (size_t)ln[rt].grow * (size_t)(a.in_ch + a.view_ch);
            const int g = ln[rt].g;
#pragma unroll
            for (int t = 0; t < 32; ++t) {
                const int ch = t < 30 ? 3 + 30 * g + t : (t == 30 ? (g ? 2 : 0) : (g ? PE_K - 1 : 1));      // = rr_pe_channel(g, t)
                asm volatile("global_load_dword %0, %1, off" : "=v"(ev[rt][t]) : "v"(row + (ch < a.in_ch ? ch : 0)) : "memory");
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int ch = t < 12 ? 3 + 12 * g + t : (g ? 28 + (t - 12) : (t < 15 ? t - 12 : 27));        // = rr_dpe_channel(g, t)
                asm volatile("global_load_dword %0, %1, off" : "=v"(eu[rt][t]) : "v"(row + (ch < a.view_ch ? a.in_ch + ch : 0)) : "memory");
            }
        } else {
            const float* pp = a.pts + 3 * (size_t)ln[rt].grow;
            const float* pv = a.viewdirs + 3 * (size_t)(ln[rt].grow / a.spr);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                asm volatile("global_load_dword %0, %1, off" : "=v"(in6[rt][c]) : "v"(pp + c) : "memory");
                asm volatile("global_load_dword %0, %1, off" : "=v"(in6[rt][3 + c]) : "v"(pv + c) : "memory");
            }
        }
    }
#pragma unroll
    for (int p = 0; p < HEAD_PIECES / 4; ++p)
        dma_1k(reinterpret_cast<const unsigned char*>(a.packed) + (size_t)(4 * p + wave) * 1024, lane * 16,
               lds_base + NSLOTS * SLOT_BYTES + (4 * p + wave) * 1024);
    issue_unit<NS, 0>(a, lds_base, wave, lane);
    issue_unit<NS, 1>(a, lds_base, wave, lane);
    issue_unit<NS, 2>(a, lds_base, wave, lane);
    wait_vm<HEAD_PIECES / 4 + pieces_per_wave<NS>(0) + pieces_per_wave<NS>(1) + pieces_per_wave<NS>(2)>();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {      // (nothing reads them above the wait)
        if constexpr (EMB) {
            const int g = ln[rt].g;
#pragma unroll
            for (int t = 0; t < 32; ++t) {
                asm volatile("" : "+v"(ev[rt][t]));
                const int ch = t < 30 ? 3 + 30 * g + t : (t == 30 ? (g ? 2 : 0) : (g ? PE_K - 1 : 1));
                ev[rt][t] = ch < a.in_ch ? ev[rt][t] : 0.0f;
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                asm volatile("" : "+v"(eu[rt][t]));
                const int ch = t < 12 ? 3 + 12 * g + t : (g ? 28 + (t - 12) : (t < 15 ? t - 12 : 27));
                eu[rt][t] = ch < a.view_ch ? eu[rt][t] : 0.0f;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 6; ++c) asm volatile("" : "+v"(in6[rt][c]));
        }
    }
#ifdef RR_TRACE
    if (blockIdx.x == (RR_TRACE) && tid == 0) g_rr_trace[4] = clock64();
#endif
    Wave<NS> w[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        Wave<NS>& wt = w[rt];
        const Lane& lt = ln[rt];
        wt.sig = wt.o0 = wt.o1 = wt.o2 = wt.amax = 0.0f;
        // ---- this lane's row: position and view direction stay in six registers; the encodings (32 of the 64 xyz
        // channels, 16 of the 32 direction channels per lane) are evaluated where they are consumed -- before L0 and
        // again before the skip layer, before the view layer -- rather than held in 48 registers across the network ----
        if constexpr (EMB) {
            split_xyz<NS>(wt, ev[rt]);
#pragma unroll
            for (int t = 0; t < 16; ++t) wt.ud[t] = eu[rt][t];
        } else {
            wt.px = in6[rt][0]; wt.py = in6[rt][1]; wt.pz = in6[rt][2];
            wt.dx = in6[rt][3]; wt.dy = in6[rt][4]; wt.dz = in6[rt][5];
            encode_xyz<NS>(wt, lt);
        }
        if constexpr (SAVE) {
            // saved encoding planes, original channel order: through this wave's corner of ring slot 3 (unused until
            // unit 3's weights arrive, which the sync point of unit 0 -- after every wave's prologue -- requests); the
            // wave's row tiles take turns (LDS operations of one wave execute in order)
            float v[32], u[16];
            if constexpr (EMB) {
#pragma unroll
                for (int t = 0; t < 32; ++t) v[t] = ev[rt][t];
#pragma unroll
                for (int t = 0; t < 16; ++t) u[t] = eu[rt][t];
            } else {
                xyz_values(wt.px, wt.py, wt.pz, lt.g, v);
                dir_values(wt.dx, wt.dy, wt.dz, lt.g, u);
            }
            _Float16* st = reinterpret_cast<_Float16*>(smem + 3 * SLOT_BYTES) + (size_t)wave * 32 * (PE_K + DPE_K);
            _Float16* prow = st + (size_t)(lane & 31) * (PE_K + DPE_K);
#pragma unroll
            for (int t = 0; t < 32; ++t) prow[rr_pe_channel(lt.g, t)] = (_Float16)v[t];
#pragma unroll
            for (int t = 0; t < 16; ++t) prow[PE_K + rr_dpe_channel(lt.g, t)] = (_Float16)u[t];   // (every channel has a slot)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            _Float16* pe_plane = plane_ptr(a, 0, lt.N) + (size_t)SV_PE_OFF * lt.N;
            _Float16* dpe_plane = plane_ptr(a, 0, lt.N) + (size_t)SV_DPE_OFF * lt.N;
#pragma unroll
            for (int q = 0; q < 4; ++q) {            // 32 rows x 8 pieces of 16 B
                const int idx = q * 64 + lane, r = idx >> 3, c = idx & 7;
                *reinterpret_cast<u32x4*>(pe_plane + (size_t)(lt.wrow0 + r) * PE_K + 8 * c) =
                    *reinterpret_cast<const u32x4*>(st + (size_t)r * (PE_K + DPE_K) + 8 * c);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {            // 32 rows x 4 pieces
                const int idx = q * 64 + lane, r = idx >> 2, c = idx & 3;
                *reinterpret_cast<u32x4*>(dpe_plane + (size_t)(lt.wrow0 + r) * DPE_K + 8 * c) =
                    *reinterpret_cast<const u32x4*>(st + (size_t)r * (PE_K + DPE_K) + PE_K + 8 * c);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
#ifdef RR_TRACE
    if (blockIdx.x == (RR_TRACE) && tid == 0) g_rr_trace[5] = clock64();
#endif
    // unit 0's weights and the head block are in place for everyone
    // (training: the prologue's six encoding-plane stores per row tile were issued after all three units' pieces)
    wait_vm<pieces_per_wave<NS>(1) + pieces_per_wave<NS>(2) + (SAVE ? 6 * RT : 0)>();
    __syncthreads();
#ifdef RR_TRACE
    if (blockIdx.x == (RR_TRACE) && tid == 0) g_rr_trace[6] = clock64();
#endif
    h16x8 carry[PFD][NS];
    {
        constexpr Unit d0 = unit_desc(NS, 0);
#pragma unroll
        for (int p = 0; p < PFD; ++p) load_a<NS>(carry[p], ln[0], 0, d0.nk, p % d0.nj, p / d0.nj);
    }
    run_units<NS, RT, SAVE, EMB, 0>(w, carry, a, ln, lds_base);
    // the last unit's slab epilogues (view layer) and the heads
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        constexpr Unit ld = unit_desc(NS, n_units(NS) - 1);
#pragma unroll
        for (int c = 0; c < 8 * ld.nj; ++c) epi_chunk<NS, SAVE, ld.layer>(w[rt], ld.j0 + (c >> 3), c & 7, a, ln[rt]);
        if constexpr (SAVE) store_mask_row<NS, 9>(w[rt], a, ln[rt]);
    }
#ifdef RR_TRACE
    if (blockIdx.x == (RR_TRACE) && tid == 0) { g_rr_trace[2] = clock64(); g_rr_trace[3] = wall_clock64(); }
#endif
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const Wave<NS>& wt = w[rt];
        const Lane& lt = ln[rt];
        const u32x2 s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(wt.sig), __float_as_uint(wt.sig), false, false);
        const u32x2 s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(wt.o0), __float_as_uint(wt.o0), false, false);
        const u32x2 s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(wt.o1), __float_as_uint(wt.o1), false, false);
        const u32x2 s3 = __builtin_amdgcn_permlane32_swap(__float_as_uint(wt.o2), __float_as_uint(wt.o2), false, false);
        if (a.status && !(wt.amax <= H16_MAX)) atomicOr(a.status, PLNERF_RANGE_ACTIVATION);   // (rare; sticky until cleared)
        if (lt.g == 0 && lt.rowv) {
            const float sg = (__uint_as_float(s0[0]) + __uint_as_float(s0[1])) + hd[H_BA];
            const float r = (__uint_as_float(s1[0]) + __uint_as_float(s1[1])) + hd[H_BR + 0];
            const float g = (__uint_as_float(s2[0]) + __uint_as_float(s2[1])) + hd[H_BR + 1];
            const float b = (__uint_as_float(s3[0]) + __uint_as_float(s3[1])) + hd[H_BR + 2];
            reinterpret_cast<float4*>(a.raw_out)[row0 + lt.trow] = make_float4(r, g, b, sg);
        }
    }
}

template <int NS, bool SAVE, bool EMB>
int launch(const FwdArgs& a, hipStream_t st) {
    const size_t lds = rr_lds_bytes();
    constexpr int ROWS = RR_ROWS * row_tiles(NS);
    (void)hipFuncSetAttribute((const void*)mlp_fwd_rr_kernel<NS, SAVE, EMB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    hipLaunchKernelGGL((mlp_fwd_rr_kernel<NS, SAVE, EMB>), dim3((a.n_rows + ROWS - 1) / ROWS), dim3(RR_THREADS), lds, st, a);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

}  // namespace
}  // namespace plnerf_rr

#ifdef RR_TRACE
extern "C" int plnerf_debug_rr_trace(unsigned long long* out8) {
    return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(plnerf_rr::g_rr_trace), 128 * sizeof(unsigned long long));
}
#endif

extern "C" int plnerf_build_flags_rr(void) {
    int f = RR_ABLATE ? 1 : 0;
#ifdef RR_TRACE
    f |= 4;
#endif
    return f;
}

namespace plnerf {
namespace impl {

size_t rr_packed_bytes(int ns) { return (size_t)lay::FWD_FLOATS * ns * 2; }

int rr_pack(const float* const* params, int xyz_ch, int dir_ch, int ns, void* section, hipStream_t st) {
    ParamPtrs P;
    P.xyz_ch = xyz_ch; P.dir_ch = dir_ch;
    for (int i = 0; i < PLNERF_N_PARAM_TENSORS; ++i) P.p[i] = params[i];
    const int groups = lay::FWD_FLOATS / 8, threads = 256, blocks = (groups + threads - 1) / threads;
    if (ns == 1) hipLaunchKernelGGL(plnerf_rr::rr_pack_kernel<1>, dim3(blocks), dim3(threads), 0, st, P, (unsigned char*)section);
    else hipLaunchKernelGGL(plnerf_rr::rr_pack_kernel<2>, dim3(blocks), dim3(threads), 0, st, P, (unsigned char*)section);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

// `embedded` (a caller-supplied encoding, [n_rows][in_ch + view_ch]) is served in split mode only (ns == 2: rr_embedded_ok)
bool rr_embedded_ok(int ns) { return ns == 2; }
int rr_fwd(const void* packed, const void* section, int ns, const float* pts, const float* viewdirs, const float* embedded,
           int in_ch, int view_ch, int n_rows, int samples_per_ray, float* raw_out, void* saved, unsigned* status,
           hipStream_t st) {
    plnerf_rr::FwdArgs a{packed, section, pts, viewdirs, n_rows, samples_per_ray < 1 ? 1 : samples_per_ray, raw_out, saved,
                         status, embedded, in_ch, view_ch};
    if (embedded) {
        if (ns != 2) return PLNERF_EINVAL;
        return saved ? plnerf_rr::launch<2, true, true>(a, st) : plnerf_rr::launch<2, false, true>(a, st);
    }
    if (ns == 1) return saved ? plnerf_rr::launch<1, true, false>(a, st) : plnerf_rr::launch<1, false, false>(a, st);
    return saved ? plnerf_rr::launch<2, true, false>(a, st) : plnerf_rr::launch<2, false, false>(a, st);
}

}  // namespace impl
}  // namespace plnerf
