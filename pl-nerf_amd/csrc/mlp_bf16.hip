// The PL-NeRF MLP on v_mfma_f32_32x32x16_bf16: PLNERF_PREC_BF16 (NS = 1, plain bf16
// operands) and PLNERF_PREC_BF16X3 (NS = 2: every operand is split x = hi + lo with
// hi = bf16(x), lo = bf16(x - hi), and each product is evaluated as hi*hi + lo*hi + hi*lo with
// fp32 accumulation -- ~2^-16 relative operand error, i.e. fp32-class results, at 3 MFMAs per
// product instead of 16 fp32-MFMA passes).
//
// Reference: run_network (run_plnerf.py:78-92) = Embedder (run_nerf_helpers.py:24-54) +
// NeRF.forward (:105-128) and its autograd backward.
//
// Formulation: OUT^T[feature][sample] = W[feature][k] . X^T[k][sample].  The MFMA "A" operand is
// the weight tile (pre-packed in fragment order, streamed from L2 by the one wave that owns that
// 32-feature slab -- no LDS staging, no duplicate fetch), the "B" operand is the activation tile
// in LDS, [sample][feature] row-major bf16, read with one 16-byte ds_read per fragment (row stride
// 528 B == 16 mod 256 -> conflict-free).  With features on the accumulator rows, a lane's four
// consecutive registers are four consecutive features of ONE sample, i.e. four consecutive k of
// the next layer: the bias+ReLU epilogue converts them to bf16 and writes one 8-byte LDS store
// per plane, in place.
//
// A 512-thread workgroup (8 waves, 2 per SIMD) owns a tile of TM samples (128 for bf16, 64 for
// bf16x3; 94 KB of LDS either way) and walks all 12 layers without leaving the CU.  For
// training, each layer's activation tile is additionally copied LDS -> HBM as coalesced fp32
// rows (the same plane layout as the fp32 mode), so the fp32-MFMA weight-gradient stage is shared.
#include <type_traits>

#include "common.h"
#include "mlp_internal.h"
#include "mlp_layout.h"
#include "mlp_pack_src.h"

using namespace plnerf;
using namespace plnerf::lay;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int HEAD_FLOATS = HB_END - HB;            // biases + sigma/rgb rows, fp32, first in the pack
constexpr size_t HEAD_BYTES = (size_t)HEAD_FLOATS * 4;
constexpr int BLDA = 264;   // bf16 elements per activation row (256 + 8): 528 B
constexpr int BLDP = 72;    // xyz-encoding row (64 + 8)
constexpr int BLDD = 40;    // direction-encoding row (32 + 8)
constexpr int NTHREADS = 512;
#ifndef PLNERF_ABLATE
#define PLNERF_ABLATE 0   // timing experiments (results wrong): 1 cheap sincos, 2 no heads, 4 no epilogue,
                          // 16 no LDS operand reads in the K loop, 32 no weight loads in the K loop,
                          // 64 pin load/MFMA phases with sched_barrier, 128 no k rotation (these two keep results right)
#endif

__host__ __device__ constexpr int tile_rows(int ns) { return ns == 1 ? 128 : 64; }

// head-block accessors (same order as the fp32 pack's head block)
__device__ __forceinline__ const float* head(const void* packed) { return (const float*)packed; }
constexpr int H_BIAS = HB_BIAS - HB, H_BF = HB_BF - HB, H_BV = HB_BV - HB, H_WA = HB_WA - HB, H_BA = HB_BA - HB,
              H_WR = HB_WR - HB, H_BR = HB_BR - HB;

// ------------------------------------------------------------------------------------
// packing: [head fp32][fwd GEMMs][dgrad GEMMs]; a GEMM's weight operand is stored as blocks
// (32-feature tile ft, 16-deep k step ks), block order [ft][ks], NS planes per block, each plane
// 64 lanes x 8 bf16 (1 KiB): lane l = (g<<5 | ff) holds W[32 ft + ff][16 ks + 8 g + 0..7].
// ------------------------------------------------------------------------------------
template <int NS>
__global__ void pack_bf16_kernel(ParamPtrs P, unsigned char* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int FWD_GROUPS = FWD_FLOATS / 8;
    constexpr int BWD_GROUPS = (PACKED_FLOATS - BWD) / 8;
    if (idx < HEAD_FLOATS) {
        // head block: identical content to the fp32 pack's
        const int h = idx + HB;
        float v;
        if (h < HB_BF) { const int r = h - HB_BIAS; v = P.p[2 * (r >> 8) + 1][r & 255]; }
        else if (h < HB_BV) v = P.p[P_BF][h - HB_BF];
        else if (h < HB_WA) v = P.p[P_BV][h - HB_BV];
        else if (h < HB_BA) v = P.p[P_WA][h - HB_WA];
        else if (h < HB_WR) v = (h == HB_BA) ? P.p[P_BA][0] : 0.0f;
        else if (h < HB_BR) v = P.p[P_WR][h - HB_WR];
        else v = (h - HB_BR) < 3 ? P.p[P_BR][h - HB_BR] : 0.0f;
        reinterpret_cast<float*>(out)[idx] = v;
    }
    if (idx >= FWD_GROUPS + BWD_GROUPS) return;
    const bool is_fwd = idx < FWD_GROUPS;
    int rem = is_fwd ? idx : idx - FWD_GROUPS;      // group index within the fwd / bwd region
    int g = 0, off = 0;                              // off in groups
    if (is_fwd) {
        while (g < N_FWD - 1 && rem >= off + fwd_K[g] * fwd_N[g] / 8) { off += fwd_K[g] * fwd_N[g] / 8; ++g; }
    } else {
        while (g < N_BWD - 1 && rem >= off + bwd_K[g] * bwd_N[g] / 8) { off += bwd_K[g] * bwd_N[g] / 8; ++g; }
    }
    const int r = rem - off;                         // group within the GEMM: (ft*KS + ks)*64 + lane
    const int KS = (is_fwd ? fwd_K[g] : bwd_K[g]) / 16;
    const int blk = r >> 6, lane = r & 63;
    const int ft = blk / KS, ks = blk - ft * KS;
    const int row = ft * 32 + (lane & 31), k0 = ks * 16 + 8 * (lane >> 5);
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = is_fwd ? fwd_src(P, g, k0 + e, row) : bwd_src(P, g, k0 + e, row);
        hi[e] = (__bf16)v;
        lo[e] = (__bf16)(v - (float)hi[e]);
    }
    // element offset of the GEMM's region: planes are interleaved per block
    const size_t region = (size_t)(is_fwd ? 0 : FWD_FLOATS) * NS + (size_t)off * 8 * NS;
    bf16x8* dst = reinterpret_cast<bf16x8*>(out + HEAD_BYTES) + region / 8 + ((size_t)blk * NS) * 64 + lane;
    dst[0] = hi;
    if (NS == 2) dst[64] = lo;
}

// WAVE-UNIFORM pointer (bf16x8 units) to block (ft, ks = ks0) of a GEMM; lanes add their own index, so
// every weight load is "scalar base + one shared lane offset" and needs no per-load address registers
template <int NS>
__device__ __forceinline__ const bf16x8* wblock(const void* packed, bool fwd, int elem_off, int KS, int ft, int ks0) {
    const bf16x8* base = reinterpret_cast<const bf16x8*>((const unsigned char*)packed + HEAD_BYTES);
    const size_t region = ((size_t)(fwd ? 0 : FWD_FLOATS) + (size_t)elem_off) * NS / 8;
    return base + region + ((size_t)(ft * KS + ks0) * NS) * 64;
}

// ------------------------------------------------------------------------------------
// tile GEMM: acc[t] (32 features x 32 samples) += W[ft rows][k range] . X^T[k range][sample tile t]
//   a      : weight blocks of this wave's feature tile, starting at the first k step (lane included)
//   b_lane : &X_plane0[(lane & 31) * ld + 8 * (lane >> 5)] at the first k of the range
// Software pipeline, KC k-steps fully unrolled (all indices static):
//   weight fragments (global, L2-resident) are fetched PF steps ahead into a PF+1 slot ring,
//   activation fragments (LDS) one step ahead into a 2-slot ring,
// The source order is "issue next loads, then this step's MFMAs"; hipcc schedules around it and
// inserts the counted vmcnt/lgkmcnt waits.  (Pinning the two phases with sched_barrier(0) measured
// 24 % SLOWER in bf16 and neutral in bf16x3, so it is off; PLNERF_ABLATE=64 re-enables it.)
// ------------------------------------------------------------------------------------
// Deferred write-back of the PREVIOUS layer's tile, interleaved into a 16-step K loop: the tile
// (bf16 planes in LDS, also being read as this GEMM's B operand) is copied to its fp32 plane in HBM
// (+ ReLU bit mask) a chunk at a time -- LDS reads in step 4c+1, convert + global stores in step
// 4c+2 -- so the HBM write stream overlaps the MFMAs instead of following them.  Chunk = 8 features
// of one sample; a 512-thread workgroup moves TM*32 chunks, i.e. TM/16 per thread = CPS per slot.
struct NoSide {
    static constexpr bool ACTIVE = false;
    __device__ __forceinline__ void load(int) {}
    __device__ __forceinline__ void store(int) {}
};

template <int NS, int TMROWS>
struct PlaneCopy {
    static constexpr bool ACTIVE = true;
    static constexpr int CPS = TMROWS / 64;          // chunks per thread per slot (4 slots per K loop)
    const __bf16* tile;                               // plane 0 of the LDS tile
    int ld, plane_stride;
    float* plane;                                     // fp32 [rows][256] destination (nullptr = nothing pending)
    unsigned char* mask;                              // [rows][32] or nullptr
    int row0, rows_valid, tid;
    bf16x8 h[CPS], l[CPS];
    __device__ __forceinline__ void load(int slot) {
        if (!plane) return;
#pragma unroll
        for (int j = 0; j < CPS; ++j) {
            const int idx = tid + NTHREADS * (slot * CPS + j);
            const int s = idx >> 5, c = idx & 31;
            const __bf16* src = tile + (size_t)s * ld + c * 8;
            h[j] = *reinterpret_cast<const bf16x8*>(src);
            if (NS == 2) l[j] = *reinterpret_cast<const bf16x8*>(src + plane_stride);
        }
    }
    __device__ __forceinline__ void store(int slot) {
        if (!plane) return;
#pragma unroll
        for (int j = 0; j < CPS; ++j) {
            const int idx = tid + NTHREADS * (slot * CPS + j);
            const int s = idx >> 5, c = idx & 31;
            if (s >= rows_valid) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = (float)h[j][e];
                if (NS == 2) v[e] += (float)l[j][e];
            }
            float4* dst = reinterpret_cast<float4*>(plane + (size_t)(row0 + s) * W + c * 8);
            dst[0] = make_float4(v[0], v[1], v[2], v[3]);
            dst[1] = make_float4(v[4], v[5], v[6], v[7]);
            if (mask) {
                unsigned m = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) m |= (v[e] > 0.0f ? 1u : 0u) << e;
                mask[(size_t)(row0 + s) * 32 + c] = (unsigned char)m;
            }
        }
    }
};

#ifndef PLNERF_WPF
#define PLNERF_WPF 4
#endif
constexpr int WPF = PLNERF_WPF;   // weight prefetch distance (k-steps)

template <int NS>
struct WQueue {
    bf16x8 q[WPF + 1][NS];   // one ring, reused by every GEMM of the tile walk
};

// request the first min(KC, WPF) weight blocks of a GEMM.
// `rot` rotates the order in which the k-steps of a GEMM are visited (step i handles k-step
// (i + rot) mod KC).  Every workgroup walks the same layers at the same pace, so without it all
// CUs of an XCD ask the L2 for the same few KB of weights at the same moment and only the one or
// two L2 channels that own those lines are busy; rotating by workgroup spreads the simultaneous
// requests over the whole layer (all channels).  Only the fp32 summation order changes.
template <int NS, int KC>
__device__ __forceinline__ void wq_prime(WQueue<NS>& w, const bf16x8* __restrict__ a, const int rot, const int lane) {
    constexpr int PF = KC < WPF ? KC : WPF;
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const int ks = (p + rot) & (KC - 1);
#pragma unroll
        for (int s = 0; s < NS; ++s) w.q[p][s] = a[(ks * NS + s) * 64 + lane];
    }
}

template <int NS, int NT, int KC, class Side>
__device__ __forceinline__ void mma_bf16(f32x16 (&acc)[NT], WQueue<NS>& w, const bf16x8* __restrict__ a,
                                         const __bf16* b_lane, const int ld, const int plane_stride,
                                         const int rot, const int lane, Side& side) {
    static_assert((KC & (KC - 1)) == 0, "KC must be a power of two");
    constexpr int PF = KC < WPF ? KC : WPF;
    bf16x8 bq[2][NT][NS];
    {
        const __bf16* b0 = b_lane + (rot & (KC - 1)) * 16;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < NS; ++s)
                bq[0][t][s] = *reinterpret_cast<const bf16x8*>(b0 + s * plane_stride + (size_t)t * 32 * ld);
    }
#pragma unroll
    for (int i = 0; i < KC; ++i) {
        // keep the rotated addresses from being hoisted out of the (outer) layer loop as 2*KC
        // loop-invariant address registers: recompute them per step from an opaque copy of rot
        int rot_i = rot;
        asm volatile("" : "+s"(rot_i));
        if (i + PF < KC && !(PLNERF_ABLATE & 32)) {
            const int ks = (i + PF + rot_i) & (KC - 1);
#pragma unroll
            for (int s = 0; s < NS; ++s) w.q[(i + PF) % (PF + 1)][s] = a[(ks * NS + s) * 64 + lane];
        }
        if (i + 1 < KC && !(PLNERF_ABLATE & 16)) {
            const __bf16* b1 = b_lane + ((i + 1 + rot_i) & (KC - 1)) * 16;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    bq[(i + 1) & 1][t][s] =
                        *reinterpret_cast<const bf16x8*>(b1 + s * plane_stride + (size_t)t * 32 * ld);
        }
        if (Side::ACTIVE && KC == 16) {
            if ((i & 3) == 2) side.store(i >> 2);     // data requested one step ago has landed
            if ((i & 3) == 1) side.load(i >> 2);
        }
        if (PLNERF_ABLATE & 64) __builtin_amdgcn_sched_barrier(0);
        const bf16x8 ah = w.q[i % (PF + 1)][0];
        const bf16x8 al = w.q[i % (PF + 1)][NS - 1];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[i & 1][t][0], acc[t], 0, 0, 0);
            if (NS == 2) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bq[i & 1][t][0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[i & 1][t][NS - 1], acc[t], 0, 0, 0);
            }
        }
        if (PLNERF_ABLATE & 64) __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
}

// accumulators start at the bias of their feature rows (rows = features in this formulation), so
// the epilogue is only relu + convert + store
template <int NT>
__device__ __forceinline__ void init_acc(f32x16 (&acc)[NT], const float* __restrict__ bias, const int f_base,
                                         const int lane) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + f_base + 8 * rg + 4 * (lane >> 5));
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][4 * rg + j] = b[j];
    }
}

template <int NS>
__device__ __forceinline__ void split4(const float (&v)[4], __bf16* p0, const int plane_stride) {
    bf16x4 hi;
#pragma unroll
    for (int j = 0; j < 4; ++j) hi[j] = (__bf16)v[j];
    *reinterpret_cast<bf16x4*>(p0) = hi;
    if (NS == 2) {
        bf16x4 lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) lo[j] = (__bf16)(v[j] - (float)hi[j]);
        *reinterpret_cast<bf16x4*>(p0 + plane_stride) = lo;
    }
}

// the value the GEMMs see for one stored element: hi (+ lo)
template <int NS>
__device__ __forceinline__ void load8(const __bf16* p0, const int plane_stride, float (&v)[8]) {
    const bf16x8 hi = *reinterpret_cast<const bf16x8*>(p0);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)hi[e];
    if (NS == 2) {
        const bf16x8 lo = *reinterpret_cast<const bf16x8*>(p0 + plane_stride);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)lo[e];
    }
}

// coalesced copy of an LDS tile [rows][WIDTH] (bf16 planes) to an fp32 plane in HBM
template <int NS, int WIDTH>
__device__ __forceinline__ void tile_to_plane(const __bf16* tile, const int ld, const int plane_stride,
                                              float* __restrict__ plane, const int row0, const int rows_valid,
                                              const int rows, const int tid,
                                              unsigned char* __restrict__ mask_plane = nullptr) {
    constexpr int CPR = WIDTH / 8;   // 8-element chunks per row
    for (int idx = tid; idx < rows * CPR; idx += NTHREADS) {
        const int s = idx / CPR, c = idx - s * CPR;
        if (s >= rows_valid) continue;
        float v[8];
        load8<NS>(tile + (size_t)s * ld + c * 8, plane_stride, v);
        float4* dst = reinterpret_cast<float4*>(plane + (size_t)(row0 + s) * WIDTH + c * 8);
        dst[0] = make_float4(v[0], v[1], v[2], v[3]);
        dst[1] = make_float4(v[4], v[5], v[6], v[7]);
        if (mask_plane) {   // relu'(h) as one bit per activation, for the dgrad kernel
            unsigned m = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) m |= (v[e] > 0.0f ? 1u : 0u) << e;
            mask_plane[(size_t)(row0 + s) * CPR + c] = (unsigned char)m;
        }
    }
}

// ------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------
struct FwdArgs {
    const void* packed;
    const float* pts;
    const float* viewdirs;
    const float* embedded;
    int n_rows, spr;
    float* raw_out;
    float* saved;
};

// accumulators (features x samples) -> + bias -> (relu) -> bf16 planes in LDS, in place
template <int NS, int NT, bool RELU>
__device__ __forceinline__ void store_act(f32x16 (&acc)[NT], __bf16* act, const int plane_stride, const int f_base,
                                          const int s_base, const int lane) {
    if (PLNERF_ABLATE & 4) {   // keep the accumulators alive, skip the conversion + LDS stores
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" ::"v"(acc[t]));
#endif
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int s = s_base + t * 32 + (lane & 31);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int f0 = f_base + 8 * rg + 4 * (lane >> 5);
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = acc[t][4 * rg + j];
                if (RELU) v[j] = v[j] > 0.0f ? v[j] : 0.0f;
            }
            split4<NS>(v, act + (size_t)s * BLDA + f0, plane_stride);
        }
    }
}

template <int NS, bool SAVE>
__global__ __launch_bounds__(NTHREADS) void mlp_fwd_bf16_kernel(FwdArgs a) {
    constexpr int TM = tile_rows(NS), NT = TM / 32, TPR = NTHREADS / TM;
    constexpr int ACT_PLANE = TM * BLDA, PE_PLANE = TM * BLDP, DPE_PLANE = TM * BLDD;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* act = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* pe = act + NS * ACT_PLANE;
    __bf16* dpe = pe + NS * PE_PLANE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
    const int row0 = blockIdx.x * TM;
    const int rows_valid = min(TM, a.n_rows - row0);
    const size_t N = (size_t)a.n_rows;
    const float* hd = head(a.packed);
    const int rot = (PLNERF_ABLATE & 128) ? 0 : (int)(blockIdx.x >> 3);   // consecutive blocks of one XCD (b, b+8, ...) get consecutive rotations

    // ---- prologue: encodings -> bf16 planes ---------------------------------------------
    if (a.embedded) {
        for (int e = tid; e < TM * EMB_CH; e += NTHREADS) {
            const int row = e / EMB_CH, c = e - row * EMB_CH;
            const int grow = min(row0 + row, a.n_rows - 1);
            const float v = a.embedded[(size_t)grow * EMB_CH + c];
            __bf16* dst = c < XYZ_CH ? pe + row * BLDP + c : dpe + row * BLDD + (c - XYZ_CH);
            const int ps = c < XYZ_CH ? PE_PLANE : DPE_PLANE;
            const __bf16 h = (__bf16)v;
            dst[0] = h;
            if (NS == 2) dst[ps] = (__bf16)(v - (float)h);
        }
        for (int row = tid; row < TM; row += NTHREADS) {
            for (int s = 0; s < NS; ++s) {
                pe[s * PE_PLANE + row * BLDP + XYZ_CH] = (__bf16)0.0f;
                for (int c = DIR_CH; c < DPE_K; ++c) dpe[s * DPE_PLANE + row * BLDD + c] = (__bf16)0.0f;
            }
        }
    } else {
        auto put = [&](__bf16* dst, int ps, float v) {
            const __bf16 h = (__bf16)v;
            dst[0] = h;
            if (NS == 2) dst[ps] = (__bf16)(v - (float)h);
        };
#if PLNERF_ABLATE & 1
#define sincosf(x, s, c) (*(s) = __sinf(x), *(c) = __cosf(x))
#endif
        for (int e = tid; e < TM * 4; e += NTHREADS) {
            const int row = e % TM, q = e / TM;
            const int grow = min(row0 + row, a.n_rows - 1);
            const float px = a.pts[3 * (size_t)grow + 0], py = a.pts[3 * (size_t)grow + 1],
                        pz = a.pts[3 * (size_t)grow + 2];
            __bf16* prow = pe + row * BLDP;
            for (int f = q; f < XYZ_FREQS; f += 4) {
                const float sc = (float)(1 << f);
                float s, c;
                sincosf(px * sc, &s, &c); put(prow + 3 + 6 * f + 0, PE_PLANE, s); put(prow + 3 + 6 * f + 3, PE_PLANE, c);
                sincosf(py * sc, &s, &c); put(prow + 3 + 6 * f + 1, PE_PLANE, s); put(prow + 3 + 6 * f + 4, PE_PLANE, c);
                sincosf(pz * sc, &s, &c); put(prow + 3 + 6 * f + 2, PE_PLANE, s); put(prow + 3 + 6 * f + 5, PE_PLANE, c);
            }
            if (q == 0) {
                put(prow + 0, PE_PLANE, px); put(prow + 1, PE_PLANE, py); put(prow + 2, PE_PLANE, pz);
                put(prow + XYZ_CH, PE_PLANE, 0.0f);
            }
            const int ray = grow / a.spr;
            const float dx = a.viewdirs[3 * (size_t)ray + 0], dy = a.viewdirs[3 * (size_t)ray + 1],
                        dz = a.viewdirs[3 * (size_t)ray + 2];
            __bf16* drow = dpe + row * BLDD;
            {
                const int f = q;
                const float sc = (float)(1 << f);
                float s, c;
                sincosf(dx * sc, &s, &c); put(drow + 3 + 6 * f + 0, DPE_PLANE, s); put(drow + 3 + 6 * f + 3, DPE_PLANE, c);
                sincosf(dy * sc, &s, &c); put(drow + 3 + 6 * f + 1, DPE_PLANE, s); put(drow + 3 + 6 * f + 4, DPE_PLANE, c);
                sincosf(dz * sc, &s, &c); put(drow + 3 + 6 * f + 2, DPE_PLANE, s); put(drow + 3 + 6 * f + 5, DPE_PLANE, c);
            }
            if (q == 1) { put(drow + 0, DPE_PLANE, dx); put(drow + 1, DPE_PLANE, dy); put(drow + 2, DPE_PLANE, dz); }
            if (q == 2) {
                for (int c = DIR_CH; c < DPE_K; ++c) put(drow + c, DPE_PLANE, 0.0f);
            }
        }
    }
    WQueue<NS> wq;
    const bf16x8* wp0 = wblock<NS>(a.packed, true, fwd_off(G_L0), 4, wave, 0);
    wq_prime<NS, 4>(wq, wp0, rot, lane);
    __syncthreads();
    if (SAVE) {
        tile_to_plane<NS, PE_K>(pe, BLDP, PE_PLANE, a.saved + (size_t)SV_PE_OFF * N, row0, rows_valid, TM, tid);
        tile_to_plane<NS, DPE_K>(dpe, BLDD, DPE_PLANE, a.saved + (size_t)SV_DPE_OFF * N, row0, rows_valid, TM, tid);
    }

    const int b_off = (lane & 31), b_k = 8 * (lane >> 5);
    const __bf16* act_lane = act + b_off * BLDA + b_k;
    const __bf16* pe_lane = pe + b_off * BLDP + b_k;
    const __bf16* dpe_lane = dpe + b_off * BLDD + b_k;
    f32x16 acc[NT];
#define PLANE(p) (a.saved + (size_t)(p) * W * N)
#define MASKP(p) (reinterpret_cast<unsigned char*>(a.saved + (size_t)SV_FLOATS * N) + (size_t)(p) * (W / 8) * N)
#define WB(g, KS, ks0) wblock<NS>(a.packed, true, fwd_off(g), KS, wave, ks0)

    // Weight fragments of the NEXT GEMM are requested before the barrier + epilogue of the current
    // one (they do not depend on activations), so each K loop starts with its first blocks in flight.
    const bf16x8* wp;
    // training: layer l's tile is written back to HBM during layer l+1's K loop (PlaneCopy)
    NoSide noside;
    typename std::conditional<SAVE, PlaneCopy<NS, TM>, NoSide>::type side;
    auto pend = [&](float* plane, unsigned char* mask) {
        if constexpr (SAVE) { side.plane = plane; side.mask = mask; }
    };
    if constexpr (SAVE) {
        side.tile = act; side.ld = BLDA; side.plane_stride = ACT_PLANE; side.plane = nullptr; side.mask = nullptr;
        side.row0 = row0; side.rows_valid = rows_valid; side.tid = tid;
    }
    // L0..L7 and the feature layer as ONE loop body (l = 8 is the feature layer): an optional
    // 4-step K range over the xyz encoding (L0, L5) followed by an optional 16-step range over the
    // activation tile.  One copy of the code keeps registers and I-cache small.
    const int hrow = tid / TPR, hq = tid % TPR;
    float sigma = 0.0f;
    constexpr int NTV = NT / 2;
    const int vft = wave & 3, vs0 = (wave >> 2) * (TM / 2);
    wp = wp0;
#pragma unroll 1
    for (int l = 0; l <= 8; ++l) {
        const bool has_pe = (l == 0) || (l == 5);
        const bool has_act = l > 0;
        init_acc(acc, l < 8 ? hd + H_BIAS + l * W : hd + H_BF, wave * 32, lane);
        if (has_pe) {
            mma_bf16<NS, NT, 4>(acc, wq, wp, pe_lane, BLDP, PE_PLANE, rot, lane, noside);
            if (has_act) {
                wp = WB(G_L5, 20, 4);
                wq_prime<NS, 16>(wq, wp, rot, lane);
            }
        }
        if (has_act) mma_bf16<NS, NT, 16>(acc, wq, wp, act_lane, BLDA, ACT_PLANE, rot, lane, side);
        // request the next GEMM's first weight blocks before the barrier + epilogue of this one
        if (l == 4) {
            wp = WB(G_L5, 20, 0);
            wq_prime<NS, 4>(wq, wp, rot, lane);
        } else if (l == 8) {
            wp = wblock<NS>(a.packed, true, fwd_off(G_VIEWS), 18, vft, 0);
            wq_prime<NS, 16>(wq, wp, rot, lane);
        } else {
            // packed order: L0 | L1..L4 | L5 | L6 L7 FEAT ; next = l + 1
            const int nxt = l + 1;
            const int woff = nxt <= 4 ? fwd_off(G_L1) + (nxt - 1) * W * W : fwd_off(G_L6) + (nxt - 6) * W * W;
            wp = wblock<NS>(a.packed, true, woff, 16, wave, 0);
            wq_prime<NS, 16>(wq, wp, rot, lane);
        }
        __syncthreads();
        if (l < 8) store_act<NS, NT, true>(acc, act, ACT_PLANE, wave * 32, 0, lane);
        else store_act<NS, NT, false>(acc, act, ACT_PLANE, wave * 32, 0, lane);
        __syncthreads();
        if (SAVE) pend(PLANE(l), l < 8 ? MASKP(l) : nullptr);
        if (l == 7) {
            // sigma head (reads h7): TPR threads per sample, 8-feature chunks interleaved across them
            const float* wa = hd + H_WA;
#pragma unroll 1
            for (int c = hq; c < ((PLNERF_ABLATE & 2) ? 0 : W / 8); c += TPR) {
                float v[8];
                load8<NS>(act + (size_t)hrow * BLDA + c * 8, ACT_PLANE, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) sigma = fmaf(v[e], wa[c * 8 + e], sigma);
            }
#pragma unroll
            for (int d = 1; d < TPR; d <<= 1) sigma += __shfl_xor(sigma, d);
            sigma += hd[H_BA];
        }
    }
    // view layer: 4 feature tiles x 2 sample halves over the 8 waves
    {
        f32x16 accv[NTV];
        init_acc(accv, hd + H_BV, vft * 32, lane);
        mma_bf16<NS, NTV, 16>(accv, wq, wp, act_lane + (size_t)vs0 * BLDA, BLDA, ACT_PLANE, rot, lane, side);
        wp = wblock<NS>(a.packed, true, fwd_off(G_VIEWS), 18, vft, 16);
        wq_prime<NS, 2>(wq, wp, rot, lane);
        mma_bf16<NS, NTV, 2>(accv, wq, wp, dpe_lane + (size_t)vs0 * BLDD, BLDD, DPE_PLANE, rot, lane, noside);
        __syncthreads();
        store_act<NS, NTV, true>(accv, act, ACT_PLANE, vft * 32, vs0, lane);
        __syncthreads();
        if (SAVE) tile_to_plane<NS, HV>(act, BLDA, ACT_PLANE, a.saved + (size_t)SV_HV_OFF * N, row0, rows_valid, TM, tid, MASKP(8));
    }
    // rgb head
    {
        const float* wr = hd + H_WR;
        float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
#pragma unroll 1
        for (int c = hq; c < ((PLNERF_ABLATE & 2) ? 0 : HV / 8); c += TPR) {
            float v[8];
            load8<NS>(act + (size_t)hrow * BLDA + c * 8, ACT_PLANE, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o0 = fmaf(v[e], wr[0 * HV + c * 8 + e], o0);
                o1 = fmaf(v[e], wr[1 * HV + c * 8 + e], o1);
                o2 = fmaf(v[e], wr[2 * HV + c * 8 + e], o2);
            }
        }
#pragma unroll
        for (int d = 1; d < TPR; d <<= 1) {
            o0 += __shfl_xor(o0, d);
            o1 += __shfl_xor(o1, d);
            o2 += __shfl_xor(o2, d);
        }
        if (hq == 0 && hrow < rows_valid) {
            float4 o;
            o.x = o0 + hd[H_BR + 0];
            o.y = o1 + hd[H_BR + 1];
            o.z = o2 + hd[H_BR + 2];
            o.w = sigma;
            reinterpret_cast<float4*>(a.raw_out)[row0 + hrow] = o;
        }
    }
#undef PLANE
#undef MASKP
#undef WB
}

// ------------------------------------------------------------------------------------
// backward: dgrad chain (writes the fp32 dz planes the shared fp32 wgrad stage consumes)
// ------------------------------------------------------------------------------------
struct BwdArgs {
    const void* packed;
    const float* g_raw;
    int n_rows;
    const float* saved;
    float* dz;
};

// relu mask of a saved activation plane -> one bit per (sample, feature) in LDS
__device__ __forceinline__ void build_mask(unsigned char* maskb, const unsigned char* __restrict__ mask_plane,
                                           const int row0, const int rows_valid, const int rows, const int tid) {
    // the tile's rows are contiguous in the mask plane: a straight 32 B/row copy, 4 bytes per thread
    const unsigned* src = reinterpret_cast<const unsigned*>(mask_plane + (size_t)row0 * 32);
    unsigned* dst = reinterpret_cast<unsigned*>(maskb);
    for (int idx = tid; idx < rows * 8; idx += NTHREADS) dst[idx] = (idx >> 3) < rows_valid ? src[idx] : 0u;
}

template <int NS, int NT, bool MASK, bool ALPHA>
__device__ __forceinline__ void store_dz(f32x16 (&acc)[NT], __bf16* g, const int plane_stride, const float* gr,
                                         const float* __restrict__ wa, const unsigned char* maskb, const int f_base,
                                         const int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int s = t * 32 + (lane & 31);
        const float ga = ALPHA ? gr[s * 4 + 3] : 0.0f;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int f0 = f_base + 8 * rg + 4 * (lane >> 5);
            float v[4];
            unsigned m = 0xF;
            if (MASK) m = (maskb[s * 32 + (f0 >> 3)] >> (f0 & 7)) & 0xF;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = acc[t][4 * rg + j];
                if (ALPHA) v[j] = fmaf(ga, wa[f0 + j], v[j]);
                if (MASK && !((m >> j) & 1)) v[j] = 0.0f;
            }
            split4<NS>(v, g + (size_t)s * BLDA + f0, plane_stride);
        }
    }
}

// bf16x3: 128 VGPRs and 69 KB of LDS let two workgroups share a CU, so one's HBM phase (dz plane
// write-back, mask fetch) overlaps the other's MFMA phase; bf16's 128-row tile needs the full file.
template <int NS>
__global__ __launch_bounds__(NTHREADS, NS == 2 ? 4 : 2) void mlp_bwd_bf16_kernel(BwdArgs a) {
    constexpr int TM = tile_rows(NS), NT = TM / 32;
    constexpr int G_PLANE = TM * BLDA;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* g = reinterpret_cast<__bf16*>(smem_raw);
    float* gr = reinterpret_cast<float*>(g + NS * G_PLANE);
    unsigned char* maskb = reinterpret_cast<unsigned char*>(gr + TM * 4);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
    const int row0 = blockIdx.x * TM;
    const int rows_valid = min(TM, a.n_rows - row0);
    const size_t N = (size_t)a.n_rows;
    const float* hd = head(a.packed);
    const int rot = (PLNERF_ABLATE & 128) ? 0 : (int)(blockIdx.x >> 3);   // consecutive blocks of one XCD (b, b+8, ...) get consecutive rotations
#define SPLANE(p) (a.saved + (size_t)(p) * W * N)
#define DPLANE(p) (a.dz + (size_t)(p) * W * N)
#define MASKP(p) (reinterpret_cast<const unsigned char*>(a.saved + (size_t)SV_FLOATS * N) + (size_t)(p) * (W / 8) * N)

    for (int row = tid; row < TM; row += NTHREADS) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < rows_valid) v = reinterpret_cast<const float4*>(a.g_raw)[row0 + row];
        reinterpret_cast<float4*>(gr)[row] = v;
    }
    __syncthreads();
    // dz_view = (g_rgb W_rgb) * relu'(hv): 8 features per thread-iteration, coalesced planes
    {
        const float* wr = hd + H_WR;
        const unsigned char* hv_mask = MASKP(8);
        float* dzv_plane = a.dz + (size_t)DZ_V_OFF * N;
        for (int idx = tid; idx < TM * (HV / 8); idx += NTHREADS) {
            const int s = idx >> 4, c = idx & 15;
            const bool ok = s < rows_valid;
            const unsigned hm = ok ? hv_mask[(size_t)(row0 + s) * (HV / 8) + c] : 0u;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int i = c * 8 + e;
                float t = gr[s * 4 + 0] * wr[i];
                t = fmaf(gr[s * 4 + 1], wr[HV + i], t);
                t = fmaf(gr[s * 4 + 2], wr[2 * HV + i], t);
                v[e] = ((hm >> e) & 1u) ? t : 0.0f;
            }
            float lo4[4] = {v[0], v[1], v[2], v[3]}, hi4[4] = {v[4], v[5], v[6], v[7]};
            split4<NS>(lo4, g + (size_t)s * BLDA + c * 8, G_PLANE);
            split4<NS>(hi4, g + (size_t)s * BLDA + c * 8 + 4, G_PLANE);
            if (ok) {
                float4* dst = reinterpret_cast<float4*>(dzv_plane + (size_t)(row0 + s) * HV + c * 8);
                dst[0] = make_float4(v[0], v[1], v[2], v[3]);
                dst[1] = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
    }
    __syncthreads();
    const __bf16* g_lane = g + (lane & 31) * BLDA + 8 * (lane >> 5);
    f32x16 acc[NT];
    // d feature = dz_view . W_view[:, :256]  (K = 128)
    WQueue<NS> wq;
    // (interleaving the dz write-back into the next K loop like the forward does costs ~16 VGPRs,
    // which at the 128-register / 2-workgroups-per-CU operating point of bf16x3 turns into spills and
    // measured 15 % slower; the second resident workgroup already overlaps the write-back)
    NoSide noside;
    const bf16x8* wp = wblock<NS>(a.packed, false, bwd_off(D_VIEWS) - BWD, 8, wave, 0);
    wq_prime<NS, 8>(wq, wp, rot, lane);
    zero_acc(acc);
    mma_bf16<NS, NT, 8>(acc, wq, wp, g_lane, BLDA, G_PLANE, rot, lane, noside);
    wp = wblock<NS>(a.packed, false, bwd_off(D_FEAT) - BWD, 16, wave, 0);
    wq_prime<NS, 16>(wq, wp, rot, lane);
    __syncthreads();
    store_dz<NS, NT, false, false>(acc, g, G_PLANE, gr, nullptr, nullptr, wave * 32, lane);
    __syncthreads();
    tile_to_plane<NS, W>(g, BLDA, G_PLANE, DPLANE(DZ_FEAT), row0, rows_valid, TM, tid);
    build_mask(maskb, MASKP(7), row0, rows_valid, TM, tid);
    // d h7 = dz_feature . W_f + g_sigma w_alpha, masked by h7
    zero_acc(acc);
    mma_bf16<NS, NT, 16>(acc, wq, wp, g_lane, BLDA, G_PLANE, rot, lane, noside);
    wp = wblock<NS>(a.packed, false, bwd_off(D_L7) - BWD, 16, wave, 0);
    wq_prime<NS, 16>(wq, wp, rot, lane);
    __syncthreads();
    store_dz<NS, NT, true, true>(acc, g, G_PLANE, gr, hd + H_WA, maskb, wave * 32, lane);
    __syncthreads();
    tile_to_plane<NS, W>(g, BLDA, G_PLANE, DPLANE(7), row0, rows_valid, TM, tid);
#pragma unroll 1
    for (int l = 7; l >= 1; --l) {
        build_mask(maskb, MASKP(l - 1), row0, rows_valid, TM, tid);
        zero_acc(acc);
        mma_bf16<NS, NT, 16>(acc, wq, wp, g_lane, BLDA, G_PLANE, rot, lane, noside);
        if (l > 1) {
            wp = wblock<NS>(a.packed, false, bwd_off(D_L7) - BWD + (8 - l) * W * W, 16, wave, 0);
            wq_prime<NS, 16>(wq, wp, rot, lane);
        }
        __syncthreads();
        store_dz<NS, NT, true, false>(acc, g, G_PLANE, gr, nullptr, maskb, wave * 32, lane);
        __syncthreads();
        tile_to_plane<NS, W>(g, BLDA, G_PLANE, DPLANE(l - 1), row0, rows_valid, TM, tid);
    }
#undef SPLANE
#undef DPLANE
#undef MASKP
}

template <int NS>
size_t lds_fwd() { return (size_t)NS * tile_rows(NS) * (BLDA + BLDP + BLDD) * 2; }
template <int NS>
size_t lds_bwd() { return (size_t)NS * tile_rows(NS) * BLDA * 2 + (size_t)tile_rows(NS) * 16 + (size_t)tile_rows(NS) * 32; }

template <int NS>
int fwd_launch(const FwdArgs& a, hipStream_t st) {
    constexpr int TM = tile_rows(NS);
    const size_t lds = lds_fwd<NS>();
    dim3 grid((a.n_rows + TM - 1) / TM), block(NTHREADS);
    if (a.saved) {
        (void)hipFuncSetAttribute((const void*)mlp_fwd_bf16_kernel<NS, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((mlp_fwd_bf16_kernel<NS, true>), grid, block, lds, st, a);
    } else {
        (void)hipFuncSetAttribute((const void*)mlp_fwd_bf16_kernel<NS, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((mlp_fwd_bf16_kernel<NS, false>), grid, block, lds, st, a);
    }
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

template <int NS>
int bwd_launch(const BwdArgs& a, hipStream_t st) {
    constexpr int TM = tile_rows(NS);
    const size_t lds = lds_bwd<NS>();
    (void)hipFuncSetAttribute((const void*)mlp_bwd_bf16_kernel<NS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    hipLaunchKernelGGL((mlp_bwd_bf16_kernel<NS>), dim3((a.n_rows + TM - 1) / TM), dim3(NTHREADS), lds, st, a);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

}  // namespace

namespace plnerf {
namespace impl {

size_t bf16_packed_bytes(int ns) {
    return HEAD_BYTES + (size_t)(FWD_FLOATS + (PACKED_FLOATS - BWD)) * ns * 2;
}

int bf16_pack(const float* const* params, int ns, void* packed, hipStream_t st) {
    ParamPtrs P;
    for (int i = 0; i < PLNERF_N_PARAM_TENSORS; ++i) P.p[i] = params[i];
    const int groups = (FWD_FLOATS + (PACKED_FLOATS - BWD)) / 8;
    const int threads = 256, blocks = (groups + threads - 1) / threads;
    if (ns == 1) hipLaunchKernelGGL(pack_bf16_kernel<1>, dim3(blocks), dim3(threads), 0, st, P, (unsigned char*)packed);
    else hipLaunchKernelGGL(pack_bf16_kernel<2>, dim3(blocks), dim3(threads), 0, st, P, (unsigned char*)packed);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

int bf16_fwd(const void* packed, int ns, const float* pts, const float* viewdirs, const float* embedded, int n_rows,
             int samples_per_ray, float* raw_out, void* saved, hipStream_t st) {
    FwdArgs a{packed, pts, viewdirs, embedded, n_rows, samples_per_ray < 1 ? 1 : samples_per_ray, raw_out,
              (float*)saved};
    return ns == 1 ? fwd_launch<1>(a, st) : fwd_launch<2>(a, st);
}

int bf16_dgrad(const void* packed, int ns, const float* g_raw, int n_rows, const float* saved, float* dz,
               hipStream_t st) {
    BwdArgs a{packed, g_raw, n_rows, saved, dz};
    return ns == 1 ? bwd_launch<1>(a, st) : bwd_launch<2>(a, st);
}

}  // namespace impl
}  // namespace plnerf
