// The weight-gradient stage of the PL-NeRF MLP's backward, every precision mode:
//
//   wgrad_f32_kernel      exact fp32 (v_mfma_f32_32x32x2_f32) straight from the fp32 planes
//   wgrad_main_kernel     the 256-wide jobs of the 16-bit modes on v_mfma_f32_32x32x16_f16 over the HALF planes of
//   wgrad_thin_kernel     mlp_layout.h (LDS-staged, ds_read_b64_tr_b16); the encoding / direction columns two stages deep
//                         (tiled saved planes: the view layer's direction columns ride on the idle half of the main
//                         launch's view job instead -- X2 in wgrad_half_body -- and the thin launch has two jobs)
//   wgrad_head_kernel     sigma / rgb head rows (VALU reductions)
//   wgrad_reduce_kernel   deterministic split-K sum into the 24 gradient tensors (+ the launch scale of the half dz
//                         planes divided back out, + the range status for a data-parallel exchange)
//   absmax_kernel, absmax_act_kernel   max |g_raw| = the half dz planes' launch scale (and the density activation's
//                         derivative on the backward's entry)
//
// Reference: what autograd derives for run_nerf_helpers.py:105-128 at loss.backward() (run_plnerf.py:1300):
// dW_l = dz_l^T a_{l-1}, db_l = column sums of dz_l.  (Round 4: moved out of mlp_f32.hip, where the 16-bit modes' dominant
// backward kernel lived in a file named for another mode.)
#include "common.h"
#include "mlp_frag.h"
#include "mlp_internal.h"
#include "mlp_layout.h"
#include "mlp_pack_src.h"

using namespace plnerf;
using namespace plnerf::lay;

// row ranges of the 256-wide weight-gradient jobs.  Half planes: 8 tiles (L1..L7 and the composed view layer, mlp_layout.h;
// 9 tiles x 28 = 252 until round 5, when feature_linear had a job of its own) x 32 = 256 workgroups, ONE round on the
// 256 CUs (each keeps its 256 x 256 partial in registers for twice as many rows as with 56, and the partial sums
// written and re-read by the reduction halve: 134 -> 67 MB per network; step -1.5 %).  The fp32 kernel (17 tiles of
// 256 threads, several workgroups per CU) wants the 56 it was tuned with (28: +8 % step).
#ifndef PLNERF_WG_SPLITS
#define PLNERF_WG_SPLITS 32
#endif
#ifndef PLNERF_WG_SPLITS_F32
#define PLNERF_WG_SPLITS_F32 56
#endif

namespace {

// ------------------------------------------------------------------------------------
// backward: weight gradients, split-K TN GEMM  C[o][i] = sum_m A[m][o] B[m][i]
// ------------------------------------------------------------------------------------
struct WJob {
    const void* A;    // dz plane (fp32 or half elements), row stride lda elements
    const void* B;    // activation plane, row stride ldb elements
    int lda, ldb, O, I;
    int part_off;     // offset (floats) of this job's [O][I] partial inside a split block
    int bias_off;     // offset of the [O] bias partial, or -1
    int b_tiled;      // half planes: B is in SV_LAYOUT_TILED (mlp_layout.h) instead of row-major
    const void* B2;   // view job of the tiled layout only (else nullptr): the direction-encoding plane (32 wide, tiled) --
    int part2_off;    //   dz_view^T dpe rides on the job's four idle waves instead of re-reading dz_view in a thin job
    const float* g_raw;       // the same job (round 6): the upstream gradient [n_rows][4] and the launch scale's word -- the job's B operand
    const unsigned* gmax;     //   IS h7, so the idle waves also sum alpha_linear.weight's gradient g_sigma^T h7 (PART_SIGMA): h7 is read once per step
};
constexpr int MAX_WTILES = 20;
struct WgradArgs {
    WJob jobs[10];
    int tile_job[MAX_WTILES];
    int tile_o0[MAX_WTILES];
    int n_rows, rows_per_split;
    float* part;      // [splits][PART_PER_SPLIT]
};
// The half kernels take up to two networks per grid (round 5: plnerf_mlp_bwd_multi): row ranges [0, splits0) of the
// grid's y extent belong to n[0], the rest to n[1] -- the ONE round of workgroups is dealt out over the networks in
// proportion to their rows, so that a training step's coarse and fine networks share a launch instead of each paying its
// ramp and its tail (profiles/r05_merged_bwd_bound.txt).
struct WgradArgs2 {
    WgradArgs n[2];
    int splits0;
};

// Waves are arranged WO x WI over the workgroup tile; each owns NO x NI 32x32 MFMA tiles.
// KS k-steps (2 rows each) are loaded per iteration, one iteration ahead of the MFMAs.
template <int WO, int WI, int NO, int NI, int KS>
__global__ __launch_bounds__(256) void wgrad_f32_kernel(WgradArgs a) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const WJob job = a.jobs[a.tile_job[blockIdx.x]];
    const int wo = wave / WI, wi = wave % WI;
    const int o_base = a.tile_o0[blockIdx.x] + wo * NO * 32;
    const int i_base = wi * NI * 32;
    const int split = blockIdx.y;
    const int m_begin = split * a.rows_per_split;
    const int m_end = min(a.n_rows, m_begin + a.rows_per_split);
    const int kh = lane >> 5, ll = lane & 31;
    const float* Ap = (const float*)job.A + o_base + ll;
    const float* Bp = (const float*)job.B + i_base + ll;
    f32x16 acc[NO][NI];
    zero_acc(acc);
    float bsum[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) bsum[o] = 0.0f;
    float ac[KS][NO], bc[KS][NI], an[KS][NO], bn[KS][NI];

    auto load = [&](float (&av)[KS][NO], float (&bv)[KS][NI], int m) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int mm = m + 2 * s + kh;
            const bool ok = mm < m_end;
#pragma unroll
            for (int o = 0; o < NO; ++o) av[s][o] = ok ? Ap[(size_t)mm * job.lda + o * 32] : 0.0f;
#pragma unroll
            for (int i = 0; i < NI; ++i) bv[s][i] = ok ? Bp[(size_t)mm * job.ldb + i * 32] : 0.0f;
        }
    };
    if (m_begin < m_end) load(ac, bc, m_begin);
    for (int m = m_begin; m < m_end; m += 2 * KS) {
        if (m + 2 * KS < m_end) load(an, bn, m + 2 * KS);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                bsum[o] += ac[s][o];
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    acc[o][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[s][o], bc[s][i], acc[o][i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int o = 0; o < NO; ++o) ac[s][o] = an[s][o];
#pragma unroll
            for (int i = 0; i < NI; ++i) bc[s][i] = bn[s][i];
        }
    }
    float* part = a.part + (size_t)split * PART_PER_SPLIT;
    float* cpart = part + job.part_off;
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int orow = o_base + o * 32 + frag_row(r, lane);
                const int icol = i_base + i * 32 + ll;
                cpart[(size_t)orow * job.I + icol] = acc[o][i][r];
            }
    if (job.bias_off >= 0 && wi == 0) {
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            const float b = bsum[o] + __shfl_xor(bsum[o], 32);
            if (kh == 0) part[job.bias_off + o_base + o * 32 + ll] = b;
        }
    }
}

// The same TN GEMM for the 16-bit MFMA modes, on v_mfma_f32_32x32x16_f16 over the HALF planes of
// mlp_layout.h (both operands are stored in half by the training forward / the dgrad kernel, so one
// MFMA per product; the dz planes carry a power-of-two scale that the reduction divides out).
// A workgroup loads each 16-row slab of dz / activations ONCE with coalesced 16-byte loads and stores
// it row-major in LDS; waves then build their k-contiguous MFMA fragments with ds_read_b64_tr_b16,
// the gfx950 transposing LDS read (lane = feature column, 4 consecutive rows per read; row stride
// 576 B puts the 4 rows of a read on disjoint banks; semantics pinned by tools/probes/tr16_probe.hip).
// A stage is TR_STEPS k-steps (64 rows): stage k+1 is fetched into registers before stage k's MFMAs and
// written to LDS after them; one barrier per stage.  8 waves as 4(o) x 2(i), each 64(o) x 32 NI(i): a 256 x 256
// workgroup tile reads every plane once.  (History: a register-only version re-split fp32 operands in every wave
// and was VALU-bound, 24 % MFMA busy -- profiles/r01_bf16x3_pmc_sq_tcp_before_lds_wgrad.txt.)
typedef short v4s16 __attribute__((ext_vector_type(4)));
typedef _Float16 wh8 __attribute__((ext_vector_type(8)));
#ifndef PLNERF_TR_STEPS
#define PLNERF_TR_STEPS 4
#endif
constexpr int TR_STEPS = PLNERF_TR_STEPS;   // MFMA k-steps per LDS stage
constexpr int TR_ROWS = 16;            // rows per MFMA k-step
constexpr int TR_RS = 288;             // LDS row stride in half elements (256 + 32): 576 B
constexpr int TR_PLANE = TR_ROWS * TR_RS;

__device__ __forceinline__ wh8 tr_frag(const _Float16* plane, int lane, int col0) {
    // 8 consecutive rows (k) of column col0 + (lane & 31), rows 8 * (lane >> 5) .. +7
    const int lam = lane & 15, gam = lane >> 4;
    const int row = 8 * (gam >> 1) + (lam >> 2), col = col0 + 16 * (gam & 1) + 4 * (lam & 3);
    const _Float16* p = plane + row * TR_RS + col;
    typedef __attribute__((address_space(3))) v4s16* lds_v4;
    const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)p);
    const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p + 4 * TR_RS));
    union { v4s16 h[2]; wh8 v; } u;
    u.h[0] = lo;
    u.h[1] = hi;
    return u.v;
}

// The same fragment from a B stage held in the TILED order (mlp_layout.h: SV_LAYOUT_TILED): the stage's LDS image is
// [piece block pb = (slab, fragment, lane half)][row of the stage][8 halves] with TRB_PAD halves between blocks (the
// two lane halves a 16-lane read group touches then sit on different banks); 4 consecutive features of a row are
// contiguous there too, which is all ds_read_b64_tr_b16 asks for.
constexpr int TRB_PAD = 32;
__device__ __forceinline__ wh8 tr_frag_tiled(const _Float16* stage, int srows, int k, int lane, int col0) {
    const int lam = lane & 15, gam = lane >> 4;
    const int row = TR_ROWS * k + 8 * (gam >> 1) + (lam >> 2);
    const int pb = ((col0 >> 5) * 2 + (gam & 1)) * 2 + (lam & 1);          // col = col0 + 16 (gam & 1) + 4 (lam & 3)
    const _Float16* p = stage + (size_t)pb * (srows * 8 + TRB_PAD) + row * 8 + 4 * ((lam >> 1) & 1);
    typedef __attribute__((address_space(3))) v4s16* lds_v4;
    const v4s16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)p);
    const v4s16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p + 4 * 8));
    union { v4s16 h[2]; wh8 v; } u;
    u.h[0] = lo;
    u.h[1] = hi;
    return u.v;
}

// One body for every job shape.  The thin jobs (256 x 64 encoding columns of L0 / L5, 128 x 32 direction columns
// of the view layer) run it DEEP: a stage costs its HBM latency whatever its width, and a thin job's stage is only 20-40 KB per CU, so
// ONE stage ahead left the launch at 3.9 TB/s.  Here two stages are in flight, in two register sets -- which
// only works if the compiler can COUNT the outstanding loads: gfx9 retires loads in order on one vmcnt, and a load
// issued under a run-time predicate (slot in range? row in range?) makes the count unknown, so every wait becomes
// vmcnt(0) and the younger stage's latency is back on the critical path (a first attempt inside the generic
// kernel measured +-0 for exactly that reason).  O and I are therefore template parameters (slot counts static),
// rows past the range are loaded from the clamped last row and zeroed by a select.
template <int O, int I, int NI, bool DEEP, bool TILED = false, bool X2 = false>
__device__ __forceinline__ void wgrad_half_body(const WgradArgs& a, const WJob& job, const int split, unsigned char* smem_raw) {
    // (TILED B: the 256-wide activation planes, and since round 5 the 64- / 32-wide encoding planes of the register-resident
    // forward, whose columns are in ITS order -- the reduction kernel un-permutes them, sv_enc_channel)
    constexpr int B_CT = 4 * I;                             // 16-byte chunks per 32-row tile of a tiled B plane
    // the A operand: a half dz plane -- TILED too when it is 256 wide (written by the dgrad kernel's MFMA epilogues,
    // mlp_layout.h); dz_view (O = 128) is row-major
    constexpr bool ATILED = O == W;
    constexpr int NO = 2, WI = 2, KST = TR_STEPS, SROWS = TR_ROWS * KST, SPLANE = SROWS * TR_RS;
    static_assert(32 * (SROWS * 8 + TRB_PAD) <= SPLANE, "a tiled stage fits the operand's LDS plane");
    constexpr int A_C8 = O / 8, B_C8 = I / 8;
    constexpr int SA = SROWS * A_C8 / 512;                  // A chunks per thread and stage: 4 (O = 256) or 2
    constexpr int B_CHUNKS = SROWS * B_C8;                  // B chunks per stage: 2048 (I = 256), 512 (64), 256 (32)
    constexpr int SB = B_CHUNKS >= 512 ? B_CHUNKS / 512 : 1;
    constexpr int B_THREADS = B_CHUNKS >= 512 ? 512 : B_CHUNKS;   // threads that own a B chunk
    static_assert(SROWS * A_C8 % 512 == 0 && B_CHUNKS % B_THREADS == 0 && 512 % B_THREADS == 0, "slot layout");
    static_assert(I == NI * 32 * WI || I == 32, "two waves across I");
    _Float16* lds = reinterpret_cast<_Float16*>(smem_raw);   // [buffer 2][operand A,B][SROWS][TR_RS] (+ X2: [buffer 2][4 piece blocks])
    // X2 (the view job, O = 128, of a tiled backward; round 5): its waves 4..7 own no output tile (o_base >= O) -- they
    // multiply the SAME dz_view stage with a second, 32-wide B operand, the direction-encoding plane: the view layer's
    // direction columns, for which a thin job used to read dz_view (256 B/row) a second time.
    static_assert(!X2 || (O == HV && I == W && TILED && !DEEP), "the second B operand rides on the view job's idle waves");
    constexpr int I2 = DPE_K, B2_CT = 4 * I2, B2_CHUNKS = SROWS * I2 / 8, B2_BLOCK = SROWS * 8 + TRB_PAD;
    _Float16* lds2 = lds + (size_t)4 * SPLANE;
    // ... and (round 6) the sigma head: a third "A" operand of which only column 0 lives, g_sigma under the launch scale -- the
    // stage's LDS image is [row][4 halves] = {g_sigma, 0, 0, 0} (+ 8 zero bytes every lane whose fragment columns are not 0..3 reads)
    constexpr int G_STAGE = SROWS * 4 + 4;                  // halves per buffer (the zero quad at the end)
    _Float16* lds3 = lds2 + (size_t)2 * 4 * B2_BLOCK;
    float g_scale = 1.0f;
    if constexpr (X2) {
        const float gm = __uint_as_float(*job.gmax);       // (written by the gradient chain's launch, or the absmax pass, before this one)
        if (gm > 0.0f && gm < __builtin_inff()) {
            int e;
            (void)frexpf(gm, &e);
            g_scale = ldexpf(1.0f, (int)DZH_TARGET_EXP - e);
        }
        g_scale = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(g_scale)));      // (uniform: a scalar register)
    }
    const _Float16* B2g = (const _Float16*)job.B2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int q2 = tid % B2_CHUNKS, b2_row = (q2 / B2_CT) * 32 + (q2 & 31), b2_col = (q2 % B2_CT) >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const _Float16* Ag = (const _Float16*)job.A;
    const _Float16* Bg = (const _Float16*)job.B;
    const int wo = wave / WI, wi = wave % WI;
    const int o_base = wo * NO * 32, i_base = wi * NI * 32;
    const bool live = o_base < O && i_base < I;
    const int m_begin = split * a.rows_per_split;
    const int m_end = min(a.n_rows, m_begin + a.rows_per_split);
    int a_row[SA], a_col[SA];        // ATILED: row of the stage | piece block of the chunk (as b_row / b_col below)
#pragma unroll
    for (int j = 0; j < SA; ++j) {
        const int q = tid + 512 * j;
        if (ATILED) { a_row[j] = (q >> 10) * 32 + (q & 31); a_col[j] = (q & 1023) >> 5; }
        else { a_row[j] = q / A_C8; a_col[j] = (q % A_C8) * 8; }
    }
    int b_row[SB], b_col[SB];        // TILED: row of the stage | piece block of the chunk
#pragma unroll
    for (int j = 0; j < SB; ++j) {
        const int q = tid % B_THREADS + 512 * j;           // (threads past B_THREADS reload a neighbour's chunk, unused)
        if (TILED) {                                       // chunk q of the stage = tile q / B_CT, piece q % B_CT
            b_row[j] = (q / B_CT) * 32 + (q & 31); b_col[j] = (q % B_CT) >> 5;
        } else {
            b_row[j] = q / B_C8; b_col[j] = (q % B_C8) * 8;
        }
    }
    const int last_tile = (a.n_rows + 31) / 32 - 1;      // the last tile that holds a real row
    const bool b_owner = tid < B_THREADS;
    struct Set { wh8 a[SA]; wh8 b[SB]; wh8 b2; float g; };
    Set s0, s1;      // (s1 only in the DEEP variant)
    // bias partial = column sums of the dz stage.  Row-major A: a thread's chunks share one chunk column.  Tiled A: its
    // chunks alternate between two piece blocks (q and q + 512: blocks pb and pb + 16), one set of sums each.
    float bs[ATILED ? 2 : 1][8];
#pragma unroll
    for (int h = 0; h < (ATILED ? 2 : 1); ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[h][e] = 0.0f;
    const wh8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // (the zeroing select lives in stash, not here: a consumer right behind the load would put the new stage's
    // latency back in front of this stage's MFMAs)
    // (plain loads: non-temporal ones -- the Gate A probe's consumers reached 6.7 TB/s with them -- measured +0.7 % on the step
    // here, backward 2.94 -> 2.98 ms, profiles/r06_sigma_fold_ab.txt's lib:wgNT legs)
#define WG_LD(p) (*(p))
    auto fetch = [&](Set& s, const int m) {
        const int last = m_end - 1;
#pragma unroll
        for (int j = 0; j < SA; ++j) {
            if (ATILED)     // (m is a multiple of 32; a tile past the plane re-reads the last one and is zeroed in stash)
                s.a[j] = WG_LD(reinterpret_cast<const wh8*>(
                    Ag + ((size_t)min((m >> 5) + (a_row[j] >> 5), last_tile) * 1024 + a_col[j] * 32 + (a_row[j] & 31)) * 8));
            else
                s.a[j] = WG_LD(reinterpret_cast<const wh8*>(Ag + (size_t)min(m + a_row[j], last) * O + a_col[j]));
        }
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            if (TILED)      // (m is a multiple of 64: a stage is two whole tiles; a tile past the plane re-reads the last one)
                s.b[j] = WG_LD(reinterpret_cast<const wh8*>(
                    Bg + ((size_t)min((m >> 5) + (b_row[j] >> 5), last_tile) * B_CT + b_col[j] * 32 + (b_row[j] & 31)) * 8));
            else
                s.b[j] = WG_LD(reinterpret_cast<const wh8*>(Bg + (size_t)min(m + b_row[j], last) * I + b_col[j]));
        }
        if constexpr (X2) {    // (every thread, like B: threads past the chunk count reload a neighbour's -- static load counts)
            s.b2 = WG_LD(reinterpret_cast<const wh8*>(
                B2g + ((size_t)min((m >> 5) + (b2_row >> 5), last_tile) * B2_CT + b2_col * 32 + (b2_row & 31)) * 8));
            s.g = job.g_raw[4 * (size_t)min(m + (tid & (SROWS - 1)), last) + 3];
        }
    };
    auto stash = [&](const Set& s, const int buf, const int m) {
        _Float16* A0 = lds + (size_t)buf * 2 * SPLANE;
        _Float16* B0 = A0 + SPLANE;
#pragma unroll
        for (int j = 0; j < SA; ++j) {
            const wh8 v = m + a_row[j] < m_end ? s.a[j] : zero8;
            _Float16* dst = ATILED ? A0 + (size_t)a_col[j] * (SROWS * 8 + TRB_PAD) + a_row[j] * 8 : A0 + a_row[j] * TR_RS + a_col[j];
            *reinterpret_cast<wh8*>(dst) = v;
#pragma unroll
            for (int e = 0; e < 8; ++e) bs[ATILED ? (j & 1) : 0][e] += (float)v[e];
        }
        if (b_owner) {
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                _Float16* dst = TILED ? B0 + (size_t)b_col[j] * (SROWS * 8 + TRB_PAD) + b_row[j] * 8
                                      : B0 + b_row[j] * TR_RS + b_col[j];
                *reinterpret_cast<wh8*>(dst) = m + b_row[j] < m_end ? s.b[j] : zero8;
            }
        }
        if constexpr (X2) {
            if (tid < B2_CHUNKS)
                *reinterpret_cast<wh8*>(lds2 + (size_t)(buf * 4 + b2_col) * B2_BLOCK + b2_row * 8) = m + b2_row < m_end ? s.b2 : zero8;
            if (tid < SROWS) {
                typedef _Float16 wh4 __attribute__((ext_vector_type(4)));
                const float gs = m + tid < m_end ? __builtin_amdgcn_fmed3f(s.g * g_scale, -H16_MAX, H16_MAX) : 0.0f;
                const wh4 q = {(_Float16)gs, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
                *reinterpret_cast<wh4*>(lds3 + (size_t)buf * G_STAGE + tid * 4) = q;
            }
        }
    };
    f32x16 acc[NO][NI];
    zero_acc(acc);
    auto compute = [&](const int buf) {
        if (!live) {
            if constexpr (X2) {      // waves 4..7: output tile (wave - 4) of dz_view^T dpe, in acc[0][0] (theirs is otherwise unused)
                const _Float16* A0 = lds + (size_t)buf * 2 * SPLANE;
                const _Float16* B0 = A0 + SPLANE;
                const _Float16* B2s = lds2 + (size_t)buf * 4 * B2_BLOCK;
                const _Float16* G0 = lds3 + (size_t)buf * G_STAGE;
#pragma unroll
                for (int k = 0; k < KST; ++k) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tr_frag(A0 + k * TR_PLANE, lane, 32 * (wave - 4)),
                                                                       tr_frag_tiled(B2s, SROWS, k, lane, 0), acc[0][0], 0, 0, 0);
                    // ... and two 32-column tiles of g_sigma^T h7 (row 0 of the tile lives): the fragment of the one-column operand
                    // (g_lane / g_step: the lane's quad in the stage image, or the zero quad for lanes whose fragment columns are not 0..3)
                    // (addresses rebuilt per k-step from an opaque copy of the lane id: kept across the stage loop they cost the
                    // kernel the registers it does not have -- 256 VGPR and 56 bytes of scratch in the first form)
                    typedef __attribute__((address_space(3))) v4s16* lds_v4;
                    int ln = lane;
                    asm volatile("" : "+v"(ln));
                    const bool col0 = (ln & 0x13) == 0;                       // fragment columns 16 (gam & 1) + 4 (lam & 3) .. + 3 are 0..3
                    const int quad = col0 ? (TR_ROWS * k + 8 * (ln >> 5) + ((ln & 15) >> 2)) * 4 : SROWS * 4;
                    const _Float16* gp = G0 + quad;
                    union { v4s16 h[2]; wh8 v; } gf;
                    gf.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)gp);
                    gf.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(gp + (col0 ? 16 : 0)));
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        acc[0][1 + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gf.v, tr_frag_tiled(B0, SROWS, k, lane, 32 * (2 * (wave - 4) + t)),
                                                                               acc[0][1 + t], 0, 0, 0);
                }
            }
            return;
        }
        const _Float16* A0 = lds + (size_t)buf * 2 * SPLANE;
        const _Float16* B0 = A0 + SPLANE;
#pragma unroll
        for (int k = 0; k < KST; ++k) {
            wh8 af[NO];
#pragma unroll
            for (int o = 0; o < NO; ++o)
                af[o] = ATILED ? tr_frag_tiled(A0, SROWS, k, lane, o_base + 32 * o) : tr_frag(A0 + k * TR_PLANE, lane, o_base + 32 * o);
#pragma unroll
            for (int i = 0; i < NI; ++i) {      // one B fragment live at a time (register budget of the two sets)
                const wh8 bf = TILED ? tr_frag_tiled(B0, SROWS, k, lane, i_base + 32 * i)
                                     : tr_frag(B0 + k * TR_PLANE, lane, i_base + 32 * i);
#pragma unroll
                for (int o = 0; o < NO; ++o)
                    acc[o][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[o], bf, acc[o][i], 0, 0, 0);
            }
        }
    };
    // stage i is computed from LDS buffer i & 1 while stage i+1 sits in a register set (requested one iteration ago)
    // and stage i+2 is requested into the other set; two iterations per trip keep the sets static
    // Fetches are UNCONDITIONAL (a stage past the range re-reads the clamped last row and is never stashed): a
    // branch around a fetch would again leave the number of outstanding loads unknown at the next wait.
    if (m_begin >= m_end) return;     // (uniform; only when there are more row ranges than rows)
    fetch(s0, m_begin);
    if constexpr (DEEP) {
        fetch(s1, m_begin + SROWS);
        stash(s0, 0, m_begin);
        __syncthreads();
        for (int m = m_begin; m < m_end; m += 2 * SROWS) {
            fetch(s0, m + 2 * SROWS);
            compute(0);
            if (m + SROWS < m_end) stash(s1, 1, m + SROWS);
            __syncthreads();
            if (m + SROWS >= m_end) break;
            fetch(s1, m + 3 * SROWS);
            compute(1);
            if (m + 2 * SROWS < m_end) stash(s0, 0, m + 2 * SROWS);
            __syncthreads();
        }
    } else {
        // one stage ahead (the 256-wide jobs: their 64 KB stages already run at the HBM's read ceiling, and the
        // second register set measured +-0 there at 256 VGPR with spills)
        stash(s0, 0, m_begin);
        __syncthreads();
        int buf = 0;
        for (int m = m_begin; m < m_end; m += SROWS, buf ^= 1) {
            fetch(s0, m + SROWS);
            compute(buf);
            if (m + SROWS < m_end) stash(s0, buf ^ 1, m + SROWS);
            __syncthreads();
        }
    }
    float* part = a.part + (size_t)split * PART_PER_SPLIT;
    if (live) {
        float* cpart = part + job.part_off;
        const int ll = lane & 31;
#pragma unroll
        for (int o = 0; o < NO; ++o)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    cpart[(size_t)(o_base + o * 32 + frag_row(r, lane)) * I + i_base + i * 32 + ll] = acc[o][i][r];
    }
    if constexpr (X2) {
        if (!live) {
            float* c2 = part + job.part2_off;
#pragma unroll
            for (int r = 0; r < 16; ++r) c2[(size_t)(32 * (wave - 4) + frag_row(r, lane)) * I2 + (lane & 31)] = acc[0][0][r];
            if (lane < 32) {      // row 0 of the sigma tiles = register 0 of lanes 0..31
#pragma unroll
                for (int t = 0; t < 2; ++t) part[PART_SIGMA + 32 * (2 * (wave - 4) + t) + lane] = acc[0][1 + t][0];
            }
        }
    }
    if (job.bias_off >= 0) {
        // bias partial = column sums of the dz slabs: a thread's slots share one chunk column (512 % A_C8 == 0)
        float* red = reinterpret_cast<float*>(smem_raw);      // [RED_ROWS][O] floats, reusing the LDS
        constexpr int RED_ROWS = ATILED ? 32 : 512 / A_C8;
        __syncthreads();
        if (ATILED) {
            // chunk of piece block pb = (slab, fragment, lane half g): halves e = features 32 slab + 16 fragment + 4 g + (e & 3)
            // + 8 (e >> 2) of one row; the 32 threads (tid & 31) of a block pair each hold partial sums over their rows
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int pb = a_col[h];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    red[(tid & 31) * O + 32 * (pb >> 2) + 16 * ((pb >> 1) & 1) + 4 * (pb & 1) + (e & 3) + 8 * (e >> 2)] = bs[ATILED ? h : 0][e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[a_row[0] * O + a_col[0] + e] = bs[0][e];
        }
        __syncthreads();
        for (int f = tid; f < O; f += 512) {
            float sum = 0.0f;
            for (int r = 0; r < RED_ROWS; ++r) sum += red[r * O + f];
            part[job.bias_off + f] = sum;
        }
    }
}

__global__ __launch_bounds__(512) void wgrad_thin_kernel(WgradArgs2 p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const bool second = (int)blockIdx.y >= p.splits0;      // (uniform)
    const WgradArgs& a = p.n[second ? 1 : 0];
    const int split = (int)blockIdx.y - (second ? p.splits0 : 0);
    const int j = a.tile_job[blockIdx.x];
    if (j < 0) return;      // (this network's direction columns ride on its view job: X2)
    const WJob job = a.jobs[j];
    if (job.b_tiled) {      // (uniform per workgroup: the register-resident forward's encoding planes; O = W only -- X2 above)
        wgrad_half_body<W, PE_K, 1, true, true>(a, job, split, smem_raw);
    } else {
        if (job.O == W) wgrad_half_body<W, PE_K, 1, true>(a, job, split, smem_raw);      // encoding columns of L0 / L5
        else wgrad_half_body<HV, DPE_K, 1, true>(a, job, split, smem_raw);               // direction columns of the view layer
    }
}

// the 256-wide jobs on the same body (256 x 256 layers, 128 x 256 feature columns of the view layer)
__global__ __launch_bounds__(512) void wgrad_main_kernel(WgradArgs2 p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const bool second = (int)blockIdx.y >= p.splits0;      // (uniform)
    const WgradArgs& a = p.n[second ? 1 : 0];
    const int split = (int)blockIdx.y - (second ? p.splits0 : 0);
    const WJob job = a.jobs[a.tile_job[blockIdx.x]];
    if (job.b_tiled) {      // (uniform per workgroup)
        if (job.O == W) wgrad_half_body<W, W, 4, false, true>(a, job, split, smem_raw);
        else wgrad_half_body<HV, W, 4, false, true, true>(a, job, split, smem_raw);      // (+ the direction columns: X2)
    } else {
        if (job.O == W) wgrad_half_body<W, W, 4, false>(a, job, split, smem_raw);
        else wgrad_half_body<HV, W, 4, false>(a, job, split, smem_raw);
    }
}

// sigma / rgb heads: dW_alpha = sum_m g_sigma h7, dW_rgb[c] = sum_m g_c hv, and their biases
template <typename PT>
struct HeadArgs {
    const float* g_raw;
    const PT* h7;     // saved planes: float (fp32 mode) or _Float16 (16-bit modes)
    const PT* hv;
    int n_rows, rows_per_wg;
    float* part;  // [n_wg][HEAD_PART]
    int tiled;    // half planes in SV_LAYOUT_TILED (rows_per_wg is then a multiple of 32)
    int skip_alpha;   // dW_alpha comes from the main launch's view job (PART_SIGMA): h7 is not read here
};

template <typename PT>
struct HeadArgs2 {      // workgroups [0, wgs0) belong to n[0], the rest to n[1]
    HeadArgs<PT> n[2];
    int wgs0;
};

template <typename PT>
__global__ __launch_bounds__(256) void wgrad_head_kernel(HeadArgs2<PT> p) {
    const bool second = (int)blockIdx.x >= p.wgs0;      // (uniform)
    const HeadArgs<PT>& a = p.n[second ? 1 : 0];
    const int wg = (int)blockIdx.x - (second ? p.wgs0 : 0);
    // 16-byte loads: VEC plane elements per thread, so a row of h7 (hv) is read by T7 (TV) neighbouring threads and
    // the workgroup covers R7 (RV) rows per pass; the row classes are summed through LDS at the end
    constexpr int VEC = 16 / (int)sizeof(PT);
    typedef PT vec_t __attribute__((ext_vector_type(VEC)));
    constexpr int T7 = W / VEC, R7 = 256 / T7;
    constexpr int TV = HV / VEC, RV = 256 / TV;
    __shared__ float acc7[R7][W];
    __shared__ float accv[RV][3][HV];
    __shared__ float red[4][4];
    const int tid = threadIdx.x;
    const int m_begin = wg * a.rows_per_wg;
    const int m_end = min(a.n_rows, m_begin + a.rows_per_wg);
    const float4* g4 = reinterpret_cast<const float4*>(a.g_raw);
    // Tiled half planes (mlp_layout.h): a 16-byte piece = 8 features {4 g + 0..3, 8 + 4 g + 0..3} + 32 j + 16 f of one row,
    // 32 rows of a piece block contiguous.  Thread = (piece block, row class): eight lanes read one 128-byte line.
    const bool tiled = sizeof(PT) == 2 && a.tiled;
    if (a.skip_alpha) {      // (uniform)
        for (int r = 0; r < R7; ++r) acc7[r][tid] = 0.0f;
    } else {   // dW_alpha[c] = sum_m g_sigma[m] h7[m][c]
        const int c = tiled ? tid / R7 : tid % T7, r = tiled ? tid % R7 : tid / T7;
        float w[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) w[e] = 0.0f;
#pragma unroll 4
        for (int m = m_begin + r; m < m_end; m += R7) {
            const float gs = a.g_raw[4 * (size_t)m + 3];
            const PT* src = tiled ? a.h7 + (size_t)(m >> 5) * (32 * W) + ((size_t)c * 32 + (m & 31)) * 8
                                  : a.h7 + (size_t)m * W + c * VEC;
            const vec_t h = *reinterpret_cast<const vec_t*>(src);
#pragma unroll
            for (int e = 0; e < VEC; ++e) w[e] = fmaf(gs, (float)h[e], w[e]);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            // piece block c = (slab, fragment, lane half): features 32 j + 16 f + 4 g + (e & 3) + 8 (e >> 2)
            const int col = tiled ? 16 * (c >> 1) + 4 * (c & 1) + (e & 3) + 8 * (e >> 2) : c * VEC + e;
            acc7[r][col] = w[e];
        }
    }
    {   // dW_rgb[k][c] = sum_m g_k[m] hv[m][c]
        const int c = tiled ? tid / RV : tid % TV, r = tiled ? tid % RV : tid / TV;
        float w0[VEC], w1[VEC], w2[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) { w0[e] = 0.0f; w1[e] = 0.0f; w2[e] = 0.0f; }
#pragma unroll 4
        for (int m = m_begin + r; m < m_end; m += RV) {
            const float4 g = g4[m];
            const PT* src = tiled ? a.hv + (size_t)(m >> 5) * (32 * HV) + ((size_t)c * 32 + (m & 31)) * 8
                                  : a.hv + (size_t)m * HV + c * VEC;
            const vec_t h = *reinterpret_cast<const vec_t*>(src);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float hf = (float)h[e];
                w0[e] = fmaf(g.x, hf, w0[e]);
                w1[e] = fmaf(g.y, hf, w1[e]);
                w2[e] = fmaf(g.z, hf, w2[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int col = tiled ? 16 * (c >> 1) + 4 * (c & 1) + (e & 3) + 8 * (e >> 2) : c * VEC + e;
            accv[r][0][col] = w0[e];
            accv[r][1][col] = w1[e];
            accv[r][2][col] = w2[e];
        }
    }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int m = m_begin + tid; m < m_end; m += 256) {
        const float4 g = g4[m];
        s0 += g.x; s1 += g.y; s2 += g.z; s3 += g.w;
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
    if ((tid & 63) == 0) { red[tid >> 6][0] = s0; red[tid >> 6][1] = s1; red[tid >> 6][2] = s2; red[tid >> 6][3] = s3; }
    __syncthreads();
    float* part = a.part + (size_t)wg * HEAD_PART;
    {
        float wa = 0.0f;
#pragma unroll
        for (int r = 0; r < R7; ++r) wa += acc7[r][tid];
        part[tid] = wa;
    }
    for (int o = tid; o < 3 * HV; o += 256) {
        const int k = o / HV, c = o - k * HV;
        float v = 0.0f;
#pragma unroll
        for (int r = 0; r < RV; ++r) v += accv[r][k][c];
        part[256 + o] = v;
    }
    if (tid < 4) {
        const float s = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        // layout: [640] = b_alpha, [641..643] = b_rgb
        part[tid == 3 ? 640 : 641 + tid] = s;
    }
}

// ------------------------------------------------------------------------------------
// Gradient with respect to the network's (embedded) input rows -- off the training path (the reference's sample
// positions carry no gradient, run_plnerf.py:728), kept simple: the three places an input enters,
//   g_x[row, c]   = sum_k dz_0[row, k] W_0[k, c] + dz_5[row, k] W_5[k, c]        (c < xyz_ch: layer 0 and the skip layer)
//   g_dir[row, c] = sum_k dz_view[row, k] W_view[k, 256 + c]                     (c < dir_ch: the view layer)
// on the vector ALU from the dz planes the dgrad kernel of the mode left in the workspace (fp32 row-major, or IEEE half
// in the tiled order of mlp_layout.h under the launch scale, which is divided back out here).  One workgroup per CU
// walks 32-row tiles with the three weight blocks (fp32, [k][channel]) in LDS; thread = (row of the tile, channel
// octet).
// ------------------------------------------------------------------------------------
struct InGradArgs {
    const float* w0;        // [256][xyz_ch]              pts_linears.0.weight
    const float* w5;        // [256][xyz_ch + 256]        pts_linears.5.weight (its first xyz_ch columns)
    const float* wv;        // [128][256 + dir_ch]        views_linears.0.weight (its last dir_ch columns)
    const void* dz0;
    const void* dz5;
    const void* dzv;
    const unsigned* gmax;   // half planes: the launch scale's maximum (nullptr: fp32 planes)
    int xyz_ch, dir_ch, n_rows;
    float* g_emb;           // [n_rows][xyz_ch + dir_ch]
};

template <bool H16>
__global__ __launch_bounds__(256) void input_grad_kernel(InGradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float ig_smem[];
    float* W0t = ig_smem;                 // [256][64]
    float* W5t = W0t + W * PE_K;          // [256][64]
    float* Wvt = W5t + W * PE_K;          // [128][32]
    const int tid = threadIdx.x;
    for (int i = tid; i < W * PE_K; i += 256) {
        const int k = i >> 6, c = i & 63;
        W0t[i] = c < a.xyz_ch ? a.w0[(size_t)k * a.xyz_ch + c] : 0.0f;
        W5t[i] = c < a.xyz_ch ? a.w5[(size_t)k * (a.xyz_ch + W) + c] : 0.0f;
    }
    for (int i = tid; i < HV * DPE_K; i += 256) {
        const int k = i >> 5, c = i & 31;
        Wvt[i] = c < a.dir_ch ? a.wv[(size_t)k * (W + a.dir_ch) + W + c] : 0.0f;
    }
    __syncthreads();
    float unscale = 1.0f;
    if (H16 && a.gmax) {
        const float gm = __uint_as_float(*a.gmax);
        if (gm > 0.0f && gm < __builtin_inff()) {
            int e;
            (void)frexpf(gm, &e);
            unscale = ldexpf(1.0f, e - (int)DZH_TARGET_EXP);
        }
    }
    const int r = tid & 31, cg = tid >> 5;
    const int n_tiles = (a.n_rows + 31) / 32;
    const int out_ch = a.xyz_ch + a.dir_ch;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int row = tile * 32 + r;
        float ax[8], ad[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) ax[j] = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) ad[j] = 0.0f;
        if (row < a.n_rows) {
            for (int pb = 0; pb < 32; ++pb) {
                float v0[8], v5[8];
                int f[8];
                if (H16) {
                    const size_t off = (((size_t)tile * 32 + pb) * 32 + r) * 8;      // the tiled order: piece block pb of the tile
                    const wh8 p0 = *reinterpret_cast<const wh8*>((const _Float16*)a.dz0 + off);
                    const wh8 p5 = *reinterpret_cast<const wh8*>((const _Float16*)a.dz5 + off);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v0[e] = (float)p0[e]; v5[e] = (float)p5[e];
                        f[e] = 32 * (pb >> 2) + 16 * ((pb >> 1) & 1) + 4 * (pb & 1) + (e & 3) + 8 * (e >> 2);
                    }
                } else {
                    const float* q0 = (const float*)a.dz0 + (size_t)row * W + 8 * pb;
                    const float* q5 = (const float*)a.dz5 + (size_t)row * W + 8 * pb;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { v0[e] = q0[e]; v5[e] = q5[e]; f[e] = 8 * pb + e; }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float* w0r = W0t + f[e] * PE_K + cg * 8;
                    const float* w5r = W5t + f[e] * PE_K + cg * 8;
#pragma unroll
                    for (int j = 0; j < 8; ++j) ax[j] = fmaf(v5[e], w5r[j], fmaf(v0[e], w0r[j], ax[j]));
                }
            }
            for (int c8 = 0; c8 < HV / 8; ++c8) {      // dz_view: row-major in every mode
                float vv[8];
                if (H16) {
                    const wh8 pv = *reinterpret_cast<const wh8*>((const _Float16*)a.dzv + (size_t)row * HV + 8 * c8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) vv[e] = (float)pv[e];
                } else {
                    const float* qv = (const float*)a.dzv + (size_t)row * HV + 8 * c8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) vv[e] = qv[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float* wr = Wvt + (8 * c8 + e) * DPE_K + cg * 4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) ad[j] = fmaf(vv[e], wr[j], ad[j]);
                }
            }
            float* dst = a.g_emb + (size_t)row * out_ch;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (cg * 8 + j < a.xyz_ch) dst[cg * 8 + j] = ax[j] * unscale;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (cg * 4 + j < a.dir_ch) dst[a.xyz_ch + cg * 4 + j] = ad[j] * unscale;
        }
    }
}

constexpr int WG_SPLITS = PLNERF_WG_SPLITS;
constexpr int HEAD_OUT = 644;   // dW_alpha 256, dW_rgb 384, b_alpha 1, b_rgb 3
struct ReduceArgs {
    const float* part;
    const float* head_part;
    int splits, splits_thin, n_head;   // row ranges of the 256-wide jobs / of the three thin jobs
    const unsigned* gmax;   // 16-bit modes: the dz planes were scaled by 2^(DZH_TARGET_EXP - exponent(max |g_raw|))
    const unsigned* status; // the network's range status word (16-bit modes), or nullptr
    float* status_out;      // nullptr, or where this launch leaves (float)(*status != 0): the tail of the caller's flat gradient
    int enc_tiled;          // the encoding planes were tiled (mlp_layout.h): column c of the thin jobs' results = channel sv_enc_channel(c);
                            // the view layer's direction columns then come from the MAIN launch's row ranges (its view job)
    int sig_main;           // alpha_linear.weight's gradient comes from the main launch's view job (PART_SIGMA), not from the head partials
    float* gred;            // 16-bit modes (composed view layer, mlp_layout.h): the view job's 128 x 256 result is G = dz_view^T h7 and goes
                            // here with s = the job's bias sums behind it (compose_grads turns them into the factors' gradients);
                            // nullptr: fp32 mode, the reference's two layers
    GradPtrs G;
};

struct ReduceArgs2 {      // blockIdx.y = network
    ReduceArgs n[2];
};

__global__ void wgrad_reduce_kernel(ReduceArgs2 p) {
    const ReduceArgs& a = p.n[blockIdx.y];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= PART_PER_SPLIT + HEAD_OUT * 64) return;
    // the range status travels with the gradient it qualifies (dp.GradientBucket sums this element over the ranks; a
    // non-zero sum withholds the step everywhere)
    if (idx == 0 && a.status_out) *a.status_out = (a.status && *a.status) ? 1.0f : 0.0f;
    if (idx >= PART_PER_SPLIT) {
        // head partials: one wavefront per output element (up to 512 partials each; a single thread walking
        // them was the longest thing in this kernel)
        const int r = idx - PART_PER_SPLIT, h = r >> 6, lane = r & 63;
        float s = 0.0f;
        for (int w = lane; w < a.n_head; w += 64) s += a.head_part[(size_t)w * HEAD_PART + h];
        s = wave_sum(s);
        if (lane != 0) return;
        if (h < 256) { if (!a.sig_main) a.G.p[P_WA][h] = s; }
        else if (h < 640) a.G.p[P_WR][h - 256] = s;
        else if (h == 640) a.G.p[P_BA][0] = s;
        else a.G.p[P_BR][h - 641] = s;
        return;
    }
    // the thin jobs (encoding columns of L0 / L5, direction columns of the view layer, and L0's bias, which
    // rides on them) are launched over their own number of row ranges
    const bool thin = (idx >= PART_PE0 && idx < (a.enc_tiled ? PART_VDIR : PART_BIAS)) || (idx >= PART_BIAS && idx < PART_BIAS + W);
    // composed view layer: the feature layer's job and bias slots are not used
    if (a.gred && ((idx >= PART_MAIN + 7 * W * W && idx < PART_VMAIN) || (idx >= PART_BIAS + 8 * W && idx < PART_BIAS + 9 * W))) return;
    if (idx >= PART_SIGMA && !a.sig_main) return;
    const int ns = thin ? a.splits_thin : a.splits;
    float s = 0.0f;
    {   // a batch's loads all in flight before its first add (one dependent load per add held the kernel at 33 us;
        // 22 us this way); the additions keep their left-to-right order, so the sums are the same bits
        constexpr int BATCH = 14;
        int sp = 0;
        for (; sp + BATCH <= ns; sp += BATCH) {
            float v[BATCH];
#pragma unroll
            for (int q = 0; q < BATCH; ++q) v[q] = __builtin_nontemporal_load(a.part + (size_t)(sp + q) * PART_PER_SPLIT + idx);
#pragma unroll
            for (int q = 0; q < BATCH; ++q) s += v[q];
        }
        for (; sp < ns; ++sp) s += a.part[(size_t)sp * PART_PER_SPLIT + idx];
    }
    if (a.gmax) {   // undo the power-of-two scale of the half dz planes (exact)
        const float gm = __uint_as_float(*a.gmax);
        if (gm > 0.0f && gm < __builtin_inff()) {
            int e;
            (void)frexpf(gm, &e);
            s = ldexpf(s, e - (int)DZH_TARGET_EXP);
        }
    }
    if (idx < PART_VMAIN) {
        const int job = idx >> 16, r = idx & 65535, o = r >> 8, i = r & 255;
        // jobs: L1, L2, L3, L4, L5 (hidden part), L6, L7, feature
        if (job < 4) a.G.p[2 * (job + 1)][o * W + i] = s;
        else if (job == 4) a.G.p[10][o * (W + a.G.xyz_ch) + a.G.xyz_ch + i] = s;
        else if (job < 7) a.G.p[2 * (job + 1)][o * W + i] = s;
        else a.G.p[P_WF][o * W + i] = s;
    } else if (idx < PART_PE0) {
        const int r = idx - PART_VMAIN, o = r >> 8, i = r & 255;
        if (a.gred) a.gred[r] = s;      // G[o][i]
        else a.G.p[P_WV][o * (W + a.G.dir_ch) + i] = s;
    } else if (idx < PART_PE5) {
        const int r = idx - PART_PE0, o = r >> 6, i = a.enc_tiled ? sv_enc_channel(r & 63, false) : (r & 63);
        if (i < a.G.xyz_ch) a.G.p[0][o * a.G.xyz_ch + i] = s;
    } else if (idx < PART_VDIR) {
        const int r = idx - PART_PE5, o = r >> 6, i = a.enc_tiled ? sv_enc_channel(r & 63, false) : (r & 63);
        if (i < a.G.xyz_ch) a.G.p[10][o * (W + a.G.xyz_ch) + i] = s;
    } else if (idx < PART_BIAS) {
        const int r = idx - PART_VDIR, o = r >> 5, i = a.enc_tiled ? sv_enc_channel(r & 31, true) : (r & 31);
        if (i < a.G.dir_ch) a.G.p[P_WV][o * (W + a.G.dir_ch) + W + i] = s;
    } else if (idx >= PART_SIGMA) {
        a.G.p[P_WA][idx - PART_SIGMA] = s;
    } else {
        const int r = idx - PART_BIAS;
        if (r < 8 * W) a.G.p[2 * (r >> 8) + 1][r & 255] = s;
        else if (r < 9 * W) a.G.p[P_BF][r - 8 * W] = s;
        else {
            a.G.p[P_BV][r - 9 * W] = s;
            if (a.gred) a.gred[HV * W + (r - 9 * W)] = s;
        }
    }
}

// max |x| over n4 float4 (g_raw rows): grid-stride rows, wave + workgroup reduction, ONE atomic per workgroup
// (atomics on a single address serialise in the L2: one per wave of a one-row-per-thread grid measured slower
// than the 16-loads-per-thread loop it was meant to replace)
__global__ __launch_bounds__(256) void absmax_kernel(const float4* __restrict__ x, size_t n4, unsigned* out) {
    __shared__ float wmax[4];
    float m = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = x[i];
        const float c[4] = {fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w)};
#pragma unroll
        for (int k = 0; k < 4; ++k) m = (c[k] > m || c[k] != c[k]) ? c[k] : m;   // a NaN sticks (and compares above every float as bits)
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_xor(m, d);
        m = (o > m || o != o) ? o : m;
    }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) m = (wmax[w] > m || wmax[w] != wmax[w]) ? wmax[w] : m;
        if (m != 0.0f) atomicMax(out, __float_as_uint(m));
    }
}

// The density activation's derivative, folded into the backward's entry: g_eff = g_raw with its sigma column multiplied by
// d softplus / d sigma, evaluated from the activation's OUTPUT raw_out[., 3] (common.h) -- written to the workspace
// (the caller's gradient buffer is not touched) -- and, for the 16-bit modes, max |g_eff| in the same pass (what
// absmax_kernel does for a network without activation).
__global__ __launch_bounds__(256) void absmax_act_kernel(const float4* __restrict__ g, const float* __restrict__ raw,
                                                         const float beta, float4* __restrict__ g_eff, size_t n4,
                                                         unsigned* out) {
    __shared__ float wmax[4];
    float m = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = g[i];
        v.w = v.w * density_activation_grad(raw[4 * i + 3], beta);
        g_eff[i] = v;
        const float c[4] = {fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w)};
#pragma unroll
        for (int k = 0; k < 4; ++k) m = (c[k] > m || c[k] != c[k]) ? c[k] : m;
    }
    if (!out) return;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_xor(m, d);
        m = (o > m || o != o) ? o : m;
    }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) m = (wmax[w] > m || wmax[w] != wmax[w]) ? wmax[w] : m;
        if (m != 0.0f) atomicMax(out, __float_as_uint(m));
    }
}

inline int splits_for(int n_rows, bool h16) {
    const int cap = h16 ? WG_SPLITS : PLNERF_WG_SPLITS_F32;
    int s = (n_rows + 1023) / 1024;
    if (s < 1) s = 1;
    if (s > cap) s = cap;
    return s;
}
inline int head_wgs_for(int n_rows) {
    int s = (n_rows + 255) / 256;
    if (s < 1) s = 1;
    if (s > MAX_HEAD_WGS) s = MAX_HEAD_WGS;
    return s;
}

}  // namespace

namespace plnerf {
namespace impl {

// Weight gradients from the saved activation planes and the dz planes of the mode's dgrad kernel (fp32 planes in
// fp32 mode, half planes otherwise): split-K partials, head reductions, deterministic final sum into grads[24].
int absmax(const float* x, size_t n, unsigned* out, hipStream_t st) {
    if (hipMemsetAsync(out, 0, sizeof(unsigned), st) != hipSuccess) return PLNERF_ELAUNCH;
    if (n % 4 != 0) return PLNERF_EINVAL;   // rows of four
    const size_t n4 = n / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, st, (const float4*)x, n4, out);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

int absmax_act(const float* g_raw, const float* raw_out, float beta, int n_rows, float* g_eff, unsigned* out, hipStream_t st) {
    if (out && hipMemsetAsync(out, 0, sizeof(unsigned), st) != hipSuccess) return PLNERF_ELAUNCH;
    const size_t n4 = (size_t)n_rows;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(absmax_act_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, st, (const float4*)g_raw,
                       raw_out, beta, (float4*)g_eff, n4, out);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

namespace {
// Everything the stage's launches need for ONE network, with the number of row ranges it may use in the main / thin
// launches (`cap_main`, `cap_thin`: the whole round for a single network, its share of it when two networks share a grid).
struct WgradPlan {
    WgradArgs main_args, thin_args;
    int nt;                           // tiles (jobs) of the main launch
    int splits, splits_thin, n_head, rows_per_head_wg;
    float* head_part;
    ReduceArgs red;
};

int plan_job(const plnerf::impl::WgradJob& jb, bool h16, int cap_main, int cap_thin, WgradPlan& P) {
    const int n_rows = jb.n_rows;
    const size_t N = (size_t)n_rows;
    const size_t NS_ = h16 ? sv_rows(N) : N;       // row stride of the saved half state (padded, mlp_layout.h)
    const int tiled = h16 && jb.saved_layout == SV_LAYOUT_TILED;
    const size_t es = h16 ? sizeof(_Float16) : sizeof(float);        // plane element size
    float* part = jb.partials;
    P.head_part = part + (size_t)MAX_SPLITS * PART_PER_SPLIT;
    int splits = (n_rows + 1023) / 1024;
    splits = splits < 1 ? 1 : (splits > cap_main ? cap_main : splits);
    int rps = (n_rows + splits - 1) / splits;
    rps = h16 ? (rps + 63) & ~63 : (rps + 15) & ~15;      // half kernels: whole 64-row stages (two tiles of the tiled layout)
    // Rounding the range length up can leave the LAST ranges without a row (131,072 rows over 85 ranges of 1,600: ranges
    // 82..84 start past the end).  A workgroup without rows writes no partial, and the reduction would add whatever the
    // workspace held there (round 4: found by comparing two schedules of the same step bit for bit) -- so only ranges
    // that hold a row are launched and summed.
    splits = (n_rows + rps - 1) / rps;
    int splits_thin = splits;
    if (h16) { splits_thin = (n_rows + 1023) / 1024; splits_thin = splits_thin < 1 ? 1 : (splits_thin > cap_thin ? cap_thin : splits_thin); }
    int rps_thin = (n_rows + splits_thin - 1) / splits_thin;
    rps_thin = h16 ? (rps_thin + 63) & ~63 : (rps_thin + 15) & ~15;      // (half: whole tiles of the tiled dz planes)
    splits_thin = (n_rows + rps_thin - 1) / rps_thin;
    const unsigned char* sv = (const unsigned char*)jb.saved;
    const unsigned char* dz = (const unsigned char*)jb.dz;
    auto splane = [&](int p) { return (const void*)(sv + (size_t)p * W * NS_ * es); };
    const size_t ND = h16 ? dz_rows(N) : N;        // row stride of the dz planes (half: padded to the dgrad kernel's tiles)
    auto dplane = [&](int p) { return (const void*)(dz + (size_t)p * W * ND * es); };
    // (half state: the composed network's planes -- no feature plane, no dz_feature plane; mlp_layout.h)
    const void* pe_plane = sv + (size_t)(h16 ? SVC_PE_OFF : SV_PE_OFF) * NS_ * es;
    const void* dpe_plane = sv + (size_t)(h16 ? SVC_DPE_OFF : SV_DPE_OFF) * NS_ * es;
    const void* dzv_plane = dz + (size_t)(h16 ? DZC_V_OFF : DZ_V_OFF) * ND * es;
    {
        // main: 256 x 256 layer jobs + the view layer's feature columns (half: G = dz_view^T h7 of the composed view layer, and no
        // feature job)
        WgradArgs a{};
        const int layer_of_job[8] = {1, 2, 3, 4, 5, 6, 7, -1};
        int nt = 0;
        for (int j = 0; j < (h16 ? 7 : 8); ++j) {
            WJob& jw = a.jobs[j];
            if (j < 7) {
                const int l = layer_of_job[j];
                jw.A = dplane(l);
                jw.B = splane(l - 1);
                jw.bias_off = PART_BIAS + l * W;
            } else {
                jw.A = dplane(DZ_FEAT);
                jw.B = splane(7);
                jw.bias_off = PART_BIAS + 8 * W;
            }
            jw.lda = W; jw.ldb = W; jw.O = W; jw.I = W;
            jw.b_tiled = tiled;
            jw.part_off = PART_MAIN + j * W * W;
            a.tile_job[nt] = j; a.tile_o0[nt++] = 0;
            if (!h16) { a.tile_job[nt] = j; a.tile_o0[nt++] = 128; }   // f32 kernel: 128-row o tiles
        }
        WJob& jv = a.jobs[8];
        jv.A = dzv_plane; jv.lda = HV; jv.B = splane(h16 ? 7 : SV_FEAT); jv.ldb = W; jv.O = HV; jv.I = W;
        jv.part_off = PART_VMAIN; jv.bias_off = PART_BIAS + 9 * W; jv.b_tiled = tiled;
        if (tiled) { jv.B2 = dpe_plane; jv.part2_off = PART_VDIR; jv.g_raw = jb.g_raw; jv.gmax = jb.gmax; }      // (the direction columns and the sigma head on the job's idle waves)
        a.tile_job[nt] = 8; a.tile_o0[nt++] = 0;
        a.n_rows = n_rows; a.rows_per_split = rps; a.part = part;
        P.main_args = a; P.nt = nt;
    }
    {
        // the three thin jobs: encoding columns of L0 / L5 (256 x 64), direction columns of the view layer (128 x 32)
        // (folding them into the main launch as extra tiles measured 1-2 % slower than this second launch)
        WgradArgs a{};
        a.jobs[0] = WJob{dplane(0), pe_plane, W, PE_K, W, PE_K, PART_PE0, PART_BIAS + 0 * W, tiled};
        a.jobs[1] = WJob{dplane(5), pe_plane, W, PE_K, W, PE_K, PART_PE5, -1, tiled};
        a.jobs[2] = WJob{dzv_plane, dpe_plane, HV, DPE_K, HV, DPE_K, PART_VDIR, -1, tiled};
        for (int t = 0; t < 3; ++t) { a.tile_job[t] = t; a.tile_o0[t] = 0; }
        if (tiled) a.tile_job[2] = -1;      // (dz_view^T dpe: the view job of the main launch)
        // half: three (tiled: two) tiles only, so more row ranges than the main launch to cover the 256 CUs (3 x 85 = 255, 2 x 127).
        // (Equal ROW counts per workgroup, not equal bytes: a stage costs its latency whatever its width, so
        // giving the 128 x 32 job half as many, twice as long ranges measured 0.27 ms slower per step.)
        a.n_rows = n_rows; a.rows_per_split = h16 ? rps_thin : rps; a.part = part;
        P.thin_args = a;
    }
    P.splits = splits; P.splits_thin = splits_thin;
    P.n_head = head_wgs_for(n_rows);
    P.rows_per_head_wg = h16 ? ((((n_rows + P.n_head - 1) / P.n_head) + 31) & ~31) : (n_rows + P.n_head - 1) / P.n_head;
    {
        ReduceArgs a{};
        a.part = part; a.head_part = P.head_part; a.splits = splits; a.splits_thin = splits_thin; a.n_head = P.n_head;
        a.gmax = h16 ? jb.gmax : nullptr;
        a.status = jb.status; a.status_out = jb.status_out;
        a.enc_tiled = tiled;
        a.sig_main = tiled;
        a.gred = h16 ? jb.gred : nullptr;
        if (h16 && (!jb.gred || !jb.cb)) return PLNERF_EINVAL;
        a.G.xyz_ch = jb.xyz_ch; a.G.dir_ch = jb.dir_ch;
        for (int i = 0; i < PLNERF_N_PARAM_TENSORS; ++i) {
            if (!jb.grads[i]) return PLNERF_EINVAL;
            a.G.p[i] = jb.grads[i];
        }
        P.red = a;
    }
    return PLNERF_OK;
}

int launch_reduce(const WgradPlan* P, int n, hipStream_t st) {
    ReduceArgs2 r{};
    for (int j = 0; j < 2; ++j) r.n[j] = P[j < n ? j : 0].red;
    const int total = PART_PER_SPLIT + HEAD_OUT * 64;   // PART_PER_SPLIT is a multiple of 64: head waves stay whole
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 255) / 256, n), dim3(256), 0, st, r);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}
}  // namespace

// fp32 mode (and the single-network entry of the 16-bit modes)
int wgrad(const float* g_raw, int n_rows, const void* saved, const void* dzv, const unsigned* gmax, float* part,
          float* const* grads, int xyz_ch, int dir_ch, bool h16, int saved_layout, const unsigned* status,
          float* status_out, hipStream_t st) {
    const WgradJob jb{g_raw, n_rows, saved, dzv, gmax, part, grads, xyz_ch, dir_ch, saved_layout, status, status_out};
    if (h16) return wgrad_h16_multi(1, &jb, st);
    WgradPlan P;
    const int rc = plan_job(jb, false, PLNERF_WG_SPLITS_F32, PLNERF_WG_SPLITS_F32, P);
    if (rc) return rc;
    hipLaunchKernelGGL((wgrad_f32_kernel<2, 2, 2, 4, 4>), dim3(P.nt, P.splits), dim3(256), 0, st, P.main_args);
    PLNERF_CHECK_LAUNCH();
    hipLaunchKernelGGL((wgrad_f32_kernel<4, 1, 2, 2, 4>), dim3(2, P.splits), dim3(256), 0, st, P.thin_args);
    PLNERF_CHECK_LAUNCH();
    WgradArgs v{};
    v.jobs[0] = P.thin_args.jobs[2];
    v.tile_job[0] = 0; v.tile_o0[0] = 0;
    v.n_rows = n_rows; v.rows_per_split = P.thin_args.rows_per_split; v.part = part;
    hipLaunchKernelGGL((wgrad_f32_kernel<4, 1, 1, 1, 8>), dim3(1, P.splits), dim3(256), 0, st, v);
    PLNERF_CHECK_LAUNCH();
    HeadArgs2<float> h{};
    h.n[0] = h.n[1] = HeadArgs<float>{g_raw, (const float*)P.main_args.jobs[7].B, (const float*)((const unsigned char*)saved + (size_t)SV_HV_OFF * (size_t)n_rows * sizeof(float)),
                                      n_rows, P.rows_per_head_wg, P.head_part, 0, 0};
    h.wgs0 = P.n_head;
    hipLaunchKernelGGL(wgrad_head_kernel<float>, dim3(P.n_head), dim3(256), 0, st, h);
    PLNERF_CHECK_LAUNCH();
    return launch_reduce(&P, 1, st);
}

int wgrad_h16_multi(int n, const WgradJob* jobs, hipStream_t st) {
    if (n < 1 || n > MAX_BWD_JOBS) return PLNERF_EINVAL;
    // the one round of workgroups, dealt out in proportion to the networks' rows (a single network: all of it, as before)
    bool all_tiled = true;
    for (int j = 0; j < n; ++j) all_tiled = all_tiled && jobs[j].saved_layout == SV_LAYOUT_TILED;
    const int thin_tiles = all_tiled ? 2 : 3;               // (tiled: the direction columns ride on the main launch's view job)
    const int thin_round = all_tiled ? 127 : 85;            // row ranges of a thin job: 2 x 127 = 254 / 3 x 85 = 255 workgroups
    int cap_main[MAX_BWD_JOBS] = {WG_SPLITS, 0}, cap_thin[MAX_BWD_JOBS] = {thin_round, 0};
    if (n == 2) {
        const double f = (double)jobs[0].n_rows / ((double)jobs[0].n_rows + (double)jobs[1].n_rows);
        auto deal = [&](int total, int* cap) {
            int s0 = (int)(total * f + 0.5);
            s0 = s0 < 1 ? 1 : (s0 > total - 1 ? total - 1 : s0);
            cap[0] = s0; cap[1] = total - s0;
        };
        deal(WG_SPLITS, cap_main);
        deal(thin_round, cap_thin);
    }
    WgradPlan P[MAX_BWD_JOBS];
    for (int j = 0; j < n; ++j) {
        const int rc = plan_job(jobs[j], true, cap_main[j], cap_thin[j], P[j]);
        if (rc) return rc;
    }
    const size_t lds = (size_t)2 * 2 * TR_STEPS * TR_PLANE * sizeof(_Float16);
    {
        // (+ the view job's second B operand, two buffers of four piece blocks, and its one-column sigma operand)
        const size_t lds_main = lds + (size_t)2 * 4 * (TR_ROWS * TR_STEPS * 8 + TRB_PAD) * sizeof(_Float16) +
                                (size_t)2 * (TR_ROWS * TR_STEPS * 4 + 4) * sizeof(_Float16);
        WgradArgs2 a{};
        for (int j = 0; j < 2; ++j) a.n[j] = P[j < n ? j : 0].main_args;
        a.splits0 = P[0].splits;
        const int splits = P[0].splits + (n == 2 ? P[1].splits : 0);
        (void)hipFuncSetAttribute((const void*)wgrad_main_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_main);
        hipLaunchKernelGGL(wgrad_main_kernel, dim3(P[0].nt, splits), dim3(512), lds_main, st, a);
        PLNERF_CHECK_LAUNCH();
    }
    {
        WgradArgs2 a{};
        for (int j = 0; j < 2; ++j) a.n[j] = P[j < n ? j : 0].thin_args;
        a.splits0 = P[0].splits_thin;
        const int splits = P[0].splits_thin + (n == 2 ? P[1].splits_thin : 0);
        (void)hipFuncSetAttribute((const void*)wgrad_thin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(wgrad_thin_kernel, dim3(thin_tiles, splits), dim3(512), lds, st, a);
        PLNERF_CHECK_LAUNCH();
    }
    {
        HeadArgs2<_Float16> h{};
        for (int j = 0; j < 2; ++j) {
            const int q = j < n ? j : 0;
            const size_t NS_ = sv_rows((size_t)jobs[q].n_rows);
            h.n[j] = HeadArgs<_Float16>{jobs[q].g_raw, (const _Float16*)P[q].main_args.jobs[8].B,      // (h7: the composed view job's B plane)
                                        (const _Float16*)((const unsigned char*)jobs[q].saved + (size_t)SVC_HV_OFF * NS_ * sizeof(_Float16)),
                                        jobs[q].n_rows, P[q].rows_per_head_wg, P[q].head_part,
                                        jobs[q].saved_layout == SV_LAYOUT_TILED, jobs[q].saved_layout == SV_LAYOUT_TILED};
        }
        h.wgs0 = P[0].n_head;
        hipLaunchKernelGGL(wgrad_head_kernel<_Float16>, dim3(P[0].n_head + (n == 2 ? P[1].n_head : 0)), dim3(256), 0, st, h);
        PLNERF_CHECK_LAUNCH();
    }
    const int rc = launch_reduce(P, n, st);
    if (rc) return rc;
    // the composed view layer's factors: dW_f, db_f and the feature columns of dW_v from the reduced G and s
    ComposeGradJob cj[MAX_BWD_JOBS];
    for (int j = 0; j < n; ++j) cj[j] = ComposeGradJob{jobs[j].cb, jobs[j].gred, jobs[j].grads, jobs[j].dir_ch};
    return compose_grads(n, cj, st);
}

int input_grad(const float* const* params, int n_rows, const void* dzv, const unsigned* gmax, bool h16, int xyz_ch,
               int dir_ch, float* g_emb, hipStream_t st) {
    const size_t N = (size_t)n_rows;
    const size_t es = h16 ? sizeof(_Float16) : sizeof(float);
    const size_t ND = h16 ? dz_rows(N) : N;
    const unsigned char* dz = (const unsigned char*)dzv;
    InGradArgs a{};
    a.w0 = params[0]; a.w5 = params[10]; a.wv = params[P_WV];
    a.dz0 = dz; a.dz5 = dz + (size_t)5 * W * ND * es; a.dzv = dz + (size_t)(h16 ? DZC_V_OFF : DZ_V_OFF) * ND * es;
    a.gmax = h16 ? gmax : nullptr;
    a.xyz_ch = xyz_ch; a.dir_ch = dir_ch; a.n_rows = n_rows; a.g_emb = g_emb;
    const size_t lds = ((size_t)2 * W * PE_K + (size_t)HV * DPE_K) * sizeof(float);
    const int n_tiles = (n_rows + 31) / 32;
    const int grid = n_tiles < 256 ? n_tiles : 256;
    if (h16) {
        (void)hipFuncSetAttribute((const void*)input_grad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(input_grad_kernel<true>, dim3(grid), dim3(256), lds, st, a);
    } else {
        (void)hipFuncSetAttribute((const void*)input_grad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(input_grad_kernel<false>, dim3(grid), dim3(256), lds, st, a);
    }
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

}  // namespace impl
}  // namespace plnerf
