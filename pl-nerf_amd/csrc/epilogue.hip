// The coarse pass's epilogue as ONE per-ray kernel (run_plnerf.py:714-735, piecewise-linear mode):
//
//     raw2outputs(raw, z)                       -> rgb0, disp0, acc0, depth0        (:553-624, weights :516-550)
//     sample_pdf_reformulation(z, w, tau, T, u) -> z_samples                        (run_nerf_helpers.py:364-445)
//     clamp(z_samples, near, far); z_std                                            (:731, :752)
//     sort(cat(z, z_samples))                   -> z_fine                            (:733-734)
//     pts = o + d * z_fine                                                          (:735)
//
// One wavefront owns one ray.  weights, tau, T and the cdf never leave the CU: they are LDS rows of the wave (the
// separate launches write them to HBM for the next launch to read back).  HBM sees raw, z and u in, the coarse maps,
// z_fine, pts and z_std out: algorithmic bytes per ray 20 S + 4 N (+44) in, 16 (S + N) + 32 out.  Every device
// function is the one the separate kernels use (ray_dev.h), so the results are bit-identical to
// plnerf_quad_fwd -> plnerf_sample_pl -> plnerf_merge_sort -> plnerf_ray_points
// (tests/test_gpu_parity.py::test_fused_coarse_epilogue_equals_separate_launches).
#include "common.h"
#include "philox.h"
#include "ray_dev.h"

using namespace plnerf;

namespace {

constexpr int WAVES = 4;

struct EpiArgs {
    RayIn in;
    const float* rays_o;
    const float* u;          // [R, N] (stride N), one shared row [N] (stride 0), or null = drawn here (rng)
    int u_row_stride;
    RngArgs rng;
    int R, S, N;
    int color_mode, white_bkgd, farcolorfix;
    float zero_tol, eps;
    int lds_stride;
    float* rgb_map;
    float* disp_map;
    float* acc_map;
    float* depth_map;
    float* weights;          // optional [R, S+1]
    float* tau;              // optional [R, S+2]
    float* T;                // optional [R, S+2]
    float* z_fine;           // [R, S+N]
    float* pts;              // [R, S+N, 3]
    float* z_std;            // [R]
    // hypothesis mode (HYP: the depth-supervised variant's final stage, plnerf_fine_epilogue): the samples themselves
    float* samples;          // [R, N], NOT clamped
    int64_t* inds;           // [R, N] searchsorted indices (plnerf_sample_pl_bwd wants them)
    float* u_out;            // [R, N] the draws used (the variant's render_rays returns them)
};

// HYP = false: the coarse pass's epilogue (samples clamped, merged, sorted, turned into positions).
// HYP = true: raw2outputs + the *_return_u sampler of the depth-supervised variant's LAST stage
// (run_nerf_sample_based_depth.py:923-934): the samples are the depth hypotheses pred_hyp -- not clamped, kept with their
// indices and draws for the sampler's backward -- and z_std is theirs (:934); nothing is sorted or positioned.
// KPL: registers per lane of the general sort network (64 KPL >= S + N); KS: of the samples' own network (64 KS >= N)
template <int KPL, bool HYP, int KS = KPL>
__global__ __launch_bounds__(256) void coarse_epilogue_kernel(const EpiArgs a) {
    constexpr int MODE = PLNERF_MODE_LINEAR;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int ray = blockIdx.x * WAVES + wave;
    const bool live = ray < a.R;
    if (!live) ray = a.R - 1;
    const int S = a.S, K = S + 2, n = S + 1, N = a.N, NF = S + N;
    float* zk = smem + (size_t)wave * a.lds_stride;   // K knots [near, z, far]
    float* tau = zk + K;                               // K
    float* col = tau + K;                              // 3 S
    float* Tr = col + 3 * S;                           // K
    float* cdf = Tr + K;                               // K
    float* smp = cdf + K;                              // N clamped samples; later the sorted row (NF, aliases col..)
    float dnorm;
    load_ray(a.in, ray, lane, zk, tau, col, dnorm);
    __syncthreads();

    // ---- quadrature (quad_fwd_kernel's loop) with the cdf's running sum riding along ----
    double carry = 1.0, ccarry = 0.0;
    double sr = 0, sg = 0, sb = 0, sd = 0, sa = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const bool valid = i < n;
        float seg = 0.f, e = 1.f, f = 1.f;
        if (valid) interval<MODE>(i, S, zk, tau, dnorm, seg, e, f);
        const double incl = wave_incl_prod((double)f);
        double excl = __shfl_up(incl, 1);
        if (lane == 0) excl = 1.0;
        const float Ti = (float)(carry * excl);
        const float Tn = (float)(carry * incl);
        carry = carry * __shfl(incl, 63);
        float w = 0.0f;
        if (valid) {
            w = (1.0f - e) * Ti;
            sr += (double)(w * elem_colour<MODE>(i, 0, S, col, a.color_mode, a.farcolorfix));
            sg += (double)(w * elem_colour<MODE>(i, 1, S, col, a.color_mode, a.farcolorfix));
            sb += (double)(w * elem_colour<MODE>(i, 2, S, col, a.color_mode, a.farcolorfix));
            sd += (double)(w * elem_depth<MODE>(i, zk));
            sa += (double)w;
            Tr[i + 1] = Tn;
            if (live) {
                if (a.weights) a.weights[(size_t)ray * n + i] = w;
                if (a.T) a.T[(size_t)ray * K + i + 1] = Tn;
            }
        }
        // cdf = [0, cumsum(weights)] (fp64 running sum, each entry rounded to fp32), as sample_pl_kernel builds it
        const double cincl = wave_incl_sum((double)w);
        if (valid) cdf[i + 1] = (float)(ccarry + cincl);
        ccarry = ccarry + __shfl(cincl, 63);
    }
    if (live) {
        if (a.T && lane == 0) a.T[(size_t)ray * K] = 1.0f;
        if (a.tau)
            for (int s = lane; s < K; s += 64) a.tau[(size_t)ray * K + s] = tau[s];
    }
    sr = wave_sum(sr); sg = wave_sum(sg); sb = wave_sum(sb); sd = wave_sum(sd); sa = wave_sum(sa);
    if (live && lane == 0) {
        const float acc = (float)sa, depth = (float)sd;
        float r = (float)sr, g = (float)sg, b = (float)sb;
        if (a.white_bkgd) {
            const float bg = 1.0f - acc;
            r += bg; g += bg; b += bg;
        }
        a.rgb_map[3 * ray + 0] = r;
        a.rgb_map[3 * ray + 1] = g;
        a.rgb_map[3 * ray + 2] = b;
        a.depth_map[ray] = depth;
        a.acc_map[ray] = acc;
        a.disp_map[ray] = 1.0f / tmax(1e-10f, depth / acc);
    }
    __syncthreads();
    if (lane == 0) { Tr[0] = 1.0f; cdf[0] = 0.0f; cdf[K - 1] = 1.0f; }
    __syncthreads();

    // ---- importance samples (sample_pl_kernel's loop), clamped to [near, far] ----
    const float lo = zk[0], hi = zk[K - 1];
    const float zt = a.zero_tol, eps = a.eps;
    double ssum = 0.0;
    for (int k = lane; k < N; k += 64) {
        const float u = a.u ? a.u[(size_t)ray * a.u_row_stride + k] : rng_uniform(a.rng, a.rng.ray_id0 + ray, k);
        const int ind = upper_bound(cdf, K, u);
        const int below = ind - 1 > 0 ? ind - 1 : 0;
        const int above = ind < K - 1 ? ind : K - 1;
        const float s0 = zk[below], s1 = zk[above];
        const float T0 = Tr[below];
        const float tau0 = tau[below], tau1 = tau[above];
        const int di = below < S ? below : S;
        const float d = tau[di + 1] - tau[di];
        float out = (d < zt && d > -zt) ? s0 : -1.0f;
        const bool rising = d >= zt;
        if (rising || d <= -zt) out = invert_segment(s0, s1, T0, tau0, tau1, u, eps, rising);      // (one evaluation, operands selected per lane)
        if (out != out) out = s0;
        if constexpr (HYP) {
            if (live) {
                a.samples[(size_t)ray * N + k] = out;
                a.inds[(size_t)ray * N + k] = ind;
                if (a.u_out) a.u_out[(size_t)ray * N + k] = u;
            }
        } else {
            out = tmin(tmax(out, lo), hi);        // torch.clamp(z_samples, near, far)
        }
        smp[k] = out;
        ssum += (double)out;
    }
    // z_std = std(z_samples, unbiased=False) over the clamped samples
    ssum = wave_sum(ssum);
    const double mean = ssum / (double)N;
    __syncthreads();
    double sq = 0.0;
    for (int k = lane; k < N; k += 64) {
        const double dv = (double)smp[k] - mean;
        sq += dv * dv;
    }
    sq = wave_sum(sq);
    if (live && lane == 0) a.z_std[ray] = (float)sqrt(sq / (double)N);
    if constexpr (HYP) return;

    // ---- sort(cat(z, samples)) ----
    // The coarse depths are ascending in every reference call (stratified bins, run_plnerf.py:683-705) -- checked per ray,
    // else the general network below.  Then only the N samples need sorting (the bitonic network on KS registers: 28 stages
    // for 128 keys instead of 36 over 256), and the two ascending runs merge by rank: two binary searches per element in
    // place of the network's last eight stages.  Same multiset, same order: the same bits as sort(cat()) (round 5; the
    // launch was VALU-issue bound, profiles/r05_pmc_stream_kernels_262144rays.txt).
    uint32_t xs[KS];
#pragma unroll
    for (int r = 0; r < KS; ++r) {
        const int q = 64 * r + lane;
        xs[r] = q < N ? sort_key(smp[q]) : 0xFFFFFFFFu;
    }
    bitonic_sort_regs<KS>(xs, lane);
    bool asc = true;
    for (int p = lane; p + 1 < S; p += 64) asc = asc && (sort_key(zk[p + 1]) <= sort_key(zk[p + 2]));
    const bool merge = __all(asc) != 0;           // (per wave = per ray; every path below meets the same barriers)
    __syncthreads();                              // the sampler's and z_std's reads of tau / smp are done
    uint32_t* zkey = reinterpret_cast<uint32_t*>(tau);
    uint32_t* skey = reinterpret_cast<uint32_t*>(smp);
    if (merge) {
        for (int p = lane; p < S; p += 64) zkey[p] = sort_key(zk[p + 1]);
#pragma unroll
        for (int r = 0; r < KS; ++r) {
            const int q = 64 * r + lane;
            if (q < N) skey[q] = xs[r];
        }
    }
    __syncthreads();
    int rank_s[KS], rank_z[KPL];
    uint32_t x[KPL];
    if (merge) {
#pragma unroll
        for (int r = 0; r < KS; ++r) rank_s[r] = 64 * r + lane + count_le(zkey, S, xs[r]);
        // (the z elements' ranks read skey, which the sorted row may overwrite: all of them before the barrier)
#pragma unroll
        for (int r = 0; r < KPL; ++r) {
            const int p = 64 * r + lane;
            rank_z[r] = p < S ? p + count_lt(skey, N, zkey[p]) : -1;
        }
    } else {
#pragma unroll
        for (int r = 0; r < KPL; ++r) {
            const int p = 64 * r + lane;
            uint32_t key = 0xFFFFFFFFu;
            if (p < S) key = sort_key(zk[p + 1]);
            else if (p < NF) key = sort_key(smp[p - S]);
            x[r] = key;
        }
        bitonic_sort_regs<KPL>(x, lane);
    }
    float* zs = col;                  // NF <= 3 S + K + K + N floats from col on (col, Tr, cdf, smp are contiguous)
    // ONE barrier for both arms (`merge` is decided per wave = per ray: a barrier inside each arm was reached at different
    // program points by the waves of one workgroup -- the counts matched, the HIP programming model calls it undefined):
    // every read of col / smp / tau is done, the sorted row may overwrite them
    __syncthreads();
    if (merge) {
#pragma unroll
        for (int r = 0; r < KPL; ++r)
            if (rank_z[r] >= 0) zs[rank_z[r]] = zk[64 * r + lane + 1];
#pragma unroll
        for (int r = 0; r < KS; ++r)
            if (64 * r + lane < N) zs[rank_s[r]] = sort_unkey(xs[r]);
    } else {
#pragma unroll
        for (int r = 0; r < KPL; ++r) {
            const int p = 64 * r + lane;
            if (p < NF) zs[p] = sort_unkey(x[r]);
        }
    }
    __syncthreads();
    if (live)
        for (int p = lane; p < NF; p += 64) a.z_fine[(size_t)ray * NF + p] = zs[p];
    if (!live) return;
    // ---- positions of the merged samples ----
    const float o[3] = {a.rays_o[3 * (size_t)ray], a.rays_o[3 * (size_t)ray + 1], a.rays_o[3 * (size_t)ray + 2]};
    const float dd[3] = {a.in.rays_d[3 * (size_t)ray], a.in.rays_d[3 * (size_t)ray + 1], a.in.rays_d[3 * (size_t)ray + 2]};
    float* prow = a.pts + (size_t)ray * 3 * NF;
    if ((3 * NF) % 4 == 0 && ((uintptr_t)a.pts & 15) == 0) {
        for (int q = lane; q < 3 * NF / 4; q += 64) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = 4 * q + k, si = e / 3, c = e - 3 * si;
                const float oc = c == 0 ? o[0] : (c == 1 ? o[1] : o[2]);
                const float dc = c == 0 ? dd[0] : (c == 1 ? dd[1] : dd[2]);
                v[k] = oc + dc * zs[si];
            }
            reinterpret_cast<float4*>(prow)[q] = make_float4(v[0], v[1], v[2], v[3]);
        }
    } else {
        for (int e = lane; e < 3 * NF; e += 64) {
            const int si = e / 3, c = e - 3 * si;
            prow[e] = (c == 0 ? o[0] : (c == 1 ? o[1] : o[2])) + (c == 0 ? dd[0] : (c == 1 ? dd[1] : dd[2])) * zs[si];
        }
    }
}

template <int KPL, bool HYP = false, int KS = KPL>
int launch(const EpiArgs& a, size_t lds, hipStream_t st) {
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)coarse_epilogue_kernel<KPL, HYP, KS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
    hipLaunchKernelGGL((coarse_epilogue_kernel<KPL, HYP, KS>), dim3((a.R + WAVES - 1) / WAVES), dim3(WAVES * 64), lds, st, a);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

}  // namespace

extern "C" int plnerf_coarse_epilogue(const float* raw, const float* z, const float* near, const float* far,
                                      const float* rays_o, const float* rays_d, const float* noise, const float* u,
                                      int u_row_stride, uint64_t seed, uint32_t step, int ray_id0, int R, int S, int N,
                                      int color_mode, int white_bkgd, int farcolorfix, float zero_tol, float epsilon,
                                      float* rgb_map, float* disp_map, float* acc_map, float* depth_map, float* weights,
                                      float* tau, float* T, float* z_fine, float* pts, float* z_std,
                                      plnerf_stream_t stream) {
    if (R < 0 || S < 2 || N < 1) return PLNERF_EINVAL;
    if (u && u_row_stride != 0 && u_row_stride != N) return PLNERF_EINVAL;
    if (color_mode != PLNERF_COLOR_MIDPOINT && color_mode != PLNERF_COLOR_LEFT) return PLNERF_EINVAL;
    if (S > PLNERF_MAX_SAMPLES || S + N > 1024) return PLNERF_ERANGE;
    if (R == 0) return PLNERF_OK;
    if (!raw || !z || !near || !far || !rays_o || !rays_d || !rgb_map || !disp_map || !acc_map || !depth_map ||
        !z_fine || !pts || !z_std)
        return PLNERF_EINVAL;
    EpiArgs a{};
    a.in = RayIn{raw, z, near, far, rays_d, noise, S};
    a.rays_o = rays_o; a.u = u; a.u_row_stride = u_row_stride;
    a.rng = RngArgs{(uint32_t)seed, (uint32_t)(seed >> 32), 1u, step, ray_id0, u ? 0 : 1};
    a.R = R; a.S = S; a.N = N; a.color_mode = color_mode; a.white_bkgd = white_bkgd; a.farcolorfix = farcolorfix;
    a.zero_tol = zero_tol; a.eps = epsilon;
    a.rgb_map = rgb_map; a.disp_map = disp_map; a.acc_map = acc_map; a.depth_map = depth_map;
    a.weights = weights; a.tau = tau; a.T = T; a.z_fine = z_fine; a.pts = pts; a.z_std = z_std;
    // zk, tau, Tr, cdf: 4 (S+2); col 3 S; samples N   (the sorted row of S + N <= 3 S + 2 (S+2) + N reuses col onwards)
    a.lds_stride = ((4 * (S + 2) + 3 * S + N) + 3) & ~3;
    const size_t lds = (size_t)WAVES * a.lds_stride * sizeof(float);
    if (lds > 160 * 1024) return PLNERF_ERANGE;
    hipStream_t st = (hipStream_t)stream;
    const int nf = S + N;
    // (KS: the samples' own network; the small shapes get every size, the large ones sort their samples on the full width)
    if (nf <= 64) return launch<1>(a, lds, st);
    if (nf <= 128) return N <= 64 ? launch<2, false, 1>(a, lds, st) : launch<2>(a, lds, st);
    if (nf <= 256) return N <= 64 ? launch<4, false, 1>(a, lds, st) : (N <= 128 ? launch<4, false, 2>(a, lds, st) : launch<4>(a, lds, st));
    if (nf <= 512) return launch<8>(a, lds, st);
    return launch<16>(a, lds, st);
}


extern "C" int plnerf_fine_epilogue(const float* raw, const float* z, const float* near, const float* far,
                                    const float* rays_d, const float* noise, const float* u, int u_row_stride,
                                    uint64_t seed, uint32_t step, int ray_id0, int R, int S, int N, int color_mode,
                                    int white_bkgd, int farcolorfix, float zero_tol, float epsilon, float* rgb_map,
                                    float* disp_map, float* acc_map, float* depth_map, float* weights, float* tau,
                                    float* T, float* samples, int64_t* inds, float* u_out, float* z_std,
                                    plnerf_stream_t stream) {
    if (R < 0 || S < 2 || N < 1) return PLNERF_EINVAL;
    if (u && u_row_stride != 0 && u_row_stride != N) return PLNERF_EINVAL;
    if (color_mode != PLNERF_COLOR_MIDPOINT && color_mode != PLNERF_COLOR_LEFT) return PLNERF_EINVAL;
    if (S > PLNERF_MAX_SAMPLES || N > 1024) return PLNERF_ERANGE;
    if (R == 0) return PLNERF_OK;
    if (!raw || !z || !near || !far || !rays_d || !rgb_map || !disp_map || !acc_map || !depth_map || !weights || !tau ||
        !T || !samples || !inds || !z_std)
        return PLNERF_EINVAL;
    EpiArgs a{};
    a.in = RayIn{raw, z, near, far, rays_d, noise, S};
    a.u = u; a.u_row_stride = u_row_stride;
    // (stream id 4: the hypotheses' draw is not the coarse pass's importance draw, stream 1)
    a.rng = RngArgs{(uint32_t)seed, (uint32_t)(seed >> 32), 4u, step, ray_id0, u ? 0 : 1};
    a.R = R; a.S = S; a.N = N; a.color_mode = color_mode; a.white_bkgd = white_bkgd; a.farcolorfix = farcolorfix;
    a.zero_tol = zero_tol; a.eps = epsilon;
    a.rgb_map = rgb_map; a.disp_map = disp_map; a.acc_map = acc_map; a.depth_map = depth_map;
    a.weights = weights; a.tau = tau; a.T = T; a.z_std = z_std;
    a.samples = samples; a.inds = inds; a.u_out = u_out;
    a.lds_stride = ((4 * (S + 2) + 3 * S + N) + 3) & ~3;
    const size_t lds = (size_t)WAVES * a.lds_stride * sizeof(float);
    if (lds > 160 * 1024) return PLNERF_ERANGE;
    return launch<1, true>(a, lds, (hipStream_t)stream);      // (KPL sizes the sort network only: unused here)
}
