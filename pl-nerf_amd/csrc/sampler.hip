// Hierarchical samplers and the per-ray merge sort.
//
//   plnerf_sample_const : sample_pdf                    (run_nerf_helpers.py:241-284)
//   plnerf_sample_pl    : sample_pdf_reformulation      (run_nerf_helpers.py:364-445)
//                         + pw_linear_sample_{in,de}creasing (:340-361)
//   plnerf_merge_sort   : clamp + cat + sort            (run_plnerf.py:731-734)
//
// One wavefront per ray.  The ray's cdf / knots / tau / T rows are read once with
// coalesced loads and kept in LDS; the cdf is a wave prefix scan; every lane then inverts
// the cdf for its own u by bisection in LDS.  HBM-bound; algorithmic bytes per ray:
// 4(4S+7+N)+4N (PL sampler), 8(S+N) (merge sort)  (SURVEY.md section 8d).
//
// Compiled with -ffp-contract=off: the reference evaluates every product and sum as a
// separately rounded fp32 op (eager PyTorch), and the bit-exact index contract of
// plnerf_sample_const depends on that.
#include "common.h"
#include "ray_dev.h"

using namespace plnerf;

namespace {

constexpr int WAVES = 4;

// fp32 row sum with the association order of torch.sum's vectorised CPU kernel (8-lane
// vectors, 4 interleaved accumulators, leftover vectors into accumulator 0,
// ((a0+a1)+a2)+a3, then the scalar tail first and the 8 vector lanes after it).
// Verified against torch 2.10 CPU for every n in 1..510 (tests/test_gpu_parity.py::test_sample_pdf_every_bin_count).
// 4 <= n <= 7 is its own case there (round 5: found by tools/fuzz_samplers.py on weights that put u ON a cdf entry -- 24
// search indices of 32,634 differed at 5 weights): one 4-lane vector, the tail added to lane 0 in order, then the lanes in
// order -- (((((x0 + x4) + x5) + x6) + x1) + x2) + x3.
// Lanes 0..7 of the wave play the 8 SIMD lanes; result is broadcast to the wave.
__device__ __forceinline__ float torch_row_sum(const float* x, int n, int lane) {
    if (n >= 4 && n < 8) {      // (uniform)
        float t = x[0];
        for (int e = 4; e < n; ++e) t = t + x[e];
        t = t + x[1];
        t = t + x[2];
        return t + x[3];
    }
    const int nvec = n >> 3;
    const int g4 = nvec & ~3;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (lane < 8) {
        for (int v = 0; v < g4; v += 4) {
            a0 = a0 + x[(v + 0) * 8 + lane];
            a1 = a1 + x[(v + 1) * 8 + lane];
            a2 = a2 + x[(v + 2) * 8 + lane];
            a3 = a3 + x[(v + 3) * 8 + lane];
        }
        for (int v = g4; v < nvec; ++v) a0 = a0 + x[v * 8 + lane];
        a0 = a0 + a1;
        a0 = a0 + a2;
        a0 = a0 + a3;
    }
    float total = 0.f;
    for (int e = nvec * 8; e < n; ++e) total = total + x[e];   // uniform across lanes
#pragma unroll
    for (int l = 0; l < 8; ++l) total = total + __shfl(a0, l);
    return total;
}

struct SampleConstArgs {
    const float* bins;
    const float* weights;
    const float* u;
    int u_row_stride;
    int R, B, N;
    int lds_stride;
    float* samples;
    int64_t* inds;
};

__global__ __launch_bounds__(256) void sample_const_kernel(SampleConstArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int ray = blockIdx.x * WAVES + wave;
    const bool live = ray < a.R;
    if (!live) ray = a.R - 1;
    const int B = a.B, n = a.B - 1;
    float* cdf = smem + wave * a.lds_stride;   // B entries
    float* bins = cdf + B;                      // B entries
    float* wv = bins + B;                       // n entries (weights + 1e-5)
    for (int j = lane; j < B; j += 64) bins[j] = a.bins[(size_t)ray * B + j];
    for (int j = lane; j < n; j += 64) wv[j] = a.weights[(size_t)ray * n + j] + 1e-5f;
    __syncthreads();
    const float total = torch_row_sum(wv, n, lane);
    // cdf = [0, cumsum(pdf)]: fp64 running sum, each entry rounded to fp32
    double carry = 0.0;
    for (int base = 0; base < n; base += 64) {
        const int j = base + lane;
        const float pdf = (j < n) ? wv[j] / total : 0.0f;
        const double incl = wave_incl_sum((double)pdf);
        if (j < n) cdf[j + 1] = (float)(carry + incl);
        carry = carry + __shfl(incl, 63);
    }
    if (lane == 0) cdf[0] = 0.0f;
    __syncthreads();
    if (!live) return;
    const float* urow = a.u + (size_t)ray * a.u_row_stride;
    for (int k = lane; k < a.N; k += 64) {
        const float u = urow[k];
        const int ind = upper_bound(cdf, B, u);
        const int below = ind - 1 > 0 ? ind - 1 : 0;
        const int above = ind < B - 1 ? ind : B - 1;
        const float c0 = cdf[below], c1 = cdf[above];
        float denom = c1 - c0;
        if (denom < 1e-5f) denom = 1.0f;
        const float t = (u - c0) / denom;
        const float b0 = bins[below], b1 = bins[above];
        a.samples[(size_t)ray * a.N + k] = b0 + t * (b1 - b0);
        if (a.inds) a.inds[(size_t)ray * a.N + k] = (int64_t)ind;
    }
}


struct SampleConstBwdArgs {
    const float* bins;
    const float* weights;
    const float* u;
    int u_row_stride;
    const int64_t* inds;
    const float* g_samples;
    int R, B, N;
    int lds_stride;
    float* g_weights;
};

// Backward of sample_const_kernel with respect to `weights` (what autograd derives for sample_pdf_return_u,
// depth_supervised_exps/model/run_nerf_helpers.py:343-394, when pred_hyp carries a loss in constant mode):
//   sample = b0 + t (b1 - b0),  t = (u - c0) / denom,  denom = c1 - c0 (or 1 where that is < 1e-5)
//   cdf[j] = sum_{i<j} pdf[i],  pdf = w' / sum(w'),  w' = w + 1e-5
// Per-knot sums run in sample order and the suffix sums are wave scans: deterministic, no atomics.
__global__ __launch_bounds__(256) void sample_const_bwd_kernel(SampleConstBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int ray = blockIdx.x * WAVES + wave;
    const bool live = ray < a.R;
    if (!live) ray = a.R - 1;
    const int B = a.B, n = a.B - 1, N = a.N;
    float* cdf = smem + wave * a.lds_stride;   // B
    float* bins = cdf + B;                      // B
    float* wv = bins + B;                       // n  (weights + 1e-5), later g_pdf
    float* g0 = wv + B;                         // N: gradient landing on cdf[below]
    float* g1 = g0 + N;                         // N: on cdf[above]
    int* lo = reinterpret_cast<int*>(g1 + N);   // N
    int* hi = lo + N;                           // N
    float* gc = reinterpret_cast<float*>(hi + N);   // B: g_cdf
    // the cdf before its rounding to fp32 (round 6): c1 - c0 of two fp32 roundings carries 6e-8 / (c1 - c0) -- 1e-3 of the gradient
    // on a narrow bin, in the reference's fp32 autograd and in this kernel until then; the derivative's VALUE now comes from these,
    // which bins count as empty (c1 - c0 < 1e-5) stays the forward's fp32 decision (as plnerf_sample_pl_bwd: LABNOTES R6-13)
    double* cdf64 = reinterpret_cast<double*>(gc + B);      // B  (4 B + 4 N floats precede it: 8-byte aligned)
    for (int j = lane; j < B; j += 64) bins[j] = a.bins[(size_t)ray * B + j];
    for (int j = lane; j < n; j += 64) wv[j] = a.weights[(size_t)ray * n + j] + 1e-5f;
    __syncthreads();
    const float total = torch_row_sum(wv, n, lane);
    double total64 = 0.0;      // (the same row in fp64, from the fp32 weights: w + 1e-5 and its sum before any rounding)
    for (int j = lane; j < n; j += 64) total64 += (double)a.weights[(size_t)ray * n + j] + 1e-5;
    total64 = wave_sum(total64);
    double carry = 0.0, carry64 = 0.0;
    for (int base = 0; base < n; base += 64) {
        const int j = base + lane;
        const float pdf = (j < n) ? wv[j] / total : 0.0f;
        const double incl = wave_incl_sum((double)pdf);
        const double incl64 = wave_incl_sum((j < n) ? ((double)a.weights[(size_t)ray * n + j] + 1e-5) / total64 : 0.0);
        if (j < n) { cdf[j + 1] = (float)(carry + incl); cdf64[j + 1] = carry64 + incl64; }
        carry = carry + __shfl(incl, 63);
        carry64 = carry64 + __shfl(incl64, 63);
    }
    if (lane == 0) { cdf[0] = 0.0f; cdf64[0] = 0.0; }
    __syncthreads();
    const float* urow = a.u + (size_t)ray * a.u_row_stride;
    for (int k = lane; k < N; k += 64) {
        const size_t o = (size_t)ray * N + k;
        const float u = urow[k];
        const int ind = (int)a.inds[o];
        const int below = ind - 1 > 0 ? ind - 1 : 0;
        const int above = ind < B - 1 ? ind : B - 1;
        const float c0 = cdf[below], c1 = cdf[above];
        const float d = c1 - c0;
        const bool active = !(d < 1e-5f);
        const double c0d = cdf64[below];
        const double denom = active ? cdf64[above] - c0d : 1.0;
        const double gt = (double)a.g_samples[o] * ((double)bins[above] - (double)bins[below]);
        const double q = ((double)u - c0d) / denom;            // = t
        // dt/dc0 = -1/denom + [active] t/denom ;  dt/dc1 = -[active] t/denom
        g0[k] = (float)(gt * ((-1.0 / denom) + (active ? q / denom : 0.0)));
        g1[k] = active ? (float)(gt * (-(q / denom))) : 0.0f;
        lo[k] = below; hi[k] = above;
    }
    __syncthreads();
    for (int j = lane; j < B; j += 64) {
        float sacc = 0.0f;
        for (int k = 0; k < N; ++k) {
            if (lo[k] == j) sacc += g0[k];
            if (hi[k] == j) sacc += g1[k];
        }
        gc[j] = sacc;
    }
    __syncthreads();
    // g_pdf[i] = sum_{j > i} g_cdf[j]  (i = 0..n-1): suffix sums, scanned from the top in fp64
    double tail = 0.0, dot = 0.0;
    for (int base = 0; base < n; base += 64) {
        const int p = base + lane;                  // position from the top: i = n - 1 - p, adds g_cdf[i + 1]
        const int i = n - 1 - p;
        const double v = (p < n) ? (double)gc[i + 1] : 0.0;
        const double incl = wave_incl_sum(v);
        if (p < n) {
            const float gp = (float)(tail + incl);
            dot += (double)gp * (double)(wv[i] / total);   // sum_k g_pdf[k] pdf[k]
            cdf[i] = gp;                                      // g_pdf (the cdf row is no longer needed)
        }
        tail = tail + __shfl(incl, 63);
    }
    dot = wave_sum(dot);
    __syncthreads();
    if (!live) return;
    for (int i = lane; i < n; i += 64) a.g_weights[(size_t)ray * n + i] = (cdf[i] - (float)dot) / total;
}

struct SamplePlArgs {
    const float* z;
    const float* weights;
    const float* tau;
    const float* T;
    const float* near;
    const float* far;
    const float* u;
    int u_row_stride;
    int R, S, N;
    int lds_stride;
    float zero_tol, eps;
    float* samples;
    float* T_below;
    float* tau_below;
    float* bin_below;
    int64_t* inds;
};

__global__ __launch_bounds__(256) void sample_pl_kernel(SamplePlArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int ray = blockIdx.x * WAVES + wave;
    const bool live = ray < a.R;
    if (!live) ray = a.R - 1;
    const int S = a.S, K = a.S + 2;            // K knots / cdf entries
    float* cdf = smem + wave * a.lds_stride;
    float* knots = cdf + K;
    float* tau = knots + K;
    float* Tr = tau + K;
    for (int j = lane; j < K; j += 64) {
        tau[j] = a.tau[(size_t)ray * K + j];
        Tr[j] = a.T[(size_t)ray * K + j];
        float kn;
        if (j == 0) kn = a.near[ray];
        else if (j == K - 1) kn = a.far[ray];
        else kn = a.z[(size_t)ray * S + j - 1];
        knots[j] = kn;
    }
    // cdf = [0, cumsum(weights)] (fp64 running sum), last entry forced to 1
    double carry = 0.0;
    const int n = S + 1;
    for (int base = 0; base < n; base += 64) {
        const int j = base + lane;
        const float w = (j < n) ? a.weights[(size_t)ray * n + j] : 0.0f;
        const double incl = wave_incl_sum((double)w);
        if (j < n) cdf[j + 1] = (float)(carry + incl);
        carry = carry + __shfl(incl, 63);
    }
    __syncthreads();
    if (lane == 0) { cdf[0] = 0.0f; cdf[K - 1] = 1.0f; }
    __syncthreads();
    if (!live) return;
    const float* urow = a.u + (size_t)ray * a.u_row_stride;
    const float zt = a.zero_tol, eps = a.eps;
    for (int k = lane; k < a.N; k += 64) {
        const float u = urow[k];
        const int ind = upper_bound(cdf, K, u);
        const int below = ind - 1 > 0 ? ind - 1 : 0;
        const int above = ind < K - 1 ? ind : K - 1;
        const float s0 = knots[below], s1 = knots[above];
        const float T0 = Tr[below];
        const float tau0 = tau[below], tau1 = tau[above];
        const int di = below < S ? below : S;               // H4: reference reads out of bounds at u == 1
        const float d = tau[di + 1] - tau[di];
        float out = (d < zt && d > -zt) ? s0 : -1.0f;
        const bool rising = d >= zt;
        if (rising || d <= -zt) out = invert_segment(s0, s1, T0, tau0, tau1, u, eps, rising);      // (one evaluation, operands selected per lane)
        if (out != out) out = s0;
        const size_t o = (size_t)ray * a.N + k;
        a.samples[o] = out;
        if (a.T_below) a.T_below[o] = T0;
        if (a.tau_below) a.tau_below[o] = tau0;
        if (a.bin_below) a.bin_below[o] = s0;
        if (a.inds) a.inds[o] = (int64_t)ind;
    }
}


struct SamplePlBwdArgs {
    const float* z;
    const float* tau;
    const float* T;
    const float* near;
    const float* far;
    const float* u;
    int u_row_stride;
    const int64_t* inds;
    const float* g_samples;
    int R, S, N;
    int lds_stride;
    float zero_tol, eps;
    float* g_tau;
    float* g_T;
    float* g_knots;   // (nullable; plnerf_sample_pl_bwd_rays) [R,S+2]: gradient of the knots [near, z, far] themselves
};

// Gradient of invert_segment (above) with respect to (T0, tau0, tau1), following torch's rules for
// the guards: max(eps, x) passes the gradient to x where x > eps, clamp(t, lo, hi) where
// lo <= t <= hi.  Returns false when nothing flows (clamped).
// Which side of every guard the sample stands on is decided on the fp32 values the FORWARD computed (the same expressions, op
// for op) -- the gradient belongs to the branch the forward took.  The derivative's VALUE is then evaluated in fp64 from the
// same fp32 inputs: d t / d tau_r is the difference of two terms of size t / (tau_r - tau_l) that cancel to first order
// (-t / den against (L / den) ln / (span sqrt(disc))), and in fp32 -- the reference's autograd, and this kernel until round 6's
// third-seed campaign -- that leaves 1e-4 ... 1e-2 of max |g| on intervals with a small density difference (tools/fuzz_train_step_depth.py:
// the kernel 3-7x further from fp64 than torch's fp32 chain on 4 of 400 steps, 3x nearer on others).  A few dozen fp64 operations per
// hypothesis, R x N hypotheses per step: nothing next to the networks.  The result is the exact gradient of the forward's fp32 values
// to fp32 rounding (1e-7 of max |g|), where the reference's own is its fp32 chain's.
__device__ __forceinline__ bool invert_segment_grad(float s0, float s1, float T0, float tau0, float tau1,
                                                    float u, float eps, bool rising, float g, float& g_T0,
                                                    float& g_a, float& g_b) {
    g_T0 = g_a = g_b = 0.0f;
    bool disc_live, diff_live, ln_live;
    {   // the forward's fp32 values: predicates only
        const float L = s1 - s0;
        const float ratio = (1.0f - u) / tmax(eps, T0);
        const float ln_term = -logf(tmax(eps, ratio));
        const float span = tmax(eps, L);
        const float disc = tau0 * tau0 + (rising ? (2.0f * (tau1 - tau0) * ln_term) / span
                                                 : -((2.0f * (tau0 - tau1) * ln_term) / span));
        const float sq = sqrtf(tmax(eps, disc));
        const float diff = rising ? tau1 - tau0 : tau0 - tau1;
        const float den = tmax(eps, diff);
        const float t_raw = rising ? (L * (-tau0 + sq)) / den : (L * (tau0 - sq)) / den;
        if (!(t_raw >= eps && t_raw <= L)) return false;
        disc_live = disc > eps;
        diff_live = diff > eps;
        ln_live = ratio > eps && T0 > eps;
    }
    const double e = (double)eps, a0 = (double)tau0, a1 = (double)tau1, gd = (double)g;
    const double L = (double)s1 - (double)s0;
    const double m0 = (double)T0 > e ? (double)T0 : e;
    const double ratio = (1.0 - (double)u) / m0;
    const double ln_term = -log(ratio > e ? ratio : e);
    const double span = L > e ? L : e;
    const double diff = rising ? a1 - a0 : a0 - a1;
    const double disc = a0 * a0 + (rising ? (2.0 * (a1 - a0) * ln_term) / span : -((2.0 * (a0 - a1) * ln_term) / span));
    const double sq = sqrt(disc > e ? disc : e);
    const double den = diff > e ? diff : e;
    const double t_raw = rising ? (L * (-a0 + sq)) / den : (L * (a0 - sq)) / den;
    const double sgn = rising ? 1.0 : -1.0;
    const double g_sq = gd * sgn * (L / den);
    const double g_disc = disc_live ? g_sq * (0.5 / sq) : 0.0;
    const double g_den = diff_live ? -gd * (t_raw / den) : 0.0;      // d t_raw / d den = -t_raw / den
    // direct terms, the denominator (d den / d tau0 = -sgn, d den / d tau1 = +sgn), the discriminant
    g_a = (float)(gd * (-sgn) * (L / den) - sgn * g_den + g_disc * (2.0 * a0 - (2.0 * ln_term) / span));
    g_b = (float)(sgn * g_den + g_disc * ((2.0 * ln_term) / span));
    const double g_ln = g_disc * ((2.0 * (a1 - a0)) / span);
    // ln_term = -log(max(eps, (1-u) / max(eps, T0)))
    g_T0 = ln_live ? (float)(g_ln / (double)T0) : 0.0f;
    return true;
}

// d sample / d (s0, s1) of sample = s0 + clamp(t_raw(L), eps, L), L = s1 - s0, as autograd derives it for the reference
// (run_nerf_helpers.py:340-361; torch.clamp with a tensor bound: a value inside [eps, L] carries the gradient, beyond L --
// or when eps > L, where clamp returns its upper bound -- the bound does).  t_raw = L num / den depends on L directly and
// through the discriminant's 1 / span.  (Sides of the guards from the forward's fp32 values, the value in fp64: as above.)
__device__ __forceinline__ void invert_segment_knot_grad(float s0, float s1, float T0, float tau0, float tau1, float u,
                                                         float eps, bool rising, float g, float& g_s0, float& g_s1) {
    bool inner_live;
    {
        const float L = s1 - s0;
        const float ln_term = -logf(tmax(eps, (1.0f - u) / tmax(eps, T0)));
        const float span = tmax(eps, L);
        const float diff = rising ? tau1 - tau0 : tau0 - tau1;
        const float q = (2.0f * diff * ln_term) / span;
        const float disc = rising ? tau0 * tau0 + q : tau0 * tau0 - q;
        const float sq = sqrtf(tmax(eps, disc));
        const float den = tmax(eps, diff);
        const float num = rising ? -tau0 + sq : tau0 - sq;
        const float t_raw = (L * num) / den;
        if (eps > L || t_raw > L) { g_s0 = 0.0f; g_s1 = g; return; }      // sample = s0 + (s1 - s0)
        if (t_raw < eps) { g_s0 = g; g_s1 = 0.0f; return; }               // sample = s0 + eps
        inner_live = disc > eps && L > eps;
    }
    const double e = (double)eps, a0 = (double)tau0, a1 = (double)tau1;
    const double L = (double)s1 - (double)s0;
    const double m0 = (double)T0 > e ? (double)T0 : e;
    const double ratio = (1.0 - (double)u) / m0;
    const double ln_term = -log(ratio > e ? ratio : e);
    const double span = L > e ? L : e;
    const double diff = rising ? a1 - a0 : a0 - a1;
    const double q = (2.0 * diff * ln_term) / span;
    const double disc = rising ? a0 * a0 + q : a0 * a0 - q;
    const double sq = sqrt(disc > e ? disc : e);
    const double den = diff > e ? diff : e;
    const double num = rising ? -a0 + sq : a0 - sq;
    double dt = num / den;
    if (inner_live) dt -= (L / den) * ((0.5 / sq) * (q / span));
    g_s1 = (float)((double)g * dt);
    g_s0 = g - g_s1;
}

__global__ __launch_bounds__(256) void sample_pl_bwd_kernel(SamplePlBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int ray = blockIdx.x * WAVES + wave;
    const bool live = ray < a.R;
    if (!live) ray = a.R - 1;
    const int S = a.S, K = a.S + 2, N = a.N;
    float* knots = smem + wave * a.lds_stride;
    float* tau = knots + K;
    float* Tr = tau + K;
    float* gT0 = Tr + K;          // per sample: gradient landing on T[below]
    float* ga = gT0 + N;          //             on tau[below]
    float* gb = ga + N;           //             on tau[above]
    int* lo = reinterpret_cast<int*>(gb + N);   // below (above = min(below + 1, K - 1) unless clamped at 0)
    int* hi = lo + N;
    float* gs0 = reinterpret_cast<float*>(hi + N);      // (only with g_knots) on knot[below]
    float* gs1 = gs0 + N;                               //                     on knot[above]
    for (int j = lane; j < K; j += 64) {
        tau[j] = a.tau[(size_t)ray * K + j];
        Tr[j] = a.T[(size_t)ray * K + j];
        float kn;
        if (j == 0) kn = a.near[ray];
        else if (j == K - 1) kn = a.far[ray];
        else kn = a.z[(size_t)ray * S + j - 1];
        knots[j] = kn;
    }
    __syncthreads();
    const float* urow = a.u + (size_t)ray * a.u_row_stride;
    const float zt = a.zero_tol, eps = a.eps;
    for (int k = lane; k < N; k += 64) {
        const size_t o = (size_t)ray * N + k;
        const float u = urow[k];
        const int ind = (int)a.inds[o];
        const int below = ind - 1 > 0 ? ind - 1 : 0;
        const int above = ind < K - 1 ? ind : K - 1;
        const float s0 = knots[below], s1 = knots[above];
        const int di = below < S ? below : S;
        const float d = tau[di + 1] - tau[di];
        float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;
        const float g = a.g_samples[o];
        float k0 = g, k1 = 0.0f;      // (a flat interval and a NaN sample are the left knot itself, run_nerf_helpers.py:425, 432)
        if (d >= zt || d <= -zt) {
            const bool rising = d >= zt;
            // (a NaN sample falls back to s_left in the forward: no gradient)
            const float t = invert_segment(s0, s1, Tr[below], tau[below], tau[above], u, eps, rising);
            if (t == t) {
                invert_segment_grad(s0, s1, Tr[below], tau[below], tau[above], u, eps, rising, g, g0, g1, g2);
                if (a.g_knots) invert_segment_knot_grad(s0, s1, Tr[below], tau[below], tau[above], u, eps, rising, g, k0, k1);
            }
        }
        gT0[k] = g0; ga[k] = g1; gb[k] = g2;
        if (a.g_knots) { gs0[k] = k0; gs1[k] = k1; }
        lo[k] = below; hi[k] = above;
    }
    __syncthreads();
    if (!live) return;
    // per-knot sums in sample order (deterministic); all reads are LDS broadcasts
    for (int j = lane; j < K; j += 64) {
        float st = 0.0f, sT = 0.0f;
        for (int k = 0; k < N; ++k) {
            const int b = lo[k], t = hi[k];
            if (b == j) { sT += gT0[k]; st += ga[k]; }
            if (t == j) st += gb[k];
        }
        a.g_tau[(size_t)ray * K + j] = st;
        a.g_T[(size_t)ray * K + j] = sT;
        if (a.g_knots) {
            float sk = 0.0f;
            for (int k = 0; k < N; ++k) {
                if (lo[k] == j) sk += gs0[k];
                if (hi[k] == j) sk += gs1[k];
            }
            a.g_knots[(size_t)ray * K + j] = sk;
        }
    }
}

struct MergeArgs {
    const float* z;
    const float* z_new;
    const float* near;
    const float* far;
    int R, S, N;
    int lds_stride;
    float* out;
};

// Bitonic sort of one ray's keys in REGISTERS, one wavefront per ray: position p = 64 r + lane holds key x[r],
// the row is padded with the maximum key to 64 KPL.  A compare-exchange distance j >= 64 pairs two registers of
// the same lane; j < 64 pairs lanes l and l ^ j (one cross-lane read per key).  log2(n)(log2(n)+1)/2 stages of
// one min/max per key: ~540 instructions per lane at n = 192 where the rank sort it replaces needed ~1340
// (and no LDS).  Values are sorted as torch.sort sorts them (ascending, NaN last); ties are values, so order
// among equals does not show.
template <int KPL>
__global__ __launch_bounds__(256) void merge_sort_kernel(MergeArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int ray = blockIdx.x * WAVES + wave;
    const bool live = ray < a.R;
    if (!live) ray = a.R - 1;
    const int S = a.S, N = a.N, n = S + N;
    const float lo = a.near[ray], hi = a.far[ray];
    uint32_t x[KPL];
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
        const int p = 64 * r + lane;
        uint32_t k = 0xFFFFFFFFu;
        if (p < S) k = sort_key(a.z[(size_t)ray * S + p]);
        else if (p < n) k = sort_key(tmin(tmax(a.z_new[(size_t)ray * N + (p - S)], lo), hi));
        x[r] = k;
    }
    bitonic_sort_regs<KPL>(x, lane);
    if (!live) return;
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
        const int p = 64 * r + lane;
        if (p < n) a.out[(size_t)ray * n + p] = sort_unkey(x[r]);
    }
}

inline int set_lds(const void* fn, size_t lds) {
    if (lds > 160 * 1024) return PLNERF_ERANGE;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return PLNERF_OK;
}

}  // namespace

extern "C" int plnerf_sample_const(const float* bins, const float* weights, const float* u,
                                   int u_row_stride, int R, int B, int N, float* samples, int64_t* inds,
                                   plnerf_stream_t stream) {
    if (!bins || !weights || !u || !samples) return PLNERF_EINVAL;
    if (R < 0 || B < 2 || N < 1 || (u_row_stride != 0 && u_row_stride != N)) return PLNERF_EINVAL;
    if (B > PLNERF_MAX_SAMPLES + 2) return PLNERF_ERANGE;
    if (R == 0) return PLNERF_OK;
    SampleConstArgs a{bins, weights, u, u_row_stride, R, B, N, 0, samples, inds};
    a.lds_stride = ((3 * B) + 3) & ~3;
    const size_t lds = (size_t)WAVES * a.lds_stride * sizeof(float);
    int rc = set_lds((const void*)sample_const_kernel, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(sample_const_kernel, dim3((R + WAVES - 1) / WAVES), dim3(WAVES * 64), lds,
                       (hipStream_t)stream, a);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}


extern "C" int plnerf_sample_const_bwd(const float* bins, const float* weights, const float* u, int u_row_stride,
                                       const int64_t* inds, const float* g_samples, int R, int B, int N,
                                       float* g_weights, plnerf_stream_t stream) {
    if (R < 0 || B < 2 || N < 1 || (u_row_stride != 0 && u_row_stride != N)) return PLNERF_EINVAL;
    if (B > PLNERF_MAX_SAMPLES + 2) return PLNERF_ERANGE;
    if (R == 0) return PLNERF_OK;
    if (!bins || !weights || !u || !inds || !g_samples || !g_weights) return PLNERF_EINVAL;
    SampleConstBwdArgs a{bins, weights, u, u_row_stride, inds, g_samples, R, B, N, 0, g_weights};
    a.lds_stride = ((6 * B + 4 * N) + 3) & ~3;      // (+ the fp64 cdf)
    const size_t lds = (size_t)WAVES * a.lds_stride * sizeof(float);
    int rc = set_lds((const void*)sample_const_bwd_kernel, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(sample_const_bwd_kernel, dim3((R + WAVES - 1) / WAVES), dim3(WAVES * 64), lds,
                       (hipStream_t)stream, a);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

extern "C" int plnerf_sample_pl(const float* z, const float* weights, const float* tau, const float* T,
                                const float* near, const float* far, const float* u, int u_row_stride,
                                int R, int S, int N, float zero_tol, float epsilon, float* samples,
                                float* T_below, float* tau_below, float* bin_below, int64_t* inds,
                                plnerf_stream_t stream) {
    if (!z || !weights || !tau || !T || !near || !far || !u || !samples) return PLNERF_EINVAL;
    if (R < 0 || S < 1 || N < 1 || (u_row_stride != 0 && u_row_stride != N)) return PLNERF_EINVAL;
    if (S > PLNERF_MAX_SAMPLES) return PLNERF_ERANGE;
    if (R == 0) return PLNERF_OK;
    SamplePlArgs a{z, weights, tau, T, near, far, u, u_row_stride, R, S, N, 0, zero_tol, epsilon,
                   samples, T_below, tau_below, bin_below, inds};
    a.lds_stride = ((4 * (S + 2)) + 3) & ~3;
    const size_t lds = (size_t)WAVES * a.lds_stride * sizeof(float);
    int rc = set_lds((const void*)sample_pl_kernel, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(sample_pl_kernel, dim3((R + WAVES - 1) / WAVES), dim3(WAVES * 64), lds,
                       (hipStream_t)stream, a);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}


namespace {
int sample_pl_bwd_launch(const float* z, const float* tau, const float* T, const float* near,
                         const float* far, const float* u, int u_row_stride, const int64_t* inds,
                         const float* g_samples, int R, int S, int N, float zero_tol, float epsilon,
                         float* g_tau, float* g_T, float* g_knots, plnerf_stream_t stream) {
    if (R < 0 || S < 1 || N < 1 || (u_row_stride != 0 && u_row_stride != N)) return PLNERF_EINVAL;
    if (S > PLNERF_MAX_SAMPLES) return PLNERF_ERANGE;
    if (R == 0) return PLNERF_OK;
    if (!z || !tau || !T || !near || !far || !u || !inds || !g_samples || !g_tau || !g_T) return PLNERF_EINVAL;
    SamplePlBwdArgs a{z, tau, T, near, far, u, u_row_stride, inds, g_samples, R, S, N, 0, zero_tol, epsilon,
                      g_tau, g_T, g_knots};
    a.lds_stride = ((3 * (S + 2) + (g_knots ? 7 : 5) * N) + 3) & ~3;
    const size_t lds = (size_t)WAVES * a.lds_stride * sizeof(float);
    int rc = set_lds((const void*)sample_pl_bwd_kernel, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(sample_pl_bwd_kernel, dim3((R + WAVES - 1) / WAVES), dim3(WAVES * 64), lds,
                       (hipStream_t)stream, a);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}
}  // namespace

extern "C" int plnerf_sample_pl_bwd(const float* z, const float* tau, const float* T, const float* near,
                                    const float* far, const float* u, int u_row_stride, const int64_t* inds,
                                    const float* g_samples, int R, int S, int N, float zero_tol, float epsilon,
                                    float* g_tau, float* g_T, plnerf_stream_t stream) {
    return sample_pl_bwd_launch(z, tau, T, near, far, u, u_row_stride, inds, g_samples, R, S, N, zero_tol, epsilon, g_tau, g_T,
                                nullptr, stream);
}

extern "C" int plnerf_sample_pl_bwd_rays(const float* z, const float* tau, const float* T, const float* near,
                                         const float* far, const float* u, int u_row_stride, const int64_t* inds,
                                         const float* g_samples, int R, int S, int N, float zero_tol, float epsilon,
                                         float* g_tau, float* g_T, float* g_knots, plnerf_stream_t stream) {
    if (R > 0 && !g_knots) return PLNERF_EINVAL;
    return sample_pl_bwd_launch(z, tau, T, near, far, u, u_row_stride, inds, g_samples, R, S, N, zero_tol, epsilon, g_tau, g_T,
                                g_knots, stream);
}

extern "C" int plnerf_merge_sort(const float* z, const float* z_new, const float* near, const float* far,
                                 int R, int S, int N, float* out, plnerf_stream_t stream) {
    if (!z || !z_new || !near || !far || !out) return PLNERF_EINVAL;
    if (R < 0 || S < 0 || N < 0 || S + N < 1) return PLNERF_EINVAL;
    if (S + N > 1024) return PLNERF_ERANGE;
    if (R == 0) return PLNERF_OK;
    MergeArgs a{z, z_new, near, far, R, S, N, 0, out};
    const dim3 grid((R + WAVES - 1) / WAVES), block(WAVES * 64);
    const int n = S + N;
    if (n <= 64) hipLaunchKernelGGL(merge_sort_kernel<1>, grid, block, 0, (hipStream_t)stream, a);
    else if (n <= 128) hipLaunchKernelGGL(merge_sort_kernel<2>, grid, block, 0, (hipStream_t)stream, a);
    else if (n <= 256) hipLaunchKernelGGL(merge_sort_kernel<4>, grid, block, 0, (hipStream_t)stream, a);
    else if (n <= 512) hipLaunchKernelGGL(merge_sort_kernel<8>, grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(merge_sort_kernel<16>, grid, block, 0, (hipStream_t)stream, a);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

// ------------------------------------------------------------------------------------
// Coarse sample depths (run_plnerf.py:683-705) and sample positions (:708, :735) of a ray batch: the dozen
// element-wise torch launches of the reference's prologue as two kernels.  Every product and sum is rounded
// separately, in the reference's order (the file is built with -ffp-contract=off), so z and pts are bit-identical
// to the torch expressions.
// ------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ float coarse_depth(const float nr, const float fr, const float t, const int lindisp) {
    const float omt = 1.0f - t;
    if (!lindisp) return nr * omt + fr * t;                 // near * (1 - t) + far * t
    return 1.0f / (1.0f / nr * omt + 1.0f / fr * t);        // 1 / (1/near * (1 - t) + 1/far * t)
}

__global__ __launch_bounds__(256) void stratified_z_kernel(const float* __restrict__ near, const float* __restrict__ far,
                                                           const float* __restrict__ t_vals,
                                                           const float* __restrict__ t_rand, const int R, const int S,
                                                           const int lindisp, float* __restrict__ z_out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)R * S) return;
    const int r = (int)(idx / S), s = (int)(idx - (size_t)r * S);
    const float nr = near[r], fr = far[r];
    const float z = coarse_depth(nr, fr, t_vals[s], lindisp);
    if (!t_rand) { z_out[idx] = z; return; }
    // mids = .5 * (z[1:] + z[:-1]); upper = [mids, z[-1]]; lower = [z[0], mids]; z = lower + (upper - lower) * t_rand
    const float zl = s > 0 ? coarse_depth(nr, fr, t_vals[s - 1], lindisp) : z;
    const float zu = s + 1 < S ? coarse_depth(nr, fr, t_vals[s + 1], lindisp) : z;
    const float lower = s > 0 ? 0.5f * (z + zl) : z;
    const float upper = s + 1 < S ? 0.5f * (zu + z) : z;
    z_out[idx] = lower + (upper - lower) * t_rand[idx];
}

__global__ __launch_bounds__(256) void ray_points_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                         const float* __restrict__ z, const int R, const int S,
                                                         float* __restrict__ pts) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // one (ray, sample, coordinate)
    if (idx >= (size_t)R * S * 3) return;
    const size_t rs = idx / 3;
    const int c = (int)(idx - rs * 3);
    const int r = (int)(rs / S);
    pts[idx] = rays_o[3 * (size_t)r + c] + rays_d[3 * (size_t)r + c] * z[rs];
}

// the same, four consecutive output floats per thread (one 16-byte store); needs 3 S % 4 == 0
__global__ __launch_bounds__(256) void ray_points4_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                          const float* __restrict__ z, const int R, const int S,
                                                          float4* __restrict__ pts) {
    const size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int per_ray = 3 * S / 4;
    if (i4 >= (size_t)R * per_ray) return;
    const int r = (int)(i4 / per_ray);
    const int e0 = (int)(i4 - (size_t)r * per_ray) * 4;                     // first element within the ray's S x 3 block
    const float o[3] = {rays_o[3 * (size_t)r], rays_o[3 * (size_t)r + 1], rays_o[3 * (size_t)r + 2]};
    const float d[3] = {rays_d[3 * (size_t)r], rays_d[3 * (size_t)r + 1], rays_d[3 * (size_t)r + 2]};
    const float* zr = z + (size_t)r * S;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = e0 + k, sidx = e / 3, c = e - 3 * sidx;
        const float oc = c == 0 ? o[0] : (c == 1 ? o[1] : o[2]);
        const float dc = c == 0 ? d[0] : (c == 1 ? d[1] : d[2]);
        v[k] = oc + dc * zr[sidx];
    }
    pts[i4] = make_float4(v[0], v[1], v[2], v[3]);
}

}  // namespace

extern "C" int plnerf_stratified_z(const float* near, const float* far, const float* t_vals, const float* t_rand,
                                   int R, int S, int lindisp, float* z_vals, plnerf_stream_t stream) {
    if (R < 0 || S < 1) return PLNERF_EINVAL;
    if (R == 0) return PLNERF_OK;
    if (!near || !far || !t_vals || !z_vals) return PLNERF_EINVAL;
    const size_t n = (size_t)R * S;
    hipLaunchKernelGGL(stratified_z_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, near,
                       far, t_vals, t_rand, R, S, lindisp, z_vals);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

extern "C" int plnerf_ray_points(const float* rays_o, const float* rays_d, const float* z_vals, int R, int S,
                                 float* pts, plnerf_stream_t stream) {
    if (R < 0 || S < 1) return PLNERF_EINVAL;
    if (R == 0) return PLNERF_OK;
    if (!rays_o || !rays_d || !z_vals || !pts) return PLNERF_EINVAL;
    const size_t n = (size_t)R * S * 3;
    if ((3 * S) % 4 == 0 && ((uintptr_t)pts & 15) == 0)
        hipLaunchKernelGGL(ray_points4_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           rays_o, rays_d, z_vals, R, S, (float4*)pts);
    else
        hipLaunchKernelGGL(ray_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rays_o,
                           rays_d, z_vals, R, S, pts);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}
