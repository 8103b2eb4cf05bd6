// Quadrature kernels: raw2outputs forward and its backward with respect to `raw`.
//
// Reference semantics: run_plnerf.py:553-624 (raw2outputs), :516-550
// (compute_weights_piecewise_linear), :504-513 (compute_weights).  HBM-bound streaming
// scan: one 64-lane wavefront owns one ray, reads its (z, raw) rows with coalesced
// 16-byte loads, keeps the ray's knots in LDS, and does the transmittance prefix product
// as a wave scan (fp64 carry, like the reference's CPU cumprod) -- nothing is re-read
// from HBM.  Algorithmic bytes per ray (linear): 32*S+64.
#include "common.h"
#include "ray_dev.h"

using namespace plnerf;

namespace {

constexpr int WAVES = 4;  // rays per 256-thread workgroup

struct QuadArgs {
    const float* raw;
    const float* z;
    const float* near;
    const float* far;
    const float* rays_d;
    const float* noise;
    int R, S;
    int color_mode, white_bkgd, farcolorfix;
    int lds_stride;  // floats per wave
    // forward outputs
    float* rgb_map;
    float* disp_map;
    float* acc_map;
    float* depth_map;
    float* weights;
    float* tau;
    float* T;
    // backward inputs / output
    const float* g_rgb;
    const float* g_depth;
    const float* g_acc;
    const float* g_weights;
    const float* g_tau;   // linear mode: upstream gradient of the returned tau [R,S+2] (or nullptr)
    const float* g_T;     // linear mode: upstream gradient of the returned T   [R,S+2] (or nullptr)
    float* g_raw;
    // backward by-product (may be null): max |g_raw| of each workgroup's rays as fp32 bits, [ceil(R / WAVES)] -- the
    // candidates of the half dz planes' launch scale, which plnerf_mlp_bwd otherwise finds with a pass of its own over g_raw
    unsigned* absmax_out;
    // backward, the ray geometry's gradient (all four or none; plnerf_quad_bwd_rays): what autograd gives the reference
    // for z_vals [R,S], near, far [R] (the outer knots of the piecewise-linear rule; zero in constant mode) and for
    // |rays_d| [R], which scales every interval
    float* g_z;
    float* g_near;
    float* g_far;
    float* g_dnorm;
};

template <int MODE>
__global__ __launch_bounds__(256) void quad_fwd_kernel(QuadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int ray = blockIdx.x * WAVES + wave;
    const bool live = ray < a.R;
    if (!live) ray = a.R - 1;
    const int S = a.S;
    float* zk = smem + wave * a.lds_stride;
    float* tau = zk + (S + 2);
    float* col = tau + (S + 2);
    float dnorm;
    load_ray(RayIn{a.raw, a.z, a.near, a.far, a.rays_d, a.noise, a.S}, ray, lane, zk, tau, col, dnorm);
    __syncthreads();

    const int n = (MODE == PLNERF_MODE_LINEAR) ? S + 1 : S;
    double carry = 1.0;
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
        if (valid) {
            const float w = (1.0f - e) * Ti;
            sr += (double)(w * elem_colour<MODE>(i, 0, S, col, a.color_mode, a.farcolorfix));
            sg += (double)(w * elem_colour<MODE>(i, 1, S, col, a.color_mode, a.farcolorfix));
            sb += (double)(w * elem_colour<MODE>(i, 2, S, col, a.color_mode, a.farcolorfix));
            sd += (double)(w * elem_depth<MODE>(i, zk));
            sa += (double)w;
            if (live) {
                if (a.weights) a.weights[(size_t)ray * n + i] = w;
                if (MODE == PLNERF_MODE_LINEAR && a.T) a.T[(size_t)ray * (S + 2) + i + 1] = Tn;
            }
        }
    }
    if (MODE == PLNERF_MODE_LINEAR && live) {
        if (a.T && lane == 0) a.T[(size_t)ray * (S + 2)] = 1.0f;
        if (a.tau)
            for (int s = lane; s < S + 2; s += 64) a.tau[(size_t)ray * (S + 2) + s] = tau[s];
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
}

// Backward.  With w_i = (1-e_i) T_i and T_i = prod_{j<i} f_j:
//   dL/de_k = T_k (X_k - G_k),  X_k = sum_{i>k} G_i (1-e_i) prod_{k<j<i} f_j
// (division-free, so exact even when some e_k underflows to 0, like autograd's cumprod
// backward).  X obeys the reverse recurrence X_k = G_{k+1}(1-e_{k+1}) + f_{k+1} X_{k+1},
// evaluated as a wave scan over affine maps.
// The returned transmittances T_m = prod_{j<m} f_j (m = 1..S+1) can carry their own upstream
// gradient H_m (the depth-supervised variant differentiates through the sampler,
// depth_supervised_exps/run_nerf_sample_based_depth.py:923-934): that adds T_k Y_k to dL/de_k with
// Y_k = H_{k+1} + f_{k+1} Y_{k+1}, the same recurrence -- so H_m simply joins the additive term.
// The returned tau[s+1] = relu(sigma_s + noise_s) passes its upstream gradient straight to sigma_s.
template <int MODE>
__global__ __launch_bounds__(256) void quad_bwd_kernel(QuadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ unsigned wave_max_bits[WAVES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int ray = blockIdx.x * WAVES + wave;
    const bool live = ray < a.R;
    if (!live) ray = a.R - 1;
    const int S = a.S;
    const int n = (MODE == PLNERF_MODE_LINEAR) ? S + 1 : S;
    float* zk = smem + wave * a.lds_stride;
    float* tau = zk + (S + 2);
    float* col = tau + (S + 2);
    float* fv = col + 3 * S;     // f_i (fv[n] = 1)
    float* av = fv + (n + 1);    // G_i (1 - e_i)  (av[n] = 0)
    float* wv = av + (n + 1);    // w_i
    float* qv = wv + (n + 1);    // T_i, then Q_i = dL/de_i * seg_i * e_i
    float* sv = qv + (n + 1);    // (only with g_z) dL/dseg_i = dL/de_i * e_i * (-density of the interval)
    float dnorm;
    load_ray(RayIn{a.raw, a.z, a.near, a.far, a.rays_d, a.noise, a.S}, ray, lane, zk, tau, col, dnorm);
    const float gr = a.g_rgb[3 * ray + 0], gg = a.g_rgb[3 * ray + 1], gb = a.g_rgb[3 * ray + 2];
    const float gdep = a.g_depth ? a.g_depth[ray] : 0.0f;
    float gacc = a.g_acc ? a.g_acc[ray] : 0.0f;
    if (a.white_bkgd) gacc -= (gr + gg + gb);
    __syncthreads();

    // pass 1: forward scan, stash per-element terms
    double carry = 1.0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const bool valid = i < n;
        float seg = 0.f, e = 1.f, f = 1.f;
        if (valid) interval<MODE>(i, S, zk, tau, dnorm, seg, e, f);
        const double incl = wave_incl_prod((double)f);
        double excl = __shfl_up(incl, 1);
        if (lane == 0) excl = 1.0;
        const float Ti = (float)(carry * excl);
        carry = carry * __shfl(incl, 63);
        if (valid) {
            float G = gr * elem_colour<MODE>(i, 0, S, col, a.color_mode, a.farcolorfix) +
                      gg * elem_colour<MODE>(i, 1, S, col, a.color_mode, a.farcolorfix) +
                      gb * elem_colour<MODE>(i, 2, S, col, a.color_mode, a.farcolorfix) +
                      gdep * elem_depth<MODE>(i, zk) + gacc;
            if (a.g_weights) G += a.g_weights[(size_t)ray * n + i];
            fv[i] = f;
            av[i] = G * (1.0f - e) + ((MODE == PLNERF_MODE_LINEAR && a.g_T) ? a.g_T[(size_t)ray * (S + 2) + i] : 0.0f);
            wv[i] = (1.0f - e) * Ti;
            qv[i] = Ti;   // G_i and seg_i are recomputed in pass 2 (cheaper than two more LDS rows)
        }
    }
    if (lane == 0) {
        fv[n] = 1.0f;
        av[n] = (MODE == PLNERF_MODE_LINEAR && a.g_T) ? a.g_T[(size_t)ray * (S + 2) + n] : 0.0f;
    }
    __syncthreads();

    // pass 2: reverse affine scan.  Position p = n-1-i ascending <=> i descending;
    // y_p = A_p + F_p y_{p-1} with A_p = av[i+1], F_p = fv[i+1], y_{-1} = 0.
    float ycarry = 0.0f;
    for (int base = 0; base < n; base += 64) {
        const int p = base + lane;
        const bool valid = p < n;
        const int i = n - 1 - p;
        float A = 0.0f, F = 1.0f;
        if (valid) { A = av[i + 1]; F = fv[i + 1]; }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float Alo = __shfl_up(A, d), Flo = __shfl_up(F, d);
            if (lane >= d) { A = A + F * Alo; F = F * Flo; }
        }
        const float X = A + F * ycarry;            // X_i
        ycarry = __shfl(X, 63);
        if (valid) {
            float seg, e, f;
            interval<MODE>(i, S, zk, tau, dnorm, seg, e, f);
            // (G_i = av[i] / (1-e_i) would divide by zero when e_i == 1: recompute it)
            float G = gr * elem_colour<MODE>(i, 0, S, col, a.color_mode, a.farcolorfix) +
                      gg * elem_colour<MODE>(i, 1, S, col, a.color_mode, a.farcolorfix) +
                      gb * elem_colour<MODE>(i, 2, S, col, a.color_mode, a.farcolorfix) +
                      gdep * elem_depth<MODE>(i, zk) + gacc;
            if (a.g_weights) G += a.g_weights[(size_t)ray * n + i];
            const float Ti = qv[i];
            const float dLde = Ti * (X - G);
            qv[i] = dLde * seg * e;                // Q_i
            if (a.g_z) {                           // e_i = exp(-dens_i seg_i)
                const float dens = (MODE == PLNERF_MODE_LINEAR) ? 0.5f * (tau[i + 1] + tau[i]) : tau[i + 1];
                sv[i] = (dLde * e) * (-dens);
            }
        }
    }
    __syncthreads();

    // pass 3: per-sample gradients
    if (live) {
        float4* out = reinterpret_cast<float4*>(a.g_raw) + (size_t)ray * S;
        float gmax = 0.0f;
        for (int s = lane; s < S; s += 64) {
            float gtau, coef;
            if (MODE == PLNERF_MODE_LINEAR) {
                gtau = -0.5f * (qv[s + 1] + qv[s]);
                if (a.g_tau) gtau += a.g_tau[(size_t)ray * (S + 2) + s + 1];
                if (a.color_mode == PLNERF_COLOR_MIDPOINT) {
                    coef = 0.5f * (wv[s] + wv[s + 1]);
                    if (s == 0) coef += 0.5f * wv[0];
                    if (s == S - 1 && !a.farcolorfix) coef += 0.5f * wv[S];
                } else {
                    coef = wv[s + 1];
                    if (s == 0) coef += wv[0];
                }
            } else {
                gtau = -qv[s];
                coef = wv[s];
            }
            const float c0 = col[3 * s + 0], c1 = col[3 * s + 1], c2 = col[3 * s + 2];
            float4 g;
            g.x = gr * coef * (c0 * (1.0f - c0));
            g.y = gg * coef * (c1 * (1.0f - c1));
            g.z = gb * coef * (c2 * (1.0f - c2));
            g.w = (tau[s + 1] > 0.0f) ? gtau : 0.0f;
            out[s] = g;
            const float c[4] = {fabsf(g.x), fabsf(g.y), fabsf(g.z), fabsf(g.w)};
#pragma unroll
            for (int k = 0; k < 4; ++k) gmax = (c[k] > gmax || c[k] != c[k]) ? c[k] : gmax;   // a NaN sticks (and orders above every float as bits)
        }
        if (a.absmax_out) {      // (uniform)
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const float o = __shfl_xor(gmax, d);
                gmax = (o > gmax || o != o) ? o : gmax;
            }
            if (lane == 0) wave_max_bits[wave] = __float_as_uint(gmax);
        }
    }
    if (a.g_z && live) {
        // seg_i = (knot_{i+1} - knot_i) |d|, the depth map weighs the elements' depths: a knot collects from the two
        // elements it bounds.  Linear: knots [near, z, far], element i between knots i and i + 1, depth = their mean.
        // Constant: element i from z_i to z_{i+1} (the last one is 1e10 |d| long), depth = z_i.
        double gdn = 0.0;      // (up to 1023 terms that cancel: fp64, rounded once -- off every training path)
        if (MODE == PLNERF_MODE_LINEAR) {
            for (int k = lane; k < S + 2; k += 64) {
                const float s_lo = k > 0 ? sv[k - 1] : 0.0f, s_hi = k <= S ? sv[k] : 0.0f;
                const float w_lo = k > 0 ? wv[k - 1] : 0.0f, w_hi = k <= S ? wv[k] : 0.0f;
                const float g = dnorm * (s_lo - s_hi) + gdep * (0.5f * (w_lo + w_hi));
                if (k == 0) a.g_near[ray] = g;
                else if (k == S + 1) a.g_far[ray] = g;
                else a.g_z[(size_t)ray * S + k - 1] = g;
                if (k <= S) gdn += (double)sv[k] * (double)(zk[k + 1] - zk[k]);
            }
        } else {
            for (int k = lane; k < S; k += 64) {
                const float s_lo = k > 0 ? sv[k - 1] : 0.0f, s_hi = k < S - 1 ? sv[k] : 0.0f;
                a.g_z[(size_t)ray * S + k] = dnorm * (s_lo - s_hi) + gdep * wv[k];
                gdn += (double)sv[k] * (double)(k < S - 1 ? zk[k + 2] - zk[k + 1] : 1e10f);
            }
            if (lane == 0) { a.g_near[ray] = 0.0f; a.g_far[ray] = 0.0f; }
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) gdn += __shfl_xor(gdn, d);
        if (lane == 0) a.g_dnorm[ray] = (float)gdn;
    }
    if (a.absmax_out) {
        // One plain store per workgroup, no atomic: 4096 atomicMax on one address cost the launch 46 us of serialised L2
        // round trips (14 -> 60 us, round 5) -- every workgroup of this grid is resident at once, so "skip if the word
        // already holds more" skips nothing.  The consumer (the dgrad kernel's prologue) takes the maximum of the array.
        if (!live && lane == 0) wave_max_bits[wave] = 0u;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned m = wave_max_bits[0];      // (non-negative floats and NaNs order as unsigned integers; a NaN sticks)
#pragma unroll
            for (int w = 1; w < WAVES; ++w) m = wave_max_bits[w] > m ? wave_max_bits[w] : m;
            a.absmax_out[blockIdx.x] = m;
        }
    }
}

int check_common(const float* raw, const float* z, const float* near, const float* far,
                 const float* rays_d, int R, int S, int mode, int color_mode) {
    if (R < 0 || S < 2) return PLNERF_EINVAL;
    if (R > 0 && (!raw || !z || !near || !far || !rays_d)) return PLNERF_EINVAL;
    if (S > PLNERF_MAX_SAMPLES) return PLNERF_ERANGE;
    if (mode != PLNERF_MODE_LINEAR && mode != PLNERF_MODE_CONSTANT) return PLNERF_EINVAL;
    if (color_mode != PLNERF_COLOR_MIDPOINT && color_mode != PLNERF_COLOR_LEFT) return PLNERF_EINVAL;
    return PLNERF_OK;
}

}  // namespace

extern "C" int plnerf_quad_fwd(const float* raw, const float* z, const float* near, const float* far,
                               const float* rays_d, const float* noise, int R, int S, int mode,
                               int color_mode, int white_bkgd, int farcolorfix, float* rgb_map,
                               float* disp_map, float* acc_map, float* depth_map, float* weights,
                               float* tau, float* T, plnerf_stream_t stream) {
    int rc = check_common(raw, z, near, far, rays_d, R, S, mode, color_mode);
    if (rc) return rc;
    if (R == 0) return PLNERF_OK;
    if (!rgb_map || !disp_map || !acc_map || !depth_map) return PLNERF_EINVAL;
    QuadArgs a{};
    a.raw = raw; a.z = z; a.near = near; a.far = far; a.rays_d = rays_d; a.noise = noise;
    a.R = R; a.S = S; a.color_mode = color_mode; a.white_bkgd = white_bkgd; a.farcolorfix = farcolorfix;
    a.rgb_map = rgb_map; a.disp_map = disp_map; a.acc_map = acc_map; a.depth_map = depth_map;
    a.weights = weights; a.tau = tau; a.T = T;
    a.lds_stride = ((5 * S + 4) + 3) & ~3;
    const size_t lds = (size_t)WAVES * a.lds_stride * sizeof(float);
    dim3 grid((R + WAVES - 1) / WAVES), block(WAVES * 64);
    hipStream_t st = (hipStream_t)stream;
    if (lds > 160 * 1024) return PLNERF_ERANGE;
    if (lds > 64 * 1024) {
        if (mode == PLNERF_MODE_LINEAR)
            (void)hipFuncSetAttribute((const void*)quad_fwd_kernel<PLNERF_MODE_LINEAR>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        else
            (void)hipFuncSetAttribute((const void*)quad_fwd_kernel<PLNERF_MODE_CONSTANT>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (mode == PLNERF_MODE_LINEAR)
        hipLaunchKernelGGL(quad_fwd_kernel<PLNERF_MODE_LINEAR>, grid, block, lds, st, a);
    else
        hipLaunchKernelGGL(quad_fwd_kernel<PLNERF_MODE_CONSTANT>, grid, block, lds, st, a);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

namespace {
int quad_bwd_launch(const float* raw, const float* z, const float* near, const float* far, const float* rays_d,
                    const float* noise, int R, int S, int mode, int color_mode, int white_bkgd, int farcolorfix,
                    const float* g_rgb, const float* g_depth, const float* g_acc, const float* g_weights,
                    const float* g_tau, const float* g_T, float* g_raw, uint32_t* absmax_out, float* g_z, float* g_near,
                    float* g_far, float* g_dnorm, plnerf_stream_t stream) {
    int rc = check_common(raw, z, near, far, rays_d, R, S, mode, color_mode);
    if (rc) return rc;
    if (R == 0) return PLNERF_OK;
    if (!g_rgb || !g_raw) return PLNERF_EINVAL;
    if ((g_tau || g_T) && mode != PLNERF_MODE_LINEAR) return PLNERF_EINVAL;
    QuadArgs a{};
    a.raw = raw; a.z = z; a.near = near; a.far = far; a.rays_d = rays_d; a.noise = noise;
    a.R = R; a.S = S; a.color_mode = color_mode; a.white_bkgd = white_bkgd; a.farcolorfix = farcolorfix;
    a.g_rgb = g_rgb; a.g_depth = g_depth; a.g_acc = g_acc; a.g_weights = g_weights; a.g_tau = g_tau; a.g_T = g_T; a.g_raw = g_raw;
    a.absmax_out = absmax_out;
    a.g_z = g_z; a.g_near = g_near; a.g_far = g_far; a.g_dnorm = g_dnorm;
    a.lds_stride = ((5 * S + 4 + (g_z ? 5 : 4) * (S + 2)) + 3) & ~3;
    const size_t lds = (size_t)WAVES * a.lds_stride * sizeof(float);
    dim3 grid((R + WAVES - 1) / WAVES), block(WAVES * 64);
    hipStream_t st = (hipStream_t)stream;
    if (lds > 160 * 1024) return PLNERF_ERANGE;
    if (lds > 64 * 1024) {
        // opt in to the large dynamic-LDS carve-out (S > ~400)
        if (mode == PLNERF_MODE_LINEAR)
            (void)hipFuncSetAttribute((const void*)quad_bwd_kernel<PLNERF_MODE_LINEAR>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        else
            (void)hipFuncSetAttribute((const void*)quad_bwd_kernel<PLNERF_MODE_CONSTANT>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (mode == PLNERF_MODE_LINEAR)
        hipLaunchKernelGGL(quad_bwd_kernel<PLNERF_MODE_LINEAR>, grid, block, lds, st, a);
    else
        hipLaunchKernelGGL(quad_bwd_kernel<PLNERF_MODE_CONSTANT>, grid, block, lds, st, a);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}
}  // namespace

extern "C" int plnerf_quad_bwd(const float* raw, const float* z, const float* near, const float* far,
                               const float* rays_d, const float* noise, int R, int S, int mode,
                               int color_mode, int white_bkgd, int farcolorfix, const float* g_rgb,
                               const float* g_depth, const float* g_acc, const float* g_weights,
                               const float* g_tau, const float* g_T, float* g_raw, uint32_t* absmax_out,
                               plnerf_stream_t stream) {
    return quad_bwd_launch(raw, z, near, far, rays_d, noise, R, S, mode, color_mode, white_bkgd, farcolorfix, g_rgb, g_depth,
                           g_acc, g_weights, g_tau, g_T, g_raw, absmax_out, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int plnerf_quad_bwd_rays(const float* raw, const float* z, const float* near, const float* far,
                                    const float* rays_d, const float* noise, int R, int S, int mode,
                                    int color_mode, int white_bkgd, int farcolorfix, const float* g_rgb,
                                    const float* g_depth, const float* g_acc, const float* g_weights,
                                    const float* g_tau, const float* g_T, float* g_raw, float* g_z, float* g_near,
                                    float* g_far, float* g_dnorm, plnerf_stream_t stream) {
    if (R > 0 && (!g_z || !g_near || !g_far || !g_dnorm)) return PLNERF_EINVAL;
    return quad_bwd_launch(raw, z, near, far, rays_d, noise, R, S, mode, color_mode, white_bkgd, farcolorfix, g_rgb, g_depth,
                           g_acc, g_weights, g_tau, g_T, g_raw, nullptr, g_z, g_near, g_far, g_dnorm, stream);
}
