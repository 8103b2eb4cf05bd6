// Device functions shared by the per-ray kernels (quad.hip, sampler.hip, epilogue.hip): one definition of the
// quadrature's element rules, the cdf inversion and the register sort, so that the fused coarse epilogue computes
// bit for bit what the separate launches compute.  Everything that includes this is built with -ffp-contract=off.
#pragma once
#include "common.h"

namespace plnerf {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Loads one ray into LDS: knots zk[0..S+1] = [near, z, far], tau[0..S+1] =
// relu([1e-10, sigma+noise, 1e10]), col[3*s+c] = sigmoid(raw rgb).
struct RayIn {            // the per-ray inputs of the quadrature (device pointers; noise may be null)
    const float* raw;
    const float* z;
    const float* near;
    const float* far;
    const float* rays_d;
    const float* noise;
    int S;
};

__device__ __forceinline__ void load_ray(const RayIn& a, int ray, int lane, float* zk, float* tau,
                                         float* col, float& dnorm) {
    const int S = a.S;
    const float4* raw4 = reinterpret_cast<const float4*>(a.raw) + (size_t)ray * S;
    const float* zrow = a.z + (size_t)ray * S;
    const float* nrow = a.noise ? a.noise + (size_t)ray * S : nullptr;
    for (int s = lane; s < S; s += 64) {
        const float4 r = raw4[s];
        float sg = r.w;
        if (nrow) sg = sg + nrow[s];
        col[3 * s + 0] = sigmoidf_(r.x);
        col[3 * s + 1] = sigmoidf_(r.y);
        col[3 * s + 2] = sigmoidf_(r.z);
        tau[s + 1] = tmax(sg, 0.0f);
        zk[s + 1] = zrow[s];
    }
    if (lane == 0) {
        zk[0] = a.near[ray];
        zk[S + 1] = a.far[ray];
        tau[0] = 1e-10f;
        tau[S + 1] = 1e10f;
    }
    const float dx = a.rays_d[3 * ray + 0], dy = a.rays_d[3 * ray + 1], dz = a.rays_d[3 * ray + 2];
    dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
}

// Element i of the scan: e_i (the interval's exp term) and f_i (its transmittance factor).
template <int MODE>
__device__ __forceinline__ void interval(int i, int S, const float* zk, const float* tau, float dnorm,
                                         float& seg, float& e, float& f) {
    if (MODE == PLNERF_MODE_LINEAR) {
        seg = (zk[i + 1] - zk[i]) * dnorm;
        const float ave = 0.5f * (tau[i + 1] + tau[i]);
        e = expf((-ave) * seg);
        f = e;
    } else {
        seg = ((i < S - 1) ? (zk[i + 2] - zk[i + 1]) : 1e10f) * dnorm;
        e = expf((-tau[i + 1]) * seg);
        const float alpha = 1.0f - e;
        f = 1.0f - alpha + 1e-10f;
    }
}

// Colour attached to element i, per channel c (the reference's padded-colour rules).
template <int MODE>
__device__ __forceinline__ float elem_colour(int i, int c, int S, const float* col, int color_mode,
                                             int farcolorfix) {
    if (MODE == PLNERF_MODE_LINEAR) {
        const float left = col[3 * (i > 0 ? i - 1 : 0) + c];  // padded[i]
        if (color_mode == PLNERF_COLOR_LEFT) return left;
        float right;                                           // padded[i+1]
        if (i < S) right = col[3 * i + c];
        else right = farcolorfix ? 0.0f : col[3 * (S - 1) + c];
        return 0.5f * (right + left);
    }
    return col[3 * i + c];
}

template <int MODE>
__device__ __forceinline__ float elem_depth(int i, const float* zk) {
    return (MODE == PLNERF_MODE_LINEAR) ? 0.5f * (zk[i + 1] + zk[i]) : zk[i + 1];
}

// torch.searchsorted(cdf, u, right=True): the same upper-bound bisection as ATen's
// (mid = start + ((end-start) >> 1); !(cdf[mid] > u) -> go right), so the result agrees
// even on a cdf that is non-monotone by an ulp.
__device__ __forceinline__ int upper_bound(const float* cdf, int len, float u) {
    int start = 0, end = len;
    while (start < end) {
        const int mid = start + ((end - start) >> 1);
        if (!(cdf[mid] > u)) start = mid + 1;
        else end = mid;
    }
    return start;
}

// Closed-form inverse of T0 * exp(-(tau0 t + (tau1-tau0) t^2 / (2 (s1-s0)))) = 1-u on
// one interval, with the reference's epsilon guards, op for op (run_nerf_helpers.py:340-349 rising, 352-361 falling).
// `rising` is a per-lane value: the two directions differ only in which operand of a subtraction comes first and in the
// sign of one addend, so ONE evaluation with selected operands serves a wave whose lanes go both ways -- one logf, one
// sqrtf, two divisions per lane instead of two of each under divergent branches (round 5; every operation is the
// direction's own, so the bits are the reference's either way).
__device__ __forceinline__ float invert_segment(float s0, float s1, float T0, float tau0, float tau1,
                                                float u, float eps, bool rising) {
    const float ln_term = -logf(tmax(eps, (1.0f - u) / tmax(eps, T0)));
    const float span = tmax(eps, s1 - s0);
    const float dt = rising ? tau1 - tau0 : tau0 - tau1;              // the direction's positive slope
    const float q = (2.0f * dt * ln_term) / span;
    const float disc = rising ? tau0 * tau0 + q : tau0 * tau0 - q;
    const float sq = sqrtf(tmax(eps, disc));
    const float num = rising ? -tau0 + sq : tau0 - sq;
    float t = ((s1 - s0) * num) / tmax(eps, dt);
    t = tmin(tmax(t, eps), s1 - s0);   // torch.clamp(t, eps, s1-s0)
    return s0 + t;
}

// Total order on fp32 as torch.sort uses it for values: ascending, NaN last.
__device__ __forceinline__ uint32_t sort_key(float f) {
    if (f != f) return 0xFFFFFFFFu;
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ float sort_unkey(uint32_t k) {
    if (k == 0xFFFFFFFFu) return __uint_as_float(0x7FC00000u);        // NaN (and the padding, never written)
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// Bitonic sort of 64 KPL keys held in REGISTERS by one wavefront: position p = 64 r + lane holds x[r] (pad with the
// maximum key).  A compare-exchange distance j >= 64 pairs two registers of the same lane; j < 64 pairs lanes l and
// l ^ j (one cross-lane read per key).
template <int KPL>
__device__ __forceinline__ void bitonic_sort_regs(uint32_t (&x)[KPL], const int lane) {
    constexpr int NP = 64 * KPL;
#pragma unroll
    for (int k = 2; k <= NP; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
#pragma unroll
                for (int r = 0; r < KPL; ++r) {
                    const int q = r ^ (j >> 6);
                    if (q > r) {
                        // the pair (r, q) of this lane: ascending where bit k of the position is clear
                        const bool up = ((64 * r) & k) == 0;
                        const uint32_t mn = min(x[r], x[q]), mx = max(x[r], x[q]);
                        x[r] = up ? mn : mx;
                        x[q] = up ? mx : mn;
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < KPL; ++r) {
                    const int p = 64 * r + lane;
                    const uint32_t other = (uint32_t)__shfl_xor((int)x[r], j);
                    const bool up = (p & k) == 0, lower = (lane & j) == 0;
                    x[r] = (up == lower) ? min(x[r], other) : max(x[r], other);
                }
            }
        }
    }
}

// Positions in an ASCENDING run of keys a[0..n): how many are < k (count_lt) / <= k (count_le).  Two ascending runs merge by
// rank with these: an element's place = its index in its own run + the other run's count in front of it (ties: the first
// run's element first), a permutation of 0 .. n1 + n2 - 1.
__device__ __forceinline__ int count_lt(const uint32_t* a, int n, uint32_t k) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < k) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ int count_le(const uint32_t* a, int n, uint32_t k) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= k) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

}  // namespace plnerf
