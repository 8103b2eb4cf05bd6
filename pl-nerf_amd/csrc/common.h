// Shared device/host helpers for libplnerf_hip.so (gfx950 only; wavefront = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/plnerf_hip.h"

#define PLNERF_WAVE 64

#define PLNERF_CHECK_LAUNCH()                                   \
    do {                                                        \
        hipError_t e__ = hipGetLastError();                     \
        if (e__ != hipSuccess) return PLNERF_ELAUNCH;           \
    } while (0)

namespace plnerf {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// torch.max / torch.min / torch.clamp semantics: NaN propagates (fmaxf would drop it).
__device__ __forceinline__ float tmax(float a, float b) {
    return (a != a || b != b) ? __builtin_nanf("") : (a > b ? a : b);
}
__device__ __forceinline__ float tmin(float a, float b) {
    return (a != a || b != b) ? __builtin_nanf("") : (a < b ? a : b);
}

// The density activation of the depth-supervised variant (depth_supervised_exps/model/run_nerf_helpers.py:200:
// F.softplus(sigma, beta=10)), torch's formula and threshold: x beta > 20 ? x : log1p(exp(x beta)) / beta.
// beta <= 0: no activation.
__device__ __forceinline__ float density_activation(const float x, const float beta) {
    if (!(beta > 0.0f)) return x;
    const float bx = x * beta;
    return bx > 20.0f ? x : log1pf(expf(bx)) / beta;
}
// d softplus / dx = sigmoid(beta x), from the activation's OUTPUT y: 1 - exp(-beta y)  (exp(beta y) = 1 + exp(beta x))
__device__ __forceinline__ float density_activation_grad(const float y, const float beta) {
    return beta > 0.0f ? -expm1f(-beta * y) : 1.0f;
}

// Inclusive wave scans over 64 lanes (Hillis-Steele on __shfl_up, which lowers to
// DPP row shifts / ds_bpermute on gfx950).
__device__ __forceinline__ double wave_incl_prod(double v) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double o = __shfl_up(v, d);
        if (l >= d) v *= o;
    }
    return v;
}
__device__ __forceinline__ double wave_incl_sum(double v) {
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double o = __shfl_up(v, d);
        if (l >= d) v += o;
    }
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

}  // namespace plnerf
