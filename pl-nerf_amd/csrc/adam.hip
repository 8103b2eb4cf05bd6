// Adam as the reference configures it (run_plnerf.py:446-447, 1302-1303: torch.optim.Adam, betas (0.9, 0.999), eps 1e-8)
// over ONE flat parameter buffer per launch (optim.FlatAdam), guarded by up to two range status words and with the
// data-parallel 1 / world factor and the depth loop's clip_grad_value_ folded in.
#include <math.h>

#include "common.h"

namespace {

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float step_size, float b1, float b2, float eps,
                            float bc2_sqrt, float gscale, float clip, const unsigned* __restrict__ guard,
                            const unsigned* __restrict__ guard2, unsigned* __restrict__ withheld) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // the step's gradients come from a forward that left the half range: keep the weights (and say so: the host's step
    // count, hence its bias corrections, must not advance for a step that did not happen)
    if ((guard && *guard) || (guard2 && *guard2)) {
        if (i == 0 && withheld) atomicAdd(withheld, 1u);
        return;
    }
    // torch.optim.Adam (single-tensor path): m = lerp(m, g, 1-b1); v = b2 v + (1-b2) g^2;
    // denom = sqrt(v)/sqrt(bc2) + eps; p -= (lr/bc1) * m / denom
    float gi = g[i] * gscale;
    if (clip > 0.0f) gi = fminf(fmaxf(gi, -clip), clip);      // torch.nn.utils.clip_grad_value_ (a NaN passes through, as there)
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
    const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
}

}  // namespace

extern "C" int plnerf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                int64_t n, float lr, float beta1, float beta2, float eps, int step,
                                float grad_scale, float clip_value, const unsigned* skip_if_set,
                                const unsigned* skip_if_set2, unsigned* withheld, plnerf_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return PLNERF_EINVAL;
    if (n == 0) return PLNERF_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const int threads = 256;
    const int64_t blocks = (n + threads - 1) / threads;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream, param, grad,
                       exp_avg, exp_avg_sq, n, (float)((double)lr / bc1), beta1, beta2, eps, (float)sqrt(bc2), grad_scale,
                       clip_value, skip_if_set, skip_if_set2, withheld);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}
