// Counter-based uniform draws for the path's random numbers (stratified jitter t_rand, run_plnerf.py:700-705; the
// sampler's u, run_nerf_helpers.py:384-392): Philox4x32-10 keyed by (seed), counter = (global ray id, column block,
// stream id, step).  A draw depends only on WHICH ray of the global batch it belongs to, not on how the batch is
// split over ranks or chunks -- results are invariant to the world size.
//
// The reference draws with torch.rand (its own Philox stream, not reproducible across devices either); parity tests
// inject identical draws on both sides (pytest=True or explicit tensors), so the generator only has to be uniform.
#pragma once
#include <stdint.h>

#include "common.h"

namespace plnerf {

struct RngArgs {
    uint32_t seed_lo, seed_hi;   // key
    uint32_t stream;             // which draw of the step: 0 = t_rand (coarse jitter), 1 = u (importance samples), ...
    uint32_t step;               // optimisation step (or any per-call counter)
    int ray_id0;                 // global id of the launch's first ray (rank offset + chunk offset)
    int enabled;                 // 0: the kernel reads its draws from a caller tensor instead
};

__host__ __device__ __forceinline__ void philox_round(uint32_t (&c)[4], const uint32_t k0, const uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1;
    c[3] = (uint32_t)p0;
    c[0] = n0;
    c[2] = n2;
}

// four 32-bit words for counter (c0, c1, c2, c3)
__host__ __device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

// uniform in [0, 1) with 24 random bits (what torch.rand produces for fp32)
__host__ __device__ __forceinline__ float u01(const uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }

// element `col` of global ray `ray_id`'s row
__host__ __device__ __forceinline__ float rng_uniform(const RngArgs& g, const int ray_id, const int col) {
    uint32_t c[4] = {(uint32_t)ray_id, (uint32_t)(col >> 2), g.stream, g.step};
    philox4x32_10(c, g.seed_lo, g.seed_hi);
    return u01(c[col & 3]);
}

}  // namespace plnerf
