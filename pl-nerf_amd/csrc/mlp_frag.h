// MFMA accumulator helpers shared by the exact-fp32 MLP kernels (mlp_f32.hip) and the weight-gradient stage
// (mlp_wgrad.hip).
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace plnerf {

template <int NI, int NJ>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NI][NJ]) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
}

// C/D fragment coordinates of v_mfma_f32_32x32x*: col = lane & 31,
// row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
__device__ __forceinline__ int frag_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

}  // namespace plnerf
