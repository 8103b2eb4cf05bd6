// Internal launchers behind the MLP entry points of the C ABI (mlp_api.hip dispatches on the
// precision mode).  All precisions share one saved-state / workspace layout (fp32 planes,
// mlp_layout.h), so the weight-gradient stage is common.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace plnerf {
namespace impl {

size_t f32_packed_bytes();
int f32_pack(const float* const* params, int xyz_ch, int dir_ch, void* packed, hipStream_t st);
// xyz_ch / dir_ch = input_ch / input_ch_views of the network (columns of `embedded`)
int f32_fwd(const void* packed, const float* pts, const float* viewdirs, const float* embedded, int xyz_ch,
            int dir_ch, int n_rows, int samples_per_ray, float* raw_out, void* saved, hipStream_t st);
int f32_dgrad(const void* packed, const float* g_raw, int n_rows, const float* saved, float* dz, hipStream_t st);
// ns = 0: fp32 MFMA; ns = 1 / 2: the main 256x256 jobs on bf16 / bf16x3 MFMA (operands split in registers)
int f32_wgrad(const float* g_raw, int n_rows, const float* saved, float* dz, float* const* grads, int xyz_ch,
              int dir_ch, int ns, int f16, hipStream_t st);

// 16-bit-operand MFMA modes.  ns = 1: plain operands; ns = 2: 3-term split (hi/lo planes).
// f16 = 0: bf16 elements; f16 = 1: IEEE half elements.
size_t bf16_packed_bytes(int ns);
int bf16_pack(const float* const* params, int xyz_ch, int dir_ch, int ns, int f16, void* packed, hipStream_t st);
int bf16_fwd(const void* packed, int ns, int f16, const float* pts, const float* viewdirs, const float* embedded,
             int xyz_ch, int dir_ch, int n_rows, int samples_per_ray, float* raw_out, void* saved, hipStream_t st);
int bf16_dgrad(const void* packed, int ns, int f16, const float* g_raw, int n_rows, const float* saved, float* dz,
               hipStream_t st);

}  // namespace impl
}  // namespace plnerf
