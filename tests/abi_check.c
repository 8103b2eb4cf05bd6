/* The boundary is a C ABI: this file is plain C99, includes include/plnerf_hip.h as a foreign caller would, and assigns
 * every entry point to a function pointer of the type a binding (cgo / JNI / ctypes) would declare for it -- so a drift
 * between the header's prototypes and the binder's view of them is a COMPILE error (-Werror=incompatible-pointer-types
 * is part of -pedantic -Werror), a symbol missing from the library a link error -- then calls the ones that need no
 * GPU.  Compiled and run by tests/test_host_cpu.py::test_header_is_plain_c_and_links; the same prototypes are parsed
 * from the header and compared with the ctypes signatures of pl-nerf_amd/_lib.py by
 * test_ctypes_signatures_match_the_header. */
#include <stdio.h>
#include "plnerf_hip.h"

int main(void) {
    int (*p_version)(void) = plnerf_version;
    int (*p_build_flags)(void) = plnerf_build_flags;
    const char* (*p_error_string)(int) = plnerf_error_string;
    int (*p_quad_fwd)(const float*, const float*, const float*, const float*, const float*, const float*, int, int, int, int, int, int, float*, float*, float*, float*, float*, float*, float*, plnerf_stream_t) = plnerf_quad_fwd;
    int (*p_quad_bwd)(const float*, const float*, const float*, const float*, const float*, const float*, int, int, int, int, int, int, const float*, const float*, const float*, const float*, const float*, const float*, float*, uint32_t*, plnerf_stream_t) = plnerf_quad_bwd;
    int (*p_quad_bwd_rays)(const float*, const float*, const float*, const float*, const float*, const float*, int, int, int, int, int, int, const float*, const float*, const float*, const float*, const float*, const float*, float*, float*, float*, float*, float*, plnerf_stream_t) = plnerf_quad_bwd_rays;
    int (*p_sample_const)(const float*, const float*, const float*, int, int, int, int, float*, int64_t*, plnerf_stream_t) = plnerf_sample_const;
    int (*p_sample_const_bwd)(const float*, const float*, const float*, int, const int64_t*, const float*, int, int, int, float*, plnerf_stream_t) = plnerf_sample_const_bwd;
    int (*p_sample_pl)(const float*, const float*, const float*, const float*, const float*, const float*, const float*, int, int, int, int, float, float, float*, float*, float*, float*, int64_t*, plnerf_stream_t) = plnerf_sample_pl;
    int (*p_sample_pl_bwd)(const float*, const float*, const float*, const float*, const float*, const float*, int, const int64_t*, const float*, int, int, int, float, float, float*, float*, plnerf_stream_t) = plnerf_sample_pl_bwd;
    int (*p_sample_pl_bwd_rays)(const float*, const float*, const float*, const float*, const float*, const float*, int, const int64_t*, const float*, int, int, int, float, float, float*, float*, float*, plnerf_stream_t) = plnerf_sample_pl_bwd_rays;
    int (*p_stratified_z)(const float*, const float*, const float*, const float*, int, int, int, float*, plnerf_stream_t) = plnerf_stratified_z;
    int (*p_ray_points)(const float*, const float*, const float*, int, int, float*, plnerf_stream_t) = plnerf_ray_points;
    int (*p_merge_sort)(const float*, const float*, const float*, const float*, int, int, int, float*, plnerf_stream_t) = plnerf_merge_sort;
    int (*p_coarse_epilogue)(const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, int, uint64_t, uint32_t, int, int, int, int, int, int, int, float, float, float*, float*, float*, float*, float*, float*, float*, float*, float*, float*, plnerf_stream_t) = plnerf_coarse_epilogue;
    int (*p_fine_epilogue)(const float*, const float*, const float*, const float*, const float*, const float*, const float*, int, uint64_t, uint32_t, int, int, int, int, int, int, int, float, float, float*, float*, float*, float*, float*, float*, float*, float*, int64_t*, float*, float*, plnerf_stream_t) = plnerf_fine_epilogue;
    int (*p_uniform)(uint64_t, uint32_t, uint32_t, int, int, int, float*, plnerf_stream_t) = plnerf_uniform;
    int (*p_normal)(uint64_t, uint32_t, uint32_t, int, int, int, float*, plnerf_stream_t) = plnerf_normal;
    int (*p_select_rays)(int, int, float, float, float, float, const float*, const float*, int, int, int, int, uint64_t, uint32_t, int, int, float, float, float*, float*, float*, float*, float*, float*, int*, plnerf_stream_t) = plnerf_select_rays;
    int (*p_ndc_rays)(int, int, double, double, const float*, const float*, int, float*, float*, plnerf_stream_t) = plnerf_ndc_rays;
    int (*p_coarse_samples)(const float*, const float*, const float*, const float*, const float*, const float*, uint64_t, uint32_t, int, int, int, int, int, float*, float*, plnerf_stream_t) = plnerf_coarse_samples;
    int (*p_image_loss)(const float*, const float*, const float*, int, float*, float*, float*, const float*, void*, plnerf_stream_t) = plnerf_image_loss;
    int (*p_depth_loss)(const float*, const float*, const float*, const float*, const float*, const float*, int, int, int, int, int, const int*, float, float, float*, float*, float*, float*, void*, plnerf_stream_t) = plnerf_depth_loss;
    int (*p_depth_joint_sums)(const float*, const float*, const float*, int, int, int, int, float, double*, plnerf_stream_t) = plnerf_depth_joint_sums;
    int (*p_embed_rows)(const float*, const float*, const float*, int, int, int, int, int, float, const float*, float, float*, plnerf_stream_t) = plnerf_embed_rows;
    int (*p_gemm_f32)(const float*, int64_t, int64_t, const float*, int64_t, int64_t, const float*, const float*, int, int, int, int, int, int, float*, int64_t, int, float*, plnerf_stream_t) = plnerf_gemm_f32;
    size_t (*p_mlp_packed_bytes)(int) = plnerf_mlp_packed_bytes;
    size_t (*p_mlp_status_offset)(int) = plnerf_mlp_status_offset;
    int (*p_mlp_pack_weights)(const float* const*, int, int, int, void*, plnerf_stream_t) = plnerf_mlp_pack_weights;
    size_t (*p_mlp_saved_bytes)(int, int) = plnerf_mlp_saved_bytes;
    size_t (*p_mlp_bwd_workspace_bytes)(int, int) = plnerf_mlp_bwd_workspace_bytes;
    int (*p_mlp_saved_layout)(int, int, int) = plnerf_mlp_saved_layout;
    int (*p_mlp_fwd)(const void*, int, const float*, const float*, const float*, int, int, int, int, float, float, float*, void*, int, plnerf_stream_t) = plnerf_mlp_fwd;
    int (*p_mlp_bwd)(const void*, int, const float*, const uint32_t*, int, int, int, int, const void*, int, const float*, float, void*, float* const*, float*, plnerf_stream_t) = plnerf_mlp_bwd;
    int (*p_mlp_bwd_multi)(int, const void* const*, int, const float* const*, const uint32_t* const*, const int*, int, int, const int*, const void* const*, const int*, const float* const*, float, void* const*, float* const*, float* const*, plnerf_stream_t) = plnerf_mlp_bwd_multi;
    int (*p_mlp_input_grad)(const float* const*, int, int, int, int, const void*, float*, plnerf_stream_t) = plnerf_mlp_input_grad;
    int (*p_adam_step)(float*, const float*, float*, float*, int64_t, float, float, float, float, int, float, float, const uint32_t*, const uint32_t*, uint32_t*, plnerf_stream_t) = plnerf_adam_step;
    const void* entry[] = {
        (const void*)&p_version, (const void*)&p_build_flags, (const void*)&p_error_string, (const void*)&p_quad_fwd,
        (const void*)&p_quad_bwd, (const void*)&p_quad_bwd_rays, (const void*)&p_sample_const, (const void*)&p_sample_const_bwd, (const void*)&p_sample_pl,
        (const void*)&p_sample_pl_bwd, (const void*)&p_sample_pl_bwd_rays, (const void*)&p_stratified_z, (const void*)&p_ray_points, (const void*)&p_merge_sort,
        (const void*)&p_coarse_epilogue, (const void*)&p_fine_epilogue, (const void*)&p_uniform, (const void*)&p_normal,
        (const void*)&p_select_rays, (const void*)&p_ndc_rays, (const void*)&p_coarse_samples, (const void*)&p_image_loss, (const void*)&p_depth_loss, (const void*)&p_depth_joint_sums,
        (const void*)&p_embed_rows, (const void*)&p_gemm_f32, (const void*)&p_mlp_packed_bytes, (const void*)&p_mlp_status_offset, (const void*)&p_mlp_pack_weights,
        (const void*)&p_mlp_saved_bytes, (const void*)&p_mlp_bwd_workspace_bytes, (const void*)&p_mlp_saved_layout, (const void*)&p_mlp_fwd,
        (const void*)&p_mlp_bwd, (const void*)&p_mlp_bwd_multi, (const void*)&p_mlp_input_grad, (const void*)&p_adam_step,
    };
    size_t i, n = sizeof entry / sizeof entry[0];
    for (i = 0; i < n; ++i)
        if (!*(void* const*)entry[i]) return 2;
    if (p_version() != PLNERF_VERSION) return 3;
    if (p_build_flags() != 0) return 4;
    if (p_mlp_saved_layout(PLNERF_PREC_F16X3, 0, PLNERF_FWD_KERNEL_AUTO) != 1) return 5;
    if (p_mlp_saved_layout(PLNERF_PREC_F16X3, 0, 99) >= 0) return 6;
    /* argument validation runs before any device work: a null pointer is PLNERF_EINVAL, not a crash */
    if (p_mlp_fwd(NULL, PLNERF_PREC_FP32, NULL, NULL, NULL, 63, 27, 8, 1, 1.0f, 0.0f, NULL, NULL, PLNERF_FWD_KERNEL_AUTO, NULL) !=
        PLNERF_EINVAL)
        return 7;
    if (p_adam_step(NULL, NULL, NULL, NULL, 4, 1e-3f, 0.9f, 0.999f, 1e-8f, 1, 1.0f, 0.0f, NULL, NULL, NULL, NULL) !=
        PLNERF_EINVAL)
        return 8;
    if (p_image_loss(NULL, NULL, NULL, 4, NULL, NULL, NULL, NULL, NULL, NULL) != PLNERF_EINVAL) return 9;
    if (p_mlp_bwd(NULL, PLNERF_PREC_F16X3, NULL, NULL, 0, 63, 27, 8, NULL, 1, NULL, 0.0f, NULL, NULL, NULL, NULL) != PLNERF_EINVAL) return 10;
    if (p_mlp_bwd_multi(PLNERF_MAX_BWD_JOBS + 1, NULL, PLNERF_PREC_F16X3, NULL, NULL, NULL, 63, 27, NULL, NULL, NULL, NULL, 0.0f, NULL, NULL, NULL, NULL) != PLNERF_EINVAL) return 11;
    if (p_mlp_bwd_multi(2, NULL, PLNERF_PREC_F16X3, NULL, NULL, NULL, 63, 27, NULL, NULL, NULL, NULL, 0.0f, NULL, NULL, NULL, NULL) != PLNERF_EINVAL) return 12;
    printf("%u entry points, version %d, packed bytes fp32 %zu f16x3 %zu, saved bytes per 256 rows (f16x3) %zu: %s\n",
           (unsigned)n, p_version(), p_mlp_packed_bytes(PLNERF_PREC_FP32), p_mlp_packed_bytes(PLNERF_PREC_F16X3),
           p_mlp_saved_bytes(256, PLNERF_PREC_F16X3), p_error_string(PLNERF_EINVAL));
    return 0;
}
