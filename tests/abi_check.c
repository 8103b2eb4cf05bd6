/* The boundary is a C ABI: this file is plain C99, includes include/plnerf_hip.h as a foreign caller would, takes the
 * address of every entry point with its declared prototype (a mismatch between header and library is a link error,
 * a header that is not valid C a compile error) and calls the ones that need no GPU.  Compiled and run by
 * tests/test_host_cpu.py::test_header_is_plain_c_and_links. */
#include <stdio.h>
#include "plnerf_hip.h"

int main(void) {
    typedef void (*any_fn)(void);
    any_fn entry[] = {
        (any_fn)plnerf_version, (any_fn)plnerf_build_flags, (any_fn)plnerf_error_string, (any_fn)plnerf_image_loss,
        (any_fn)plnerf_depth_loss, (any_fn)plnerf_embed_rows, (any_fn)plnerf_quad_fwd, (any_fn)plnerf_quad_bwd,
        (any_fn)plnerf_sample_const, (any_fn)plnerf_sample_const_bwd, (any_fn)plnerf_sample_pl, (any_fn)plnerf_sample_pl_bwd,
        (any_fn)plnerf_stratified_z, (any_fn)plnerf_ray_points, (any_fn)plnerf_merge_sort, (any_fn)plnerf_coarse_epilogue,
        (any_fn)plnerf_uniform, (any_fn)plnerf_normal, (any_fn)plnerf_select_rays, (any_fn)plnerf_coarse_samples, (any_fn)plnerf_mlp_packed_bytes,
        (any_fn)plnerf_mlp_status_offset, (any_fn)plnerf_mlp_pack_weights, (any_fn)plnerf_mlp_saved_bytes,
        (any_fn)plnerf_mlp_bwd_workspace_bytes, (any_fn)plnerf_mlp_saved_layout, (any_fn)plnerf_mlp_fwd, (any_fn)plnerf_mlp_bwd,
        (any_fn)plnerf_adam_step};
    size_t i, n = sizeof entry / sizeof entry[0];
    for (i = 0; i < n; ++i)
        if (!entry[i]) return 2;
    if (plnerf_version() != PLNERF_VERSION) return 3;
    if (plnerf_build_flags() != 0) return 4;
    if (plnerf_mlp_saved_layout(PLNERF_PREC_F16X3, 0, PLNERF_FWD_KERNEL_AUTO) != 1) return 5;
    if (plnerf_mlp_saved_layout(PLNERF_PREC_F16X3, 0, 99) >= 0) return 6;
    /* argument validation runs before any device work: a null pointer is PLNERF_EINVAL, not a crash */
    if (plnerf_mlp_fwd(NULL, PLNERF_PREC_FP32, NULL, NULL, NULL, 63, 27, 8, 1, 1.0f, NULL, NULL, PLNERF_FWD_KERNEL_AUTO, NULL) !=
        PLNERF_EINVAL)
        return 7;
    if (plnerf_adam_step(NULL, NULL, NULL, NULL, 4, 1e-3f, 0.9f, 0.999f, 1e-8f, 1, 1.0f, 0.0f, NULL, NULL, NULL, NULL) !=
        PLNERF_EINVAL)
        return 8;
    printf("%u entry points, version %d, packed bytes fp32 %zu f16x3 %zu, saved bytes per 256 rows (f16x3) %zu: %s\n",
           (unsigned)n, plnerf_version(), plnerf_mlp_packed_bytes(PLNERF_PREC_FP32), plnerf_mlp_packed_bytes(PLNERF_PREC_F16X3),
           plnerf_mlp_saved_bytes(256, PLNERF_PREC_F16X3), plnerf_error_string(PLNERF_EINVAL));
    return 0;
}
