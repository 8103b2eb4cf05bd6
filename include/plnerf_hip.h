/*
 * plnerf_hip.h -- C ABI of libplnerf_hip.so, the MI355X (gfx950) implementation of
 * PL-NeRF's ray-batched volume-rendering hot path.
 *
 * The reference (mikacuy/PL-NeRF) is pure Python on PyTorch and has no FFI layer; the
 * entry points below are what a binding for this path would bind, one per reference
 * function (cited per entry as file:line into the reference tree).  All functions
 *   - take raw DEVICE pointers (fp32 unless noted) plus explicit sizes,
 *   - allocate nothing: the caller owns every input, output and workspace,
 *   - only ENQUEUE work on `stream` (a hipStream_t passed as void*; NULL = default),
 *   - are re-entrant and keep no global mutable state,
 *   - return 0 on success or a negative PLNERF_E* code (never throw).
 *
 * Row-major contiguous layouts throughout.  R = rays, S = samples per ray in this
 * pass, N = number of new samples to draw.
 */
#ifndef PLNERF_HIP_H
#define PLNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLNERF_VERSION 600 /* major*10000 + minor*100 + patch */

/* error codes */
#define PLNERF_OK 0
#define PLNERF_EINVAL (-1)   /* bad size / null pointer / unsupported combination */
#define PLNERF_ELAUNCH (-2)  /* hipGetLastError() after the launch was not hipSuccess */
#define PLNERF_ERANGE (-3)   /* size outside the compiled limits (e.g. S > PLNERF_MAX_SAMPLES) */
#define PLNERF_ENOSYS (-4)   /* precision mode not built */

/* quadrature mode: run_plnerf.py:579 (linear) / :607 (constant) */
#define PLNERF_MODE_CONSTANT 0
#define PLNERF_MODE_LINEAR 1
/* colour rule of the linear mode: run_plnerf.py:581 / :593 */
#define PLNERF_COLOR_MIDPOINT 0
#define PLNERF_COLOR_LEFT 1
/* arithmetic of the MLP FORWARD contractions (the 1e-5 parity contract is on forward outputs) */
#define PLNERF_PREC_FP32 0   /* v_mfma_f32_32x32x2_f32: exact fp32 fma chains; fp32 backward too   */
#define PLNERF_PREC_BF16X3 1 /* 3-term bf16 split on v_mfma_f32_32x32x16_bf16 (16 mantissa bits)  */
#define PLNERF_PREC_BF16 2   /* plain bf16 operands, fp32 accumulate                              */
#define PLNERF_PREC_F16X3 3  /* 3-term IEEE-half split on v_mfma_f32_32x32x16_f16 (22 bits)        */
#define PLNERF_PREC_F16 4    /* plain half operands, fp32 accumulate                              */
/* Range of the half-element modes (3, 4): conversions saturate at the IEEE-half maximum, 65,504.  A forward that
 * meets an activation, or a pack that meets a weight, beyond it sets a bit of the STATUS WORD -- the uint32 at
 * plnerf_mlp_status_offset(precision) bytes into the packed-weight buffer (sticky: the library only ever ORs into it;
 * the caller zeroes it when it allocates the buffer and after reading it).  Results computed with a bit set are
 * clamped, i.e. wrong: re-run in mode 1 (bf16x3) or 0.  plnerf_adam_step can be handed the word and then leaves the
 * weights untouched while it is non-zero.  Mode 0 never sets it; modes 1-2 (fp32's exponent range in the forward)
 * only set PLNERF_RANGE_SAVED, and only in a training forward. */
#define PLNERF_RANGE_ACTIVATION 1u /* an activation beyond +-65,504 was split into halves (forward)  */
#define PLNERF_RANGE_WEIGHT 2u     /* a weight beyond +-65,504, or not finite, was packed             */
#define PLNERF_RANGE_SAVED 4u      /* bf16-element modes (1, 2), training forward: an activation beyond +-65,504 was
                                    * clamped on its way into the IEEE-half SAVED planes.  The forward result of that call
                                    * is right (bf16 elements carry fp32's exponent range); the gradients of the matching
                                    * plnerf_mlp_bwd are not -- use mode 0 for such a network. */
/* Backward of modes 1-4: the saved activations and the pre-activation gradients are IEEE-half
 * planes, the latter under one power-of-two scale per launch (max |g_raw| -> [8,16), saturating
 * conversion); dgrad chain and weight gradients are single half MFMAs with fp32 accumulation.
 * Weight-gradient entries agree with fp32 autograd to ~1e-3 of max |g| (cosine > 0.999999). */

#define PLNERF_MAX_SAMPLES 1022 /* S+2 knots must fit the per-wave LDS row */

/* Network geometry this library is specialised for: the reference defaults
 * (run_plnerf.py:784-825: netdepth 8, netwidth 256, skips [4], use_viewdirs; multires 10,
 * multires_views 4 -> input_ch 63, input_ch_views 27; 595,844 parameters per network).
 * The two input widths are run-time arguments (input_ch <= 64, input_ch_views <= 32) so
 * that the depth-supervised variant's network (multires 9, multires_views 0 -> 57 / 3,
 * depth_supervised_exps/run_nerf_sample_based_depth.py:550-561, 1306-1309) runs on the
 * same kernels; the in-kernel positional encoding exists for 63 / 27 only, other widths
 * pass the encoding in (`embedded`). */
#define PLNERF_N_PARAM_TENSORS 24
#define PLNERF_N_PARAMS 595844

typedef void* plnerf_stream_t;

int plnerf_version(void);
/* 0 for a product build.  Non-zero when the library was compiled with tools-only switches: bit 0 timing ablations
 * whose results are WRONG by construction, bit 1 timing switches that keep results right, bit 2 trace hooks. */
int plnerf_build_flags(void);
const char* plnerf_error_string(int code);

/* ------------------------------------------------------------------------------------
 * Quadrature -- raw2outputs (run_plnerf.py:553-624) with compute_weights_piecewise_linear
 * (:516-550) or compute_weights (:504-513) fused in.  One wavefront per ray.
 *
 *   raw    [R,S,4]  (r,g,b,sigma) pre-activation          z      [R,S] sample depths
 *   near   [R], far [R]                                    rays_d [R,3]
 *   noise  [R,S] or NULL (added to sigma before the relu)
 * outputs (any of weights/tau/T may be NULL):
 *   rgb_map [R,3], disp_map [R], acc_map [R], depth_map [R]
 *   weights [R,S+1] linear | [R,S] constant;  tau, T [R,S+2] (linear only)
 */
int plnerf_quad_fwd(const float* raw, const float* z, const float* near, const float* far,
                    const float* rays_d, const float* noise, int R, int S, int mode,
                    int color_mode, int white_bkgd, int farcolorfix, float* rgb_map,
                    float* disp_map, float* acc_map, float* depth_map, float* weights,
                    float* tau, float* T, plnerf_stream_t stream);

/* Backward of plnerf_quad_fwd with respect to `raw` (what autograd derives for the
 * reference at loss.backward(), run_plnerf.py:1300).  Upstream gradients g_rgb [R,3]
 * (required), g_depth [R], g_acc [R], g_weights [R,S+1|S], and -- linear mode only, for
 * callers that differentiate through the sampler
 * (depth_supervised_exps/run_nerf_sample_based_depth.py:923-934) -- g_tau, g_T [R,S+2]
 * (each may be NULL = zero).  disp_map's gradient is folded into g_depth/g_acc by the
 * caller.  g_raw [R,S,4].
 * absmax_out (may be NULL; ABI 500): max |g_raw| as a by-product, for the consumer that scales by it (plnerf_mlp_bwd's
 * g_absmax: the half dz planes' launch scale): [ceil(R / PLNERF_QUAD_RAYS_PER_GROUP)] uint32, one per workgroup of the
 * launch = the fp32 bit pattern of the largest |g_raw| among that workgroup's rays (plain stores, every entry written; a
 * NaN leaves NaN bits, which order above every finite value as unsigned integers). */
#define PLNERF_QUAD_RAYS_PER_GROUP 4
int plnerf_quad_bwd(const float* raw, const float* z, const float* near, const float* far,
                    const float* rays_d, const float* noise, int R, int S, int mode,
                    int color_mode, int white_bkgd, int farcolorfix, const float* g_rgb,
                    const float* g_depth, const float* g_acc, const float* g_weights,
                    const float* g_tau, const float* g_T, float* g_raw, uint32_t* absmax_out,
                    plnerf_stream_t stream);

/* The same backward plus the gradient of the RAY GEOMETRY (ABI 501) -- what autograd also gives the reference when the
 * ray batch itself requires a gradient (run_plnerf.py:707, 735: z_vals, near and far enter raw2outputs' interval lengths
 * and depth map, run_plnerf.py:516-550, 604-617; |rays_d| scales every interval): g_z [R,S], g_near, g_far [R] (the outer
 * knots of the piecewise-linear rule; written as zero in constant mode, which does not use them), g_dnorm [R] = d loss /
 * d |rays_d| (the caller applies d|d|/dd = d / |d|).  All four required.  g_raw as above.  No reference training path
 * asks for these (camera-pose optimisation would). */
int plnerf_quad_bwd_rays(const float* raw, const float* z, const float* near, const float* far,
                         const float* rays_d, const float* noise, int R, int S, int mode,
                         int color_mode, int white_bkgd, int farcolorfix, const float* g_rgb,
                         const float* g_depth, const float* g_acc, const float* g_weights,
                         const float* g_tau, const float* g_T, float* g_raw, float* g_z, float* g_near,
                         float* g_far, float* g_dnorm, plnerf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Hierarchical samplers.  `u` holds the uniform draws: [R,N] when u_row_stride == N, or
 * one shared row [N] when u_row_stride == 0 (det=True: torch.linspace(0,1,N)).
 *
 * sample_pdf (run_nerf_helpers.py:241-284): bins [R,B], weights [R,B-1].
 * Bit-exact contract: `inds` (int64 [R,N], may be NULL) equals torch.searchsorted on the
 * reference's CPU cdf for B-1 >= 8: the row sum replays torch's vectorised fp32 reduction
 * tree, the cdf is an fp64 running sum rounded per element, the divide is IEEE.
 */
int plnerf_sample_const(const float* bins, const float* weights, const float* u,
                        int u_row_stride, int R, int B, int N, float* samples, int64_t* inds,
                        plnerf_stream_t stream);

/* Backward of plnerf_sample_const with respect to `weights` (bins are detached on the reference
 * path): what autograd derives for sample_pdf_return_u
 * (depth_supervised_exps/model/run_nerf_helpers.py:343-394) when the depth-supervised variant runs
 * in piecewise-constant mode.  inds [R,N] is the forward's index output; g_weights [R,B-1] is
 * written (not accumulated), deterministically.  The derivative's value is formed from the pdf and
 * cdf in fp64 (from the fp32 weights; which bins count as empty stays the forward's fp32 decision):
 * 1e-7 of max |g| from fp64 autograd, where the reference's fp32 chain leaves up to 1e-3 on a narrow
 * bin (c1 - c0 of two roundings). */
int plnerf_sample_const_bwd(const float* bins, const float* weights, const float* u,
                            int u_row_stride, const int64_t* inds, const float* g_samples, int R,
                            int B, int N, float* g_weights, plnerf_stream_t stream);

/* sample_pdf_reformulation (run_nerf_helpers.py:364-445) with pw_linear_sample_increasing /
 * _decreasing (:340-361): z [R,S], weights [R,S+1], tau,T [R,S+2], near,far [R].
 * Outputs samples [R,N] and (each may be NULL) T_below, tau_below, bin_below [R,N],
 * inds int64 [R,N].  u == 1.0 (where the reference indexes out of bounds) is defined by
 * clamping the interval index to S. */
int plnerf_sample_pl(const float* z, const float* weights, const float* tau, const float* T,
                     const float* near, const float* far, const float* u, int u_row_stride,
                     int R, int S, int N, float zero_tol, float epsilon, float* samples,
                     float* T_below, float* tau_below, float* bin_below, int64_t* inds,
                     plnerf_stream_t stream);

/* Backward of plnerf_sample_pl with respect to tau and T (the sampler's only differentiable
 * inputs: the interval search is piecewise constant in `weights`, and z / near / far are
 * detached on the reference path).  This is what autograd derives for
 * sample_pdf_reformulation_return_u (depth_supervised_exps/model/run_nerf_helpers.py:607-692
 * with pw_linear_sample_increasing / _decreasing, :499-519) when `pred_hyp` carries the
 * space-carving loss.  inds [R,N] is the index output of the forward call; g_samples [R,N].
 * Outputs g_tau, g_T [R,S+2] are written (not accumulated); the per-knot sums run in
 * sample order, so the result is deterministic.  Which side of every guard of the closed
 * form (max(eps, .), the final clamp, the branch threshold) a sample stands on is decided on
 * the fp32 values the forward computed; the derivative's VALUE -- d t / d tau is the difference
 * of two terms that cancel to first order -- is evaluated in fp64 from the same fp32 inputs:
 * the exact gradient of the forward's values to fp32 rounding, where the reference's fp32
 * autograd sits 1e-5 ... 1e-2 of max |g| from it (LABNOTES R6-13). */
int plnerf_sample_pl_bwd(const float* z, const float* tau, const float* T, const float* near,
                         const float* far, const float* u, int u_row_stride, const int64_t* inds,
                         const float* g_samples, int R, int S, int N, float zero_tol, float epsilon,
                         float* g_tau, float* g_T, plnerf_stream_t stream);

/* The same backward plus the gradient of the KNOTS (ABI 502): g_knots [R,S+2] for [near, z, far] -- what autograd also gives
 * the reference when the sampler's bins carry a gradient (a ray batch that requires one; run_nerf_helpers.py:340-361, 425, 432:
 * the closed-form inverse depends on the interval's ends directly and through its length, a flat interval and a NaN sample are
 * the left knot itself, the clamp's upper bound is the interval's length).  g_tau, g_T as above. */
int plnerf_sample_pl_bwd_rays(const float* z, const float* tau, const float* T, const float* near,
                              const float* far, const float* u, int u_row_stride, const int64_t* inds,
                              const float* g_samples, int R, int S, int N, float zero_tol, float epsilon,
                              float* g_tau, float* g_T, float* g_knots, plnerf_stream_t stream);

/* Coarse sample depths of a ray batch (run_plnerf.py:683-705): z = near (1 - t) + far t over the table
 * t_vals [S] (= torch.linspace(0, 1, S), supplied by the caller), or the `lindisp` form 1 / (1/near (1 - t) +
 * 1/far t); with t_rand [R,S] (nullable) the stratified jitter z = lower + (upper - lower) t_rand between the
 * mid-points.  near, far [R]; z_vals [R,S].  Operation order and rounding are the reference's, so the result is
 * bit-identical to its torch expressions. */
int plnerf_stratified_z(const float* near, const float* far, const float* t_vals, const float* t_rand,
                        int R, int S, int lindisp, float* z_vals, plnerf_stream_t stream);

/* Sample positions pts[r,s,:] = rays_o[r,:] + rays_d[r,:] * z_vals[r,s] (run_plnerf.py:708, :735).
 * rays_o, rays_d [R,3]; z_vals [R,S]; pts [R,S,3]. */
int plnerf_ray_points(const float* rays_o, const float* rays_d, const float* z_vals, int R, int S,
                      float* pts, plnerf_stream_t stream);

/* clamp(z_new, near, far) ++ z, sorted ascending per ray (run_plnerf.py:731-734).
 * z [R,S], z_new [R,N] -> out [R,S+N]; S+N <= 1024. */
int plnerf_merge_sort(const float* z, const float* z_new, const float* near, const float* far,
                      int R, int S, int N, float* out, plnerf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * The coarse pass's epilogue as one launch (run_plnerf.py:714-735, piecewise-linear mode):
 * raw2outputs -> sample_pdf_reformulation -> clamp -> sort(cat) -> sample positions, plus
 * z_std = std(clamped samples, unbiased=False) (:752).  Results are bit-identical to the sequence
 * plnerf_quad_fwd, plnerf_sample_pl, plnerf_merge_sort, plnerf_ray_points on the same inputs;
 * weights, tau, T and the cdf stay on chip (their output pointers may be NULL).
 *   u: [R,N] draws (u_row_stride == N), one shared row (0), or NULL = drawn in the kernel from the
 *      counter-based generator below (stream id 1) for global ray ids ray_id0 .. ray_id0 + R - 1.
 *   outputs: rgb_map [R,3], disp_map, acc_map, depth_map [R] (the coarse maps rgb0, ...),
 *            z_fine [R,S+N] sorted, pts [R,S+N,3], z_std [R].   S + N <= 1024. */
int plnerf_coarse_epilogue(const float* raw, const float* z, const float* near, const float* far,
                           const float* rays_o, const float* rays_d, const float* noise, const float* u,
                           int u_row_stride, uint64_t seed, uint32_t step, int ray_id0, int R, int S, int N,
                           int color_mode, int white_bkgd, int farcolorfix, float zero_tol, float epsilon,
                           float* rgb_map, float* disp_map, float* acc_map, float* depth_map, float* weights,
                           float* tau, float* T, float* z_fine, float* pts, float* z_std,
                           plnerf_stream_t stream);

/* The depth-supervised variant's LAST stage as one launch (depth_supervised_exps/run_nerf_sample_based_depth.py:
 * 909-934, piecewise-linear mode): raw2outputs of the final pass, then sample_pdf_reformulation_return_u on ITS
 * weights / tau / T -> the depth hypotheses pred_hyp (not clamped), and z_std = std(pred_hyp, unbiased=False).
 * Bit-identical to plnerf_quad_fwd followed by plnerf_sample_pl on the same inputs.  Outputs as plnerf_quad_fwd
 * (weights [R,S+1], tau, T [R,S+2]: the sampler's backward and the caller need them) plus samples [R,N],
 * inds [R,N] and -- if u_out is given -- the draws used [R,N].  u as in plnerf_coarse_epilogue (NULL: drawn in the
 * kernel, counter stream 4).  Backward: plnerf_sample_pl_bwd, then plnerf_quad_bwd with its g_tau / g_T. */
int plnerf_fine_epilogue(const float* raw, const float* z, const float* near, const float* far,
                         const float* rays_d, const float* noise, const float* u, int u_row_stride,
                         uint64_t seed, uint32_t step, int ray_id0, int R, int S, int N, int color_mode,
                         int white_bkgd, int farcolorfix, float zero_tol, float epsilon, float* rgb_map,
                         float* disp_map, float* acc_map, float* depth_map, float* weights, float* tau,
                         float* T, float* samples, int64_t* inds, float* u_out, float* z_std,
                         plnerf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * The caller side of the path (run_plnerf.py:1259-1296), device-side.
 *
 * Random draws: Philox4x32-10 keyed by `seed`, counter = (global ray id, column / 4, stream id,
 * step) -> 24-bit uniforms in [0, 1) like torch.rand.  A draw depends on the ray's GLOBAL id only,
 * so a batch rendered by one rank or sharded over eight sees the same numbers.
 * Stream ids: 0 = t_rand (stratified jitter, :700-705), 1 = u (importance samples). */
int plnerf_uniform(uint64_t seed, uint32_t stream_id, uint32_t step, int ray_id0, int R, int n,
                   float* out /* [R,n] */, plnerf_stream_t stream);
/* The same counters through Box-Muller: standard normal draws [R, n] keyed on (seed, step, stream_id, ray_id0 + row,
 * column) -- the density noise of raw2outputs (run_plnerf.py:568-570: torch.randn(...) * raw_noise_std), invariant
 * to how a batch is sharded over ranks.  (The reference's own stream is torch's generator; only the distribution is
 * the reference's.) */
int plnerf_normal(uint64_t seed, uint32_t stream_id, uint32_t step, int ray_id0, int R, int n, float* out,
                  plnerf_stream_t stream);

/* Training rays of one view (run_plnerf.py:1259-1281 + run_nerf_helpers.py:162-171 + the ray
 * packing of render, run_plnerf.py:146-164): ray i (global id ray_id0 + i) looks through pixel
 * perm(ray_id0 + i) of the window [crop_r0, crop_r0 + crop_rows) x [crop_c0, crop_c0 + crop_cols),
 * where perm is a bijection of the window's pixels keyed by (seed, step) -- distinct pixels, as
 * np.random.choice(..., replace=False) gives, without building the H x W grid.  c2w_host: 12 floats
 * in HOST memory (rows of the 3x4 camera-to-world matrix, read during the call).
 * Outputs (device): rays_o, rays_d [R,3]; viewdirs [R,3] = rays_d / |rays_d| (nullable); near_out,
 * far_out [R]; target [R,3] = image[row, col, :] (nullable, image [H,W,3]); pixels [R,2] int32
 * (row, col; nullable). */
int plnerf_select_rays(int H, int W, float fx, float fy, float cx, float cy, const float* c2w_host,
                       const float* image, int crop_r0, int crop_c0, int crop_rows, int crop_cols,
                       uint64_t seed, uint32_t step, int ray_id0, int R, float near, float far,
                       float* rays_o, float* rays_d, float* viewdirs, float* near_out, float* far_out,
                       float* target, int* pixels, plnerf_stream_t stream);

/* ndc_rays (run_nerf_helpers.py:184-201; called by render, run_plnerf.py:153-155, with near = 1): forward-facing rays
 * [n,3] warped to normalised device coordinates in one launch, bit-identical to the reference's fp32 expressions
 * (focal and near are the call's Python floats: the scale factors -1 / (W / (2 focal)), -1 / (H / (2 focal)) and
 * 2 near are formed in double and rounded once, as torch does with a scalar operand).  o_out / d_out may be rays_o /
 * rays_d. */
int plnerf_ndc_rays(int H, int W, double focal, double near, const float* rays_o, const float* rays_d, int n,
                    float* o_out, float* d_out, plnerf_stream_t stream);

/* plnerf_stratified_z + plnerf_ray_points in one launch (run_plnerf.py:683-708), bit-identical to
 * them.  perturb != 0: jitter from t_rand [R,S], or (t_rand == NULL) drawn in the kernel (stream 0). */
int plnerf_coarse_samples(const float* rays_o, const float* rays_d, const float* near, const float* far,
                          const float* t_vals, const float* t_rand, uint64_t seed, uint32_t step,
                          int ray_id0, int R, int S, int lindisp, int perturb, float* z_vals, float* pts,
                          plnerf_stream_t stream);

/* img2mse(rgb, target) + img2mse(rgb0, target) (run_plnerf.py:1287-1296; run_nerf_helpers.py:17):
 * loss3 [4] = {total, fine, coarse, psnr = -10 log10(fine)}; g_rgb, g_rgb0 [R,3] = d total / d rgb,
 * d total / d rgb0.
 * rgb0 may be NULL (single-pass configuration).  coarse_loss (NULL, or the loss3 of an earlier call on the coarse
 * image alone; needs rgb0 == NULL): its [1] is taken as the coarse term -- a caller that runs the coarse network's
 * loss and backward ahead of the fine pass (the two sums of :1296 are independent) still gets the reference's total.
 * workspace: PLNERF_IMAGE_LOSS_WORKSPACE_BYTES, zeroed by the caller ONCE (each launch leaves it zeroed; one per
 * stream that launches concurrently, and not shared with plnerf_depth_loss, which lays its own out differently).  Deterministic (fp64 partial sums added in workgroup order). */
#define PLNERF_IMAGE_LOSS_WORKSPACE_BYTES 4096
int plnerf_image_loss(const float* rgb, const float* rgb0, const float* target, int R, float* loss3,
                      float* g_rgb, float* g_rgb0, const float* coarse_loss, void* workspace,
                      plnerf_stream_t stream);

/* The depth-supervised loop's loss (depth_supervised_exps/run_nerf_sample_based_depth.py:1126-1150):
 *   total = img2mse(rgb, target) + space_carving_weight * compute_space_carving_loss(pred_hyp, target_h)
 *           + img2mse(rgb0, target)
 * with compute_space_carving_loss of depth_supervised_exps/model/run_nerf_helpers.py:52-86:
 * distances[h, r, p] = mask[r] * |pred_hyp[r, p] - target_h[h, r, p]|, zeroed below `threshold` (> 0); then
 * is_joint = 0 (:79-84): min over the n_hyp hypotheses per ray and point, mean over points and rays;
 * is_joint != 0 (:72-77): mean over the rays first, min over the hypotheses per POINT column (the hypothesis is chosen
 * per image), mean over the points.  joint_choice (is_joint only; NULL = choose from this call's rays): [n_points] hypothesis
 * indices chosen by the caller -- a batch SHARDED over ranks must choose per global batch, as the reference does on its gathered
 * output (run_nerf_sample_based_depth.py:564, 585: nn.DataParallel): plnerf_depth_joint_sums on every shard, the sums added
 * over the shards, argmin over the hypotheses of (float)(sum / global rays) per column, first minimum on ties (ABI 600).
 * pred_hyp [R, n_points]; target_h [n_hyp, R, target_points] with target_points = 1 or n_points; mask [R] or NULL.
 * loss5 [5] = {total, image (fine), image (coarse), space carving (unweighted), psnr of the fine image term};
 * g_rgb, g_rgb0 [R, 3], g_hyp [R, n_points] = d total / d (rgb, rgb0, pred_hyp).  rgb0 and pred_hyp may be NULL
 * (single-pass configuration; warm-up iterations without the depth term).  workspace: PLNERF_DEPTH_LOSS_WORKSPACE_BYTES
 * of device memory, 8-byte aligned, ZEROED ONCE by the caller (the kernel leaves it zeroed; one workspace per stream that
 * may run this call concurrently).  fp64 partial sums added in a fixed order: deterministic. */
#define PLNERF_DEPTH_LOSS_WORKSPACE_BYTES 4096
int plnerf_depth_loss(const float* rgb, const float* rgb0, const float* target, const float* pred_hyp,
                      const float* target_h, const float* mask, int R, int n_points, int n_hyp, int target_points,
                      int is_joint, const int* joint_choice, float space_carving_weight, float threshold, float* loss5,
                      float* g_rgb, float* g_rgb0, float* g_hyp, void* workspace, plnerf_stream_t stream);

/* The is_joint branch's column sums over THIS call's rays (model/run_nerf_helpers.py:72-75 before the mean's division):
 * sums [n_hyp, n_points] (fp64) = sum over r of mask[r] * |pred_hyp[r, p] - target_h[h, r, p]| (zeroed below threshold).
 * Fixed summation order: deterministic.  For plnerf_depth_loss's joint_choice (ABI 600). */
int plnerf_depth_joint_sums(const float* pred_hyp, const float* target_h, const float* mask, int R, int n_points,
                            int n_hyp, int target_points, float threshold, double* sums, plnerf_stream_t stream);

/* run_network's input assembly for a caller-side encoding (depth_supervised_exps/run_nerf_sample_based_depth.py:52-68
 * with the Embedder of depth_supervised_exps/model/run_nerf_helpers.py:100-130; input_scale = 1 gives the NVS
 * script's, run_plnerf.py:78-92 + run_nerf_helpers.py:24-54):
 *   embedded[row] = [ gamma_fx((pts[row] - bb_center) * bb_scale) | gamma_fd(viewdirs[row / samples_per_ray]) | cam ]
 *   gamma_L(x) = [x, sin(x s 2^0), cos(x s 2^0), ..., sin(x s 2^(L-1)), cos(x s 2^(L-1))],  s = input_scale,
 * each argument evaluated as fl(fl(x s) 2^k), the reference's association.  viewdirs NULL: the position block only.
 * cam [n_cam] (device; the per-image camera code, run_nerf_sample_based_depth.py:1122-1123) is repeated on every row.
 * bb_center_host: three HOST floats (NULL = 0).  embedded [n_rows, 3 + 6 fx (+ 3 + 6 fd + n_cam)] is what
 * plnerf_mlp_fwd takes as `embedded`. */
int plnerf_embed_rows(const float* pts, const float* viewdirs, const float* cam, int n_rows, int samples_per_ray,
                      int n_freqs_xyz, int n_freqs_dir, int n_cam, float input_scale, const float* bb_center_host,
                      float bb_scale, float* embedded, plnerf_stream_t stream);

/* A generic dense product on the exact-fp32 MFMA (ABI 600) -- the layers of NeRF shapes the fused MLP entries below are
 * not compiled for (run_nerf_helpers.py:76-128 builds any netdepth / netwidth / skip list; pl-nerf_amd/generic.py runs
 * them layer by layer through this call):
 *   C[M,N] (row stride ldc) = gate(A)[M,K] . B[K,N] (+ bias[N]) (+ C if accumulate) (relu last if relu)
 * element (i,k) of A at a[i a_row_stride + k a_col_stride], (k,j) of B at b[k b_row_stride + j b_col_stride] (a transposed
 * view is a stride swap); gate (NULL, or indexed like A): A(i,k) counts only where gate(i,k) > 0 (the ReLU's derivative in
 * the two backward products); ones_col: B's LAST column (j = N - 1) is not read from memory but taken as all ones (the bias
 * gradient as an extra column of dW = gate(G)^T [X | 1]).  fp32 accumulation, k ascending: deterministic.  M <= 4,194,240.
 * k_splits > 1 (a product with few output tiles and a long k, i.e. a weight gradient): the k range is dealt out over up to
 * k_splits workgroups per tile, partial products in `partials` (k_splits * M * N floats, caller-owned), added in order by a
 * second launch; k_splits <= 1: partials may be NULL. */
int plnerf_gemm_f32(const float* a, int64_t a_row_stride, int64_t a_col_stride, const float* b, int64_t b_row_stride,
                    int64_t b_col_stride, const float* bias, const float* gate, int M, int N, int K, int relu, int accumulate,
                    int ones_col, float* c, int64_t ldc, int k_splits, float* partials, plnerf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * The MLP: run_network (run_plnerf.py:78-92) = Embedder (run_nerf_helpers.py:24-54) +
 * NeRF.forward (:105-128), fused: positional encoding -> 8x256 trunk with skip ->
 * sigma / feature / view layer / rgb, activations staged in LDS, contractions on MFMA.
 *
 * params[24]: DEVICE pointers to the fp32 parameter tensors in state_dict order
 *   pts_linears.{0..7}.{weight,bias}, views_linears.0.{weight,bias},
 *   feature_linear.{weight,bias}, alpha_linear.{weight,bias}, rgb_linear.{weight,bias}
 * (the host array itself lives in host memory and is read during the call).
 */

/* Size in bytes of the packed-weight buffer for a precision mode (weight sections + a 16-byte status block). */
size_t plnerf_mlp_packed_bytes(int precision);
/* Byte offset of the uint32 range status word inside the packed buffer (see PLNERF_RANGE_*). */
size_t plnerf_mlp_status_offset(int precision);
/* Re-layout the 24 parameter tensors into MFMA fragment order (call after every
 * optimizer step; cheap: one pass over 2.4 MB). */
int plnerf_mlp_pack_weights(const float* const* params, int precision, int input_ch,
                            int input_ch_views, void* packed, plnerf_stream_t stream);

/* Bytes of forward state saved for the backward pass (activations of every layer) and of
 * backward scratch (per-layer pre-activation gradients, split-K partial sums). */
size_t plnerf_mlp_saved_bytes(int n_rows, int precision);
size_t plnerf_mlp_bwd_workspace_bytes(int n_rows, int precision);
/* Layout tag of the saved state the forward of this configuration writes (has_embedded: the call passes `embedded`
 * instead of pts / viewdirs).  Opaque to the caller: hand it to plnerf_mlp_bwd with the saved buffer.  (The half
 * modes have two forward kernels; one stores its 256-wide planes row-major, the other in 32-row tiles of the MFMA
 * accumulator layout, which the weight-gradient stage reads directly.  Rows of the saved state are padded to a
 * multiple of 256: plnerf_mlp_saved_bytes accounts for it.) */
int plnerf_mlp_saved_layout(int precision, int has_embedded, int fwd_kernel);

/* Which forward kernel a 16-bit mode runs on.  AUTO is the product's choice (mlp_api.hip: the register-resident kernel
 * wherever it is the faster one); RR / PP force the register-resident / the ping-pong kernel wherever that variant
 * exists, for A/B measurements and for the test suite's passes over both.  An explicit argument of the two calls it
 * affects -- the library reads no environment variable and keeps no process-wide setting, so the saved layout is a
 * pure function of (precision, has_embedded, fwd_kernel).  Ignored by PLNERF_PREC_FP32. */
#define PLNERF_FWD_KERNEL_AUTO 0
#define PLNERF_FWD_KERNEL_RR 1
#define PLNERF_FWD_KERNEL_PP 2

/* Forward.  Either (pts [n_rows,3] AND viewdirs [n_rows/samples_per_ray, 3]) with
 * embedded == NULL -- the encoding gamma(x) = [x, sin(x s 2^0), cos(x s 2^0), ...] is computed in the kernel prologue,
 * once per sample for xyz and from the per-ray direction: input_ch = 3 + 6 L (L <= 10), input_ch_views = 3 + 6 M
 * (M <= 4), s = input_scale (1: run_nerf_helpers.py:24-54; pi: the Embedder of
 * depth_supervised_exps/model/run_nerf_helpers.py:100-130, arguments evaluated as fl(fl(x pi) 2^k)) -- or
 * embedded [n_rows, input_ch + input_ch_views] (a caller-supplied encoding; NeRF.forward's
 * own signature).  saved == NULL for inference.  raw_out [n_rows,4].  fwd_kernel: PLNERF_FWD_KERNEL_* (pass the same
 * value to plnerf_mlp_saved_layout).
 * density_beta: 0 = the four channels leave as the network computes them (run_nerf_helpers.py:124); > 0 = the density
 * channel leaves as softplus_beta(sigma) = log(1 + exp(beta sigma)) / beta (beta sigma > 20: sigma) -- the NeRF of the
 * depth-supervised variant, F.softplus(alpha, beta=10), depth_supervised_exps/model/run_nerf_helpers.py:200 -- in the
 * kernel's last store instead of three element-wise launches behind it. */
int plnerf_mlp_fwd(const void* packed, int precision, const float* pts, const float* viewdirs,
                   const float* embedded, int input_ch, int input_ch_views, int n_rows,
                   int samples_per_ray, float input_scale, float density_beta, float* raw_out, void* saved,
                   int fwd_kernel, plnerf_stream_t stream);

/* Backward: g_raw [n_rows,4] -> gradients of all 24 parameter tensors, written (not
 * accumulated) to grads[24] (device pointers, same shapes as params).  Needs the `saved`
 * buffer of the matching forward call and a workspace of
 * plnerf_mlp_bwd_workspace_bytes().  Inputs (pts / viewdirs) receive no gradient, as on
 * the reference path (they do not depend on parameters; z_samples is detached,
 * run_plnerf.py:728).  density_beta / raw_out: the forward's density activation and ITS OUTPUT [n_rows,4] (needed when
 * density_beta > 0, else may be NULL): g_raw is then the gradient with respect to the activated output, and the
 * activation's derivative sigmoid(beta sigma) = 1 - exp(-beta raw_out[., 3]) is applied on entry (into the workspace:
 * g_raw itself is not written).  status_out (may be NULL): one float, set to 1 if the network's range status word
 * (plnerf_mlp_status_offset) is non-zero when the gradients are complete, else 0 -- a data-parallel caller puts it
 * behind the gradients in the buffer it all-reduces (SUM), so "some rank's forward left the half range" reaches
 * every rank with the gradient itself and can guard plnerf_adam_step there.
 * g_absmax / n_absmax (may be NULL / 0; ABI 500; 16-bit modes without a density activation): n_absmax device uint32
 * whose maximum (as unsigned integers) is the fp32 bit pattern of max |g_raw| -- plnerf_quad_bwd's absmax_out -- the call
 * then skips its own pass over g_raw (and its memset): the gradient chain's workgroups take the maximum themselves. */
int plnerf_mlp_bwd(const void* packed, int precision, const float* g_raw, const uint32_t* g_absmax, int n_absmax, int input_ch,
                   int input_ch_views, int n_rows, const void* saved, int saved_layout,
                   const float* raw_out, float density_beta, void* workspace,
                   float* const* grads, float* status_out, plnerf_stream_t stream);

/* The backward of SEVERAL networks that share precision, input widths and density activation -- the coarse and the fine
 * network of one optimisation step, whose upstream gradients both exist before either backward starts (the samples are
 * detached, run_plnerf.py:728; the loss is a sum of two image terms, :1287-1296) -- as ONE launch sequence: one
 * gradient-chain grid and one launch of each weight-gradient kernel cover every job (16-bit modes; the exact-fp32 mode
 * runs the jobs one after the other).  Separate calls end each kernel on a partial round of the 256 CUs and pay every
 * launch ramp twice: 0.14-0.16 ms of a 6.4 ms step (profiles/r05_merged_bwd_bound.txt).  Every array has n_jobs entries
 * (host memory, read during the call) and means what the argument of the same name means to plnerf_mlp_bwd; grads holds
 * n_jobs x 24 device pointers, job after job; g_absmax (with n_absmax), raw_out and status_out may be NULL as a whole or
 * per entry.
 * plnerf_mlp_bwd(...) is plnerf_mlp_bwd_multi(1, ...), bit for bit. */
#define PLNERF_MAX_BWD_JOBS 2
int plnerf_mlp_bwd_multi(int n_jobs, const void* const* packed, int precision, const float* const* g_raw,
                         const uint32_t* const* g_absmax, const int* n_absmax, int input_ch, int input_ch_views, const int* n_rows,
                         const void* const* saved, const int* saved_layout, const float* const* raw_out,
                         float density_beta, void* const* workspace, float* const* grads,
                         float* const* status_out, plnerf_stream_t stream);

/* Gradient of the same backward with respect to the network's INPUT rows (what autograd gives the reference when the
 * rows handed to NeRF.forward, run_nerf_helpers.py:105-128, require a gradient -- no reference training path does,
 * run_plnerf.py:728 detaches the samples): g_embedded [n_rows, input_ch + input_ch_views], the position channels
 * through layer 0 and the skip layer, the direction channels through the view layer.  Call it after plnerf_mlp_bwd,
 * on the same stream, with that call's workspace (it reads the pre-activation gradients left there) and the 24
 * parameter tensors (device pointers, host table, as plnerf_mlp_pack_weights takes them). */
int plnerf_mlp_input_grad(const float* const* params, int precision, int input_ch, int input_ch_views, int n_rows,
                          const void* workspace, float* g_embedded, plnerf_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Fused Adam step over a flat parameter buffer (torch.optim.Adam semantics as used at
 * run_plnerf.py:446-447, 1302-1303: betas (0.9,0.999), eps 1e-8, no weight decay, no
 * amsgrad).  step >= 1 is the step count AFTER this update.  grad_scale multiplies the
 * gradient first (1/world_size after an all-reduce sum); clip_value > 0 then clamps every entry to
 * [-clip_value, clip_value] -- torch.nn.utils.clip_grad_value_ as the depth-supervised loop applies it between
 * backward and step (depth_supervised_exps/run_nerf_sample_based_depth.py:1156; `grad` itself is left as it is);
 * clip_value <= 0: no clipping.  skip_if_set, skip_if_set2 (nullable): device uint32 words -- the status words of the
 * networks' packed buffers (one Adam over two networks: both) -- read by the kernel; while either is non-zero the
 * launch changes nothing (a step whose forward left the half range must not reach the weights) and adds 1 to
 * *withheld (nullable device uint32), so that the caller can take the steps that did not happen out of its step count. */
int plnerf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                     int64_t n, float lr, float beta1, float beta2, float eps, int step,
                     float grad_scale, float clip_value, const uint32_t* skip_if_set, const uint32_t* skip_if_set2,
                     uint32_t* withheld, plnerf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PLNERF_HIP_H */
