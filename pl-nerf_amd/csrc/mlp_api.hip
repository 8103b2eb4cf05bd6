// MLP entry points of the C ABI (include/plnerf_hip.h): argument validation and dispatch on
// the precision mode.  Kernels live in mlp_f32.hip (exact fp32 MFMA; also the shared
// weight-gradient stage) and mlp_bf16.hip (bf16 / 3-term bf16 split MFMA).
#include "common.h"
#include "mlp_internal.h"
#include "mlp_layout.h"

using namespace plnerf;

namespace {
inline int ns_of(int precision) {
    return (precision == PLNERF_PREC_BF16 || precision == PLNERF_PREC_F16) ? 1
           : (precision == PLNERF_PREC_BF16X3 || precision == PLNERF_PREC_F16X3) ? 2 : 0;
}
inline int f16_of(int precision) { return precision == PLNERF_PREC_F16X3 || precision == PLNERF_PREC_F16; }
inline bool known(int precision) { return precision >= PLNERF_PREC_FP32 && precision <= PLNERF_PREC_F16; }
// network input widths the padded GEMMs can hold (the reference's defaults are 63 / 27)
inline bool geometry_ok(int input_ch, int input_ch_views) {
    return input_ch >= 1 && input_ch <= lay::PE_K && input_ch_views >= 1 && input_ch_views <= lay::DPE_K;
}
// Forward kernels of the half-element modes.  The split mode (f16x3) runs on the register-resident kernel (mlp_rr.hip),
// inference (+20 % over the ping-pong kernel) and training (-15 % per launch), with the in-kernel encoding or a
// caller-embedded input; its saved planes leave in the tiled layout of mlp_layout.h, which the weight-gradient stage
// reads as it is.  The plain mode (f16) uses it for inference with the in-kernel encoding (two row tiles per wave:
// +18 %) and keeps the ping-pong kernel for the training forward (its 5.3 KB of plane stores per row come out of eight
// waves per CU there, out of four here: 1.22 vs 1.39 ms) and for embedded inputs.  `fwd_kernel` (an ABI argument:
// PLNERF_FWD_KERNEL_AUTO / _RR / _PP) forces one kernel wherever it exists (A/B measurements, and the test suite's
// other passes); the library itself reads no environment.
inline bool kernel_arg_ok(int k) { return k == PLNERF_FWD_KERNEL_AUTO || k == PLNERF_FWD_KERNEL_RR || k == PLNERF_FWD_KERNEL_PP; }
inline bool use_rr(bool training, int ns, bool embedded, int forced) {
    if (embedded && !impl::rr_embedded_ok(ns)) return false;      // a caller-embedded input: split mode only
    return forced == PLNERF_FWD_KERNEL_RR || (forced == PLNERF_FWD_KERNEL_AUTO && (ns == 2 || !training));
}
// bf16 elements: inference of both modes; the split mode also for the training forward and caller-embedded inputs
inline bool use_rr_bf16(bool training, int ns, bool embedded, int forced) {
    if (forced == PLNERF_FWD_KERNEL_PP) return false;
    return ns == 2 || (!training && !embedded);
}
}  // namespace

// bytes of the weight sections; the 16-byte status block follows them
inline size_t sections_bytes(int precision) {
    if (precision == PLNERF_PREC_FP32) return impl::f32_packed_bytes();
    if (ns_of(precision))
        return impl::bf16_packed_bytes(ns_of(precision)) + impl::rr_packed_bytes(ns_of(precision));      // (+ the register-resident section)
    return 0;
}
inline unsigned* status_word(void* packed, int precision) {
    return ns_of(precision) ? (unsigned*)((unsigned char*)packed + sections_bytes(precision)) : nullptr;      // every 16-bit mode
}

inline float* compose_block(const void* packed, int precision) {      // (16-bit modes)
    return (float*)((unsigned char*)const_cast<void*>(packed) + impl::bf16_compose_offset(ns_of(precision)));
}

extern "C" size_t plnerf_mlp_packed_bytes(int precision) {
    const size_t n = known(precision) ? sections_bytes(precision) : 0;
    return n ? n + 16 : 0;
}

extern "C" size_t plnerf_mlp_status_offset(int precision) { return known(precision) ? sections_bytes(precision) : 0; }

extern "C" int plnerf_mlp_pack_weights(const float* const* params, int precision, int input_ch,
                                       int input_ch_views, void* packed, plnerf_stream_t stream) {
    if (!params || !packed || !geometry_ok(input_ch, input_ch_views)) return PLNERF_EINVAL;
    if (!known(precision)) return PLNERF_ENOSYS;
    for (int i = 0; i < PLNERF_N_PARAM_TENSORS; ++i)
        if (!params[i]) return PLNERF_EINVAL;
    if (precision == PLNERF_PREC_FP32) return impl::f32_pack(params, input_ch, input_ch_views, packed, (hipStream_t)stream);
    // 16-bit modes: the composed view layer first (W_c = W_vf W_f, b_c; mlp_layout.h) -- the operand sections are packed from it
    float* cb = compose_block(packed, precision);
    int rc = impl::compose_pack(params, input_ch_views, cb, (hipStream_t)stream);
    if (rc) return rc;
    rc = impl::bf16_pack(params, input_ch, input_ch_views, ns_of(precision), f16_of(precision), packed,
                         status_word(packed, precision), (hipStream_t)stream);
    if (rc) return rc;
    void* rr_section = (unsigned char*)packed + impl::bf16_packed_bytes(ns_of(precision));
    return f16_of(precision) ? impl::rr_pack(params, input_ch, input_ch_views, ns_of(precision), cb, rr_section, (hipStream_t)stream)
                             : impl::rr_pack_bf16(params, input_ch, input_ch_views, ns_of(precision), cb, rr_section, (hipStream_t)stream);
}

// fp32 mode: fp32 planes; 16-bit MFMA modes: half planes (mlp_layout.h)
extern "C" size_t plnerf_mlp_saved_bytes(int n_rows, int precision) {
    if (!known(precision) || n_rows < 0) return 0;
    if (precision == PLNERF_PREC_FP32) return (size_t)lay::SAVED_PER_ROW * (size_t)n_rows * sizeof(float);
    return (size_t)lay::SVH_BYTES_PER_ROW * lay::sv_rows((size_t)n_rows);      // rows padded to whole workgroup tiles (SV_ROW_PAD)
}

// layout of the 256-wide saved planes the forward of this configuration writes (lay::SV_LAYOUT_*): the backward is
// told, so that whichever forward kernel ran, the weight-gradient stage reads its planes as they are
extern "C" int plnerf_mlp_saved_layout(int precision, int has_embedded, int fwd_kernel) {
    if (!known(precision) || !kernel_arg_ok(fwd_kernel)) return PLNERF_EINVAL;
    if (!ns_of(precision)) return lay::SV_LAYOUT_ROWS;
    if (f16_of(precision)) return use_rr(true, ns_of(precision), has_embedded != 0, fwd_kernel) ? lay::SV_LAYOUT_TILED : lay::SV_LAYOUT_ROWS;
    return use_rr_bf16(true, ns_of(precision), has_embedded != 0, fwd_kernel) ? lay::SV_LAYOUT_TILED : lay::SV_LAYOUT_ROWS;
}

extern "C" size_t plnerf_mlp_bwd_workspace_bytes(int n_rows, int precision) {
    if (!known(precision) || n_rows < 0) return 0;
    const size_t partials = ((size_t)lay::MAX_SPLITS * lay::PART_PER_SPLIT + (size_t)lay::MAX_HEAD_WGS * lay::HEAD_PART +
                             (size_t)lay::GRED_FLOATS) * sizeof(float);      // (+ the reduced G and s of the composed view layer)
    const size_t g_eff = (size_t)n_rows * 16;      // the upstream gradient after the density activation's derivative
    if (precision == PLNERF_PREC_FP32) return (size_t)lay::DZ_PER_ROW * (size_t)n_rows * sizeof(float) + partials + g_eff;
    return impl::h16_dz_bytes(n_rows) + lay::WSH_SCALARS_BYTES + partials + g_eff;
}

extern "C" int plnerf_mlp_fwd(const void* packed, int precision, const float* pts, const float* viewdirs,
                              const float* embedded, int input_ch, int input_ch_views, int n_rows,
                              int samples_per_ray, float input_scale, float density_beta, float* raw_out, void* saved,
                              int fwd_kernel, plnerf_stream_t stream) {
    if (!known(precision)) return PLNERF_ENOSYS;
    if (!kernel_arg_ok(fwd_kernel)) return PLNERF_EINVAL;
    if (n_rows < 0 || !geometry_ok(input_ch, input_ch_views) || !(density_beta >= 0.0f) || !(density_beta < 1e6f)) return PLNERF_EINVAL;
    const impl::FwdOpt opt{input_scale, density_beta};
    // the in-kernel encoding: 3 + 6 L position channels (L <= 10) and 3 + 6 M direction channels (M <= 4) -- a prefix of
    // the reference default's 63 | 27 (the unused bands meet zero-padded weights) -- with the encoder's input scale
    if (!embedded && ((input_ch - 3) % 6 != 0 || input_ch < 3 || input_ch > lay::XYZ_CH || (input_ch_views - 3) % 6 != 0 ||
                      input_ch_views < 3 || input_ch_views > lay::DIR_CH || !(input_scale > 0.0f) || !(input_scale < 1e6f)))
        return PLNERF_EINVAL;
    if (n_rows == 0) return PLNERF_OK;
    if (!packed || !raw_out) return PLNERF_EINVAL;
    if (!embedded && (!pts || !viewdirs || samples_per_ray < 1)) return PLNERF_EINVAL;
    if (precision == PLNERF_PREC_FP32)
        return impl::f32_fwd(packed, pts, viewdirs, embedded, input_ch, input_ch_views, n_rows, samples_per_ray,
                             opt, raw_out, saved, (hipStream_t)stream);
    if (f16_of(precision) && use_rr(saved != nullptr, ns_of(precision), embedded != nullptr, fwd_kernel))
        return impl::rr_fwd(packed, (const unsigned char*)packed + impl::bf16_packed_bytes(ns_of(precision)),
                            ns_of(precision), pts, viewdirs, embedded, input_ch, input_ch_views, n_rows, samples_per_ray,
                            opt, raw_out, saved, status_word(const_cast<void*>(packed), precision), (hipStream_t)stream);
    // bf16 elements: the register-resident kernel serves inference with the in-kernel encoding (unless pp is forced)
    if (!f16_of(precision) && use_rr_bf16(saved != nullptr, ns_of(precision), embedded != nullptr, fwd_kernel))
        return impl::rr_fwd_bf16(packed, (const unsigned char*)packed + impl::bf16_packed_bytes(ns_of(precision)),
                                 ns_of(precision), pts, viewdirs, embedded, input_ch, input_ch_views, n_rows, samples_per_ray,
                                 opt, raw_out, saved, status_word(const_cast<void*>(packed), precision), (hipStream_t)stream);
    return impl::bf16_fwd(packed, ns_of(precision), f16_of(precision), pts, viewdirs, embedded, input_ch,
                          input_ch_views, n_rows, samples_per_ray, opt, raw_out, saved,
                          status_word(const_cast<void*>(packed), precision), (hipStream_t)stream);
}

// The backward of up to PLNERF_MAX_BWD_JOBS networks that share precision, input widths and density activation (the
// coarse and the fine network of one training step) in ONE launch sequence: one dgrad grid and one main / thin / head /
// reduce launch of the weight-gradient stage cover every job (16-bit modes; the exact-fp32 mode runs the jobs one after
// the other).  plnerf_mlp_bwd is the one-job case.
extern "C" int plnerf_mlp_bwd_multi(int n_jobs, const void* const* packed, int precision, const float* const* g_raw,
                                    const uint32_t* const* g_absmax, const int* n_absmax, int input_ch,
                                    int input_ch_views, const int* n_rows, const void* const* saved, const int* saved_layout,
                                    const float* const* raw_out, float density_beta, void* const* workspace,
                                    float* const* grads, float* const* status_out, plnerf_stream_t stream) {
    if (n_jobs < 1 || n_jobs > PLNERF_MAX_BWD_JOBS) return PLNERF_EINVAL;
    if (!known(precision)) return PLNERF_ENOSYS;
    if (!packed || !g_raw || !n_rows || !saved || !saved_layout || !workspace || !grads) return PLNERF_EINVAL;
    if (!geometry_ok(input_ch, input_ch_views)) return PLNERF_EINVAL;
    if (!(density_beta >= 0.0f) || !(density_beta < 1e6f)) return PLNERF_EINVAL;
    const bool act = density_beta > 0.0f;
    for (int j = 0; j < n_jobs; ++j) {
        if (saved_layout[j] != lay::SV_LAYOUT_ROWS && saved_layout[j] != lay::SV_LAYOUT_TILED) return PLNERF_EINVAL;
        if (saved_layout[j] == lay::SV_LAYOUT_TILED && !ns_of(precision)) return PLNERF_EINVAL;
        if (!packed[j] || !g_raw[j] || !saved[j] || !workspace[j] || n_rows[j] < 1) return PLNERF_EINVAL;
        if (act && (!raw_out || !raw_out[j])) return PLNERF_EINVAL;
        for (int i = 0; i < PLNERF_N_PARAM_TENSORS; ++i)
            if (!grads[j * PLNERF_N_PARAM_TENSORS + i]) return PLNERF_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t part_only = ((size_t)lay::MAX_SPLITS * lay::PART_PER_SPLIT + (size_t)lay::MAX_HEAD_WGS * lay::HEAD_PART) * sizeof(float);
    const size_t part_bytes = part_only + (size_t)lay::GRED_FLOATS * sizeof(float);
    if (precision == PLNERF_PREC_FP32) {
        for (int j = 0; j < n_jobs; ++j) {
            float* dz = (float*)workspace[j];
            float* partials = dz + (size_t)lay::DZ_PER_ROW * (size_t)n_rows[j];
            const float* g = g_raw[j];
            if (act) {      // the activation's derivative first: every kernel below reads the effective gradient
                float* g_eff = (float*)((unsigned char*)partials + part_bytes);
                const int rc0 = impl::absmax_act(g, raw_out[j], density_beta, n_rows[j], g_eff, nullptr, st);
                if (rc0) return rc0;
                g = g_eff;
            }
            int rc = impl::f32_dgrad(packed[j], g, n_rows[j], (const float*)saved[j], dz, st);
            if (rc) return rc;
            rc = impl::wgrad(g, n_rows[j], saved[j], dz, nullptr, partials, grads + j * PLNERF_N_PARAM_TENSORS, input_ch,
                             input_ch_views, false, lay::SV_LAYOUT_ROWS, nullptr, status_out ? status_out[j] : nullptr, st);
            if (rc) return rc;
        }
        return PLNERF_OK;
    }
    // 16-bit modes: [dz half planes][max |g_raw|][partials][g_eff]
    impl::DgradJob dj[PLNERF_MAX_BWD_JOBS];
    impl::WgradJob wj[PLNERF_MAX_BWD_JOBS];
    for (int j = 0; j < n_jobs; ++j) {
        unsigned char* ws = (unsigned char*)workspace[j];
        unsigned* gmax = (unsigned*)(ws + impl::h16_dz_bytes(n_rows[j]));
        const unsigned* cand = nullptr;
        int n_cand = 0;
        float* partials = (float*)(ws + impl::h16_dz_bytes(n_rows[j]) + lay::WSH_SCALARS_BYTES);
        const float* g = g_raw[j];
        int rc = PLNERF_OK;
        if (act) {      // one pass: the activation's derivative and the launch scale's maximum
            float* g_eff = (float*)((unsigned char*)partials + part_bytes);
            rc = impl::absmax_act(g, raw_out[j], density_beta, n_rows[j], g_eff, gmax, st);
            g = g_eff;
        } else if (g_absmax && g_absmax[j] && n_absmax && n_absmax[j] > 0) {
            // the caller's producer kernel left the candidates (plnerf_quad_bwd's absmax_out): no pass, no memset -- the
            // gradient chain's workgroups take their maximum and leave it in the workspace's word for the kernels behind
            cand = g_absmax[j];
            n_cand = n_absmax[j];
        } else {
            rc = impl::absmax(g, (size_t)n_rows[j] * 4, gmax, st);
        }
        if (rc) return rc;
        dj[j] = impl::DgradJob{packed[j], ns_of(precision), g, n_rows[j], saved[j], ws, gmax, cand, n_cand};
        wj[j] = impl::WgradJob{g, n_rows[j], saved[j], ws, gmax, partials, grads + j * PLNERF_N_PARAM_TENSORS, input_ch,
                               input_ch_views, saved_layout[j], status_word(const_cast<void*>(packed[j]), precision),
                               status_out ? status_out[j] : nullptr, compose_block(packed[j], precision),
                               (float*)((unsigned char*)partials + part_only)};
    }
    const int rc = impl::bf16_dgrad(n_jobs, dj, st);
    if (rc) return rc;
    return impl::wgrad_h16_multi(n_jobs, wj, st);
}

extern "C" int plnerf_mlp_bwd(const void* packed, int precision, const float* g_raw, const uint32_t* g_absmax, int n_absmax,
                              int input_ch, int input_ch_views, int n_rows, const void* saved, int saved_layout,
                              const float* raw_out, float density_beta, void* workspace,
                              float* const* grads, float* status_out, plnerf_stream_t stream) {
    if (!grads) return PLNERF_EINVAL;
    return plnerf_mlp_bwd_multi(1, &packed, precision, &g_raw, &g_absmax, &n_absmax, input_ch, input_ch_views, &n_rows, &saved,
                                &saved_layout, &raw_out, density_beta, &workspace, grads, &status_out, stream);
}

extern "C" int plnerf_mlp_input_grad(const float* const* params, int precision, int input_ch, int input_ch_views,
                                     int n_rows, const void* workspace, float* g_embedded, plnerf_stream_t stream) {
    if (!known(precision)) return PLNERF_ENOSYS;
    if (!params || !workspace || !g_embedded || n_rows < 1 || !geometry_ok(input_ch, input_ch_views)) return PLNERF_EINVAL;
    for (int i = 0; i < PLNERF_N_PARAM_TENSORS; ++i)
        if (!params[i]) return PLNERF_EINVAL;
    if (precision == PLNERF_PREC_FP32)
        return impl::input_grad(params, n_rows, workspace, nullptr, false, input_ch, input_ch_views, g_embedded, (hipStream_t)stream);
    const unsigned char* ws = (const unsigned char*)workspace;      // [dz half planes][max |g_raw|] ... as plnerf_mlp_bwd laid them out
    return impl::input_grad(params, n_rows, ws, (const unsigned*)(ws + impl::h16_dz_bytes(n_rows)), true, input_ch,
                            input_ch_views, g_embedded, (hipStream_t)stream);
}
