// Weight section of the register-resident forward on bf16 elements (mlp_rr_body.inc with RR_BF16): the packing kernel
// and the dispatch of its two inference instantiations (mlp_rr_kb_*_infer.hip).
#define RR_BF16 1
#include "mlp_rr_body.inc"

namespace plnerf {
namespace impl {

int rr_launch_bf16_1_infer(const RrFwdArgs& a, hipStream_t st);
int rr_launch_bf16_2_infer(const RrFwdArgs& a, hipStream_t st);
int rr_launch_bf16_2_train(const RrFwdArgs& a, hipStream_t st);
int rr_launch_bf16_2_infer_emb(const RrFwdArgs& a, hipStream_t st);
int rr_launch_bf16_2_train_emb(const RrFwdArgs& a, hipStream_t st);

int rr_pack_bf16(const float* const* params, int xyz_ch, int dir_ch, int ns, const float* cb, void* section, hipStream_t st) {
    ParamPtrs P;
    P.xyz_ch = xyz_ch; P.dir_ch = dir_ch; P.cb = cb;
    for (int i = 0; i < PLNERF_N_PARAM_TENSORS; ++i) P.p[i] = params[i];
    const int groups = lay::FWDC_FLOATS / 8, threads = 256, blocks = (groups + threads - 1) / threads;
    if (ns == 1) hipLaunchKernelGGL(plnerf_rr_bf16::rr_pack_kernel<1>, dim3(blocks), dim3(threads), 0, st, P, (unsigned char*)section);
    else hipLaunchKernelGGL(plnerf_rr_bf16::rr_pack_kernel<2>, dim3(blocks), dim3(threads), 0, st, P, (unsigned char*)section);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

int rr_fwd_bf16(const void* packed, const void* section, int ns, const float* pts, const float* viewdirs,
                const float* embedded, int in_ch, int view_ch, int n_rows, int samples_per_ray, FwdOpt opt,
                float* raw_out, void* saved, unsigned* status, hipStream_t st) {
    RrFwdArgs a{packed, section, pts, viewdirs, n_rows, samples_per_ray < 1 ? 1 : samples_per_ray, raw_out, saved,
                status, embedded, in_ch, view_ch, opt.pe_scale, opt.act_beta};
    if (embedded) {      // (split mode only, like the IEEE-half build)
        if (ns != 2) return PLNERF_EINVAL;
        return saved ? rr_launch_bf16_2_train_emb(a, st) : rr_launch_bf16_2_infer_emb(a, st);
    }
    if (saved) return ns == 2 ? rr_launch_bf16_2_train(a, st) : PLNERF_EINVAL;      // (training: split mode only)
    return ns == 1 ? rr_launch_bf16_1_infer(a, st) : rr_launch_bf16_2_infer(a, st);
}

}  // namespace impl
}  // namespace plnerf
