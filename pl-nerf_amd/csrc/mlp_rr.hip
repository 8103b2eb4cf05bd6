// Register-resident forward kernel of the 16-bit-operand MFMA MLP: packing, dispatch and the C++ entry points.  The
// kernel itself is mlp_rr_body.inc; its six instantiations live in mlp_rr_k*.hip so that they compile in parallel
// (-DRR_SINGLE_TU puts them all here: the trace builds of tools/build_rr.sh).
#include "mlp_rr_body.inc"

#ifdef RR_TRACE
extern "C" int plnerf_debug_rr_trace(unsigned long long* out8) {
    return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(plnerf_rr::g_rr_trace), 128 * sizeof(unsigned long long));
}
#endif

extern "C" int plnerf_build_flags_rr(void) {
    int f = 0;
#ifdef RR_TRACE
    f |= 4;
#endif
    return f;
}

namespace plnerf {
namespace impl {

// kernel<NS, SAVE, EMB> behind a plain function each
#ifdef RR_SINGLE_TU
#define RR_LAUNCH(name, NS, SAVE, EMB) \
    int name(const RrFwdArgs& a, hipStream_t st) { return plnerf_rr::launch<NS, SAVE, EMB>(a, st); }
#else
#define RR_LAUNCH(name, NS, SAVE, EMB) int name(const RrFwdArgs& a, hipStream_t st);
#endif
RR_LAUNCH(rr_launch_1_infer, 1, false, false)
RR_LAUNCH(rr_launch_1_train, 1, true, false)
RR_LAUNCH(rr_launch_2_infer, 2, false, false)
RR_LAUNCH(rr_launch_2_train, 2, true, false)
RR_LAUNCH(rr_launch_2_infer_emb, 2, false, true)
RR_LAUNCH(rr_launch_2_train_emb, 2, true, true)
#undef RR_LAUNCH

size_t rr_packed_bytes(int ns) { return (size_t)lay::FWDC_FLOATS * ns * 2; }

int rr_pack(const float* const* params, int xyz_ch, int dir_ch, int ns, const float* cb, void* section, hipStream_t st) {
    ParamPtrs P;
    P.xyz_ch = xyz_ch; P.dir_ch = dir_ch; P.cb = cb;
    for (int i = 0; i < PLNERF_N_PARAM_TENSORS; ++i) P.p[i] = params[i];
    const int groups = lay::FWDC_FLOATS / 8, threads = 256, blocks = (groups + threads - 1) / threads;
    if (ns == 1) hipLaunchKernelGGL(plnerf_rr::rr_pack_kernel<1>, dim3(blocks), dim3(threads), 0, st, P, (unsigned char*)section);
    else hipLaunchKernelGGL(plnerf_rr::rr_pack_kernel<2>, dim3(blocks), dim3(threads), 0, st, P, (unsigned char*)section);
    PLNERF_CHECK_LAUNCH();
    return PLNERF_OK;
}

// `embedded` (a caller-supplied encoding, [n_rows][in_ch + view_ch]) is served in split mode only (ns == 2: rr_embedded_ok)
bool rr_embedded_ok(int ns) { return ns == 2; }
int rr_fwd(const void* packed, const void* section, int ns, const float* pts, const float* viewdirs, const float* embedded,
           int in_ch, int view_ch, int n_rows, int samples_per_ray, FwdOpt opt, float* raw_out, void* saved,
           unsigned* status, hipStream_t st) {
    RrFwdArgs a{packed, section, pts, viewdirs, n_rows, samples_per_ray < 1 ? 1 : samples_per_ray, raw_out, saved,
                         status, embedded, in_ch, view_ch, opt.pe_scale, opt.act_beta};
    if (embedded) {
        if (ns != 2) return PLNERF_EINVAL;
        return saved ? rr_launch_2_train_emb(a, st) : rr_launch_2_infer_emb(a, st);
    }
    if (ns == 1) return saved ? rr_launch_1_train(a, st) : rr_launch_1_infer(a, st);
    return saved ? rr_launch_2_train(a, st) : rr_launch_2_infer(a, st);
}

}  // namespace impl
}  // namespace plnerf
