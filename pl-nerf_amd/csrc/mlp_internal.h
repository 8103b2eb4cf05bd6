// Internal launchers behind the MLP entry points of the C ABI (mlp_api.hip dispatches on the
// precision mode).  The fp32 mode keeps its state in fp32 planes; the 16-bit MFMA modes share one
// half-plane layout (mlp_layout.h) and one weight-gradient stage.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace plnerf {
namespace impl {

// what a forward launch takes besides its buffers
struct FwdOpt {
    float pe_scale;     // the in-kernel encoding's input scale (1: run_nerf_helpers.py:24-54; pi: the depth variant's Embedder)
    float act_beta;     // > 0: the density channel leaves as softplus(beta)(sigma) (common.h: density_activation); 0: as it is
};

size_t f32_packed_bytes();
int f32_pack(const float* const* params, int xyz_ch, int dir_ch, void* packed, hipStream_t st);
// xyz_ch / dir_ch = input_ch / input_ch_views of the network (columns of `embedded`)
// pe_scale: the in-kernel encoding's input scale (1: run_nerf_helpers.py:24-54; pi: the depth variant's Embedder)
int f32_fwd(const void* packed, const float* pts, const float* viewdirs, const float* embedded, int xyz_ch,
            int dir_ch, int n_rows, int samples_per_ray, FwdOpt opt, float* raw_out, void* saved, hipStream_t st);
int f32_dgrad(const void* packed, const float* g_raw, int n_rows, const float* saved, float* dz, hipStream_t st);
// Weight gradients.  h16 = false: fp32 planes on the fp32 MFMA (fp32 mode); h16 = true: half planes on
// v_mfma_f32_32x32x16_f16, partial sums divided by the dz scale derived from *gmax (16-bit modes).
// status / status_out: the network's range status word (nullptr in fp32 mode) and where the reduction leaves
// (float)(word != 0) for the caller's gradient exchange (nullptr = nowhere)
int wgrad(const float* g_raw, int n_rows, const void* saved, const void* dz, const unsigned* gmax, float* partials,
          float* const* grads, int xyz_ch, int dir_ch, bool h16, int saved_layout, const unsigned* status,
          float* status_out, hipStream_t st);
// The weight-gradient stage of the 16-bit modes for up to MAX_BWD_JOBS networks at once (plnerf_mlp_bwd_multi): the
// main / thin / head / reduce launches each cover every job -- the one round of 256 (254) workgroups is dealt out over
// the jobs in proportion to their rows -- instead of running once per network.  One job = what wgrad() takes.
constexpr int MAX_BWD_JOBS = 2;
struct WgradJob {
    const float* g_raw;
    int n_rows;
    const void* saved;
    const void* dz;
    const unsigned* gmax;
    float* partials;
    float* const* grads;
    int xyz_ch, dir_ch, saved_layout;
    const unsigned* status;
    float* status_out;
    const float* cb;          // the compose block of the job's packed weights (mlp_layout.h: CB_*): the factors W_vf, W_f, b_f
    float* gred;              // lay::GRED_FLOATS floats of workspace: the reduced G and s on their way to the factors' gradients
};
int wgrad_h16_multi(int n, const WgradJob* jobs, hipStream_t st);
// one network's share of the dgrad grid (mlp_bwd_h16_kernel, two networks per grid)
struct DgradJob {
    const void* packed;
    int fwd_ns;
    const float* g_raw;
    int n_rows;
    const void* saved;
    void* dz;
    unsigned* gmax;           // the workspace's max |g_raw| word: holds it already, or (n_cand > 0) is written by the launch
    const unsigned* cand;     // n_cand candidates whose maximum is that word, or nullptr / 0
    int n_cand;
};
// The composed view layer of the 16-bit modes (mlp_layout.h; mlp_compose.hip).
//   compose_pack: cb <- W_c = W_vf W_f, b_c = W_vf b_f + b_v (fp64 accumulation, rounded once) and copies of W_vf, W_f, b_f
//   compose_grads (per job, after the split-K reduction left G and s in `gred`): grads[P_WF] = W_vf^T G, grads[P_BF] = W_vf^T s,
//                 the feature columns of grads[P_WV] = G W_f^T + s b_f^T
int compose_pack(const float* const* params, int dir_ch, float* cb, hipStream_t st);
struct ComposeGradJob {
    const float* cb;
    const float* gred;
    float* const* grads;
    int dir_ch;
};
int compose_grads(int n, const ComposeGradJob* jobs, hipStream_t st);
// gradient with respect to the embedded input rows, from the dz planes a finished plnerf_mlp_bwd left in its workspace
int input_grad(const float* const* params, int n_rows, const void* dz, const unsigned* gmax, bool h16, int xyz_ch,
               int dir_ch, float* g_emb, hipStream_t st);
// g_eff [n_rows, 4] = g_raw with the sigma column times the density activation's derivative (from raw_out, the forward's
// activated output); out (nullable): max |g_eff| as fp32 bits, like absmax
int absmax_act(const float* g_raw, const float* raw_out, float beta, int n_rows, float* g_eff, unsigned* out, hipStream_t st);
// max |x| over n floats as fp32 bits (non-negative floats order like unsigned integers) -> *out
int absmax(const float* x, size_t n, unsigned* out, hipStream_t st);
// bytes of the half dz planes of n_rows rows, rounded up to 16
#ifndef PLNERF_BWD_TM
#define PLNERF_BWD_TM 192
#endif
inline size_t h16_dz_bytes(int n_rows) {      // (lay::dz_rows)
    return (size_t)4352 * (((size_t)n_rows + PLNERF_BWD_TM - 1) / PLNERF_BWD_TM * PLNERF_BWD_TM);
}

// 16-bit-operand MFMA modes.  ns = 1: plain operands; ns = 2: 3-term split (hi/lo planes).
// f16 = 0: bf16 elements; f16 = 1: IEEE half elements.
size_t bf16_packed_bytes(int ns);
size_t bf16_compose_offset(int ns);      // byte offset of the compose block (lay::CB_FLOATS floats) inside the packed buffer
int bf16_pack(const float* const* params, int xyz_ch, int dir_ch, int ns, int f16, void* packed, unsigned* status,
              hipStream_t st);
int bf16_fwd(const void* packed, int ns, int f16, const float* pts, const float* viewdirs, const float* embedded,
             int xyz_ch, int dir_ch, int n_rows, int samples_per_ray, FwdOpt opt, float* raw_out, void* saved,
             unsigned* status, hipStream_t st);
int bf16_dgrad(int n, const DgradJob* jobs, hipStream_t st);

// Register-resident forward kernel (mlp_rr.hip; IEEE-half elements): its own weight section (k order permuted to the
// accumulator layout) appended to the packed buffer of the half modes.  (cb: the compose block, written before the call)
size_t rr_packed_bytes(int ns);
int rr_pack(const float* const* params, int xyz_ch, int dir_ch, int ns, const float* cb, void* section, hipStream_t st);
struct RrFwdArgs {
    const void* packed;     // head block at the start
    const void* wrr;        // the register-resident section of the packed weights
    const float* pts;
    const float* viewdirs;
    int n_rows, spr;
    float* raw_out;
    void* saved;
    unsigned* status;       // range status word of the packed buffer
    const float* embedded;  // caller-supplied encoding [n_rows][in_ch + view_ch] (EMB kernels), else pts / viewdirs
    int in_ch, view_ch;
    float pe_scale;         // in-kernel encoding: sin / cos(x * pe_scale * 2^k)
    float act_beta;         // > 0: sigma leaves as softplus(beta)(sigma)
};
bool rr_embedded_ok(int ns);
int rr_fwd(const void* packed, const void* section, int ns, const float* pts, const float* viewdirs, const float* embedded,
           int in_ch, int view_ch, int n_rows, int samples_per_ray, FwdOpt opt, float* raw_out, void* saved,
           unsigned* status, hipStream_t st);
// the same kernel on bf16 elements (mlp_rr_body.inc compiled with RR_BF16): inference, and -- split mode only -- the
// training forward and caller-embedded inputs
int rr_pack_bf16(const float* const* params, int xyz_ch, int dir_ch, int ns, const float* cb, void* section, hipStream_t st);
int rr_fwd_bf16(const void* packed, const void* section, int ns, const float* pts, const float* viewdirs,
                const float* embedded, int in_ch, int view_ch, int n_rows, int samples_per_ray, FwdOpt opt,
                float* raw_out, void* saved, unsigned* status, hipStream_t st);

}  // namespace impl
}  // namespace plnerf
