// The PL-NeRF MLP on the 16-bit-operand MFMAs (v_mfma_f32_32x32x16_bf16 / _f16): precision modes
// PLNERF_PREC_BF16X3 / _F16X3 (NS = 2: every forward operand is split x = hi + lo with hi = T(x),
// lo = T(x - hi), each product evaluated as hi*hi + lo*hi + hi*lo with fp32 accumulation -- fp32-class
// results at 3 MFMAs per product instead of 16 fp32-MFMA passes; the half split's 11+11 mantissa bits reach
// ~1e-6 where the bf16 split's 8+8 reach ~8e-6) and PLNERF_PREC_BF16 / _F16 (NS = 1, plain operands).
//
// Reference: run_network (run_plnerf.py:78-92) = Embedder (run_nerf_helpers.py:24-54) +
// NeRF.forward (:105-128) and its autograd backward.
//
// The kernel source lives in mlp_h16_body.inc (+ mlp_h16_fwd_pp.inc) and is compiled once per element type:
//   forward, inference            mlp_fwd_pp_kernel<NS, false>      both element types
//   forward, training             mlp_fwd_pp_kernel<NS, true>       both element types (half: the tile's hi plane IS the
//                                                                   saved plane; bf16: hi (+ lo) converted in the copy)
//   backward, dgrad chain         mlp_bwd_h16_kernel                half elements only, shared by every mode
// Formulation: OUT^T[feature][sample] = W[feature][k] . X^T[k][sample].  The MFMA "A" operand is the weight tile
// (pre-packed in fragment order, owned by the one wave that computes that 32-feature slab -- no LDS staging, no
// duplicate fetch), the "B" operand is the activation tile in LDS, [sample][feature] row-major 16-bit planes, read
// with one 16-byte ds_read per fragment (row stride 528 B == 16 mod 256 -> conflict-free).  With features on the
// accumulator rows, a lane's four consecutive registers are four consecutive features of ONE sample, i.e. four
// consecutive k of the next layer: the bias+ReLU epilogue converts them and writes one 8-byte LDS store per plane,
// in place.  A 512-thread workgroup owns a tile of TM samples (128 plain, 64 split) and walks all 12 layers without
// leaving the CU.  State saved for the backward: IEEE-half planes + ReLU bit masks (mlp_layout.h); the weight-
// gradient stage (mlp_f32.hip: wgrad_main_kernel, wgrad_thin_kernel) is shared by all four modes.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "mlp_internal.h"
#include "mlp_layout.h"
#include "mlp_pack_src.h"
#include "pe_sincos.h"

namespace plnerf_h16_bf16 {
using namespace plnerf;
using namespace plnerf::lay;
#define H16T __bf16
#define H16_MFMA __builtin_amdgcn_mfma_f32_32x32x16_bf16
#include "mlp_h16_body.inc"
#undef H16T
#undef H16_MFMA
}  // namespace plnerf_h16_bf16

namespace plnerf_h16_f16 {
using namespace plnerf;
using namespace plnerf::lay;
#define H16T _Float16
#define H16_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#define H16_HAS_BWD   // the dgrad chain of every 16-bit mode runs on half operands
#include "mlp_h16_body.inc"
#undef H16T
#undef H16_MFMA
#undef H16_HAS_BWD
}  // namespace plnerf_h16_f16

namespace plnerf {
namespace impl {

size_t bf16_packed_bytes(int ns) { return plnerf_h16_bf16::h16_packed_bytes(ns); }
size_t bf16_compose_offset(int ns) { return plnerf_h16_bf16::h16_compose_offset(ns); }

int bf16_pack(const float* const* params, int xyz_ch, int dir_ch, int ns, int f16, void* packed, unsigned* status,
              hipStream_t st) {
    // forward section in the mode's element type (ns planes), dgrad section always one half plane: one launch
    // when the forward section is half too, two otherwise
    if (f16) return plnerf_h16_f16::h16_pack(params, xyz_ch, dir_ch, ns, 3, ns, packed, status, st);
    const int rc = plnerf_h16_bf16::h16_pack(params, xyz_ch, dir_ch, ns, 1, ns, packed, nullptr, st);
    return rc ? rc : plnerf_h16_f16::h16_pack(params, xyz_ch, dir_ch, 1, 2, ns, packed, status, st);
}

int bf16_fwd(const void* packed, int ns, int f16, const float* pts, const float* viewdirs, const float* embedded,
             int xyz_ch, int dir_ch, int n_rows, int samples_per_ray, FwdOpt opt, float* raw_out, void* saved,
             unsigned* status, hipStream_t st) {
    return f16 ? plnerf_h16_f16::h16_fwd(packed, ns, pts, viewdirs, embedded, xyz_ch, dir_ch, n_rows, samples_per_ray,
                                         opt, raw_out, saved, status, st)
               : plnerf_h16_bf16::h16_fwd(packed, ns, pts, viewdirs, embedded, xyz_ch, dir_ch, n_rows, samples_per_ray,
                                          opt, raw_out, saved, status, st);      // (bf16 elements: PLNERF_RANGE_SAVED only)
}

int bf16_dgrad(int n, const DgradJob* jobs, hipStream_t st) {
    if (n < 1 || n > MAX_BWD_JOBS) return PLNERF_EINVAL;
    return plnerf_h16_f16::h16_dgrad(n, jobs, st);
}

}  // namespace impl
}  // namespace plnerf

// bit 2: trace hooks compiled in (tools/trace_fwd.py, tools/trace_bwd.py); the results-wrong ablation switches of rounds
// 1-4 (bits 0 and 1) are gone from the product sources -- git history at a21e3cd has them
extern "C" int plnerf_build_flags_h16(void) {
    int f = 0;
#ifdef PLNERF_TRACE
    f |= 4;
#endif
    return f;
}

#ifdef PLNERF_TRACE
// profiling builds only (tools/trace_fwd.py): phase stamps of workgroup PLNERF_TRACE, bf16 element type
extern "C" int plnerf_debug_trace(unsigned long long* out64) {
    return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(plnerf_h16_bf16::g_trace), 64 * sizeof(unsigned long long));
}
// the same for the half element type (forward kernels of f16x3 / f16; the dgrad kernel of every 16-bit mode)
extern "C" int plnerf_debug_trace_f16(unsigned long long* out64) {
    return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(plnerf_h16_f16::g_trace), 64 * sizeof(unsigned long long));
}
#endif
