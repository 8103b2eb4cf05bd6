// The PL-NeRF MLP on the 16-bit-operand MFMAs (v_mfma_f32_32x32x16_bf16 / _f16).
// The kernel source lives in mlp_h16_body.inc and is compiled for two element types; the f16 split
// (11+11 mantissa bits) reaches ~3e-7 where the bf16 split (8+8) reaches ~8e-6 at the same cost.
// PLNERF_PREC_BF16 (NS = 1, plain bf16
// operands) and PLNERF_PREC_BF16X3 (NS = 2: every operand is split x = hi + lo with
// hi = bf16(x), lo = bf16(x - hi), and each product is evaluated as hi*hi + lo*hi + hi*lo with
// fp32 accumulation -- ~2^-16 relative operand error, i.e. fp32-class results, at 3 MFMAs per
// product instead of 16 fp32-MFMA passes).
//
// Reference: run_network (run_plnerf.py:78-92) = Embedder (run_nerf_helpers.py:24-54) +
// NeRF.forward (:105-128) and its autograd backward.
//
// Formulation: OUT^T[feature][sample] = W[feature][k] . X^T[k][sample].  The MFMA "A" operand is
// the weight tile (pre-packed in fragment order, streamed from L2 by the one wave that owns that
// 32-feature slab -- no LDS staging, no duplicate fetch), the "B" operand is the activation tile
// in LDS, [sample][feature] row-major bf16, read with one 16-byte ds_read per fragment (row stride
// 528 B == 16 mod 256 -> conflict-free).  With features on the accumulator rows, a lane's four
// consecutive registers are four consecutive features of ONE sample, i.e. four consecutive k of
// the next layer: the bias+ReLU epilogue converts them to bf16 and writes one 8-byte LDS store
// per plane, in place.
//
// A 512-thread workgroup (8 waves, 2 per SIMD) owns a tile of TM samples (128 for bf16, 64 for
// bf16x3; 94 KB of LDS either way) and walks all 12 layers without leaving the CU.  For
// training, each layer's activation tile is additionally copied LDS -> HBM as coalesced fp32
// rows (the same plane layout as the fp32 mode), so the fp32-MFMA weight-gradient stage is shared.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "mlp_internal.h"
#include "mlp_layout.h"
#include "mlp_pack_src.h"

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

int bf16_pack(const float* const* params, int xyz_ch, int dir_ch, int ns, int f16, void* packed, hipStream_t st) {
    // forward section in the mode's element type (ns planes), dgrad section always one half plane
    const int rc = f16 ? plnerf_h16_f16::h16_pack(params, xyz_ch, dir_ch, ns, 1, ns, packed, st)
                       : plnerf_h16_bf16::h16_pack(params, xyz_ch, dir_ch, ns, 1, ns, packed, st);
    return rc ? rc : plnerf_h16_f16::h16_pack(params, xyz_ch, dir_ch, 1, 2, ns, packed, st);
}

int bf16_fwd(const void* packed, int ns, int f16, const float* pts, const float* viewdirs, const float* embedded,
             int xyz_ch, int dir_ch, int n_rows, int samples_per_ray, float* raw_out, void* saved, hipStream_t st) {
    return f16 ? plnerf_h16_f16::h16_fwd(packed, ns, pts, viewdirs, embedded, xyz_ch, dir_ch, n_rows, samples_per_ray,
                                         raw_out, saved, st)
               : plnerf_h16_bf16::h16_fwd(packed, ns, pts, viewdirs, embedded, xyz_ch, dir_ch, n_rows, samples_per_ray,
                                          raw_out, saved, st);
}

int bf16_dgrad(const void* packed, int ns, const float* g_raw, int n_rows, const void* saved, void* dz,
               const unsigned* gmax, hipStream_t st) {
    return plnerf_h16_f16::h16_dgrad(packed, ns, g_raw, n_rows, saved, dz, gmax, st);
}

}  // namespace impl
}  // namespace plnerf

#ifdef PLNERF_TRACE
// profiling builds only (tools/trace_fwd.py): phase stamps of workgroup PLNERF_TRACE, bf16 element type
extern "C" int plnerf_debug_trace(unsigned long long* out64) {
    return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(plnerf_h16_bf16::g_trace), 64 * sizeof(unsigned long long));
}
#endif
