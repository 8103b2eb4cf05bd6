// The register-resident forward on bf16 elements (mlp_rr_body.inc with RR_BF16), split mode, caller-embedded input:
// one instantiation per translation unit, see mlp_rr.hip.
#define RR_BF16 1
#include "mlp_rr_body.inc"

namespace plnerf {
namespace impl {
int rr_launch_bf16_2_train_emb(const RrFwdArgs& a, hipStream_t st) { return plnerf_rr_bf16::launch<2, true, true>(a, st); }
}  // namespace impl
}  // namespace plnerf
