// One instantiation of the register-resident forward (mlp_rr_body.inc) per translation unit: see mlp_rr.hip.
#include "mlp_rr_body.inc"

namespace plnerf {
namespace impl {
int rr_launch_2_train_emb(const RrFwdArgs& a, hipStream_t st) { return plnerf_rr::launch<2, true, true>(a, st); }
}  // namespace impl
}  // namespace plnerf
