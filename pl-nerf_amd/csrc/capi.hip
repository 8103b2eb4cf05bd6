// Version / error-string entry points of libplnerf_hip.so.
#include "common.h"

extern "C" int plnerf_version(void) { return PLNERF_VERSION; }

extern "C" const char* plnerf_error_string(int code) {
    switch (code) {
        case PLNERF_OK: return "ok";
        case PLNERF_EINVAL: return "invalid argument (null pointer, bad size or unsupported combination)";
        case PLNERF_ELAUNCH: return "HIP kernel launch failed";
        case PLNERF_ERANGE: return "size outside the compiled limits";
        case PLNERF_ENOSYS: return "precision mode not built";
        default: return "unknown plnerf error";
    }
}

// Build hygiene: the MLP translation units carry trace hooks (PLNERF_TRACE, RR_TRACE) for tools/ builds.  A library built
// with them says so here (bit 2), and the Python binding refuses to load it as the product (pl-nerf_amd/_lib.py).  (The
// results-wrong ablation switches of rounds 1-4 -- bits 0 and 1 -- left the product sources in round 5.)
extern "C" int plnerf_build_flags_h16(void);
extern "C" int plnerf_build_flags_rr(void);
extern "C" int plnerf_build_flags(void) { return plnerf_build_flags_h16() | plnerf_build_flags_rr(); }
