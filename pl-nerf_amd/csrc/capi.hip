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
