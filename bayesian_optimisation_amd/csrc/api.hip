// Small host-only entry points of libgpbo (version, error strings, padding rule).
#include "gpbo_internal.h"

extern "C" int gpbo_version(void) { return GPBO_VERSION; }

extern "C" const char *gpbo_strerror(int status) {
    switch (status) {
        case GPBO_OK: return "ok";
        case GPBO_ERR_ARG: return "invalid argument (null pointer, size, alignment or unsupported feature count)";
        case GPBO_ERR_LAUNCH: return "HIP launch/runtime error";
        case GPBO_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown status";
    }
}

extern "C" int64_t gpbo_padded_n(int64_t N) {
    if (N < 1) N = 1;
    return (N + GPBO_NPAD - 1) / GPBO_NPAD * GPBO_NPAD;
}
