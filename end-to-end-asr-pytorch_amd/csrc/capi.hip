// ABI bookkeeping entry points.
#include "las_common.h"

extern "C" int las_abi_version(void) { return LAS_ABI_VERSION; }

extern "C" const char* las_error_string(int code) {
    switch (code) {
        case LAS_OK: return "ok";
        case LAS_E_BADARG: return "bad argument";
        case LAS_E_UNSUPPORTED: return "unsupported shape";
        case LAS_E_WORKSPACE: return "workspace too small";
        case LAS_E_TIMEOUT: return "persistent kernel spin timeout";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown";
    }
}
