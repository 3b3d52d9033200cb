// api.hip — version / error reporting of libfpc_hip.so.
#include "common.hpp"

#include <string.h>

namespace fpc {
static thread_local char g_last_hip_error[256] = "";
void set_hip_error(hipError_t e) {
    const char* s = hipGetErrorString(e);
    strncpy(g_last_hip_error, s ? s : "unknown", sizeof(g_last_hip_error) - 1);
    g_last_hip_error[sizeof(g_last_hip_error) - 1] = 0;
}
void clear_hip_error() { g_last_hip_error[0] = 0; }
}  // namespace fpc

extern "C" int fpc_abi_version(void) { return FPC_ABI_VERSION; }

extern "C" const char* fpc_error_string(int code) {
    switch (code) {
        case FPC_OK: return "ok";
        case FPC_EINVAL: return "invalid argument";
        case FPC_EWORKSPACE: return "workspace too small or misaligned";
        case FPC_ELAUNCH: return "HIP launch error";
        case FPC_EDEVICE: return "no usable gfx950 device";
        case FPC_EFORMAT: return "not a PNG this decoder reads";
        default: return "unknown error code";
    }
}

extern "C" const char* fpc_last_hip_error(void) { return fpc::g_last_hip_error; }
