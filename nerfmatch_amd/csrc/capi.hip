// ABI version / error strings of the C boundary (include/nerfmatch_amd.h).
#include "common.h"

extern "C" int nm_abi_version(void) { return 1; }

extern "C" const char* nm_error_string(int code) {
  switch (code) {
    case NM_OK: return "ok";
    case NM_ERR_ARG: return "invalid argument (null pointer or non-positive size)";
    case NM_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
    case NM_ERR_LAUNCH: return "HIP launch error";
    case NM_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown error";
  }
}
