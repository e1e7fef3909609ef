// Library identity + error strings.
#include "ud_common.h"

#define UD_ABI_VERSION 1

extern "C" const char* ud_version(void) { return "unidistill_hip 1 gfx950"; }
extern "C" int ud_abi_version(void) { return UD_ABI_VERSION; }
extern "C" const char* ud_error_string(int code) {
  switch (code) {
    case UD_OK: return "ok";
    case UD_ERR_INVALID_ARG: return "invalid argument";
    case UD_ERR_WORKSPACE: return "workspace missing or too small";
    case UD_ERR_HIP: return "HIP runtime / launch failure";
    case UD_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown error";
  }
}
