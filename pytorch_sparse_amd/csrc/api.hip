// Library identification and status strings of the tsamd C-ABI.
#include "common.h"

#include <hip/hip_version.h>

namespace tsamd {
thread_local int g_last_hip_error = 0;
}

extern "C" int64_t tsamd_hip_version(void) { return (int64_t)HIP_VERSION; }

extern "C" int tsamd_build_flags(void) {
#if defined(TSAMD_EXPERIMENTS)
  return 1;
#else
  return 0;
#endif
}

extern "C" int tsamd_last_hip_error(void) { return tsamd::g_last_hip_error; }

extern "C" const char *tsamd_status_string(int status) {
  switch (status) {
    case TSAMD_OK: return "ok";
    case TSAMD_ERR_INVALID: return "invalid argument";
    case TSAMD_ERR_UNSUPPORTED: return "unsupported dtype, reduction or size";
    case TSAMD_ERR_HIP: return "HIP runtime error";
    case TSAMD_ERR_WORKSPACE: return "workspace missing or too small";
    default: return "unknown status";
  }
}
