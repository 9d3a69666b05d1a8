#include "common.h"
extern "C" const char* fmi_status_string(int status) {
  switch (status) {
    case FMI_OK: return "ok";
    case FMI_ERR_BAD_ARG: return "bad argument";
    case FMI_ERR_UNSUPPORTED: return "unsupported shape or mode";
    case FMI_ERR_LAUNCH: return "kernel launch failed";
    default: return "unknown status";
  }
}
extern "C" int fmi_version(void) { return 1; }

// debug launch counter of the GEMM-family launcher (gemm_core.h: FMI_DMA_OFF_RANGE); touched only when that variable is set,
// and then with an atomic increment -- the library has no other mutable global
extern "C" long fmi_debug_launch_counter = 0;
