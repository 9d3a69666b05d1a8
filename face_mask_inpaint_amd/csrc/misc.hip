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

// ---- reproducible mode (FMI_DETERMINISTIC=1 in the environment, or fmi_set_deterministic) -------------------------------------------
// The fast default lets partial sums of split reductions meet through fp32 atomics, whose arrival order -- and therefore the rounding --
// changes from run to run.  In this mode every launcher picks a decomposition in which each accumulated address has exactly ONE
// contributing workgroup (no split reductions, one-block tails), the attention backward writes its query-side partial tiles to a workspace
// that is added in a fixed order, and scatter-style adjoints run as gathers: two runs of the same step are bit-identical (slower: it is
// a checking mode, tests/test_gpu_deterministic.py).
#include <cstdlib>
static int det_from_env() {
  const char* e = getenv("FMI_DETERMINISTIC");
  return (e && e[0] && e[0] != '0') ? 1 : 0;
}
extern "C" int fmi_deterministic_flag = det_from_env();
extern "C" int fmi_set_deterministic(int on) {
  const int prev = fmi_deterministic_flag;
  fmi_deterministic_flag = on ? 1 : 0;
  return prev;
}
extern "C" int fmi_get_deterministic(void) { return fmi_deterministic_flag; }
