// Entry points declared in include/draco_amd.h whose kernels are not built yet.
// They fail loudly (never a silent CPU fallback).
#include "dmm_internal.h"

extern "C" {
int64_t dmm_wiener_workspace_bytes(const dmm_plan*) { return 0; }
int dmm_wiener_run(dmm_plan*, const void*, const void*, const double*, double, double, void*, void*) {
  return dmm_set_error(DMM_E_UNSUPPORTED, "dmm_wiener_run: not built yet");
}
int64_t dmm_ml_workspace_bytes(const dmm_plan*) { return 0; }
int dmm_ml_run(dmm_plan*, const void*, const void*, const double*, double, double, void*, void*) {
  return dmm_set_error(DMM_E_UNSUPPORTED, "dmm_ml_run: not built yet");
}
}
