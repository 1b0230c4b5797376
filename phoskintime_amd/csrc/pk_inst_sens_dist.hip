#include "pk_inst_sens.inc"
hipError_t launch_sens_dist(const SensArgs& a, hipStream_t st) { return launch_sens_chain<M_DIST>(a, st); }
}  // namespace pk
