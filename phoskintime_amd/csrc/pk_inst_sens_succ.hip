#include "pk_inst_sens.inc"
hipError_t launch_sens_succ(const SensArgs& a, hipStream_t st) { return launch_sens_chain<M_SUCC>(a, st); }
}  // namespace pk
