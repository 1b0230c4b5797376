// Instantiations of the rows-per-lane forward-sensitivity kernel (pk_sens_rows.hpp) for the successive model.
#include "pk_sens_rows.hpp"
#include "pk_launch.hpp"
namespace pk {
hipError_t launch_sens_rows_succ(const SensArgs& a, hipStream_t st) { return launch_sens_rows_model<M_SUCC>(a, st); }
}  // namespace pk
