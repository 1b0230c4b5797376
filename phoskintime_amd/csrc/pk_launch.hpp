// pk_launch.hpp -- host-callable launchers, one translation unit per (model, group width) so the kernels
// compile in parallel (pk_inst.inc is the body; pk_inst_m<M>_g<G>.hip the instantiations).
#pragma once
#include "pk_solve_kernel.hpp"

namespace pk {

using SolveLauncher = void (*)(const SolveArgs&, int method, bool structured, dim3 grid, hipStream_t);
using RhsLauncher = void (*)(const double* theta, const double* y, double* dydt, long long B, int n, int S, int P, dim3 grid, hipStream_t);
using JacLauncher = void (*)(const double* theta, double* J, long long B, int n, int S, int P, dim3 grid, hipStream_t);
using SteadyLauncher = void (*)(const double* theta, double* yss, int32_t* status, long long B, int n, int S, int P, dim3 grid, hipStream_t);

#define PK_DECL(M, G)                                                                                     \
  void launch_solve_m##M##_g##G(const SolveArgs&, int, bool, dim3, hipStream_t);                           \
  void launch_rhs_m##M##_g##G(const double*, const double*, double*, long long, int, int, int, dim3, hipStream_t); \
  void launch_jac_m##M##_g##G(const double*, double*, long long, int, int, int, dim3, hipStream_t);          \
  void launch_steady_m##M##_g##G(const double*, double*, int32_t*, long long, int, int, int, dim3, hipStream_t);
PK_DECL(0, 8) PK_DECL(0, 16) PK_DECL(0, 32) PK_DECL(0, 64)
PK_DECL(1, 8) PK_DECL(1, 16) PK_DECL(1, 32) PK_DECL(1, 64)
PK_DECL(2, 8) PK_DECL(2, 16) PK_DECL(2, 32) PK_DECL(2, 64)
#undef PK_DECL

// distributive-model throughput kernel (pk_dist_fast.hpp): RODAS4, arrow elimination, 4-16 lanes per replica
void launch_dist_fast(const SolveArgs&, int method, hipStream_t);
// random-model throughput kernel (pk_rand_fast.hpp): RODAS4 / LRP8, in-register Gauss-Jordan on the 2^n coupled rows
void launch_rand_fast(const SolveArgs&, int method, hipStream_t);
// thread-per-replica kernels for small distributive / successive systems (pk_tpr.hpp), LRP12
bool tpr_available(int model, int n_sites);
hipError_t launch_tpr(const SolveArgs&, int model, hipStream_t);   // sets the dynamic-LDS limit per device; its error is the caller's
// forward parameter sensitivities, one column per lane (pk_sens.hpp), LRP12
struct SensArgs;
bool sens_available(int model, int n_sites);
hipError_t launch_sens(const SensArgs&, int model, hipStream_t);
// workgroup-per-replica kernels for systems beyond 64 rows (pk_wide.hpp)
bool wide_chain_fits(int S, int n_sites);                                   // distmod / succmod: 15 LDS vectors of S doubles
hipError_t launch_wide_chain(const SolveArgs&, int model, hipStream_t);     // LRP12, exact solves
bool rand_dense_available(int n_sites);                                    // randmod n = 7: LRP12 with the dense inverse in registers (pk_rand_dense.hpp)
hipError_t launch_rand_dense(const SolveArgs&, hipStream_t);
// randmod n = 8 (and n = 6 on request): LRP12 with exact solves by twisted block elimination over the popcount levels, Schur complements in LDS (pk_rand_level.hpp)
hipError_t launch_rand_level(const SolveArgs&, hipStream_t);
// randmod n = 8, the default: LRP12 with exact solves -- odd-popcount states eliminated (diagonal block), even Schur complement inverted in registers (pk_rand_parity.hpp)
hipError_t launch_rand_parity(const SolveArgs&, hipStream_t, bool pinned_family = false);
bool wide_rand_in_lds(int n_sites);                                         // randmod n >= 7: 9 LDS vectors of 2^n + 1 doubles (n <= 10 / 11)
size_t wide_rand_scratch_bytes(int n_sites, long long B);                   // 0 when the vectors fit LDS
hipError_t launch_wide_rand(const SolveArgs&, double* scratch, hipStream_t);   // ROS34PW2-W on the n-cube
hipError_t launch_wide_steady(int model, const double* theta, double* yss, int32_t* status, long long B, int n, int S, int P, hipStream_t);
void launch_chain_rhs_wide(int model, const double* theta, const double* y, double* dydt, long long B, int n, int S, int P, hipStream_t);
void launch_chain_jac_wide(int model, const double* theta, double* J, long long B, int n, int S, int P, hipStream_t);
void launch_rand_rhs_wide(const double* theta, const double* y, double* dydt, long long B, int n, int S, int P, hipStream_t);
void launch_rand_jac_wide(const double* theta, double* J, long long B, int n, int S, int P, hipStream_t);

}  // namespace pk
