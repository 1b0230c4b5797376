#include "pk_inst_sens.inc"
hipError_t launch_sens_dist(const SensArgs&, hipStream_t);
hipError_t launch_sens_succ(const SensArgs&, hipStream_t);

// sizes with a sensitivity kernel: distmod / succmod n <= 14 (S <= 16 rows in one lane, 1 + P = 5 + 2 n <= 64 columns in one wave),
// randmod n <= 3 (2^n <= 8 coupled rows inverted in registers; 1 + P <= 16 columns) and n = 4, 5 (the inverse shared by the group in LDS)
bool sens_available(int model, int n_sites) {
  if (model == M_RAND) return n_sites <= 5;
  return n_sites <= 14;
}

hipError_t launch_sens(const SensArgs& a, int model, hipStream_t st) {
  if (model == M_DIST) return launch_sens_dist(a, st);
  if (model == M_SUCC) return launch_sens_succ(a, st);
  const int n = a.s.n_sites;
  if (n == 1) return launch_sens_one<CubeSys<1>, 8>(a, st);          // 1 + P = 7
  if (n == 2) return launch_sens_one<CubeSys<2>, 16>(a, st);         // 10
  if (n == 3) return launch_sens_one<CubeSys<3>, 16>(a, st);         // 15
  if (n == 4) return launch_sens_one<CubeLdsSys<4, 32>, 32>(a, st);  // 1 + P = 24 columns, 17 rows
  return launch_sens_one<CubeLdsSys<5, 64>, 64>(a, st);              // 41 columns, 33 rows
}

}  // namespace pk
