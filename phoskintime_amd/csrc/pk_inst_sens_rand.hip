#include <cstdlib>
#include "pk_inst_sens.inc"
hipError_t launch_sens_dist(const SensArgs&, hipStream_t);
hipError_t launch_sens_succ(const SensArgs&, hipStream_t);
hipError_t launch_sens_rows_dist(const SensArgs&, hipStream_t);
hipError_t launch_sens_rows_succ(const SensArgs&, hipStream_t);
hipError_t launch_rand_sens(const SensArgs&, hipStream_t);          // randmod n = 6, 7 (pk_rand_sens.hpp, instantiated with the workgroup-per-replica kernels)

// sizes with a sensitivity kernel: distmod / succmod n <= 14 (S <= 16 rows in one lane, 1 + P = 5 + 2 n <= 64 columns in one wave: the
// default below n = 10 / 6) and up to n = 62 (rows across the lanes of a group, eight columns per lane, the columns of a replica cut into
// chunks: pk_sens_rows.hpp; PK_SENS_ROWS=1 forces it everywhere, =2 forbids it below n = 15),
// randmod n = 6, 7 (parity-eliminated inverse in registers serving eight columns per workgroup: pk_rand_sens.hpp),
// randmod n <= 3 (2^n <= 8 coupled rows inverted in registers; 1 + P <= 16 columns) and n = 4, 5 (the inverse shared by the group in LDS)
bool sens_available(int model, int n_sites) {
  if (model == M_RAND) return n_sites <= 7;
  return n_sites <= 62;
}

hipError_t launch_sens(const SensArgs& a, int model, hipStream_t st) {
  // PK_SENS_ROWS=1 (read once): the rows-per-lane kernel at every size -- the A/B switch the tests use to hold the two kernels to agreement
  static const int rows_env = [] { const char* v = getenv("PK_SENS_ROWS"); return v ? atoi(v) : 0; }();
  // PK_SENS_ROWS_MIN (dev, read once): smallest size the rows-per-lane kernel takes by default
  // [r3] measured crossover (tools/gpu_sens_one.py, B = 32 768, column kernel vs rows kernel): distmod n = 9: 4.6 vs 6.4 ms, n = 10: 7.1 vs
  // 6.5, n = 14: 7.2 vs 4.3 (B = 16 384);  succmod n = 4: 1.5 vs 3.1, n = 6: 5.0 vs 4.6, n = 10: 8.5 vs 6.2, n = 14: 8.1 vs 4.7 (B = 16 384)
  static const int rows_min_env = [] { const char* v = getenv("PK_SENS_ROWS_MIN"); return v ? atoi(v) : 0; }();
  const int rows_min = rows_min_env > 0 ? rows_min_env : (model == M_DIST ? 10 : 6);
  if (model != M_RAND && (a.s.n_sites > 14 || rows_env == 1 || (rows_env != 2 && a.s.n_sites >= rows_min))) return model == M_DIST ? launch_sens_rows_dist(a, st) : launch_sens_rows_succ(a, st);
  if (model == M_DIST) return launch_sens_dist(a, st);
  if (model == M_SUCC) return launch_sens_succ(a, st);
  const int n = a.s.n_sites;
  if (n >= 6) return launch_rand_sens(a, st);                        // 74 / 139 columns of 65 / 129 rows: chunks of eight columns per workgroup
  if (n == 1) return launch_sens_one<CubeSys<1>, 8>(a, st);          // 1 + P = 7
  if (n == 2) return launch_sens_one<CubeSys<2>, 16>(a, st);         // 10
  if (n == 3) return launch_sens_one<CubeSys<3>, 16>(a, st);         // 15
  if (n == 4) return launch_sens_one<CubeLdsSys<4, 32>, 32>(a, st);  // 1 + P = 24 columns, 17 rows
  return launch_sens_one<CubeLdsSys<5, 64>, 64>(a, st);              // 41 columns, 33 rows
}

}  // namespace pk
