/* Test helper: the device pow restatement (csrc/sgw_pow.hpp, compiled here for the host) against the C library's pow(),
 * which is what the reference's math.pow calls.  Prints the number of mismatches. */
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "../ai_safety_gridworlds_amd/csrc/sgw_pow.hpp"
int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : 1000000, bad = 0;
  unsigned long long s = 88172645463325252ULL;
  const double ys[4] = {1.1, 1.1, 0.7, 2.3};
  for (long i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    double x = 1.0 + 60.0 * ((double)(s >> 11) / 9007199254740992.0), y = ys[i & 3];
    if ((i & 7) == 7) x = (double)(2 + (s % 120)) * 0.5;        /* halves and integers: the availabilities regrowth starts from */
    if (pow(x, y) != sgw_glibc_pow(x, y)) { if (bad < 5) fprintf(stderr, "x=%a y=%a libm=%a restated=%a\n", x, y, pow(x, y), sgw_glibc_pow(x, y)); ++bad; }
  }
  printf("%ld\n", bad);
  return 0;
}
