/* x / C == fma(fma(-C, q, x), RN(1/C), q) with q = x * RN(1/C) -- the division by a constant the packed photometric forward uses
 * (csrc/loss.hip: div_by<C>) -- compared bit for bit with the IEEE division for every `stride`-th non-negative float
 * (stride 1 = all 2 139 095 040 of them, denormals included: ~30 s on one core).
 * gcc -O2 -mfma -ffp-contract=off div_by_check.c -o div_by_check && ./div_by_check [stride]   -> exit code 0 iff no mismatch */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline float div_by(float x, float c, float rc) {
  const float q = x * rc;
  return fmaf(fmaf(-c, q, x), rc, q);
}

int main(int argc, char** argv) {
  const long stride = argc > 1 ? atol(argv[1]) : 1;
  const float cs[2] = {9.f, 3.f};
  long bad = 0, n = 0;
  for (int ci = 0; ci < 2; ++ci) {
    const float c = cs[ci], rc = 1.0f / c;
    for (long i = 0; i < 0x7f800000L; i += stride) {
      const uint32_t u = (uint32_t)i;
      float x;
      memcpy(&x, &u, 4);
      const float a = x / c, b = div_by(x, c, rc);
      uint32_t ua, ub;
      memcpy(&ua, &a, 4);
      memcpy(&ub, &b, 4);
      bad += ua != ub;
      ++n;
    }
  }
  printf("%ld values checked (stride %ld, C = 9 and 3): %ld mismatches\n", n, stride, bad);
  return bad != 0;
}
