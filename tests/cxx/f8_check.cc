// Every input of the 8-bit float arithmetic the kernels use (rccl_amd/csrc/f8_math.h) against the CPU oracle's restatement
// of the reference's conversions (oracle/oracle.c, pinned by vectors made from the reference's own header):
//   all 256 codes: decode;  all 2^32 floats (or every k-th, argv[1]): encode, and quantise == decode(encode).
// Exit code 0 and a line "ok ..." when nothing differs.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../rccl_amd/csrc/f8_math.h"

extern "C" {
float orc_f8e4m3_to_f32(uint8_t);
float orc_f8e5m2_to_f32(uint8_t);
uint8_t orc_f32_to_f8e4m3(float);
uint8_t orc_f32_to_f8e5m2(float);
}

static bool same(float a, float b) { return (isnan(a) && isnan(b)) || raF2U(a) == raF2U(b) || (a == 0.0f && b == 0.0f); }

template <int WM> static long check(uint64_t stride) {
  long bad = 0;
  for (uint32_t x = 0; x < 256; x++) {
    const float want = WM == 3 ? orc_f8e4m3_to_f32((uint8_t)x) : orc_f8e5m2_to_f32((uint8_t)x);
    const float got = f8Decode<WM>(x);
    if (!same(got, want) && bad++ < 5) printf("decode<%d>(0x%02x) = %g, oracle %g\n", WM, x, got, want);
    if (x != 0x80 && f8EncodeOnGrid<WM>(got) != x && bad++ < 5) printf("encode-on-grid<%d>(decode(0x%02x)) = 0x%02x\n", WM, x, f8EncodeOnGrid<WM>(got));
  }
#pragma omp parallel for reduction(+ : bad) schedule(static)
  for (int64_t k = 0; k < (int64_t)((1ull << 32) / stride); k++) {
    const uint32_t u = (uint32_t)((uint64_t)k * stride);
    const float v = raU2F(u);
    const uint8_t want = WM == 3 ? orc_f32_to_f8e4m3(v) : orc_f32_to_f8e5m2(v);
    const uint32_t got = f8Encode<WM>(v);
    const float q = f8Quantize<WM>(v), wq = WM == 3 ? orc_f8e4m3_to_f32(want) : orc_f8e5m2_to_f32(want);
    if (got != want || !same(q, wq)) {
      if (bad < 5) printf("float 0x%08x (%g): encode<%d> 0x%02x oracle 0x%02x, quantise %g oracle %g\n", u, v, WM, got, want, q, wq);
      bad++;
    }
  }
  return bad;
}

int main(int argc, char** argv) {
  const uint64_t stride = argc > 1 ? strtoull(argv[1], nullptr, 0) : 1;
  const long bad = check<3>(stride) + check<2>(stride);
  if (bad) { printf("%ld mismatches\n", bad); return 1; }
  printf("ok: 256 codes and every %llu-th float, both formats\n", (unsigned long long)stride);
  return 0;
}
