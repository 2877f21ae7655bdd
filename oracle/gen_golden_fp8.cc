// Generates tests/golden/ref_fp8_decode.csv and ref_fp8_encode.csv from the reference's OWN fp8
// implementation, compiled from where it lies (src/include/rccl_float8.h: rccl_float8 = E4M3,
// rccl_bfloat8 = E5M2, both in the "negative zero is NaN" encoding, software conversion path with
// clipping (rccl_float8.h:48) — the path every architecture other than gfx94x takes, rccl_float8.h:402-421,462-468).
// Build + run (CPU only):  make -C oracle golden_fp8
// TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>
#include "rccl_float8.h"
#include <cstdint>
#include <cstdio>
#include <cstring>

static uint64_t lcg = 0xF8F8F8F8ULL;
static uint32_t next32() { lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL; return (uint32_t)(lcg >> 32); }

int main(int argc, char** argv) {
  const char* dir = argc > 1 ? argv[1] : ".";
  char path[512];
  snprintf(path, sizeof path, "%s/ref_fp8_decode.csv", dir);
  FILE* f = fopen(path, "w");
  if (!f) return 1;
  fprintf(f, "code,e4m3_f32_bits,e5m2_f32_bits\n");
  for (int c = 0; c < 256; c++) {
    rccl_float8 a; a.data = (uint8_t)c;
    rccl_bfloat8 b; b.data = (uint8_t)c;
    float fa = float(a), fb = float(b);
    uint32_t ua, ub; memcpy(&ua, &fa, 4); memcpy(&ub, &fb, 4);
    fprintf(f, "%02x,%08x,%08x\n", c, ua, ub);
  }
  fclose(f);
  snprintf(path, sizeof path, "%s/ref_fp8_encode.csv", dir);
  f = fopen(path, "w");
  if (!f) return 1;
  fprintf(f, "f32_bits,e4m3_code,e5m2_code\n");
  auto emit = [&](float x) {
    uint32_t u; memcpy(&u, &x, 4);
    rccl_float8 a(x); rccl_bfloat8 b(x);
    fprintf(f, "%08x,%02x,%02x\n", u, a.data, b.data);
  };
  for (int v = 0; v < 256; v++) emit((float)(1.0 / ((double)v + 1.0)));          // the test pattern values
  for (int c = 0; c < 256; c++) {                                                 // sums/products of code pairs
    rccl_float8 a; a.data = (uint8_t)c; rccl_bfloat8 b; b.data = (uint8_t)(255 - c);
    if (c != 0x80) { emit(float(a) * 1.0625f); emit(float(a) + float(a) * 0.0625f); }
    if ((255 - c) != 0x80) { emit(float(b) * 1.125f); emit(float(b) * 0.875f); }
  }
  emit(0.0f); emit(240.0f); emit(-240.0f); emit(248.0f); emit(256.0f); emit(1e9f); emit(-1e9f); emit(57344.0f); emit(65536.0f);
  for (int i = 0; i < 6000; i++) {
    // random values with exponents in the range where both formats are exercised (2^-22 .. 2^18)
    uint32_t u = (next32() & 0x807fffffu) | ((uint32_t)(105 + next32() % 41) << 23);
    if (i % 5 == 0) u &= 0xfff00000u;             // few mantissa bits: exact values and ties
    if (i % 7 == 0) u = (u & 0xfff00000u) | 0x00080000u;
    float x; memcpy(&x, &u, 4);
    emit(x);
  }
  fclose(f);
  return 0;
}
