// Generates tests/golden/hip_f16_bf16_convert.csv and hip_f16_bf16_add.csv from
// the ROCm HIP host headers (hip_bfloat16 / __half), i.e. from the third-party
// arithmetic the reference calls for 16-bit floats:
//   device : src/device/reduce_kernel.h:166 (half +), 260-262 (bf16 through float)
//   host   : test/common/PtrUnion.cpp:181,185 (Set), 271,275 (Reduce)
// Build + run (CPU only):  make -C oracle golden
// TEST INFRASTRUCTURE ONLY — never linked into the shipped library.
#include <hip/hip_runtime.h>
#include <hip/hip_bfloat16.h>
#include <hip/hip_fp16.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

static uint64_t lcg = 0x5EED0000ULL;
static uint32_t next32() { lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL; return (uint32_t)(lcg >> 32); }

static uint16_t bits16(const void* p) { uint16_t u; memcpy(&u, p, 2); return u; }

int main(int argc, char** argv) {
  const char* dir = argc > 1 ? argv[1] : ".";
  char path[512];
  snprintf(path, sizeof path, "%s/hip_f16_bf16_convert.csv", dir);
  FILE* f = fopen(path, "w");
  if (!f) return 1;
  fprintf(f, "f32_bits,bf16_bits,f16_bits\n");
  // the reference's test pattern values 1/(v+1), v = 0..255 (PtrUnion.cpp:151-152) ...
  for (int v = 0; v < 256; v++) {
    float x = (float)(1.0 / ((double)v + 1.0));
    uint32_t u; memcpy(&u, &x, 4);
    hip_bfloat16 b(x); __half h = __float2half(x);
    fprintf(f, "%08x,%04x,%04x\n", u, bits16(&b), bits16(&h));
  }
  // ... plus random finite bit patterns over the whole exponent range (ties, subnormals, overflow)
  for (int i = 0; i < 6000; i++) {
    uint32_t u = next32();
    if ((u & 0x7f800000u) == 0x7f800000u) u &= 0xff800000u;  // inf instead of NaN payloads
    if (i % 7 == 0) u &= 0xffffe000u;                        // exactly representable in f16 mantissa
    if (i % 11 == 0) u = (u & 0xffffe000u) | 0x1000u;        // f16 tie
    if (i % 13 == 0) u = (u & 0xffff0000u) | 0x8000u;        // bf16 tie
    if (i % 5 == 0) u = (u & 0x807fffffu) | ((uint32_t)(100 + next32() % 40) << 23);  // f16 normal/subnormal range
    float x; memcpy(&x, &u, 4);
    hip_bfloat16 b(x); __half h = __float2half(x);
    fprintf(f, "%08x,%04x,%04x\n", u, bits16(&b), bits16(&h));
  }
  fclose(f);

  snprintf(path, sizeof path, "%s/hip_f16_bf16_add.csv", dir);
  f = fopen(path, "w");
  if (!f) return 1;
  fprintf(f, "a_bits,b_bits,f16_sum,f16_prod,bf16_sum,bf16_prod\n");
  for (int i = 0; i < 6000; i++) {
    uint16_t a = (uint16_t)next32(), b = (uint16_t)next32();
    // keep away from NaN encodings in either format
    if ((a & 0x7c00u) == 0x7c00u) a &= 0xbfffu;
    if ((b & 0x7c00u) == 0x7c00u) b &= 0xbfffu;
    if ((a & 0x7f80u) == 0x7f80u) a &= 0xff7fu;
    if ((b & 0x7f80u) == 0x7f80u) b &= 0xff7fu;
    if (i % 3 == 0) b = (uint16_t)((a & 0x7fffu) + (next32() % 64) - 32);  // close magnitudes
    if ((b & 0x7c00u) == 0x7c00u) b &= 0xbfffu;
    if ((b & 0x7f80u) == 0x7f80u) b &= 0xff7fu;
    __half ha, hb; memcpy(&ha, &a, 2); memcpy(&hb, &b, 2);
    __half hs = __float2half(__half2float(ha) + __half2float(hb));   // PtrUnion.cpp:271
    __half hp = __float2half(__half2float(ha) * __half2float(hb));
    hip_bfloat16 ba, bb; memcpy(&ba, &a, 2); memcpy(&bb, &b, 2);
    hip_bfloat16 bs = ba + bb;                                        // PtrUnion.cpp:275
    hip_bfloat16 bp = ba * bb;
    fprintf(f, "%04x,%04x,%04x,%04x,%04x,%04x\n", a, b, bits16(&hs), bits16(&hp), bits16(&bs), bits16(&bp));
  }
  fclose(f);
  return 0;
}
