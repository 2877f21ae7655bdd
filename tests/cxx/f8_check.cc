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

// The finite-path arithmetic of the sum / product folds and the fold of whole dwords, NaN codes included, against the chain
// decode -> op -> encode of the oracle's conversions (what the reference computes pair by pair).
static uint64_t rngState = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rngState ^= rngState << 13; rngState ^= rngState >> 7; rngState ^= rngState << 17; return (uint32_t)(rngState >> 16); }

template <int WM> static long checkFold(uint64_t stride, long nfolds) {
  long bad = 0;
  auto dec = [](uint8_t c) { return WM == 3 ? orc_f8e4m3_to_f32(c) : orc_f8e5m2_to_f32(c); };
  auto enc = [](float v) { return WM == 3 ? orc_f32_to_f8e4m3(v) : orc_f32_to_f8e5m2(v); };
  for (uint32_t c = 0; c < 256; c++) {
    if (c == 0x80) continue;
    for (int i = 0; i < 4; i++) {
      const uint32_t w = (c << (8 * i)) | (0x5a5a5a5au & ~(0xffu << (8 * i)));
      if (!same(f8DecodeFinite<WM>(w, i), dec((uint8_t)c)) && bad++ < 5) printf("decode-finite<%d>(0x%02x at byte %d) = %g\n", WM, c, i, f8DecodeFinite<WM>(w, i));
    }
    if (f8EncodeFinite<WM>(dec((uint8_t)c)) != c && bad++ < 5) printf("encode-finite<%d>(decode(0x%02x)) = 0x%02x\n", WM, c, f8EncodeFinite<WM>(dec((uint8_t)c)));
  }
  if (f8EncodeFinite<WM>(-0.0f) != 0) { bad++; printf("encode-finite<%d>(-0) != 0\n", WM); }
#pragma omp parallel for reduction(+ : bad) schedule(static)
  for (int64_t k = 0; k < (int64_t)((1ull << 32) / stride); k++) {
    const uint32_t u = (uint32_t)((uint64_t)k * stride);
    if ((u & 0x7f800000u) == 0x7f800000u) continue;                    // infinities and NaNs never reach the finite path
    const float v = raU2F(u), got = f8QuantizeFinite<WM>(v), want = f8Quantize<WM>(v);
    if (raF2U(got) != raF2U(want)) {
      if (bad < 5) printf("quantise-finite<%d>(0x%08x = %g) = %g, want %g\n", WM, u, v, got, want);
      bad++;
    }
  }
  for (long f = 0; f < nfolds; f++) {
    const int ns = 1 + (int)(rnd() % 8);
    const bool prod = rnd() & 1;
    const int preIdx = prod || (rnd() & 1) ? -1 : (int)(rnd() % (uint32_t)ns);
    const uint32_t kCode = (rnd() % 64 == 0) ? 0x80u : (rnd() & 0xffu);
    uint32_t ws[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < ns; j++) {
      ws[j] = rnd() ^ (rnd() << 16);
      if (rnd() % 16 == 0) ws[j] = (ws[j] & ~(0xffu << (8 * (rnd() % 4)))) | (0x80u << (8 * (rnd() % 4)));
      if (rnd() % 4 == 0) ws[j] &= 0x3f3f3f3fu;                        // small magnitudes: subnormal codes, no clipping
    }
    uint32_t want = 0;
    for (int i = 0; i < 4; i++) {
      uint8_t acc = (uint8_t)(ws[0] >> (8 * i));
      if (preIdx == 0) acc = enc(dec(acc) * dec((uint8_t)kCode));
      for (int j = 1; j < ns; j++) {
        uint8_t x = (uint8_t)(ws[j] >> (8 * i));
        if (preIdx == j) x = enc(dec(x) * dec((uint8_t)kCode));
        acc = enc(prod ? dec(acc) * dec(x) : dec(acc) + dec(x));
      }
      want |= (uint32_t)acc << (8 * i);
    }
    const uint32_t got = prod ? f8FoldDword<WM, true, 8>(ws, ns, preIdx, kCode) : f8FoldDword<WM, false, 8>(ws, ns, preIdx, kCode);
    if (got != want) {
      if (bad < 5) printf("fold<%d> %s of %d sources (pre %d, k 0x%02x): 0x%08x, want 0x%08x; first source 0x%08x\n", WM, prod ? "product" : "sum", ns, preIdx, kCode, got, want, ws[0]);
      bad++;
    }
  }
  return bad;
}

int main(int argc, char** argv) {
  const uint64_t stride = argc > 1 ? strtoull(argv[1], nullptr, 0) : 1;
  const long nfolds = argc > 2 ? atol(argv[2]) : 2000000;
  const long bad = check<3>(stride) + check<2>(stride) + checkFold<3>(stride, nfolds) + checkFold<2>(stride, nfolds);
  if (bad) { printf("%ld mismatches\n", bad); return 1; }
  printf("ok: 256 codes and every %llu-th float, both formats\n", (unsigned long long)stride);
  return 0;
}
