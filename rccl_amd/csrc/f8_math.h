// Arithmetic on the reference's 8-bit floats (rccl_float8 = E4M3, rccl_bfloat8 = E5M2 in the encoding where 0x80 is NaN,
// there is no negative zero and no infinity, and the exponent bias is 2^(we-1); src/include/rccl_float8.h:61-329), written
// so that the same text compiles for the GPU and, for an exhaustive host-side check of every input, for the host.
//
// The reference computes every pairwise operation as decode -> float op -> encode (src/device/reduce_kernel.h:267-272).
// A fold over several sources therefore re-encodes and re-decodes its running value at every step.  The same bits come out
// when the running value stays a float that is ROUNDED ONTO THE 8-BIT GRID after every step (f8Quantize: what encode
// followed by decode would give), and is encoded once at the end: that is how the copy loops fold these types.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define RA_F8_FN __host__ __device__ __forceinline__
#define RA_F8_UNROLL _Pragma("unroll")
#else
#define RA_F8_FN inline
#define RA_F8_UNROLL
#endif

RA_F8_FN uint32_t raF2U(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
RA_F8_FN float raU2F(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

template <int WM> struct F8Fmt {
  static constexpr int WE = 7 - WM, BIAS = 1 << (WE - 1);
  static constexpr uint32_t DROP = 23 - WM;                                            // float mantissa bits below the grid
  static constexpr uint32_t MIN_NORMAL = (uint32_t)(127 + 1 - BIAS) << 23;             // smallest normal magnitude, as float bits
  static constexpr uint32_t MAX_FINITE = ((uint32_t)(127 - BIAS + (1 << WE) - 1) << 23) | (((1u << WM) - 1u) << DROP);
  static constexpr uint32_t REBIAS = (uint32_t)(127 - BIAS) << WM;                     // code of a normal = (bits >> DROP) - REBIAS
  // adding 2^(1-BIAS-WM+23) makes the float adder round at the fixed quantum 2^(1-BIAS-WM) of the codes below MIN_NORMAL
  static constexpr uint32_t MAGIC = (uint32_t)(127 + 1 - BIAS - WM + 23) << 23;
  static constexpr uint32_t DECODE_SCALE = (uint32_t)(127 + 127 - BIAS) << 23;         // 2^(127-BIAS)
};

// code (low 8 bits of x) -> float.  The 7 magnitude bits, placed at the top of a float's exponent and mantissa, read as
// 2^(E-127) x 1.m (or, with E = 0, as the float denormal m x 2^(-126-WM)); one multiplication by 2^(127-BIAS) turns
// both into the code's value (the kernels run with float denormals on).
template <int WM> RA_F8_FN float f8Decode(uint32_t x) {
  typedef F8Fmt<WM> F;
  const float mag = raU2F((x & 0x7fu) << F::DROP) * raU2F(F::DECODE_SCALE);
  const float f = raU2F(raF2U(mag) | ((x & 0x80u) << 24));
  return (x & 0xffu) == 0x80u ? raU2F(0x7fc00000u) : f;
}

// float -> the nearest value of the 8-bit grid, as a float: round to nearest even, magnitudes beyond the largest finite
// value clip to it, infinity and NaN give NaN.  (The sign of a zero result is kept; f8Encode drops it.)
template <int WM> RA_F8_FN float f8Quantize(float v) {
  typedef F8Fmt<WM> F;
  const uint32_t u = raF2U(v), a = u & 0x7fffffffu;
  const uint32_t rn = (a + ((a >> F::DROP) & 1u) + ((1u << (F::DROP - 1)) - 1u)) & ~((1u << F::DROP) - 1u);
  const uint32_t rs = raF2U((raU2F(a) + raU2F(F::MAGIC)) - raU2F(F::MAGIC));
  uint32_t r = a >= F::MIN_NORMAL ? rn : rs;
  r = r > F::MAX_FINITE ? F::MAX_FINITE : r;
  r = a >= 0x7f800000u ? 0x7fc00000u : r;
  return raU2F((u & 0x80000000u) | r);
}

// a float that already lies on the grid (f8Decode's or f8Quantize's result) -> code
template <int WM> RA_F8_FN uint32_t f8EncodeOnGrid(float v) {
  typedef F8Fmt<WM> F;
  const uint32_t u = raF2U(v), a = u & 0x7fffffffu;
  const uint32_t cn = (a >> F::DROP) - F::REBIAS;
  const uint32_t cs = raF2U(raU2F(a) + raU2F(F::MAGIC)) - F::MAGIC;
  const uint32_t c = a >= F::MIN_NORMAL ? cn : cs;
  return a >= 0x7f800000u ? 0x80u : (c ? (((u >> 24) & 0x80u) | c) : 0u);
}

// any float -> code (rccl_float8.h:48,221-229,412-415: round to nearest even, clip, NaN and infinity -> 0x80)
template <int WM> RA_F8_FN uint32_t f8Encode(float v) { return f8EncodeOnGrid<WM>(f8Quantize<WM>(v)); }

// ---------------------------------------------------------------------------------------------------------------------
// The fold of one dword (four codes) over its sources for sums and products, with the NaN code handled beside the
// arithmetic: a NaN input makes the result NaN whatever else is folded (NaN + x, NaN * x, and a NaN scale factor), so
// the float pipeline only ever sees finite values and needs no NaN or infinity cases (the largest product of two grid
// values, and of a grid value with the scale factor, is far inside float's range), and the bytes of the result where
// any source held the NaN code are set to it at the end.  Maxima and minima do not use this (fmaxf drops a NaN).

// byte i of w -> float, for codes other than NaN (the NaN code 0x80 comes out as -0.0)
template <int WM> RA_F8_FN float f8DecodeFinite(uint32_t w, int i) {
  typedef F8Fmt<WM> F;
  const uint32_t t = i == 3 ? w : w << (24 - 8 * i);                         // sign at bit 31, magnitude bits below it
  const uint32_t m = (uint32_t)((int32_t)t >> (24 - F::DROP)) & (0x80000000u | (0x7fu << F::DROP));
  return raU2F(m) * raU2F(F::DECODE_SCALE);
}

// finite float -> nearest grid value (round to nearest even, clipped), sign kept.  The float adder rounds: adding
// 2^(E + 23 - WM) to |v| (E = v's exponent, at least that of the smallest normal) leaves exactly WM fraction bits.
template <int WM> RA_F8_FN float f8QuantizeFinite(float v) {
  typedef F8Fmt<WM> F;
  const uint32_t u = raF2U(v);
  uint32_t e = u & 0x7f800000u;
  e = e > F::MIN_NORMAL ? e : F::MIN_NORMAL;
  const float c = raU2F(e + (F::DROP << 23));
  float r = (raU2F(u & 0x7fffffffu) + c) - c;
  const float mx = raU2F(F::MAX_FINITE);
  r = r < mx ? r : mx;
  return raU2F((u & 0x80000000u) | raF2U(r));
}

// finite grid value -> code.  Dividing by the decode scale puts the 7 magnitude bits back at the top of a float's
// exponent and mantissa fields (exactly, float denormals included); a zero magnitude has no sign.
template <int WM> RA_F8_FN uint32_t f8EncodeFinite(float v) {
  typedef F8Fmt<WM> F;
  const uint32_t u = raF2U(v);
  const float q = raU2F(u & 0x7fffffffu) * raU2F((uint32_t)(127 - (127 - F::BIAS)) << 23);     // 2^-(127 - BIAS)
  const uint32_t m7 = (raF2U(q) >> F::DROP) & 0x7fu;
  const uint32_t s = (u >> 24) & 0x80u;
  return m7 | (s & (m7 + 0x7fu));
}

// bit 7 of every byte of the result: set where the byte of w is NOT the NaN code
RA_F8_FN uint32_t f8NotNanBytes(uint32_t w) {
  const uint32_t t = w ^ 0x80808080u;
  return ((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t;
}

// R: 0 sum, 1 product (RA_OP_SUM / RA_OP_PROD).  ws[0..ns): the dword of every source; the source at preIdx is multiplied
// by the scale factor whose code is kCode first (preIdx < 0: none).
template <int WM, bool PROD, int NMAX>
RA_F8_FN uint32_t f8FoldDword(const uint32_t (&ws)[NMAX], int ns, int preIdx, uint32_t kCode) {
  const float k = f8DecodeFinite<WM>(kCode & 0xffu, 0);
  uint32_t ok = f8NotNanBytes(ws[0]);
  float st[4];
  RA_F8_UNROLL for (int i = 0; i < 4; i++) {
    st[i] = f8DecodeFinite<WM>(ws[0], i);
    if (preIdx == 0) st[i] = f8QuantizeFinite<WM>(st[i] * k);
  }
  RA_F8_UNROLL for (int j = 1; j < NMAX; j++)
    if (j < ns) {
      ok &= f8NotNanBytes(ws[j]);
      RA_F8_UNROLL for (int i = 0; i < 4; i++) {
        float x = f8DecodeFinite<WM>(ws[j], i);
        if (preIdx == j) x = f8QuantizeFinite<WM>(x * k);
        st[i] = f8QuantizeFinite<WM>(PROD ? st[i] * x : st[i] + x);
      }
    }
  uint32_t out = f8EncodeFinite<WM>(st[0]) | (f8EncodeFinite<WM>(st[1]) << 8) | (f8EncodeFinite<WM>(st[2]) << 16) |
                 (f8EncodeFinite<WM>(st[3]) << 24);
  uint32_t nan = (ok & 0x80808080u) ^ 0x80808080u;                 // 0x80 in every byte where a source held the NaN code
  if (preIdx >= 0 && (kCode & 0xffu) == 0x80u) nan = 0x80808080u;  // a NaN scale factor
  const uint32_t full = (nan >> 7) * 0xffu;
  return (out & ~full) | nan;
}
