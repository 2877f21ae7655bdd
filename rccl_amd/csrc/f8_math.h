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
#else
#define RA_F8_FN inline
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
