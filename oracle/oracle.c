/*
 * CPU oracle for the all-reduce / reduce-scatter / all-gather hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library links, loads or
 * calls this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it, and only as the checker / reported baseline.
 *
 * It restates, in plain C, the arithmetic of the reference (ROCm/rccl @
 * 2024-10-24).  Every function cites the reference file:line it follows.
 *
 * Pinning: checked in tests/test_oracle.py against
 *   - the reference's known-answer test (test/StandaloneTests.cpp:191-246:
 *     2 ranks, int32, N=1250, in[i]=i -> out[i]=2i),
 *   - golden vectors produced by the ROCm 7.2.0 HIP host headers
 *     (hip_bfloat16 / __half conversions and adds, the third-party arithmetic
 *     the reference calls at reduce_kernel.h:166,260-262) under tests/golden/,
 *   - the reference's tolerance rules (test/common/PtrUnion.cpp:313-342).
 * The reference's own host oracle (test/common/PtrUnion.cpp) does not build
 * here (it needs the cmake-generated rccl.h), so there is no oracle/_ref.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ncclDataType_t values, src/nccl.h.in:364-377 */
enum { DT_I8 = 0, DT_U8 = 1, DT_I32 = 2, DT_U32 = 3, DT_I64 = 4, DT_U64 = 5,
       DT_F16 = 6, DT_F32 = 7, DT_F64 = 8, DT_BF16 = 9, DT_F8E4M3 = 10, DT_F8E5M2 = 11 };
/* ncclRedOp_t values, src/nccl.h.in:353-358 */
enum { OP_SUM = 0, OP_PROD = 1, OP_MAX = 2, OP_MIN = 3, OP_AVG = 4 };

size_t orc_type_size(int dt) {
  /* test/common/PtrUnion.cpp:11-31 */
  switch (dt) {
    case DT_I8: case DT_U8: case DT_F8E4M3: case DT_F8E5M2: return 1;
    case DT_F16: case DT_BF16: return 2;
    case DT_I32: case DT_U32: case DT_F32: return 4;
    case DT_I64: case DT_U64: case DT_F64: return 8;
    default: return 0;
  }
}

/* ---- 16-bit float formats -------------------------------------------------
 * bf16: round-to-nearest-even truncation of the f32 bits, NaN kept quiet
 *       (hip_bfloat16 float_to_bfloat16, amd_hip_bfloat16.h:96-118).
 * f16 : IEEE binary16 RNE (__float2half / __half2float, hip_fp16.h).        */
static inline uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

uint16_t orc_f32_to_bf16(float f) {
  uint32_t u = f32_bits(f);
  if (~u & 0x7f800000u) {
    u += 0x7fffu + ((u >> 16) & 1u);          /* RNE on the dropped 16 bits */
  } else if (u & 0xffffu) {
    u |= 0x10000u;                            /* NaN: keep it a NaN */
  }
  return (uint16_t)(u >> 16);
}
float orc_bf16_to_f32(uint16_t h) { return bits_f32((uint32_t)h << 16); }

uint16_t orc_f32_to_f16(float f) {
  uint32_t x = f32_bits(f);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t mant = x & 0x7fffffu;
  int32_t  exp  = (int32_t)((x >> 23) & 0xffu);
  if (exp == 0xff) return (uint16_t)(sign | 0x7c00u | (mant ? (0x200u | (mant >> 13)) : 0u));
  exp = exp - 127 + 15;
  if (exp >= 0x1f) return (uint16_t)(sign | 0x7c00u);           /* overflow -> inf */
  if (exp <= 0) {                                               /* subnormal / zero */
    if (exp < -10) return (uint16_t)sign;
    mant |= 0x800000u;
    uint32_t shift = (uint32_t)(14 - exp);
    uint32_t h = mant >> shift;
    uint32_t rem = mant & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (h & 1u))) h++;
    return (uint16_t)(sign | h);
  }
  uint32_t h = ((uint32_t)exp << 10) | (mant >> 13);
  uint32_t rem = mant & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;       /* may carry into exp: ok */
  return (uint16_t)(sign | h);
}
float orc_f16_to_f32(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t mant = h & 0x3ffu;
  if (exp == 0x1f) return bits_f32(sign | 0x7f800000u | (mant << 13));
  if (exp == 0) {
    if (!mant) return bits_f32(sign);
    int e = -1;
    do { mant <<= 1; e++; } while (!(mant & 0x400u));
    mant &= 0x3ffu;
    return bits_f32(sign | ((uint32_t)(127 - 15 - e) << 23) | (mant << 13));
  }
  return bits_f32(sign | ((exp - 15 + 127) << 23) | (mant << 13));
}

/* ---- 8-bit floats ------------------------------------------------------------
 * The reference's rccl_float8 (E4M3, wm=3) and rccl_bfloat8 (E5M2, wm=2) in the encoding where
 * 0x80 is NaN and there is no negative zero or infinity: exponent bias 2^(we-1)
 * (src/include/rccl_float8.h:137,301-302).  Conversion from float rounds to nearest even;
 * NaN/Inf -> 0x80; a magnitude beyond the largest finite value saturates to it, because the
 * reference builds with clipping on (rccl_float8.h:48,221-229,412-415).
 * Restated arithmetically; pinned code by code against tests/golden/ref_fp8_*.csv, which
 * oracle/gen_golden_fp8.cc produces from the reference header itself.                          */
static float f8_decode(uint8_t x, int wm) {
  const int we = 7 - wm, bias = 1 << (we - 1);
  if (x == 0) return 0.0f;
  if (x == 0x80) return bits_f32(0x7f800001u);
  const int sign = x >> 7, e = (x & 0x7f) >> wm, m = x & ((1 << wm) - 1);
  const float mag = e ? ldexpf((float)((1 << wm) | m), e - bias - wm) : ldexpf((float)m, 1 - bias - wm);
  return sign ? -mag : mag;
}
static uint8_t f8_encode(float f, int wm) {
  const int we = 7 - wm, bias = 1 << (we - 1), maxExp = (1 << we) - 1;
  const uint32_t u = f32_bits(f);
  if ((u & 0x7f800000u) == 0x7f800000u) return 0x80;
  const uint32_t sign = u >> 31;
  const float a = fabsf(f);
  if (a == 0.0f) return 0;
  int e2;
  (void)frexpf(a, &e2);
  int E = e2 - 1;                                   /* a = 1.xxx * 2^E */
  int expField, mant;
  if (E < 1 - bias) {                               /* below the smallest normal: fixed quantum */
    const float r = nearbyintf(ldexpf(a, -(1 - bias - wm)));
    const int q = (int)r;
    if (q == 0) return 0;
    if (q >= (1 << wm)) { expField = 1; mant = q - (1 << wm); }
    else { expField = 0; mant = q; }
  } else {
    int q = (int)nearbyintf(ldexpf(a, -(E - wm)));  /* in [2^wm, 2^(wm+1)] */
    if (q == (1 << (wm + 1))) { E++; q = 1 << wm; }
    expField = E + bias;
    mant = q - (1 << wm);
  }
  if (expField > maxExp) return (uint8_t)((sign << 7) | 0x7f);     /* clip to the largest finite value */
  return (uint8_t)((sign << 7) | ((uint32_t)expField << wm) | (uint32_t)mant);
}
float orc_f8e4m3_to_f32(uint8_t x) { return f8_decode(x, 3); }
float orc_f8e5m2_to_f32(uint8_t x) { return f8_decode(x, 2); }
uint8_t orc_f32_to_f8e4m3(float f) { return f8_encode(f, 3); }
uint8_t orc_f32_to_f8e5m2(float f) { return f8_encode(f, 2); }

/* ---- input pattern --------------------------------------------------------
 * test/common/PtrUnion.cpp:135-168 (FillPattern) + 170-191 (Set):
 * valueI = (rank+i)%256 ; valueF = 1/(valueI+1) in double, narrowed to T.   */
int orc_fill_pattern(int dt, size_t n, int rank, void* out) {
  for (size_t i = 0; i < n; i++) {
    int vi = (int)((rank + (int)i) % 256);       /* reference index is an int */
    double vf = 1.0 / ((double)vi + 1.0);
    switch (dt) {
      case DT_I8:  ((int8_t*)out)[i]   = (int8_t)vi; break;
      case DT_U8:  ((uint8_t*)out)[i]  = (uint8_t)vi; break;
      case DT_I32: ((int32_t*)out)[i]  = vi; break;
      case DT_U32: ((uint32_t*)out)[i] = (uint32_t)vi; break;
      case DT_I64: ((int64_t*)out)[i]  = vi; break;
      case DT_U64: ((uint64_t*)out)[i] = (uint64_t)vi; break;
      case DT_F16: ((uint16_t*)out)[i] = orc_f32_to_f16((float)vf); break;
      case DT_F32: ((float*)out)[i]    = (float)vf; break;
      case DT_F64: ((double*)out)[i]   = vf; break;
      case DT_BF16:((uint16_t*)out)[i] = orc_f32_to_bf16((float)vf); break;
      case DT_F8E4M3: ((uint8_t*)out)[i] = orc_f32_to_f8e4m3((float)vf); break;
      case DT_F8E5M2: ((uint8_t*)out)[i] = orc_f32_to_f8e5m2((float)vf); break;
      default: return -1;
    }
  }
  return 0;
}

/* ---- element ops ----------------------------------------------------------
 * Device semantics: src/device/reduce_kernel.h
 *   Sum/Prod in T (163-174); bf16 through float with one rounding (260-262);
 *   f16 native half add == float add + RNE (innocuous double rounding, p=24>=2*11+2);
 *   integer min/max by xor-mask compare == plain min/max on T (175-180 with the
 *   masks of src/enqueue.cc:1995-2005); float min/max fminf/fmaxf (246-249).
 * Host test oracle: test/common/PtrUnion.hpp:17-30 (ReduceOp) — same results for
 *   non-NaN data.                                                            */
#define INT_OP(T) \
static inline T op_##T(int op, T a, T b) { \
  switch (op) { case OP_SUM: return (T)(a + b); case OP_PROD: return (T)(a * b); \
    case OP_MAX: return a > b ? a : b; case OP_MIN: return a < b ? a : b; default: return a; } }
typedef int8_t i8; typedef uint8_t u8; typedef int32_t i32; typedef uint32_t u32;
typedef int64_t i64; typedef uint64_t u64;
/* signed overflow must wrap like the device's two's-complement ALU */
static inline i8  op_i8 (int op, i8 a, i8 b)  { switch (op) { case OP_SUM: return (i8)((u8)a + (u8)b); case OP_PROD: return (i8)((u8)a * (u8)b); case OP_MAX: return a > b ? a : b; case OP_MIN: return a < b ? a : b; default: return a; } }
static inline i32 op_i32(int op, i32 a, i32 b){ switch (op) { case OP_SUM: return (i32)((u32)a + (u32)b); case OP_PROD: return (i32)((u32)a * (u32)b); case OP_MAX: return a > b ? a : b; case OP_MIN: return a < b ? a : b; default: return a; } }
static inline i64 op_i64(int op, i64 a, i64 b){ switch (op) { case OP_SUM: return (i64)((u64)a + (u64)b); case OP_PROD: return (i64)((u64)a * (u64)b); case OP_MAX: return a > b ? a : b; case OP_MIN: return a < b ? a : b; default: return a; } }
INT_OP(u8) INT_OP(u32) INT_OP(u64)

static inline float op_f32(int op, float a, float b) {
  switch (op) { case OP_SUM: return a + b; case OP_PROD: return a * b;
    case OP_MAX: return fmaxf(a, b); case OP_MIN: return fminf(a, b); default: return a; } }
static inline double op_f64(int op, double a, double b) {
  switch (op) { case OP_SUM: return a + b; case OP_PROD: return a * b;
    case OP_MAX: return fmax(a, b); case OP_MIN: return fmin(a, b); default: return a; } }
static inline uint16_t op_f16(int op, uint16_t a, uint16_t b) {
  return orc_f32_to_f16(op_f32(op, orc_f16_to_f32(a), orc_f16_to_f32(b))); }
static inline uint16_t op_bf16(int op, uint16_t a, uint16_t b) {
  return orc_f32_to_bf16(op_f32(op, orc_bf16_to_f32(a), orc_bf16_to_f32(b))); }
/* fp8: every op goes through float and back (reduce_kernel.h:267-272; PtrUnion.cpp:270,274) */
typedef uint8_t f8a; typedef uint8_t f8b;
static inline uint8_t op_f8a(int op, uint8_t a, uint8_t b) { return f8_encode(op_f32(op, f8_decode(a, 3), f8_decode(b, 3)), 3); }
static inline uint8_t op_f8b(int op, uint8_t a, uint8_t b) { return f8_encode(op_f32(op, f8_decode(a, 2), f8_decode(b, 2)), 2); }

/* acc[i] = acc[i] (op) in[i] — one hop of any fold.  op is Sum/Prod/Max/Min. */
int orc_reduce_into(int dt, int op, size_t n, void* acc, const void* in) {
#define LOOP(T, F) { T* a = (T*)acc; const T* b = (const T*)in; _Pragma("omp parallel for schedule(static) if(n > 65536)") for (size_t i = 0; i < n; i++) a[i] = F(op, a[i], b[i]); } break
  switch (dt) {
    case DT_I8:  LOOP(i8, op_i8);
    case DT_U8:  LOOP(u8, op_u8);
    case DT_I32: LOOP(i32, op_i32);
    case DT_U32: LOOP(u32, op_u32);
    case DT_I64: LOOP(i64, op_i64);
    case DT_U64: LOOP(u64, op_u64);
    case DT_F16: LOOP(uint16_t, op_f16);
    case DT_F32: LOOP(float, op_f32);
    case DT_F64: LOOP(double, op_f64);
    case DT_BF16:LOOP(uint16_t, op_bf16);
    case DT_F8E4M3: LOOP(uint8_t, op_f8a);
    case DT_F8E5M2: LOOP(uint8_t, op_f8b);
    default: return -1;
  }
#undef LOOP
  return 0;
}

/* acc[i] = in[i] (op) acc[i] — the device's hop: local input is source 0, the
 * received partial is source 1 (prims_simple.h:229,133-134; common_kernel.h:81-133). */
static int reduce_local_first(int dt, int op, size_t n, void* acc, const void* in) {
#define LOOP(T, F) { T* a = (T*)acc; const T* b = (const T*)in; _Pragma("omp parallel for schedule(static) if(n > 65536)") for (size_t i = 0; i < n; i++) a[i] = F(op, b[i], a[i]); } break
  switch (dt) {
    case DT_I8:  LOOP(i8, op_i8);
    case DT_U8:  LOOP(u8, op_u8);
    case DT_I32: LOOP(i32, op_i32);
    case DT_U32: LOOP(u32, op_u32);
    case DT_I64: LOOP(i64, op_i64);
    case DT_U64: LOOP(u64, op_u64);
    case DT_F16: LOOP(uint16_t, op_f16);
    case DT_F32: LOOP(float, op_f32);
    case DT_F64: LOOP(double, op_f64);
    case DT_BF16:LOOP(uint16_t, op_bf16);
    case DT_F8E4M3: LOOP(uint8_t, op_f8a);
    case DT_F8E5M2: LOOP(uint8_t, op_f8b);
    default: return -1;
  }
#undef LOOP
  return 0;
}

/* Host test-oracle average: sum, then divide every element by nranks
 * (test/common/PtrUnion.cpp:285-311; half/bf16 divide in float then narrow). */
int orc_divide_by_int(int dt, size_t n, void* buf, int d) {
  if (orc_type_size(dt) == 0) return -1;
#pragma omp parallel for schedule(static) if(n > 65536)
  for (size_t i = 0; i < n; i++) {
    switch (dt) {
      case DT_I8:  ((i8*)buf)[i]  = (i8)(((i8*)buf)[i] / d); break;
      case DT_U8:  ((u8*)buf)[i]  = (u8)(((u8*)buf)[i] / d); break;
      case DT_I32: ((i32*)buf)[i] /= d; break;
      case DT_U32: ((u32*)buf)[i] /= d; break;   /* u32 / int -> unsigned arithmetic */
      case DT_I64: ((i64*)buf)[i] /= d; break;
      case DT_U64: ((u64*)buf)[i] /= d; break;
      case DT_F16: ((uint16_t*)buf)[i] = orc_f32_to_f16(orc_f16_to_f32(((uint16_t*)buf)[i]) / d); break;
      case DT_F32: ((float*)buf)[i] /= d; break;
      case DT_F64: ((double*)buf)[i] /= d; break;
      case DT_BF16:((uint16_t*)buf)[i] = orc_f32_to_bf16(orc_bf16_to_f32(((uint16_t*)buf)[i]) / d); break;
      case DT_F8E4M3: ((uint8_t*)buf)[i] = f8_encode(f8_decode(((uint8_t*)buf)[i], 3) / d, 3); break;
      case DT_F8E5M2: ((uint8_t*)buf)[i] = f8_encode(f8_decode(((uint8_t*)buf)[i], 2) / d, 2); break;
      default: break;
    }
  }
  return 0;
}

/* Device average, integer types: sum then divide by nranks at the last hop
 * (FuncSumPostDiv, reduce_kernel.h:546-573; divisor is an int, so unsigned T
 * follows C's usual conversions exactly as orc_divide_by_int does).
 * Device average, float types: every rank's input is multiplied by T(1/nranks)
 * BEFORE summation (FuncPreMulSum, reduce_kernel.h:364-531; scalar built at
 * src/enqueue.cc:2013-2040).                                                */
int orc_premul(int dt, size_t n, void* buf, int nranks) {
  switch (dt) {
    case DT_F32: { float s = (float)(1.0 / nranks); float* p = buf; _Pragma("omp parallel for schedule(static) if(n > 65536)") for (size_t i = 0; i < n; i++) p[i] = p[i] * s; } break;
    case DT_F64: { double s = 1.0 / nranks; double* p = buf; _Pragma("omp parallel for schedule(static) if(n > 65536)") for (size_t i = 0; i < n; i++) p[i] = p[i] * s; } break;
    case DT_F16: { float s = orc_f16_to_f32(orc_f32_to_f16((float)(1.0 / nranks))); uint16_t* p = buf;
                   _Pragma("omp parallel for schedule(static) if(n > 65536)")
                   for (size_t i = 0; i < n; i++) p[i] = orc_f32_to_f16(orc_f16_to_f32(p[i]) * s); } break;
    case DT_BF16:{ float s = orc_bf16_to_f32(orc_f32_to_bf16((float)(1.0 / nranks))); uint16_t* p = buf;
                   _Pragma("omp parallel for schedule(static) if(n > 65536)")
                   for (size_t i = 0; i < n; i++) p[i] = orc_f32_to_bf16(orc_bf16_to_f32(p[i]) * s); } break;
    case DT_F8E4M3: { float s = f8_decode(f8_encode((float)(1.0 / nranks), 3), 3); uint8_t* p = buf;
                      for (size_t i = 0; i < n; i++) p[i] = f8_encode(f8_decode(p[i], 3) * s, 3); } break;
    case DT_F8E5M2: { float s = f8_decode(f8_encode((float)(1.0 / nranks), 2), 2); uint8_t* p = buf;
                      for (size_t i = 0; i < n; i++) p[i] = f8_encode(f8_decode(p[i], 2) * s, 2); } break;
    default: return -1;
  }
  return 0;
}
static int is_float_dt(int dt) { return dt == DT_F16 || dt == DT_F32 || dt == DT_F64 || dt >= DT_BF16; }

/* ---- the reference's HOST expected value ---------------------------------
 * test/common/PrepDataFuncs.cpp:95-146: result = in[0]; for r=1..n-1
 * result = result (op) in[r]; Avg = Sum then DivideByInt(nranks).           */
int orc_host_fold(int dt, int op, int nranks, size_t count, const void* const* in, void* out) {
  size_t bytes = count * orc_type_size(dt);
  int base = (op == OP_AVG) ? OP_SUM : op;
  memcpy(out, in[0], bytes);
  for (int r = 1; r < nranks; r++)
    if (orc_reduce_into(dt, base, count, out, in[r])) return -1;
  if (op == OP_AVG) return orc_divide_by_int(dt, count, out, nranks);
  return 0;
}

/* ---- device-order folds ---------------------------------------------------
 * `order` lists the ranks in the order their inputs enter the fold of one
 * element block: the seed is in[order[0]], then each later rank computes
 * local (op) partial.  With order = 0..n-1 and a left fold (rank_first=0) this
 * is the association the all-pairs kernel uses, which equals orc_host_fold
 * for Sum/Prod/Min/Max.  With rank_first=1 it is one ring/chain hop sequence:
 * in[o_{k}] (op) ( ... (in[o_1] (op) in[o_0])) — all_reduce.h:82-144,
 * reduce_scatter.h:42-56.  Average follows the device rule (see orc_premul). */
int orc_device_fold(int dt, int op, int nranks, size_t count, const void* const* in,
                    const int* order, int rank_first, void* out) {
  size_t es = orc_type_size(dt), bytes = count * es;
  int base = (op == OP_AVG) ? OP_SUM : op;
  int premul = (op == OP_AVG) && is_float_dt(dt);
  void* tmp = premul ? malloc(bytes ? bytes : 1) : NULL;
  memcpy(out, in[order[0]], bytes);
  if (premul) orc_premul(dt, count, out, nranks);
  for (int k = 1; k < nranks; k++) {
    const void* src = in[order[k]];
    if (premul) { memcpy(tmp, src, bytes); orc_premul(dt, count, tmp, nranks); src = tmp; }
    int rc = rank_first ? reduce_local_first(dt, base, count, out, src)
                        : orc_reduce_into(dt, base, count, out, src);
    if (rc) { free(tmp); return -1; }
  }
  free(tmp);
  if (op == OP_AVG && !is_float_dt(dt)) return orc_divide_by_int(dt, count, out, nranks);
  return 0;
}

/* ---- ring all-reduce, one channel, reference association order -----------
 * src/device/all_reduce.h:71-199.  `ring[p]` = rank at ring position p.
 * Data is walked in loops of nranks*chunk elements; within a loop, chunk c is
 * seeded by position c+1, travels forward and is finished at position c:
 *   in_c + (in_{c-1} + ( ... + (in_{c+2} + in_{c+1})))   (positions mod n).
 * The last, partial loop uses `last_chunk` as its chunk size
 * (all_reduce.h:75, src/enqueue.cc:277-297).                                */
int orc_ring_allreduce(int dt, int op, int nranks, size_t count, const void* const* in,
                       const int* ring, size_t chunk, size_t last_chunk, void* out) {
  size_t es = orc_type_size(dt);
  int* order = malloc(sizeof(int) * (size_t)nranks);
  const void** ptrs = malloc(sizeof(void*) * (size_t)nranks);
  size_t loop = (size_t)nranks * chunk;
  for (size_t off = 0; off < count; off += loop) {
    size_t rem = count - off;
    size_t ck = (rem < loop) ? last_chunk : chunk;
    for (int c = 0; c < nranks; c++) {
      size_t co = (size_t)c * ck;
      if (co >= rem) continue;
      size_t ne = rem - co < ck ? rem - co : ck;
      for (int k = 0; k < nranks; k++) order[k] = ring[(c + 1 + k) % nranks];
      for (int r = 0; r < nranks; r++) ptrs[r] = (const char*)in[r] + (off + co) * es;
      if (orc_device_fold(dt, op, nranks, ne, ptrs, order, 1, (char*)out + (off + co) * es)) {
        free(order); free(ptrs); return -1; }
    }
  }
  free(order); free(ptrs);
  return 0;
}

/* ---- chain ("intra-node tree") all-reduce --------------------------------
 * src/device/all_reduce.h:269-289 with the chain of src/graph/connect.cc:61-62:
 * leaf sends, middles compute local + received, root finishes:
 *   in_root + (in_1 + ( ... + in_leaf)), chain[0] = root.                    */
int orc_chain_allreduce(int dt, int op, int nranks, size_t count, const void* const* in,
                        const int* chain, void* out) {
  int* order = malloc(sizeof(int) * (size_t)nranks);
  for (int k = 0; k < nranks; k++) order[k] = chain[nranks - 1 - k];
  int rc = orc_device_fold(dt, op, nranks, count, in, order, 1, out);
  free(order);
  return rc;
}

/* ---- comparison with the reference's tolerances --------------------------
 * test/common/PtrUnion.cpp:313-342: ints exact; f32 |d|<1e-5; f64 |d|<1e-12;
 * f16/bf16 |d|<9e-2.  Returns the index of the first mismatch, or -1.       */
long orc_first_mismatch(int dt, size_t n, const void* got, const void* want) {
  for (size_t i = 0; i < n; i++) {
    int ok;
    switch (dt) {
      case DT_I8: case DT_U8:   ok = ((const u8*)got)[i]  == ((const u8*)want)[i]; break;
      case DT_I32: case DT_U32: ok = ((const u32*)got)[i] == ((const u32*)want)[i]; break;
      case DT_I64: case DT_U64: ok = ((const u64*)got)[i] == ((const u64*)want)[i]; break;
      case DT_F16:  ok = fabs(orc_f16_to_f32(((const uint16_t*)got)[i]) - orc_f16_to_f32(((const uint16_t*)want)[i])) < 9e-2; break;
      case DT_BF16: ok = fabs(orc_bf16_to_f32(((const uint16_t*)got)[i]) - orc_bf16_to_f32(((const uint16_t*)want)[i])) < 9e-2; break;
      case DT_F8E4M3: ok = fabs(f8_decode(((const u8*)got)[i], 3) - f8_decode(((const u8*)want)[i], 3)) < 9e-2; break;
      case DT_F8E5M2: ok = fabs(f8_decode(((const u8*)got)[i], 2) - f8_decode(((const u8*)want)[i], 2)) < 9e-2; break;
      case DT_F32:  ok = fabs(((const float*)got)[i] - ((const float*)want)[i]) < 1e-5; break;
      case DT_F64:  ok = fabs(((const double*)got)[i] - ((const double*)want)[i]) < 1e-12; break;
      default: return 0;
    }
    if (!ok) return (long)i;
  }
  return -1;
}

/* ---- synthetic buckets for runs at full size --------------------------------
 * Not part of the reference: its pattern above repeats every 256 elements, so a piece that lands a multiple of 256
 * elements away from its place would go unnoticed.  Here element i of rank r is a hash of (seed, r, i), which lets any
 * process produce any rank's bucket (or any slice of it: `first`) without moving it, and every value sits on a grid the
 * element type holds exactly (k / 2^b in [-1, 1); bf16: [-1/4, 1/4)), so sums round and the association order shows in the bits.       */
static inline uint64_t orc_hash64(uint64_t i, uint32_t rank, uint32_t seed) {
  uint64_t z = i + 0x9E3779B97F4A7C15ull * ((uint64_t)rank + 1) + ((uint64_t)seed << 32);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
int orc_fill_hashed(int dt, size_t n, size_t first, int rank, uint32_t seed, void* out) {
  if (orc_type_size(dt) == 0) return -1;
#pragma omp parallel for schedule(static) if(n > 65536)
  for (size_t i = 0; i < n; i++) {
    const uint64_t h = orc_hash64(first + i, (uint32_t)rank, seed);
    switch (dt) {
      case DT_I8: case DT_U8: ((uint8_t*)out)[i] = (uint8_t)h; break;
      case DT_I32: case DT_U32: ((uint32_t*)out)[i] = (uint32_t)h; break;
      case DT_I64: case DT_U64: ((uint64_t*)out)[i] = h; break;
      case DT_F32: ((float*)out)[i] = (float)((int32_t)(h & 0xffffffu) - 0x800000) * (1.0f / 8388608.0f); break;
      case DT_F64: ((double*)out)[i] = (double)((int64_t)(h & 0xffffffffull) - 0x80000000ll) * (1.0 / 2147483648.0); break;
      case DT_F16: ((uint16_t*)out)[i] = orc_f32_to_f16((float)((int32_t)(h & 0x7ffu) - 1024) * (1.0f / 1024.0f)); break;
      case DT_BF16: ((uint16_t*)out)[i] = orc_f32_to_bf16((float)((int32_t)(h & 0xffu) - 128) * (1.0f / 512.0f)); break;   /* [-1/4, 1/4): an 8-way sum stays below 2, so any two association orders stay within the reference's 9e-2 */
      default: { uint8_t c = (uint8_t)h; ((uint8_t*)out)[i] = c == 0x80 ? 0 : c; } break;    /* 8-bit floats: any code but NaN */
    }
  }
  return 0;
}

/* Elements outside the reference's tolerance (see orc_first_mismatch); *first gets the lowest such index. */
long orc_count_mismatch(int dt, size_t n, const void* got, const void* want, long* first) {
  long bad = 0, lowest = -1;
  const size_t block = 1u << 20;
  const size_t nblocks = (n + block - 1) / block;
#pragma omp parallel for schedule(static) reduction(+ : bad)
  for (size_t b = 0; b < nblocks; b++) {
    const size_t lo = b * block, len = lo + block < n ? block : n - lo;
    const size_t es = orc_type_size(dt);
    size_t at = 0;
    while (at < len) {
      long k = orc_first_mismatch(dt, len - at, (const char*)got + (lo + at) * es, (const char*)want + (lo + at) * es);
      if (k < 0) break;
      bad++;
#pragma omp critical
      { if (lowest < 0 || (long)(lo + at) + k < lowest) lowest = (long)(lo + at) + k; }
      at += (size_t)k + 1;
    }
  }
  if (first) *first = lowest;
  return bad;
}

/* ---- CPU baseline: OpenMP element-wise sum of n fp32 buffers -------------
 * The "host-core OpenMP sum-reduce over the same buffers": out = in[0]+...+in[n-1]
 * folded in rank order (bit-identical to orc_host_fold for fp32 sum).       */
void orc_omp_sum_f32(int nranks, const float* const* in, float* out, size_t count) {
#pragma omp parallel for simd schedule(static)
  for (size_t i = 0; i < count; i++) {
    float a = in[0][i];
    for (int r = 1; r < nranks; r++) a += in[r][i];
    out[i] = a;
  }
}
int orc_omp_threads(void) {
#ifdef _OPENMP
  extern int omp_get_max_threads(void);
  return omp_get_max_threads();
#else
  return 1;
#endif
}
/* The reference's own CPU path for the same sum: the single-threaded fold of
 * test/common/PtrUnion.cpp:260-281, one rank at a time.                     */
void orc_ref_style_sum_f32(int nranks, const float* const* in, float* out, size_t count) {
  memcpy(out, in[0], count * sizeof(float));
  for (int r = 1; r < nranks; r++)
    for (size_t i = 0; i < count; i++) out[i] = out[i] + in[r][i];
}

/* Array converters (used to build random bf16/f16 test inputs). */
void orc_cvt_f32_to_bf16(size_t n, const float* in, uint16_t* out) { for (size_t i = 0; i < n; i++) out[i] = orc_f32_to_bf16(in[i]); }
void orc_cvt_f32_to_f16(size_t n, const float* in, uint16_t* out) { for (size_t i = 0; i < n; i++) out[i] = orc_f32_to_f16(in[i]); }
void orc_cvt_f16_to_f32(size_t n, const uint16_t* in, float* out) { for (size_t i = 0; i < n; i++) out[i] = orc_f16_to_f32(in[i]); }
