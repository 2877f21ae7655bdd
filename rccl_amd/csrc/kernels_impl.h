// HIP kernels for gfx950 (MI355X): the reduce-copy core, the step/credit FIFO
// protocol between GPUs, and the collective schedules built from them.
//
// What this file replaces in the reference (ROCm/rccl):
//   reduceCopy / reduceCopyPacks            src/device/common_kernel.h:30-265
//   element ops, pre/post ops               src/device/reduce_kernel.h:44-573
//   Primitives<ProtoSimple> wait/post/op    src/device/prims_simple.h:111-381
//   ring / chain schedules                  src/device/all_reduce.h:17-356,
//                                           reduce_scatter.h:13-58, all_gather.h:13-152
//   one-rank path                           src/device/onerank.cu:21-90
// The all-pairs ("direct") schedule has no counterpart there: it is the native
// schedule for a node whose GPUs are all joined by their own xGMI link.
//
// Element-wise, bandwidth-bound work: 16-byte nontemporal accesses, no MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.h"
#include "f8_math.h"
#include "launch.h"

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ element types
struct bf16_t { uint16_t v; };
typedef _Float16 f16_t;

// bf16 <-> f32 exactly as hip_bfloat16 does it (round to nearest even, NaN stays NaN),
// the arithmetic the reference uses for bf16 (reduce_kernel.h:260-262).
__device__ __forceinline__ float bf2f(bf16_t a) { return __uint_as_float((uint32_t)a.v << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if (~u & 0x7f800000u) u += 0x7fffu + ((u >> 16) & 1u);
  else if (u & 0xffffu) u |= 0x10000u;
  bf16_t r; r.v = (uint16_t)(u >> 16); return r;
}

// 8-bit floats in the reference's encoding (rccl_float8 = E4M3, rccl_bfloat8 = E5M2; 0x80 is NaN, no
// negative zero, no infinity, bias 2^(we-1); src/include/rccl_float8.h:61-329).  Conversion from float
// rounds to nearest even and clips to the largest finite value (rccl_float8.h:48).  The arithmetic lives in f8_math.h
// (branch-free, checked on the host for all 256 codes and all 2^32 floats).
template <int WM> struct f8_t { uint8_t v; };
typedef f8_t<3> f8e4m3_t;
typedef f8_t<2> f8e5m2_t;
template <int WM> __device__ __forceinline__ float f8ToF(f8_t<WM> a) { return f8Decode<WM>(a.v); }
template <int WM> __device__ __forceinline__ f8_t<WM> fToF8(float v) { f8_t<WM> out; out.v = (uint8_t)f8Encode<WM>(v); return out; }

template <typename T> struct IsF8 { static constexpr bool value = false; };
template <int WM> struct IsF8<f8_t<WM>> { static constexpr bool value = true; };
template <typename T> struct IsInt { static constexpr bool value = true; };
template <> struct IsInt<float> { static constexpr bool value = false; };
template <> struct IsInt<double> { static constexpr bool value = false; };
template <> struct IsInt<f16_t> { static constexpr bool value = false; };
template <> struct IsInt<bf16_t> { static constexpr bool value = false; };

// One binary step of the reduction, in element type T (reduce_kernel.h:163-180,246-272).
template <typename T, int OP> struct Elem {
  static __device__ __forceinline__ T red(T a, T b) {
    if (OP == RA_OP_PROD) return (T)(a * b);
    if (OP == RA_OP_MAX) return a > b ? a : b;
    if (OP == RA_OP_MIN) return a < b ? a : b;
    return (T)(a + b);
  }
};
template <int OP> struct Elem<float, OP> {
  static __device__ __forceinline__ float red(float a, float b) {
    if (OP == RA_OP_PROD) return a * b;
    if (OP == RA_OP_MAX) return fmaxf(a, b);
    if (OP == RA_OP_MIN) return fminf(a, b);
    return a + b;
  }
};
template <int OP> struct Elem<double, OP> {
  static __device__ __forceinline__ double red(double a, double b) {
    if (OP == RA_OP_PROD) return a * b;
    if (OP == RA_OP_MAX) return fmax(a, b);
    if (OP == RA_OP_MIN) return fmin(a, b);
    return a + b;
  }
};
template <int OP> struct Elem<f16_t, OP> {
  static __device__ __forceinline__ f16_t red(f16_t a, f16_t b) {
    if (OP == RA_OP_PROD) return a * b;  // native f16 ops: one rounding, as the reference's half ops
    if (OP == RA_OP_MAX) return (f16_t)fmaxf((float)a, (float)b);
    if (OP == RA_OP_MIN) return (f16_t)fminf((float)a, (float)b);
    return a + b;
  }
};
template <int OP> struct Elem<bf16_t, OP> {
  static __device__ __forceinline__ bf16_t red(bf16_t a, bf16_t b) {
    return f2bf(Elem<float, OP>::red(bf2f(a), bf2f(b)));
  }
};

template <int WM, int OP> struct Elem<f8_t<WM>, OP> {   // through float and back (reduce_kernel.h:267-272)
  static __device__ __forceinline__ f8_t<WM> red(f8_t<WM> a, f8_t<WM> b) {
    return fToF8<WM>(Elem<float, OP>::red(f8ToF<WM>(a), f8ToF<WM>(b)));
  }
};

// Pre-multiply (FuncPreMulSum, reduce_kernel.h:364-531) and post-divide
// (FuncSumPostDiv, reduce_kernel.h:546-573).  `arg` carries the scalar's bits.
template <typename T> __device__ __forceinline__ T argAs(uint64_t arg) {
  union { uint64_t u; T t; } x; x.u = arg; return x.t;
}
template <typename T> struct Scale {
  static __device__ __forceinline__ T pre(T a, uint64_t arg) { return (T)(a * argAs<T>(arg)); }
  static __device__ __forceinline__ T post(T a, uint64_t arg) { return (T)(a / (int)arg); }
};
template <> struct Scale<bf16_t> {
  static __device__ __forceinline__ bf16_t pre(bf16_t a, uint64_t arg) { return f2bf(bf2f(a) * bf2f(argAs<bf16_t>(arg))); }
  static __device__ __forceinline__ bf16_t post(bf16_t a, uint64_t) { return a; }
};
template <int WM> struct Scale<f8_t<WM>> {
  static __device__ __forceinline__ f8_t<WM> pre(f8_t<WM> a, uint64_t arg) { return fToF8<WM>(f8ToF<WM>(a) * f8ToF<WM>(argAs<f8_t<WM>>(arg))); }
  static __device__ __forceinline__ f8_t<WM> post(f8_t<WM> a, uint64_t) { return a; }
};
template <> struct Scale<f16_t> {
  static __device__ __forceinline__ f16_t pre(f16_t a, uint64_t arg) { return a * argAs<f16_t>(arg); }
  static __device__ __forceinline__ f16_t post(f16_t a, uint64_t) { return a; }
};
template <> struct Scale<float> {
  static __device__ __forceinline__ float pre(float a, uint64_t arg) { return a * argAs<float>(arg); }
  static __device__ __forceinline__ float post(float a, uint64_t) { return a; }
};
template <> struct Scale<double> {
  static __device__ __forceinline__ double pre(double a, uint64_t arg) { return a * argAs<double>(arg); }
  static __device__ __forceinline__ double post(double a, uint64_t) { return a; }
};

// ------------------------------------------------------------------ 16-byte packs
template <typename T> union Pack {
  u32x4 v;
  T e[16 / sizeof(T)];
};
template <typename T, int OP> struct PackOps {
  static __device__ __forceinline__ u32x4 red(u32x4 a, u32x4 b) {
    constexpr int R = (OP == RA_OP_PREMULSUM || OP == RA_OP_SUMPOSTDIV) ? RA_OP_SUM : OP;
    Pack<T> x, y; x.v = a; y.v = b;
#pragma unroll
    for (int i = 0; i < (int)(16 / sizeof(T)); i++) x.e[i] = Elem<T, R>::red(x.e[i], y.e[i]);
    return x.v;
  }
  static __device__ __forceinline__ u32x4 pre(u32x4 a, uint64_t arg) {
    Pack<T> x; x.v = a;
#pragma unroll
    for (int i = 0; i < (int)(16 / sizeof(T)); i++) x.e[i] = Scale<T>::pre(x.e[i], arg);
    return x.v;
  }
  static __device__ __forceinline__ u32x4 post(u32x4 a, uint64_t arg) {
    Pack<T> x; x.v = a;
#pragma unroll
    for (int i = 0; i < (int)(16 / sizeof(T)); i++) x.e[i] = Scale<T>::post(x.e[i], arg);
    return x.v;
  }
};
// 8-bit integers, four to a dword, without taking the dword apart (the reference's SWAR byte sum:
// src/device/reduce_kernel.h:183-193): sums carry-free through a 7-bit add plus the top bits' exclusive or; products,
// maxima and minima on the even and the odd bytes as two lanes of 16 bits each (v_pk_mul_lo_u16, v_pk_max / v_pk_min;
// signed bytes ride in the high byte of their lane, where a 16-bit signed compare orders them).
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
typedef int16_t i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t asU32(u16x2 v) { union { u16x2 p; uint32_t u; } x; x.p = v; return x.u; }
__device__ __forceinline__ uint32_t asU32(i16x2 v) { union { i16x2 p; uint32_t u; } x; x.p = v; return x.u; }
__device__ __forceinline__ u16x2 asU16x2(uint32_t u) { union { u16x2 p; uint32_t u; } x; x.u = u; return x.p; }
__device__ __forceinline__ i16x2 asI16x2(uint32_t u) { union { i16x2 p; uint32_t u; } x; x.u = u; return x.p; }
template <bool SIGNED, int R> __device__ __forceinline__ uint32_t byteRed4(uint32_t a, uint32_t b) {
  if (R == RA_OP_SUM) return ((a & 0x7f7f7f7fu) + (b & 0x7f7f7f7fu)) ^ ((a ^ b) & 0x80808080u);
  if (R == RA_OP_PROD) {
    const u16x2 e = asU16x2(a & 0x00ff00ffu) * asU16x2(b & 0x00ff00ffu);
    const u16x2 o = asU16x2((a >> 8) & 0x00ff00ffu) * asU16x2((b >> 8) & 0x00ff00ffu);
    return (asU32(e) & 0x00ff00ffu) | ((asU32(o) & 0x00ff00ffu) << 8);
  }
  if (SIGNED) {
    const i16x2 ae = asI16x2((a << 8) & 0xff00ff00u), be = asI16x2((b << 8) & 0xff00ff00u);
    const i16x2 ao = asI16x2(a & 0xff00ff00u), bo = asI16x2(b & 0xff00ff00u);
    const i16x2 e = R == RA_OP_MAX ? __builtin_elementwise_max(ae, be) : __builtin_elementwise_min(ae, be);
    const i16x2 o = R == RA_OP_MAX ? __builtin_elementwise_max(ao, bo) : __builtin_elementwise_min(ao, bo);
    return ((asU32(e) >> 8) & 0x00ff00ffu) | (asU32(o) & 0xff00ff00u);
  }
  const u16x2 ae = asU16x2(a & 0x00ff00ffu), be = asU16x2(b & 0x00ff00ffu);
  const u16x2 ao = asU16x2((a >> 8) & 0x00ff00ffu), bo = asU16x2((b >> 8) & 0x00ff00ffu);
  const u16x2 e = R == RA_OP_MAX ? __builtin_elementwise_max(ae, be) : __builtin_elementwise_min(ae, be);
  const u16x2 o = R == RA_OP_MAX ? __builtin_elementwise_max(ao, bo) : __builtin_elementwise_min(ao, bo);
  return asU32(e) | (asU32(o) << 8);
}
// every byte times the byte `s` (FuncPreMulSum on 8-bit integers: the product's low byte)
__device__ __forceinline__ uint32_t byteMul4(uint32_t a, uint32_t s) {
  const u16x2 k = {(uint16_t)s, (uint16_t)s};
  const u16x2 e = asU16x2(a & 0x00ff00ffu) * k, o = asU16x2((a >> 8) & 0x00ff00ffu) * k;
  return (asU32(e) & 0x00ff00ffu) | ((asU32(o) & 0x00ff00ffu) << 8);
}
// every byte divided by n, truncating as C does (FuncSumPostDiv, src/device/reduce_kernel.h:546-573).  |x| * ceil(2^16 / n)
// >> 16 is floor(|x| / n) for |x| <= 255 and n <= 256 (the error term |x| * (n ceil(2^16/n) - 2^16) stays below 2^16).
template <bool SIGNED> __device__ __forceinline__ uint32_t byteDiv4(uint32_t a, uint32_t n) {
  const uint32_t m = (65536u + n - 1u) / n;
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    if (SIGNED) {
      const int x = (int)(int8_t)(a >> (8 * i));
      const uint32_t q = (__umul24((uint32_t)(x < 0 ? -x : x), m)) >> 16;
      r |= ((uint32_t)(x < 0 ? -(int)q : (int)q) & 0xffu) << (8 * i);
    } else {
      r |= ((__umul24((a >> (8 * i)) & 0xffu, m) >> 16) & 0xffu) << (8 * i);
    }
  }
  return r;
}
template <bool SIGNED, int OP> struct BytePackOps {
  static constexpr int R = (OP == RA_OP_PREMULSUM || OP == RA_OP_SUMPOSTDIV) ? RA_OP_SUM : OP;
  static __device__ __forceinline__ u32x4 red(u32x4 a, u32x4 b) {
    u32x4 r;
    r.x = byteRed4<SIGNED, R>(a.x, b.x); r.y = byteRed4<SIGNED, R>(a.y, b.y);
    r.z = byteRed4<SIGNED, R>(a.z, b.z); r.w = byteRed4<SIGNED, R>(a.w, b.w);
    return r;
  }
  static __device__ __forceinline__ u32x4 pre(u32x4 a, uint64_t arg) {
    const uint32_t s = (uint32_t)arg & 0xffu;
    u32x4 r; r.x = byteMul4(a.x, s); r.y = byteMul4(a.y, s); r.z = byteMul4(a.z, s); r.w = byteMul4(a.w, s);
    return r;
  }
  static __device__ __forceinline__ u32x4 post(u32x4 a, uint64_t arg) {
    const uint32_t n = (uint32_t)arg;
    u32x4 r; r.x = byteDiv4<SIGNED>(a.x, n); r.y = byteDiv4<SIGNED>(a.y, n); r.z = byteDiv4<SIGNED>(a.z, n); r.w = byteDiv4<SIGNED>(a.w, n);
    return r;
  }
};
template <int OP> struct PackOps<int8_t, OP> : BytePackOps<true, OP> {};
template <int OP> struct PackOps<uint8_t, OP> : BytePackOps<false, OP> {};

template <typename T, int OP> __device__ __forceinline__ u32x4 packRed(u32x4 a, u32x4 b) { return PackOps<T, OP>::red(a, b); }
template <typename T> __device__ __forceinline__ u32x4 packPre(u32x4 a, uint64_t arg) { return PackOps<T, RA_OP_SUM>::pre(a, arg); }
template <typename T> __device__ __forceinline__ u32x4 packPost(u32x4 a, uint64_t arg) { return PackOps<T, RA_OP_SUM>::post(a, arg); }

// The fold of one pack position over its sources, left to right: src[0] (op) src[1] (op) ... (op) src[ns-1], entries from
// ns on ignored; the source at preIdx is pre-multiplied (FuncPreMulSum), the result post-divided when `post`
// (FuncSumPostDiv).  NMAX is a compile-time bound, so the source array is only ever indexed by constants.
template <typename T, int OP> struct PackFold {
  template <int NMAX>
  static __device__ __forceinline__ u32x4 fold(const u32x4 (&src)[NMAX], int ns, int preIdx, uint64_t opArg, bool post) {
    u32x4 acc = src[0];
    if (OP == RA_OP_PREMULSUM && preIdx == 0) acc = packPre<T>(acc, opArg);
#pragma unroll
    for (int j = 1; j < NMAX; j++)
      if (j < ns) {
        u32x4 x = src[j];
        if (OP == RA_OP_PREMULSUM && preIdx == j) x = packPre<T>(x, opArg);
        acc = packRed<T, OP>(acc, x);
      }
    if (OP == RA_OP_SUMPOSTDIV && post) acc = packPost<T>(acc, opArg);
    return acc;
  }
};
// The 8-bit floats keep the running value as floats that are rounded onto the 8-bit grid after every operation
// (f8_math.h: the same bits as the reference's encode after every pairwise operation, src/device/reduce_kernel.h:267-272,
// at about half the instructions) and encode once, when the fold is over.  One dword (four elements) at a time through
// all its sources, so that no more than eight floats are live.
template <int WM, int OP> struct PackFold<f8_t<WM>, OP> {
  static constexpr int R = (OP == RA_OP_PREMULSUM || OP == RA_OP_SUMPOSTDIV) ? RA_OP_SUM : OP;
  template <int NMAX>
  static __device__ __forceinline__ u32x4 fold(const u32x4 (&src)[NMAX], int ns, int preIdx, uint64_t opArg, bool) {
    if constexpr (R == RA_OP_SUM || R == RA_OP_PROD) {
      // sums and products: the finite-path fold of f8_math.h, NaN codes tracked beside the arithmetic
      u32x4 out;
#pragma unroll
      for (int w = 0; w < 4; w++) {
        uint32_t ws[NMAX];
#pragma unroll
        for (int j = 0; j < NMAX; j++) ws[j] = src[j][w];
        out[w] = f8FoldDword<WM, R == RA_OP_PROD, NMAX>(ws, ns, OP == RA_OP_PREMULSUM ? preIdx : -1, (uint32_t)opArg);
        __builtin_amdgcn_sched_barrier(0);       // one dword after the other: keeps the float temporaries of one dword apart from the next one's
      }
      return out;
    }
    const float k = OP == RA_OP_PREMULSUM ? f8Decode<WM>((uint32_t)opArg) : 1.0f;
    u32x4 out;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      float st[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        st[i] = f8Decode<WM>(src[0][w] >> (8 * i));
        if (OP == RA_OP_PREMULSUM && preIdx == 0) st[i] = f8Quantize<WM>(st[i] * k);
      }
#pragma unroll
      for (int j = 1; j < NMAX; j++)
        if (j < ns) {
#pragma unroll
          for (int i = 0; i < 4; i++) {
            float x = f8Decode<WM>(src[j][w] >> (8 * i));
            if (OP == RA_OP_PREMULSUM && preIdx == j) x = f8Quantize<WM>(x * k);
            const float v = Elem<float, R>::red(st[i], x);
            st[i] = (R == RA_OP_MAX || R == RA_OP_MIN) ? v : f8Quantize<WM>(v);   // a maximum of grid values lies on the grid
          }
        }
      out[w] = f8EncodeOnGrid<WM>(st[0]) | (f8EncodeOnGrid<WM>(st[1]) << 8) | (f8EncodeOnGrid<WM>(st[2]) << 16) |
               (f8EncodeOnGrid<WM>(st[3]) << 24);
      __builtin_amdgcn_sched_barrier(0);       // one dword after the other: keeps the float temporaries of one dword apart from the next one's
    }
    return out;
  }
};

// Global-address-space accesses (never flat): nontemporal 16-byte loads and stores.
typedef __attribute__((address_space(1))) const u32x4 gc_u32x4;
typedef __attribute__((address_space(1))) u32x4 g_u32x4;
__device__ __forceinline__ u32x4 ldPack(const u32x4* p) { return __builtin_nontemporal_load((gc_u32x4*)(uint64_t)p); }
__device__ __forceinline__ void stPack(u32x4* p, u32x4 v) { __builtin_nontemporal_store(v, (g_u32x4*)(uint64_t)p); }
// scalar 64-bit base + 32-bit per-lane byte offset
__device__ __forceinline__ u32x4 ldAt(uint64_t base, uint32_t off) { return __builtin_nontemporal_load((gc_u32x4*)(base + off)); }
__device__ __forceinline__ void stAt(uint64_t base, uint32_t off, u32x4 v) { __builtin_nontemporal_store(v, (g_u32x4*)(base + off)); }
// Write-through stores (system scope: sc0 sc1, streaming: nt).  The bytes are at their destination once the wave's
// vmcnt drains, so no cache write-back is needed before a flag that announces them, and nothing stays dirty in L2.
// Used for every byte that leaves for a peer's window and, inside the collective kernels, for the local receive
// buffer too (an L2 with no dirty lines makes the release fence in front of every flag post cheap).
// The bulk form goes through a buffer descriptor (base in SGPRs, per-lane 32-bit offset): unlike inline asm it is a
// memory instruction the compiler knows, so its s_waitcnt vmcnt(N) counts stay exact and a tile's stores can still be
// in flight while the next tile's loads are awaited.
// The descriptor carries the whole uniform part of the address and the scalar-offset field stays the constant 0.
// With a REGISTER in that field hipcc (ROCm 7.2) treats a 16-byte buffer store as free of the store-data hazard and
// lets the next VALU instruction overwrite the data registers; on gfx950 the store then ships the overwritten dwords
// of some lanes (seen with the byte-wise adds of the 8-bit integer types, whose VALU work sits right behind the
// stores: a few wrong dwords per launch).  With a constant there the compiler spaces the two by itself.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t bufRsrc(uint64_t base) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, -1, 0x00020000);   // raw buffer, no range check
}
#define RA_AUX_WT 19   // sc0 (1) | nt (2) | sc1 (16)
__device__ __forceinline__ void stSysAt(uint64_t base, uint32_t off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, bufRsrc(base), (int)off, 0, RA_AUX_WT);
}
// Single packs of ragged ends: hipcc does not count this store (every step ends with an explicit s_waitcnt vmcnt(0))
// and does not see its data registers either, hence the wait states behind it.
__device__ __forceinline__ void stSys(u32x4* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

// ------------------------------------------------------------------ reduce-copy
// dst[k][i] = post( pre?(src[0][i]) (op) src[1][i] (op) ... ), folded left to right.
// `preIdx` is the one source that gets the pre-multiply (this rank's own input; the
// others arrive already scaled), or -1.
//
// An argument block lives in registers.  That only works while every access to src[] / dst[] uses an index the
// compiler knows: an index computed at run time would force the arrays into per-lane scratch memory, and no kernel of
// this library has a stack frame (see tests/test_stack_size.py).  So the arrays are filled and read in loops that are
// fully unrolled over RA_MAX_SRCS with the run-time count as a predicate (RC_EACH below), never `for (j < ns)`.
struct RcArgs {
  const char* src[RA_MAX_SRCS];
  char* dst[RA_MAX_SRCS];
  int ns, nd, preIdx;
  bool post;
  int nLocal = 0;  // destinations [0, nLocal) are this GPU's own memory, written with plain streaming stores (0, 1, or all
                   // of them); the rest are windows or peers' buffers, written with write-through stores
  uint64_t opArg;
};
// for (J = FIRST; J < COUNT; J++) BODY with J a compile-time constant in every copy of BODY
#define RC_EACH(J, FIRST, COUNT, ...)                          \
  _Pragma("unroll") for (int J = (FIRST); J < RA_MAX_SRCS; J++) \
    if (J < (COUNT)) { __VA_ARGS__ }

// Values that are the same in every lane but reach us through memory or a call (so the
// compiler cannot know it) are moved to scalar registers: addresses then take the
// scalar-base + 32-bit lane-offset form and cost one VGPR per pack instead of two per stream.
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ int uniformInt(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const char* uniformPtr(const char* p) { return (const char*)uniform64((uint64_t)p); }
__device__ __forceinline__ char* uniformPtr(char* p) { return (char*)uniform64((uint64_t)p); }

// Packs [first, npacks) in tiles of nthr*U packs; tile t starts at first + t*tileStep.
//
// LOC: how many leading destinations are local memory written with plain streaming stores (a.nLocal: 0 or 1 here; the
// receive buffer, when it is a destination, comes first); every other destination is a window slot or a peer's buffer
// and is written with write-through stores (descriptor base = destination + tile offset, per-lane offset below one tile).
//
// NS > 0 (source and destination counts known at compile time): software-pipelined.  All NS*U loads of a tile are
// issued together, and the loads of the NEXT tile are issued before this tile's stores, so a wave waits for a tile's
// loads with the previous tile's stores and the next tile's loads still in flight: store acknowledgements never stall
// the loop and a wave keeps between NS*U and 2*NS*U 16-byte loads in flight.  Two register sets swap roles from tile
// to tile.  The waits are the compiler's own s_waitcnt vmcnt(N); they are exact because every path through the loop
// issues the same number of memory instructions: the counts are template arguments, the tile behind the last one is
// "loaded" from one already-read line instead of being skipped, and the first tile is peeled so that the loop is entered
// in its steady state.  !PIPE (the local kernels, whose workgroups mostly own a single tile): one tile at a time.
// NS == 0: any counts, one tile at a time.
template <typename T, int OP, int LOC, int NS, int U, int ND, bool PIPE>
__device__ __forceinline__ void rcPacks(const RcArgs& a, size_t npacksIn, size_t firstIn, size_t tileStepIn, int tid, int nthr) {
  constexpr int NSA = NS > 0 ? NS : 1;
  constexpr int NDA = NS > 0 ? ND : 1;
  static_assert(NS == 0 || ND > 0, "pipelined form needs both counts");
  const size_t npacks = uniform64(npacksIn), first = uniform64(firstIn), tileStep = uniform64(tileStepIn);
  const int ns = __builtin_amdgcn_readfirstlane(a.ns);
  const int nd = NS > 0 ? ND : __builtin_amdgcn_readfirstlane(a.nd);
  const int preIdx = __builtin_amdgcn_readfirstlane(a.preIdx);
  const uint64_t opArg = uniform64(a.opArg);
  const bool post = __builtin_amdgcn_readfirstlane((int)a.post) != 0;
  const uint32_t nthrU = (uint32_t)__builtin_amdgcn_readfirstlane(nthr);
  // this thread's byte offset inside a row of nthr packs: a copy the compiler cannot trace back to the lane number, so what
  // is derived from it is computed in this call instead of once per kernel and kept in registers across every other loop
  uint32_t lane16 = (uint32_t)tid * 16u;
  asm volatile("" : "+v"(lane16));
  const uint32_t row = nthrU * 16u;                     // bytes between a thread's consecutive packs
  const size_t tile = (size_t)nthrU * U;
  size_t base = first;

  const int nLocal = NS > 0 ? LOC : __builtin_amdgcn_readfirstlane(a.nLocal);
  auto storeTo = [&](uint64_t dk, bool wt, const u32x4 (&acc)[U], size_t b16) {
    if (wt) {
#pragma unroll
      for (int u = 0; u < U; u++) stSysAt(dk + b16, lane16 + (uint32_t)u * row, acc[u]);
    } else {
#pragma unroll
      for (int u = 0; u < U; u++) stAt(dk + b16, lane16 + (uint32_t)u * row, acc[u]);
    }
  };

  if (NS > 0) {
    // every stream's base goes to scalar registers once: the argument block itself lives in scratch memory, and a
    // scratch load inside the loop would be one more memory instruction to wait for
    uint64_t s[NSA], d[NDA];
#pragma unroll
    for (int j = 0; j < NSA; j++) s[j] = (uint64_t)uniformPtr(a.src[j]);
#pragma unroll
    for (int k = 0; k < NDA; k++) d[k] = (uint64_t)uniformPtr(a.dst[k]);
    u32x4 va[NSA][U], vb[NSA][U];
    // the tile at `at`; !real: the same number of loads, all lanes from the first pack of the tile at `at` (a line this
    // workgroup has just read: no branch, no traffic to speak of)
    auto load = [&](u32x4 (&v)[NSA][U], size_t at, bool real) {
#pragma unroll
      for (int u = 0; u < U; u++)
#pragma unroll
        for (int j = 0; j < NSA; j++) v[j][u] = ldAt(s[j] + at * 16, real ? lane16 + (uint32_t)u * row : 0u);
    };
    // one pack position at a time through all its sources (loads are issued in the same order, so the first fold
    // waits for the first NS loads only)
    auto fold = [&](u32x4 (&cur)[NSA][U], size_t at) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        u32x4 col[NSA];
#pragma unroll
        for (int j = 0; j < NSA; j++) col[j] = cur[j][u];
        cur[0][u] = PackFold<T, OP>::template fold<NSA>(col, NSA, preIdx, opArg, post);
      }
#pragma unroll
      for (int k = 0; k < NDA; k++) storeTo(d[k], k >= LOC, cur[0], at * 16);
    };
    if (!PIPE) {
      for (; base + tile <= npacks; base += tileStep) { load(va, base, true); fold(va, base); }
    } else if (base + tile <= npacks) {
      load(va, base, true);
      size_t nxt = base + tileStep;
      bool more = nxt + tile <= npacks;
      load(vb, more ? nxt : base, more);
      fold(va, base);
      base = nxt;
      while (more) {                                   // the tile at `base` is in vb
        nxt = base + tileStep;
        more = nxt + tile <= npacks;
        load(va, more ? nxt : base, more);
        fold(vb, base);
        base = nxt;
        if (!more) break;
        nxt = base + tileStep;
        more = nxt + tile <= npacks;
        load(vb, more ? nxt : base, more);
        fold(va, base);
        base = nxt;
      }
    }
  } else {
    for (; base + tile <= npacks; base += tileStep) {
      const size_t b16 = base * 16;
      u32x4 t[U][RA_MAX_SRCS];
#pragma unroll
      for (int u = 0; u < U; u++) t[u][0] = ldAt((uint64_t)uniformPtr(a.src[0]) + b16, lane16 + (uint32_t)u * row);
      RC_EACH(j, 1, ns,
        const uint64_t sj = (uint64_t)uniformPtr(a.src[j]) + b16;
        _Pragma("unroll") for (int u = 0; u < U; u++) t[u][j] = ldAt(sj, lane16 + (uint32_t)u * row);
      )
      u32x4 acc[U];
#pragma unroll
      for (int u = 0; u < U; u++) acc[u] = PackFold<T, OP>::template fold<RA_MAX_SRCS>(t[u], ns, preIdx, opArg, post);
      RC_EACH(k, 0, nd, storeTo((uint64_t)uniformPtr(a.dst[k]), k >= nLocal, acc, b16);)
    }
  }
  // Ragged end: the one tile that is not full, one pack per thread per pass.  Only the
  // workgroup whose walk reaches it handles it.
  if (base < npacks) {
    for (size_t i = base + tid; i < npacks; i += nthrU) {
      u32x4 t[RA_MAX_SRCS];
      t[0] = ldPack((const u32x4*)uniformPtr(a.src[0]) + i);
      RC_EACH(j, 1, ns, t[j] = ldPack((const u32x4*)uniformPtr(a.src[j]) + i);)
      const u32x4 acc = PackFold<T, OP>::template fold<RA_MAX_SRCS>(t, ns, preIdx, opArg, post);
      RC_EACH(k, 0, nd,
        if (k >= nLocal) stSys((u32x4*)uniformPtr(a.dst[k]) + i, acc);
        else stPack((u32x4*)uniformPtr(a.dst[k]) + i, acc);
      )
    }
  }
}

// Four and eight sources inside the collective kernels: one source after the other.  A tile is 4 packs per thread of ONE
// source (64 KiB contiguous with 1024 threads), added into an accumulator tile while the next source's tile is in flight;
// the destinations are written when all sources are in.  Same left-to-right association as the loop above.  Why: a
// collective's workgroups each walk their own step, megabytes apart, so a launch keeps (workgroups x streams) DRAM
// pages open at once; with one pack of every source per thread (16 KiB bursts per source) the 8-source fold of 240
// workgroups measured 4.4-5.0 TB/s standalone, with 64 KiB bursts 5.1-5.9 TB/s (tools/mix_bench.hip).  (The local
// kernels keep the loop above: their workgroups take consecutive tiles, which is sequential for the DRAM anyway.)
// The bits of 1 in T: multiplying by it is the identity in every element type here, so a fold can scale every source,
// the ones that need no scaling by 1, instead of choosing per pack between a scaled and an unscaled copy.
template <typename T> __device__ __forceinline__ uint64_t oneBits() {
  union { uint64_t u; T t; } x; x.u = 0; x.t = (T)1; return x.u;
}
template <> __device__ __forceinline__ uint64_t oneBits<bf16_t>() { return 0x3f80u; }
template <> __device__ __forceinline__ uint64_t oneBits<f8e4m3_t>() { return 0; }     // (the 8-bit floats never take this path)
template <> __device__ __forceinline__ uint64_t oneBits<f8e5m2_t>() { return 0; }

// A uniform address pinned in scalar registers: the compiler may not fold it into per-lane 64-bit arithmetic, so the access
// keeps the scalar-base + 32-bit lane-offset form.
__device__ __forceinline__ uint64_t scalarBase(uint64_t v) {
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  asm volatile("" : "+s"(lo), "+s"(hi));
  return ((uint64_t)hi << 32) | lo;
}

template <typename T, int OP, int LOC, int NS, int ND>
__device__ __forceinline__ void rcPacksSerial(const RcArgs& a, size_t npacksIn, size_t firstIn, size_t tileStepIn, int tid, int nthr) {
  constexpr int U = 4;
  static_assert(NS >= 3, "two sources are folded pack by pack");
  const size_t npacks = uniform64(npacksIn), first = uniform64(firstIn), tileStep = uniform64(tileStepIn);
  const int preIdx = __builtin_amdgcn_readfirstlane(a.preIdx);
  const uint64_t opArg = uniform64(a.opArg);
  const bool post = __builtin_amdgcn_readfirstlane((int)a.post) != 0;
  const uint32_t nthrU = (uint32_t)__builtin_amdgcn_readfirstlane(nthr);
  const uint32_t lane16 = (uint32_t)tid * 16u;
  const uint32_t row = nthrU * 16u;
  const size_t tile = (size_t)nthrU * U;
  uint64_t s[NS], d[ND];
#pragma unroll
  for (int j = 0; j < NS; j++) s[j] = (uint64_t)uniformPtr(a.src[j]);
#pragma unroll
  for (int k = 0; k < ND; k++) d[k] = (uint64_t)uniformPtr(a.dst[k]);
  size_t base = first;
  for (; base + tile <= npacks; base += tileStep) {
    const size_t b16 = base * 16;
    // a copy of the lane offset the compiler cannot trace: address arithmetic stays inside the loop (scalar base + this one
    // 32-bit offset) instead of being kept as per-lane 64-bit pointers across the whole kernel
    uint32_t off = lane16;
    asm volatile("" : "+v"(off));
    // (likewise the index of the scaled source: the per-source choice of the factor is then made here, tile by tile, in two
    // scalar selects, instead of once for all sources in front of the loop with every factor kept in registers across it)
    int preJ = preIdx;
    asm volatile("" : "+s"(preJ));
    u32x4 acc[U], x[U], y[U];
    // (the row goes into the scalar base: one lane offset.  `after`: the loads may not be issued before that value exists —
    // an empty asm ties the lane offset to it — so that never more than three tiles are live)
    auto load = [&](u32x4 (&v)[U], int j, const u32x4& after) {
      uint32_t o = off;
      asm volatile("" : "+v"(o) : "v"(after));
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = ldAt(scalarBase(s[j] + b16 + (size_t)u * row), o);
    };
    auto fold = [&](const u32x4 (&v)[U], int j) {
      // the factor of this source, chosen and kept in scalar registers (1 where the source needs no scaling)
      const uint64_t factor = OP == RA_OP_PREMULSUM ? scalarBase(preJ == j ? opArg : oneBits<T>()) : 0;
#pragma unroll
      for (int u = 0; u < U; u++) {
        u32x4 t = v[u];
        if (OP == RA_OP_PREMULSUM) t = packPre<T>(t, factor);
        acc[u] = packRed<T, OP>(acc[u], t);
      }
    };
    const u32x4 none = {0, 0, 0, 0};
    load(acc, 0, none);
    load(x, 1, none);
    if (OP == RA_OP_PREMULSUM) {
#pragma unroll
      for (int u = 0; u < U; u++) acc[u] = packPre<T>(acc[u], scalarBase(preJ == 0 ? opArg : oneBits<T>()));
    }
#pragma unroll
    for (int j = 1; j < NS; j += 2) {
      if (j + 1 < NS) load(y, j + 1, j > 1 ? acc[U - 1] : none);      // behind the fold that freed y
      fold(x, j);
      if (j + 1 < NS) {
        if (j + 2 < NS) load(x, j + 2, acc[U - 1]);                     // behind the fold that freed x
        fold(y, j + 1);
      }
    }
    if (OP == RA_OP_SUMPOSTDIV && post) {
#pragma unroll
      for (int u = 0; u < U; u++) acc[u] = packPost<T>(acc[u], opArg);
    }
#pragma unroll
    for (int k = 0; k < ND; k++) {
      if (k >= LOC) {
#pragma unroll
        for (int u = 0; u < U; u++) stSysAt(scalarBase(d[k] + b16 + (size_t)u * row), off, acc[u]);
      } else {
#pragma unroll
        for (int u = 0; u < U; u++) stAt(scalarBase(d[k] + b16 + (size_t)u * row), off, acc[u]);
      }
    }
  }
  // ragged end: the one tile that is not full, one pack per thread per pass
  if (base < npacks) {
    for (size_t i = base + tid; i < npacks; i += nthrU) {
      u32x4 t[NS];
#pragma unroll
      for (int j = 0; j < NS; j++) t[j] = ldPack((const u32x4*)s[j] + i);
      const u32x4 r = PackFold<T, OP>::template fold<NS>(t, NS, preIdx, opArg, post);
#pragma unroll
      for (int k = 0; k < ND; k++) {
        if (k >= LOC) stSys((u32x4*)d[k] + i, r);
        else stPack((u32x4*)d[k] + i, r);
      }
    }
  }
}

// One element into a peer's window, system-scope write-through (see stSysAt).
template <typename T> __device__ __forceinline__ void stSysElem(T* p, T v) {
  if constexpr (sizeof(T) == 1) { union { T t; uint8_t u; } x; x.t = v; const uint32_t w = x.u;
    asm volatile("global_store_byte %0, %1, off sc0 sc1" : : "v"(p), "v"(w) : "memory"); }
  else if constexpr (sizeof(T) == 2) { union { T t; uint16_t u; } x; x.t = v; const uint32_t w = x.u;
    asm volatile("global_store_short %0, %1, off sc0 sc1" : : "v"(p), "v"(w) : "memory"); }
  else if constexpr (sizeof(T) == 4) { union { T t; uint32_t u; } x; x.t = v;
    asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(p), "v"(x.u) : "memory"); }
  else { union { T t; uint64_t u; } x; x.t = v;
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(p), "v"(x.u) : "memory"); }
}

// Elements [first, nelem) one at a time (unaligned pointers, or the sub-pack tail).
template <typename T, int OP>
__device__ __forceinline__ void rcElems(const RcArgs& a, size_t nelem, size_t first, size_t step, int tid) {
  constexpr int R = (OP == RA_OP_PREMULSUM || OP == RA_OP_SUMPOSTDIV) ? RA_OP_SUM : OP;
  const int ns = uniformInt(a.ns), nd = uniformInt(a.nd), nLocal = uniformInt(a.nLocal), preIdx = uniformInt(a.preIdx);
  const uint64_t opArg = uniform64(a.opArg);
  for (size_t i = first + tid; i < nelem; i += step) {
    T acc = ((const T*)a.src[0])[i];
    if (OP == RA_OP_PREMULSUM && preIdx == 0) acc = Scale<T>::pre(acc, opArg);
    RC_EACH(j, 1, ns,
      T t = ((const T*)a.src[j])[i];
      if (OP == RA_OP_PREMULSUM && preIdx == j) t = Scale<T>::pre(t, opArg);
      acc = Elem<T, R>::red(acc, t);
    )
    if (OP == RA_OP_SUMPOSTDIV && a.post) acc = Scale<T>::post(acc, opArg);
    RC_EACH(k, 0, nd,
      if (k >= nLocal) stSysElem<T>((T*)a.dst[k] + i, acc);
      else ((T*)a.dst[k])[i] = acc;
    )
  }
}

__device__ __forceinline__ bool rcAligned(const RcArgs& a) {
  uintptr_t m = 0;
  RC_EACH(j, 0, a.ns, m |= (uintptr_t)a.src[j];)
  RC_EACH(k, 0, a.nd, m |= (uintptr_t)a.dst[k];)
  return (uniform64(m) & 15) == 0;
}

// One (source count, destination count) case: full tiles of nthr*U packs are dealt round-robin to the `nparts`
// workgroups taking part; the ragged last tile goes to whoever owns that tile index.
template <typename T, int OP, int LOC, int NS, int U, int ND, bool PIPE = true>
__device__ __forceinline__ void rcCase(const RcArgs& a, size_t npacks, int tid, int nthr, int part, int nparts) {
  const size_t tile = (size_t)nthr * U;
  const size_t nfull = npacks / tile;
  const bool ownsRagged = (nfull % nparts) == (size_t)part;
  const size_t lim = ownsRagged ? npacks : nfull * tile;
  rcPacks<T, OP, LOC, NS, U, ND, PIPE>(a, lim, (size_t)part * tile, (size_t)nparts * tile, tid, nthr);
}

template <typename T, int OP, int LOC, int NS, int ND>
__device__ __forceinline__ void rcCaseSerial(const RcArgs& a, size_t npacks, int tid, int nthr, int part, int nparts) {
  const size_t tile = (size_t)nthr * 4;
  const size_t nfull = npacks / tile;
  const bool ownsRagged = (nfull % nparts) == (size_t)part;
  const size_t lim = ownsRagged ? npacks : nfull * tile;
  rcPacksSerial<T, OP, LOC, NS, ND>(a, lim, (size_t)part * tile, (size_t)nparts * tile, tid, nthr);
}

// Work-shared reduce-copy of `nelem` elements.  `part` of `nparts` workgroups take part
// (the collectives use one workgroup: part 0 of 1; the local kernels use the whole grid).
// The (sources, destinations, local destinations) shapes the schedules use at 2, 4 and 8 ranks get the pipelined loop
// (sources x packs per thread = 8 loads, 128 B, per tile and thread in every case); every other shape takes the plain
// loop.  Everything is inlined into its caller, so a call site names the family of shapes it can produce (SH) and only
// those loops are built there.
enum {
  RC_LOCAL,      // local kernels: 1, 2, 4 or 8 sources into one local destination
  RC_SEND,       // one source into one remote destination (scatter with a pre-multiply, ring seed, chain leaf, reduce push)
  RC_HOP,        // local + received into one remote destination (ring and chain middle hops)
  RC_HOP_KEEP,   // local + received into the receive buffer and one remote destination (ring: own piece; chain: root)
  RC_HOP_END,    // local + received into the receive buffer (ring reduce-scatter: last hop)
  RC_FOLD_ALL,   // n sources into the receive buffer and n - 1 remote destinations (the all-pairs reducer)
  RC_FOLD_ONE,   // n sources into the receive buffer (reduce-scatter and reduce on the all-pairs links, one exchange)
};
template <int SH> __device__ __forceinline__ constexpr bool rcHas(int ns, int nd, int loc) {
  return SH == RC_LOCAL      ? (nd == 1 && loc == 1 && (ns == 1 || ns == 2 || ns == 4 || ns == 8))
         : SH == RC_SEND     ? (ns == 1 && nd == 1 && loc == 0)
         : SH == RC_HOP      ? (ns == 2 && nd == 1 && loc == 0)
         : SH == RC_HOP_KEEP ? (ns == 2 && nd == 2 && loc == 1)
         : SH == RC_HOP_END  ? (ns == 2 && nd == 1 && loc == 1)
         : SH == RC_FOLD_ALL ? (loc == 1 && ns == nd && (ns == 2 || ns == 4 || ns == 8))
                             : (loc == 1 && nd == 1 && (ns == 2 || ns == 4 || ns == 8));
}
template <typename T, int OP, bool PIPE, int SH>
__device__ __forceinline__ void reduceCopy(const RcArgs& a, size_t nelem, int tid, int nthr, int part, int nparts) {
  if (nelem == 0 || a.nd == 0) return;
  if (rcAligned(a)) {
    const size_t npacks = nelem * sizeof(T) / 16;
    const int shape = __builtin_amdgcn_readfirstlane(a.ns * 256 + a.nd * 16 + a.nLocal);
    bool done = false;
    // (the 8-bit floats are bound by their conversions, not by memory: they take the plain loop, one pack per thread, whose
    // code is a fraction of the pipelined loops' and leaves the registers to the float temporaries; only their sums and
    // products of one or two sources take the multi-pack loops, one tile at a time, which is as much as stays in registers)
#define RC_TRY(NS, U, ND, LOC)                                                                  \
    if constexpr (rcHas<SH>(NS, ND, LOC) && (!IsF8<T>::value || (NS <= 2 && (OP == RA_OP_SUM || OP == RA_OP_PROD))))  \
      if (!done && shape == NS * 256 + ND * 16 + LOC) {                                           \
        if constexpr (NS >= 4 && PIPE) rcCaseSerial<T, OP, LOC, NS, ND>(a, npacks, tid, nthr, part, nparts);             \
        else rcCase<T, OP, LOC, NS, U, ND, PIPE && !IsF8<T>::value>(a, npacks, tid, nthr, part, nparts);                  \
        done = true;                                                                              \
      }
    RC_TRY(1, 8, 1, 0) RC_TRY(1, 8, 1, 1) RC_TRY(2, 4, 1, 0) RC_TRY(2, 4, 1, 1) RC_TRY(4, 2, 1, 1) RC_TRY(8, 1, 1, 1)
    RC_TRY(2, 4, 2, 1) RC_TRY(4, 2, 4, 1) RC_TRY(8, 1, 8, 1)
#undef RC_TRY
    if (!done) rcCase<T, OP, 0, 0, IsF8<T>::value ? 1 : 2, 0, PIPE>(a, npacks, tid, nthr, part, nparts);
    const size_t fin = npacks * (16 / sizeof(T));
    if (fin < nelem && part == 0) rcElems<T, OP>(a, nelem, fin, nthr, tid);
  } else {
    rcElems<T, OP>(a, nelem, (size_t)part * nthr, (size_t)nparts * nthr, tid);
  }
}

// The entry point the collective schedules call: one workgroup, pipelined loops.
template <typename T, int OP, int SH>
__device__ __forceinline__ void rcWG(const RcArgs& a, size_t nelem, int tid, int nthr) {
  reduceCopy<T, OP, true, SH>(a, nelem, tid, nthr, 0, 1);
}
// One source to a.nd destinations, any element type.  Call sites name their family of shapes:
enum {
  CP_END,      // into the receive buffer only (last hop of a ring or chain, fallback copies)
  CP_FWD,      // into the receive buffer and on to one remote destination (ring and chain forwarding hops)
  CP_SPREAD,   // into 1, 2, 3, 4, 7 or 8 destinations of which the first may be local (all-gather, broadcast, one exchange)
};
template <int CS, int LOC>
static __device__ __forceinline__ void copyShapes(const RcArgs& a, size_t np, int tid, int nthr) {
  const int nd = __builtin_amdgcn_readfirstlane(a.nd);
  bool done = false;
#define CP_TRY(ND)                                                                                              \
  if constexpr (CS == CP_SPREAD || (CS == CP_END && ND == 1 && LOC == 1) || (CS == CP_FWD && ND == 2 && LOC == 1)) \
    if (!done && nd == ND) { rcCase<int8_t, RA_OP_SUM, LOC, 1, 8, ND>(a, np, tid, nthr, 0, 1); done = true; }
  CP_TRY(1) CP_TRY(2) CP_TRY(3) CP_TRY(4) CP_TRY(7) CP_TRY(8)
#undef CP_TRY
  if (!done) rcCase<int8_t, RA_OP_SUM, 0, 0, 2, 0>(a, np, tid, nthr, 0, 1);
}
template <int CS>
static __device__ __forceinline__ void copyWG(const RcArgs& a, size_t nbytes, int tid, int nthr) {
  if (nbytes == 0 || a.nd == 0) return;
  if (rcAligned(a)) {
    const int nLocal = __builtin_amdgcn_readfirstlane(a.nLocal);
    if (nLocal == 0 && CS == CP_SPREAD) copyShapes<CS, 0>(a, nbytes / 16, tid, nthr);
    else if (nLocal == 1) copyShapes<CS, 1>(a, nbytes / 16, tid, nthr);
    else rcCase<int8_t, RA_OP_SUM, 0, 0, 2, 0>(a, nbytes / 16, tid, nthr, 0, 1);
    const size_t fin = nbytes & ~(size_t)15;
    if (fin < nbytes) rcElems<int8_t, RA_OP_SUM>(a, nbytes, fin, nthr, tid);
  } else {
    rcElems<int8_t, RA_OP_SUM>(a, nbytes, 0, nthr, tid);
  }
}

// Up to RA_MAX_SRCS independent copies src[j] -> dst[j] of `nbytes` each in ONE pass: every thread keeps
// pairs x rows = up to 8 16-byte loads of a span in flight (ROWS consecutive 16 KiB rows of every pair), so the n-1
// pieces a scatter or gather role moves per round share a single load-latency ramp instead of paying it once per
// piece, and a single pair (2 ranks) still has eight loads in flight per thread.  Pipelined like rcPacks: the next
// span's loads are issued before this span's stores, two register sets, first span peeled, NP exact.
template <bool WT, int NP, int ROWS>
__device__ __forceinline__ size_t pairRows(const uint64_t* sIn, const uint64_t* dIn, size_t nbytes, uint32_t lane16, size_t row) {
  const size_t span = row * ROWS;
  const size_t full = nbytes / span * span;
  if (full == 0) return 0;
  uint64_t s[NP], d[NP];
#pragma unroll
  for (int j = 0; j < NP; j++) { s[j] = sIn[j]; d[j] = dIn[j]; }
  u32x4 va[NP][ROWS], vb[NP][ROWS];
  auto load = [&](u32x4 (&v)[NP][ROWS], size_t b, bool real) {
#pragma unroll
    for (int j = 0; j < NP; j++)
#pragma unroll
      for (int r = 0; r < ROWS; r++) v[j][r] = ldAt(s[j] + b + (real ? (size_t)r * row : 0), real ? lane16 : 0u);
  };
  auto store = [&](u32x4 (&v)[NP][ROWS], size_t b) {
#pragma unroll
    for (int j = 0; j < NP; j++) {
      if (WT) {
#pragma unroll
        for (int r = 0; r < ROWS; r++) stSysAt(d[j] + b + (size_t)r * row, lane16, v[j][r]);
      } else {
#pragma unroll
        for (int r = 0; r < ROWS; r++) stAt(d[j] + b + (size_t)r * row, lane16, v[j][r]);
      }
    }
  };
  size_t b = 0;
  load(va, 0, true);
  bool more = span < full;
  load(vb, more ? span : 0, more);
  store(va, 0);
  b = span;
  while (more) {                                       // the span at b is in vb
    more = b + span < full;
    load(va, more ? b + span : b, more);
    store(vb, b);
    b += span;
    if (!more) break;
    more = b + span < full;
    load(vb, more ? b + span : b, more);
    store(va, b);
    b += span;
  }
  return full;
}

// WT: the destinations are windows (write-through stores: the scatter role); otherwise local memory (the gather role).
template <bool WT>
static __device__ __forceinline__ void pairCopyWG(const RcArgs& a, size_t nbytesIn, int tid, int nthr) {
  const int np = __builtin_amdgcn_readfirstlane(a.ns);
  const size_t nbytes = uniform64(nbytesIn);
  if (nbytes == 0 || np == 0) return;
  uintptr_t m = 0;
  RC_EACH(j, 0, np, m |= (uintptr_t)a.src[j] | (uintptr_t)a.dst[j];)
  const bool aligned = (uniform64(m) & 15) == 0;
  size_t done = 0;
  if (aligned) {
    const uint32_t nthrU = (uint32_t)__builtin_amdgcn_readfirstlane(nthr);
    uint32_t lane16 = (uint32_t)tid * 16u;
    asm volatile("" : "+v"(lane16));                    // (see rcPacks)
    const size_t row = (size_t)nthrU * 16;
    uint64_t s[RA_MAX_SRCS], d[RA_MAX_SRCS];
#pragma unroll
    for (int j = 0; j < RA_MAX_SRCS; j++) {
      s[j] = uniform64((uint64_t)(j < np ? a.src[j] : a.src[0]));
      d[j] = uniform64((uint64_t)(j < np ? a.dst[j] : a.dst[0]));
    }
    // One or two pairs: spans of rows x pairs = 8 loads per thread, then what is left one row of every pair at a time.
    // Three pairs and more: one pair after the other in spans of 8 rows (128 KiB contiguous per stream) — a launch's
    // workgroups each walk their own step, so the DRAM sees (workgroups x streams) open pages at once, and long bursts
    // per stream beat sharing one load-latency ramp between the pairs (16 KiB bursts of 7 pairs: see rcPacksSerial).
    if (np <= 2) {
      switch (np) {
        case 1: done = pairRows<WT, 1, 8>(s, d, nbytes, lane16, row); break;
        default: done = pairRows<WT, 2, 4>(s, d, nbytes, lane16, row); break;
      }
#pragma unroll
      for (int j = 0; j < RA_MAX_SRCS; j++) { s[j] += done; d[j] += done; }
      switch (np) {
        case 1: done += pairRows<WT, 1, 1>(s, d, nbytes - done, lane16, row); break;
        default: done += pairRows<WT, 2, 1>(s, d, nbytes - done, lane16, row); break;
      }
    } else {
      // (one copy of the loops: the pair is picked with selects over constant indices, not by indexing the arrays)
      size_t big = 0, rest = 0;
#pragma nounroll
      for (int j = 0; j < np; j++) {
        uint64_t sj[1] = {s[0]}, dj[1] = {d[0]};
#pragma unroll
        for (int q = 1; q < RA_MAX_SRCS; q++)
          if (q == j) { sj[0] = s[q]; dj[0] = d[q]; }
        big = pairRows<WT, 1, 8>(sj, dj, nbytes, lane16, row);
        sj[0] += big; dj[0] += big;
        rest = pairRows<WT, 1, 1>(sj, dj, nbytes - big, lane16, row);
      }
      done = big + rest;
    }
  }
  // what is left (ragged end, or everything when a pointer is unaligned): pair by pair, bytes
  if (done < nbytes) {
    RC_EACH(j, 0, np,
      const int8_t* sj = (const int8_t*)uniformPtr(a.src[j]);
      int8_t* dj = (int8_t*)uniformPtr(a.dst[j]);
      for (size_t i = done + tid; i < nbytes; i += nthr) {
        if (WT) stSysElem<int8_t>(dj + i, sj[i]);
        else dj[i] = sj[i];
      }
    )
  }
}

// A pre-multiply scalar that lives in device memory (ncclScalarDevice) is read when the kernel runs, never on the
// host at enqueue time: a captured launch replays with whatever the scalar holds then (the reference dereferences
// it on the device as well, src/device/reduce_kernel.h:364-457 with scalarArgIsPtr).
// System-scope loads: the scalar may have been rewritten by a copy engine since an earlier launch read it, and a launch
// replayed from a graph does not necessarily start by dropping cached lines.
template <typename T> __device__ __forceinline__ uint64_t loadScalarArg(const void* p) {
  uint64_t u;
  if constexpr (sizeof(T) == 1) u = __hip_atomic_load((const uint8_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else if constexpr (sizeof(T) == 2) u = __hip_atomic_load((const uint16_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else if constexpr (sizeof(T) == 4) u = __hip_atomic_load((const uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else u = __hip_atomic_load((const uint64_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return uniform64(u);
}

// ------------------------------------------------------------------ local kernels
// One rank (src/device/onerank.cu) and the exported N-source reduce-copy.
template <typename T, int OP>
__global__ __launch_bounds__(1024) void raLocalKernel(RcArgs k, size_t nelem, const void* opArgPtr) {
  RcArgs a;                        // a copy with constant indices: the kernel-argument block itself is read-only memory
#pragma unroll
  for (int j = 0; j < RA_MAX_SRCS; j++) { a.src[j] = k.src[j]; a.dst[j] = k.dst[j]; }
  a.ns = k.ns; a.nd = k.nd; a.preIdx = k.preIdx; a.post = k.post; a.nLocal = k.nLocal;
  a.opArg = (OP == RA_OP_PREMULSUM && opArgPtr) ? loadScalarArg<T>(opArgPtr) : k.opArg;
  reduceCopy<T, OP, false, RC_LOCAL>(a, nelem, threadIdx.x, blockDim.x, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------ inter-GPU protocol
__device__ __forceinline__ uint64_t ldFlag(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void stFlag(uint64_t* p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void drainVmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Waits until *f >= want.  Every wait is bounded and watches the communicator's abort request; once one has given up
// (*aborted, a flag of the workgroup), the later ones return at once.
typedef __attribute__((address_space(3))) int LdsInt;
static __device__ __forceinline__ void raSpin(const RaDevComm* comm, LdsInt* aborted, const uint64_t* f, uint64_t want) {
  if (*aborted) return;
  const uint64_t t0 = wall_clock64();
  uint32_t spins = 0;
  while (ldFlag(f) < want) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 0x3ff) == 0) {
      const bool asked = __hip_atomic_load(comm->abortFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
      const bool late = wall_clock64() - t0 > comm->timeoutTicks;
      if (asked || late || *aborted) {
        *aborted = 1;
        uint32_t expect = RA_ERR_NONE;
        __hip_atomic_compare_exchange_strong(comm->errorFlag, &expect, asked ? RA_ERR_ABORTED : RA_ERR_TIMEOUT,
                                             __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
      }
    }
  }
}

struct ChanShared {
  uint64_t sendStep[RA_MAX_RANKS][RA_NKINDS];
  uint64_t recvStep[RA_MAX_RANKS][RA_NKINDS];
  uint32_t sendDirty[RA_NKINDS];
  uint32_t recvDirty[RA_NKINDS];
  int aborted;
  // Registered receive buffers: the word each peer sent with the first step of its FIFO in this launch, the word
  // this workgroup sends with its own first post, and where a peer's receive buffer is mapped here (0: use the FIFO).
  uint64_t auxIn[RA_MAX_RANKS];
  uint64_t auxOut[RA_MAX_RANKS];
  uint64_t directBase[RA_MAX_RANKS];
  // tuneFlags 32: ticks thread 0 spent waiting / in the body / draining / posting, steps; stamps of the step under way
  uint64_t tk[5], ts[4];
};
// The per-channel state lives in LDS and is only ever reached through LDS pointers (never generic ones), so that every
// access is a ds_ instruction and none can fall into another aperture.
typedef __attribute__((address_space(3))) ChanShared LdsChan;
#define RA_LDS_CHAN(var) ((LdsChan*)&(var))

// Everything a workgroup knows about the channel it is serving.  All members are the same in every lane; the
// kernels are built without calls, so the object never leaves the registers.
struct Ctx {
  const RaDevComm* comm;
  LdsChan* sh;
  int ch, tid, nthr, rank, nRanks;
  uint32_t stepBytes;
  int auxRecv = 0, auxSend = 0;   // this step carries the once-per-launch words (first step of a workgroup only)

  __device__ __forceinline__ uint64_t* flag(int owner, int which, int peer, int kind) const {
    return (uint64_t*)(comm->win[owner] + raFlagOffset(comm->lay, which, ch, peer, kind));
  }
  // Slot I fill next on the FIFO towards `peer` (it lives in peer's window).
  __device__ __forceinline__ char* sendSlot(int peer, int kind) const {
    return uniformPtr(comm->win[peer] + raDataOffset(comm->lay, ch, peer, rank, kind) +
                      (sh->sendStep[peer][kind] % comm->lay.nSlots) * (size_t)comm->lay.slotBytes);
  }
  // Slot I drain next on the FIFO from `peer` (it lives in my window).
  __device__ __forceinline__ const char* recvSlot(int peer, int kind) const {
    return uniformPtr((const char*)comm->win[rank] + raDataOffset(comm->lay, ch, rank, peer, kind) +
                      (sh->recvStep[peer][kind] % comm->lay.nSlots) * (size_t)comm->lay.slotBytes);
  }

  // Loads the channel's step counters into its LDS block.  The caller puts a barrier behind the last init.
  static __device__ __forceinline__ void loadChannel(const RaDevComm* c, LdsChan* s, int channel) {
    const int tid = threadIdx.x;
    const int n = c->nRanks * RA_NKINDS;
    if (tid < n) {
      const size_t ix = (size_t)channel * n + tid;
      s->sendStep[tid / RA_NKINDS][tid % RA_NKINDS] = c->sendStep[ix];
      s->recvStep[tid / RA_NKINDS][tid % RA_NKINDS] = c->recvStep[ix];
    }
    if (tid == 0) {
      s->aborted = 0;
      for (int k = 0; k < RA_NKINDS; k++) { s->sendDirty[k] = 0; s->recvDirty[k] = 0; }
    }
    if (tid < RA_MAX_RANKS) { s->auxIn[tid] = 0; s->auxOut[tid] = 0; s->directBase[tid] = 0; }
    if (tid < 5) s->tk[tid] = 0;
  }
  __device__ __forceinline__ void init(const RaDevComm* c, LdsChan* s, int channel, uint32_t stepBytes_) {
    stepBytes = stepBytes_;
    comm = c; sh = s; ch = uniformInt(channel); tid = threadIdx.x; nthr = blockDim.x;
    rank = uniformInt(c->rank); nRanks = uniformInt(c->nRanks);
    loadChannel(c, s, channel);
    __syncthreads();
  }
  // The workgroup turns to another of its channels (two-rank role pairing: a scatter or gather workgroup serves two).
  __device__ __forceinline__ void select(LdsChan* s, int channel) { sh = s; ch = uniformInt(channel); }

  // Persists the step counters this workgroup advanced on channel (s, channel).
  static __device__ __forceinline__ void storeChannel(const RaDevComm* c, LdsChan* s, int channel, int role) {
    const int tid = threadIdx.x;
    if ((c->tuneFlags & 32) && tid == 0 && role < 3)
      for (int k = 0; k < 5; k++)
        __hip_atomic_fetch_add(&c->stats[8 + role * 5 + k], s->tk[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int n = c->nRanks * RA_NKINDS;
    if (tid < n) {
      const int p = tid / RA_NKINDS, k = tid % RA_NKINDS;
      const size_t ix = (size_t)channel * n + tid;
      if (s->sendDirty[k] >> p & 1) c->sendStep[ix] = s->sendStep[p][k];
      if (s->recvDirty[k] >> p & 1) c->recvStep[ix] = s->recvStep[p][k];
    }
  }
  __device__ __forceinline__ void finish(int role = 0) {
    __syncthreads();
    storeChannel(comm, sh, ch, role);
  }

  // Fresh copies of the lane number and of the communicator pointer that the compiler cannot trace back to the
  // originals: what a step derives from them (every lane's flag addresses, window layout fields) is then computed
  // where it is used instead of once in front of the round loop, where it would stay live across the copy loops and,
  // with their 128 registers in use, end up in scratch memory.
  static __device__ __forceinline__ int freshLane(int v) { asm volatile("" : "+v"(v)); return v; }
  static __device__ __forceinline__ const RaDevComm* freshComm(const RaDevComm* c) {
    const uint64_t u = (uint64_t)c;
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return (const RaDevComm*)(((uint64_t)hi << 32) | lo);
  }
  __device__ __forceinline__ uint64_t* flagOf(const RaDevComm* c, int owner, int which, int peer, int kind) const {
    return (uint64_t*)(c->win[owner] + raFlagOffset(c->lay, which, ch, peer, kind));
  }

  // Waits until the peers in `mask` have consumed every slot this workgroup published on the FIFOs of `kind` towards them
  // (their credits have caught up with the send counters): a sender whose peers read its user buffer directly may not
  // leave the kernel before.
  __device__ __forceinline__ void awaitCredits(uint32_t mask, int kind) {
    const int t = freshLane(tid);
    const RaDevComm* c = freshComm(comm);
    __syncthreads();
    if (t < RA_MAX_RANKS && (mask >> t & 1)) raSpin(c, &sh->aborted, flagOf(c, rank, 1, t, kind), sh->sendStep[t][kind]);
    __syncthreads();
  }

  // One protocol step.  recvMask/sendMask: bit p set = peer p takes part.  Lane p of wave 0
  // waits for peer p's data (tail), lane 16+p waits for a free slot towards peer p (head).
  // `body()` runs on all threads between the waits and the posts; it reads recvSlot()s and
  // fills sendSlot()s.  Returns false once the communicator was aborted.
  template <class F>
  __device__ __forceinline__ bool step(uint32_t recvMask, int rk, uint32_t sendMask, int sk, F&& body) {
    bool ok;
    {
      const int t = freshLane(tid);
      const RaDevComm* c = freshComm(comm);
      const bool timed = (c->tuneFlags & 32) && t == 0;
      if (timed) sh->ts[0] = wall_clock64();
      if (t < RA_MAX_RANKS) {
        if (recvMask >> t & 1) {
          raSpin(c, &sh->aborted, flagOf(c, rank, 0, t, rk), sh->recvStep[t][rk] + 1);
          if (auxRecv) sh->auxIn[t] = ldFlag(flagOf(c, rank, 2 + (int)(sh->recvStep[t][rk] % c->lay.nSlots), t, rk));
          if (!(c->tuneFlags & 2)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        }
      } else if (t < 2 * RA_MAX_RANKS) {
        const int p = t - RA_MAX_RANKS;
        if (sendMask >> p & 1) {
          const uint64_t need = sh->sendStep[p][sk] + 1;   // slots published once this one is
          if (need > (uint64_t)c->lay.nSlots) raSpin(c, &sh->aborted, flagOf(c, rank, 1, p, sk), need - c->lay.nSlots);
        }
      }
      if (t < 64) drainVmem();
      __syncthreads();
      if (timed) sh->ts[1] = wall_clock64();
      ok = uniformInt(sh->aborted) == 0;
      // tuneFlags 64: every step that consumes data dawdles (about 0.1 ms) before its body, so that the senders run
      // ahead until they are held back by the credits: the slow-consumer case, which fast reducers never produce.
      if ((c->tuneFlags & 64) && recvMask)
        for (int i = 0; i < 32; i++) __builtin_amdgcn_s_sleep(127);
    }
    if (ok) body();
    {
      const int t = freshLane(tid);
      const RaDevComm* c = freshComm(comm);
      const bool timed = (c->tuneFlags & 32) && t == 0;
      if (timed) sh->ts[2] = wall_clock64();
      // Every wave: its loads have returned and its stores are acknowledged.  (A step that only receives posts nothing
      // but credits, and a wave that has issued its last store has consumed all its loads, so such a step could do
      // without the drain; tuneFlags 128 leaves it out there.  Measured with ranks sharing one GPU it changes nothing:
      // 1.58 / 4.70 / 9.86 ms with the drain, 1.59 / 4.70 / 9.87 ms without, for 1 GiB at 2 / 4 / 8 ranks.)
      if (sendMask || !(c->tuneFlags & 128)) drainVmem();
      __syncthreads();
      if (timed) sh->ts[3] = wall_clock64();
      if (t < 64) {
        // Nothing but the drain is needed before the flag.  Every byte this step handed to a peer went out as a
        // system-scope write-through store (sc0 sc1), each wave waited for its own stores to be acknowledged (vmcnt(0))
        // and all waves met at the barrier above, so those bytes are at their destination before the flag store is
        // issued; the flag takes the same path behind them.  The reference posts plain stores and therefore needs the
        // L2 write-back of __threadfence() first (prims_simple.h:168-171); here that write-back would only flush the
        // receive-buffer lines other workgroups are still producing (measured: two ranks sharing a GPU, 1 GiB: 2.02 ms
        // with an agent-scope release here, 1.77 ms without).  tuneFlags 8 puts the agent-scope release back,
        // 8 + 4 a system-scope one.
        if (sendMask && (c->tuneFlags & 8)) {
          if (c->tuneFlags & 4) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
          else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          drainVmem();
        }
        if (t < RA_MAX_RANKS) {
          if (recvMask >> t & 1) {          // slot consumed: credit to the sender (its window)
            const uint64_t s = sh->recvStep[t][rk] + 1;
            sh->recvStep[t][rk] = s;
            if (ok) stFlag(flagOf(c, t, 1, rank, rk), s);
          }
        } else if (t < 2 * RA_MAX_RANKS) {
          const int p = t - RA_MAX_RANKS;
          if (sendMask >> p & 1) {            // slot published: tail in the receiver's window
            const uint64_t s = sh->sendStep[p][sk] + 1;
            sh->sendStep[p][sk] = s;
            if (ok) {
              // host-staged transport: tell the proxy how many bytes this slot carries, then publish
              if (c->netMode) stFlag(flagOf(c, p, 2 + (int)((s - 1) % c->lay.nSlots), rank, sk), stepBytes);
              else if (auxSend) {               // the word must be in place before the tail that announces it
                stFlag(flagOf(c, p, 2 + (int)((s - 1) % c->lay.nSlots), rank, sk), sh->auxOut[p]);
                drainVmem();
              }
              stFlag(flagOf(c, p, 0, rank, sk), s);
            }
          }
        }
        if (t == 0) { sh->recvDirty[rk] |= recvMask; sh->sendDirty[sk] |= sendMask; }
        // host-staged transport: stamp the channel's word in this rank's own window behind the posts (the proxy thread
        // looks at the stamps first and walks a channel's flag lines only when its stamp has changed)
        if (c->netMode && ok && (recvMask | sendMask)) {
          drainVmem();
          if (t == 0) stFlag(flagOf(c, rank, 0, rank, 0), wall_clock64());
        }
      }
      if (timed) {
        drainVmem();
        const uint64_t t4 = wall_clock64();
        sh->tk[0] += sh->ts[1] - sh->ts[0]; sh->tk[1] += sh->ts[2] - sh->ts[1]; sh->tk[2] += sh->ts[3] - sh->ts[2];
        sh->tk[3] += t4 - sh->ts[3]; sh->tk[4] += 1;
      }
    }
    // No barrier here: the counters wave 0 has just advanced are next read by wave 0 itself (the waits of the following
    // step) and by everybody only behind that step's first barrier, which wave 0 reaches after these writes.
    return ok;
  }
};

// Round geometry shared by the all-reduce schedules: the buffer is cut into rounds of
// nRanks*chunk elements; channel c owns rounds c, c+C, ...; inside a round, piece j
// starts at j*chunk.  The final round uses lastChunkElems so that its pieces stay balanced
// (reference: src/enqueue.cc:277-297, src/device/all_reduce.h:75).
struct Rounds {
  size_t count, chunk, lastChunk, roundElems, nRounds;
  int n;
  __device__ __forceinline__ Rounds(const RaWork& w, int nRanks) {
    n = uniformInt(nRanks); count = uniform64(w.count); chunk = uniform64(w.chunkElems); lastChunk = uniform64(w.lastChunkElems);
    roundElems = (size_t)n * chunk;
    nRounds = (count + roundElems - 1) / roundElems;
  }
  // elements of the largest piece of round g
  __device__ __forceinline__ size_t pieceMax(size_t g) const { return (g + 1 == nRounds) ? lastChunk : chunk; }
  __device__ __forceinline__ void piece(size_t g, int j, size_t& off, size_t& nelem) const {
    const size_t base = g * roundElems;
    const size_t ck = (g + 1 == nRounds) ? lastChunk : chunk;
    off = base + (size_t)j * ck;
    const size_t end = off + ck < count ? off + ck : count;
    nelem = off < count ? end - off : 0;
    if (off > count) off = count;
  }
};

__device__ __forceinline__ uint32_t peersMask(int nRanks, int rank) { return ((1u << nRanks) - 1u) & ~(1u << rank); }

// Where peer p's registered receive buffer is mapped here, from the word W it announced (0: not mapped, use the FIFO).
__device__ __forceinline__ uint64_t regBaseOf(const RaDevComm* comm, int p, uint64_t W) {
  uint64_t base = 0;
  if ((W & RA_REG_VALID) && RA_REG_IDX(W) < RA_MAX_REGS) {
    const RaRegSlot slot = comm->regTable[p * RA_MAX_REGS + RA_REG_IDX(W)];
    if (slot.base && slot.serial == RA_REG_SERIAL(W)) base = slot.base + RA_REG_OFF(W);
  }
  return base;
}

// ------------------------------------------------------------------ direct (all-pairs) schedules
// All-reduce.  Piece j of every round is owned by rank j.  Three roles per channel:
//   scatter : push my input of piece p into p's window (kind 0), for every peer p
//   reduce  : fold my own piece with the n-1 arrivals in rank order, write it to recvbuff
//             and push it into every peer's window (kind 1)
//   gather  : copy the n-1 finished pieces that arrive (kind 1) into recvbuff
// The workgroup serves nCtx channels (two for the scatter and gather workgroups of the two-rank role pairing, see
// raCollKernel: channels ch0 and ch0 + 1 with LDS blocks sh0 and sh1), taking their rounds in turn, one step on each.
template <typename T, int OP>
__device__ __forceinline__ void directAllReduce(Ctx& cx, LdsChan* sh0, LdsChan* sh1, int ch0, int nCtx, const RaWork& w, int role) {
  const int n = uniformInt(cx.nRanks), r = uniformInt(cx.rank);
  role = uniformInt(role); nCtx = uniformInt(nCtx);
  const Rounds R(w, n);
  const uint32_t peers = peersMask(n, r);
  const T* in = (const T*)w.send;
  T* out = (T*)w.recv;
  // Registered receive buffers (zero-copy gather).  With its first scatter step every rank tells each peer whether
  // and where its recvbuff is registered; a reducer that has that registration mapped writes its finished piece
  // straight into the peer's recvbuff and says so with its first broadcast-phase post, and the peer's gather role
  // then has nothing to copy for it.  The words travel beside the FIFO slots, once per launch and channel.
  const bool words = !cx.comm->netMode;
  if ((role == 0 || role == 3) && cx.tid < n) {
    sh0->auxOut[cx.tid] = w.regWord;
    if (nCtx > 1) sh1->auxOut[cx.tid] = w.regWord;
  }
  // Role 3: the workgroup walks all three roles of its channel, skewed by one round each — scatter of round k, reduce of
  // round k - 1, gather of round k - 2 — so that what a step waits for was produced by the peers one iteration earlier.
  // Every rank runs the same sequence and a step depends only on steps of earlier iterations, so two slots per FIFO suffice.
  const int nPh = role == 3 ? 3 : 1;
  const size_t stride = (size_t)uniformInt(w.nChannels);
  const size_t kEnd = uniform64(R.nRounds + (size_t)(nPh - 1) * stride);
  for (size_t k = 0; (size_t)ch0 + k * stride < kEnd; k++)
   for (int q = 0; q < (role == 3 ? 3 : nCtx); q++) {     // (q: the channel of a pairing workgroup, or the phase of role 3)
    const int c = role == 3 ? 0 : q, ph = role == 3 ? q : 0;
    cx.select(c ? sh1 : sh0, ch0 + c);
    const int rl = role == 3 ? ph : role;
    if (k < (size_t)ph) continue;
    const size_t kk = k - (size_t)ph;
    const size_t g = (size_t)cx.ch + kk * stride;
    if (g >= R.nRounds) continue;
    const bool first = kk == 0;
    cx.stepBytes = (uint32_t)(R.pieceMax(g) * sizeof(T));
    if (rl == 0) {
      cx.auxSend = words && first;
      cx.step(0, 0, peers, 0, [&]() {
        // all n-1 pieces in one fused pass when they have the same length and need no scaling
        RcArgs pa; pa.ns = n - 1; pa.nd = n - 1; pa.preIdx = -1; pa.post = false; pa.opArg = 0;
        size_t ne0 = 0; bool same = OP != RA_OP_PREMULSUM;
        RC_EACH(i, 0, n - 1,
          const int p = (r + 1 + (i + cx.ch) % (n - 1)) % n;   // channels start on different links
          size_t off, ne; R.piece(g, p, off, ne);
          if (i == 0) ne0 = ne; else same = same && ne == ne0;
          pa.src[i] = uniformPtr((const char*)(in + off)); pa.dst[i] = cx.sendSlot(p, 0);
        )
        if (same) { pairCopyWG<true>(pa, ne0 * sizeof(T), cx.tid, cx.nthr); return; }
        for (int i = 0; i < n - 1; i++) {              // (a loop that indexes no argument array: one copy of the code)
          const int p = (r + 1 + (i + cx.ch) % (n - 1)) % n;
          size_t off, ne; R.piece(g, p, off, ne);
          RcArgs a; a.ns = 1; a.nd = 1; a.preIdx = 0; a.post = false; a.opArg = w.opArg;
          a.src[0] = uniformPtr((const char*)(in + off)); a.dst[0] = cx.sendSlot(p, 0);
          rcWG<T, OP, RC_SEND>(a, ne, cx.tid, cx.nthr);
        }
      });
    } else if (rl == 1) {
      cx.auxRecv = words && first && w.peerRegs;
      cx.auxSend = words && first;
      cx.step(peers, 0, peers, 1, [&]() {
        if (first && words) {
          const int p = Ctx::freshLane(cx.tid);      // (computed here, not kept in a register across every round)
          if (p < n && p != r) {
            const uint64_t base = regBaseOf(cx.comm, p, w.peerRegs ? cx.sh->auxIn[p] : 0);
            cx.sh->directBase[p] = base;
            cx.sh->auxOut[p] = base ? RA_AUX_DIRECT : 0;
          }
          __syncthreads();
        }
        size_t off, ne; R.piece(g, r, off, ne);
        RcArgs a; a.ns = n; a.nd = n; a.preIdx = r; a.post = true; a.opArg = w.opArg;
        RC_EACH(j, 0, n, a.src[j] = (j == r) ? uniformPtr((const char*)(in + off)) : cx.recvSlot(j, 0);)
        a.dst[0] = uniformPtr((char*)(out + off)); a.nLocal = 1;
        int nDirect = 0;
        RC_EACH(i, 1, n,
          const int p = (r + i) % n;
          const uint64_t base = uniform64(cx.sh->directBase[p]);
          a.dst[i] = base ? (char*)base + off * sizeof(T) : cx.sendSlot(p, 1);
          nDirect += base != 0;
        )
        rcWG<T, OP, RC_FOLD_ALL>(a, ne, cx.tid, cx.nthr);
        if (cx.tid == 0) {
          if (nDirect) __hip_atomic_fetch_add(&cx.comm->stats[0], (uint64_t)nDirect, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (n - 1 - nDirect) __hip_atomic_fetch_add(&cx.comm->stats[1], (uint64_t)(n - 1 - nDirect), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      });
    } else {
      cx.auxRecv = words && first && w.regWord != 0;
      cx.step(peers, 1, 0, 0, [&]() {
        // pieces a peer wrote straight into recvbuff need no copy
        RcArgs pa; pa.ns = n - 1; pa.nd = n - 1; pa.preIdx = -1; pa.post = false; pa.opArg = 0;
        size_t ne0 = 0; bool same = true; bool anyDirect = false;
        RC_EACH(i, 1, n,
          const int p = (r + n - i) % n;
          anyDirect = anyDirect || uniform64(cx.sh->auxIn[p]) == RA_AUX_DIRECT;
          size_t off, ne; R.piece(g, p, off, ne);
          if (i == 1) ne0 = ne; else same = same && ne == ne0;
          pa.src[i - 1] = cx.recvSlot(p, 1); pa.dst[i - 1] = uniformPtr((char*)(out + off));
        )
        if (same && !anyDirect) { pairCopyWG<false>(pa, ne0 * sizeof(T), cx.tid, cx.nthr); return; }
        for (int i = 1; i < n; i++) {
          const int p = (r + n - i) % n;
          if (uniform64(cx.sh->auxIn[p]) == RA_AUX_DIRECT) continue;
          size_t off, ne; R.piece(g, p, off, ne);
          RcArgs a; a.ns = 1; a.nd = 1; a.preIdx = -1; a.post = false; a.opArg = 0;
          a.src[0] = cx.recvSlot(p, 1); a.dst[0] = uniformPtr((char*)(out + off)); a.nLocal = 1;
          copyWG<CP_END>(a, ne * sizeof(T), cx.tid, cx.nthr);
        }
      });
    }
  }
  cx.auxRecv = 0; cx.auxSend = 0;
}

// Reduce-scatter: block j of sendbuff (count elements) belongs to rank j.  Round g covers
// elements [g*chunk, (g+1)*chunk) of every block.  Roles: scatter, reduce.
//
// Registered send buffers (zero-copy scatter phase; the reference's direct-read primitives, src/device/prims_simple.h:
// 243-283): a reducer that has a peer's sendbuff registration mapped reads that peer's contribution straight from the
// sendbuff instead of from a FIFO slot, and the peer's scatter role then moves nothing for it.  For blocks of
// RA_AG_ANNOUNCE_BYTES and more (a rule every rank evaluates alike) every scatter workgroup first sends its sendbuff's
// registration word to every peer (kind 0), every reducer answers each sender whether it will read directly (kind 1),
// and the scatter workgroup leaves the kernel only when the peers that read directly have consumed its last round (their
// credits): until then the caller's sendbuff must stay as it is.  Not with a pre-multiplied sum: contributions read
// from a sendbuff are not scaled yet, and the fold scales one source only.
template <typename T, int OP>
__device__ __forceinline__ void directReduceScatter(Ctx& cx, const RaWork& w, int role) {
  const int n = uniformInt(cx.nRanks), r = uniformInt(cx.rank);
  role = uniformInt(role);
  const uint32_t peers = peersMask(n, r);
  const T* in = (const T*)w.send;
  T* out = (T*)w.recv;
  const size_t count = uniform64(w.count), chunk = uniform64(w.chunkElems);
  const size_t nRounds = (count + chunk - 1) / chunk;
  const bool announce = !cx.comm->netMode && count * sizeof(T) >= RA_AG_ANNOUNCE_BYTES && (size_t)cx.ch < nRounds;
  if (announce && role == 0) {
    if (cx.tid < n) cx.sh->auxOut[cx.tid] = w.regWord;
    cx.auxSend = 1; cx.step(0, 0, peers, 0, []() {}); cx.auxSend = 0;
    cx.auxRecv = 1; cx.step(peers, 1, 0, 0, []() {}); cx.auxRecv = 0;      // auxIn[p]: RA_AUX_DIRECT when p reads my sendbuff itself
  } else if (announce) {
    cx.auxRecv = 1;
    cx.step(peers, 0, 0, 0, [&]() {
      if (cx.tid < n && cx.tid != r) {
        const int p = cx.tid;
        const uint64_t base = (OP == RA_OP_PREMULSUM || !w.peerRegs) ? 0 : regBaseOf(cx.comm, p, cx.sh->auxIn[p]);
        cx.sh->directBase[p] = base;
        cx.sh->auxOut[p] = base ? RA_AUX_DIRECT : 0;
      }
    });
    cx.auxRecv = 0;
    cx.auxSend = 1; cx.step(0, 0, peers, 1, []() {}); cx.auxSend = 0;
  }
  uint32_t readers = 0;          // scatter role: the peers that read my sendbuff themselves
  if (announce && role == 0) {
    RC_EACH(i, 0, n, if (i != r && uniform64(cx.sh->auxIn[i]) == RA_AUX_DIRECT) readers |= 1u << i;)
  }
  for (size_t g = cx.ch; g < nRounds; g += (size_t)uniformInt(w.nChannels)) {
    const size_t off = g * chunk;
    const size_t ne = count - off < chunk ? count - off : chunk;
    cx.stepBytes = (uint32_t)(ne * sizeof(T));       // what a slot of this round carries (the host-staged proxy ships that much)
    if (role == 0) {
      cx.step(0, 0, peers, 0, [&]() {
        if (readers == peers) return;                  // every peer reads for itself: the post alone says "round g may be read"
        if (OP != RA_OP_PREMULSUM && readers == 0) {
          RcArgs pa; pa.ns = n - 1; pa.nd = n - 1; pa.preIdx = -1; pa.post = false; pa.opArg = 0;
          RC_EACH(i, 0, n - 1,
            const int p = (r + 1 + (i + cx.ch) % (n - 1)) % n;   // channels start on different links
            pa.src[i] = uniformPtr((const char*)(in + (size_t)p * count + off)); pa.dst[i] = cx.sendSlot(p, 0);
          )
          pairCopyWG<true>(pa, ne * sizeof(T), cx.tid, cx.nthr);
          return;
        }
        for (int i = 0; i < n - 1; i++) {
          const int p = (r + 1 + (i + cx.ch) % (n - 1)) % n;
          if (readers >> p & 1) continue;
          RcArgs a; a.ns = 1; a.nd = 1; a.preIdx = 0; a.post = false; a.opArg = w.opArg;
          a.src[0] = uniformPtr((const char*)(in + (size_t)p * count + off)); a.dst[0] = cx.sendSlot(p, 0);
          rcWG<T, OP, RC_SEND>(a, ne, cx.tid, cx.nthr);
        }
      });
    } else {
      cx.step(peers, 0, 0, 0, [&]() {
        RcArgs a; a.ns = n; a.nd = 1; a.preIdx = r; a.post = true; a.opArg = w.opArg;
        int nDirect = 0;
        RC_EACH(j, 0, n,
          if (j == r) a.src[j] = uniformPtr((const char*)(in + (size_t)r * count + off));
          else {
            const uint64_t base = uniform64(cx.sh->directBase[j]);
            a.src[j] = base ? (const char*)base + ((size_t)r * count + off) * sizeof(T) : cx.recvSlot(j, 0);
            nDirect += base != 0;
          }
        )
        a.dst[0] = uniformPtr((char*)(out + off)); a.nLocal = 1;
        rcWG<T, OP, RC_FOLD_ONE>(a, ne, cx.tid, cx.nthr);
        if (announce && cx.tid == 0) {
          if (nDirect) __hip_atomic_fetch_add(&cx.comm->stats[0], (uint64_t)nDirect, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (n - 1 - nDirect) __hip_atomic_fetch_add(&cx.comm->stats[1], (uint64_t)(n - 1 - nDirect), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      });
    }
  }
  if (readers) cx.awaitCredits(readers, 0);
}

// All-gather (bytes): roles: spread (my block into every peer's window and into my own
// recvbuff slot unless in place), gather.
__device__ __forceinline__ void directAllGather(Ctx& cx, const RaWork& w, int role) {
  const int n = uniformInt(cx.nRanks), r = uniformInt(cx.rank);
  role = uniformInt(role);
  const uint32_t peers = peersMask(n, r);
  const size_t nbytes = uniform64(w.count * w.elemSize);          // per rank
  const size_t chunkB = uniform64(w.chunkElems * w.elemSize);
  const size_t nRounds = (nbytes + chunkB - 1) / chunkB;
  // Registered receive buffers (zero-copy all-gather).  A spreader may write into a peer's recvbuff only once that
  // peer's launch has begun, and it has to learn where the buffer is: for blocks of RA_AG_ANNOUNCE_BYTES and more
  // (a rule every rank evaluates alike) each gather workgroup first sends its registration word to every peer on the
  // otherwise idle kind-0 flags, and each spread workgroup waits for those words before its first chunk.  A spreader
  // that has the registration mapped then writes its block straight into the peer's recvbuff and says so with its
  // first post; that peer's gather role has nothing to copy for it.
  const bool announce = !cx.comm->netMode && nbytes >= RA_AG_ANNOUNCE_BYTES && (size_t)cx.ch < nRounds;
  if (announce && role == 1) {
    if (cx.tid < n) cx.sh->auxOut[cx.tid] = w.regWord;
    cx.auxSend = 1;
    cx.step(0, 0, peers, 0, []() {});
    cx.auxSend = 0;
  } else if (announce) {
    cx.auxRecv = 1;
    cx.step(peers, 0, 0, 0, [&]() {
      if (cx.tid < n && cx.tid != r) {
        const int p = cx.tid;
        const uint64_t base = regBaseOf(cx.comm, p, w.peerRegs ? cx.sh->auxIn[p] : 0);
        cx.sh->directBase[p] = base;
        cx.sh->auxOut[p] = base ? RA_AUX_DIRECT : 0;
      }
    });
    cx.auxRecv = 0;
  }
  for (size_t g = cx.ch; g < nRounds; g += (size_t)uniformInt(w.nChannels)) {
    const bool first = g == (size_t)cx.ch;
    const size_t off = g * chunkB;
    const size_t nb = nbytes - off < chunkB ? nbytes - off : chunkB;
    cx.stepBytes = (uint32_t)nb;
    if (role == 0) {
      cx.auxSend = announce && first;
      cx.step(0, 0, peers, 1, [&]() {
        RcArgs a; a.ns = 1; a.preIdx = -1; a.post = false; a.opArg = 0;
        a.src[0] = uniformPtr(w.send + off);
        char* mine = uniformPtr(w.recv + (size_t)r * nbytes + off);
        const int lead = mine != a.src[0] ? 1 : 0;         // out of place: my own block goes to my recvbuff as well
        int nDirect = 0;
        // destination i (i = 1 .. n-1) is peer (r + i) % n; with the local copy in front they sit at dst[i], else at dst[i-1]
        char* peerDst[RA_MAX_SRCS];
        RC_EACH(i, 1, n,
          const int p = (r + i) % n;
          const uint64_t base = uniform64(cx.sh->directBase[p]);
          peerDst[i] = base ? (char*)base + (size_t)r * nbytes + off : cx.sendSlot(p, 1);
          nDirect += base != 0;
        )
        if (lead) { a.dst[0] = mine; RC_EACH(i, 1, n, a.dst[i] = peerDst[i];) a.nd = n; a.nLocal = 1; }
        else { RC_EACH(i, 1, n, a.dst[i - 1] = peerDst[i];) a.nd = n - 1; a.nLocal = 0; }
        copyWG<CP_SPREAD>(a, nb, cx.tid, cx.nthr);
        if (announce && cx.tid == 0) {
          if (nDirect) __hip_atomic_fetch_add(&cx.comm->stats[0], (uint64_t)nDirect, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (n - 1 - nDirect) __hip_atomic_fetch_add(&cx.comm->stats[1], (uint64_t)(n - 1 - nDirect), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      });
    } else {
      cx.auxRecv = announce && first && w.regWord != 0;
      cx.step(peers, 1, 0, 0, [&]() {
        // blocks a peer wrote straight into recvbuff need no copy
        RcArgs pa; pa.ns = n - 1; pa.nd = n - 1; pa.preIdx = -1; pa.post = false; pa.opArg = 0;
        bool anyDirect = false;
        RC_EACH(i, 1, n,
          const int p = (r + n - i) % n;
          anyDirect = anyDirect || uniform64(cx.sh->auxIn[p]) == RA_AUX_DIRECT;
          pa.src[i - 1] = cx.recvSlot(p, 1); pa.dst[i - 1] = uniformPtr(w.recv + (size_t)p * nbytes + off);
        )
        if (!anyDirect) { pairCopyWG<false>(pa, nb, cx.tid, cx.nthr); return; }
        for (int i = 1; i < n; i++) {
          const int p = (r + n - i) % n;
          if (uniform64(cx.sh->auxIn[p]) == RA_AUX_DIRECT) continue;
          RcArgs a; a.ns = 1; a.nd = 1; a.preIdx = -1; a.post = false; a.opArg = 0; a.nLocal = 1;
          a.src[0] = cx.recvSlot(p, 1); a.dst[0] = uniformPtr(w.recv + (size_t)p * nbytes + off);
          copyWG<CP_END>(a, nb, cx.tid, cx.nthr);
        }
      });
    }
  }
  cx.auxRecv = 0; cx.auxSend = 0;
}

// Broadcast (bytes): the root spreads each chunk into every peer's window (and into its own recvbuff
// when out of place); every other rank copies what arrives from the root.  Role of
// src/device/broadcast.h on the all-pairs links: one hop from the root to everybody.
__device__ __forceinline__ void directBroadcast(Ctx& cx, const RaWork& w, int role) {
  const int n = uniformInt(cx.nRanks), r = uniformInt(cx.rank), root = uniformInt(w.root);
  role = uniformInt(role);
  const uint32_t peers = peersMask(n, r);
  const size_t nbytes = uniform64(w.count * w.elemSize);
  const size_t chunkB = uniform64(w.chunkElems * w.elemSize);
  const size_t nRounds = (nbytes + chunkB - 1) / chunkB;
  if ((role == 0) != (r == root)) return;
  for (size_t g = cx.ch; g < nRounds; g += (size_t)uniformInt(w.nChannels)) {
    const size_t off = g * chunkB;
    const size_t nb = nbytes - off < chunkB ? nbytes - off : chunkB;
    cx.stepBytes = (uint32_t)nb;
    if (role == 0) {
      cx.step(0, 0, peers, 1, [&]() {
        RcArgs a; a.ns = 1; a.preIdx = -1; a.post = false; a.opArg = 0;
        a.src[0] = uniformPtr(w.send + off);
        if (w.recv != w.send) {
          a.dst[0] = uniformPtr(w.recv + off); a.nLocal = 1; a.nd = n;
          RC_EACH(i, 1, n, a.dst[i] = cx.sendSlot((r + i) % n, 1);)
        } else {
          a.nLocal = 0; a.nd = n - 1;
          RC_EACH(i, 1, n, a.dst[i - 1] = cx.sendSlot((r + i) % n, 1);)
        }
        copyWG<CP_SPREAD>(a, nb, cx.tid, cx.nthr);
      });
    } else {
      cx.step(1u << root, 1, 0, 0, [&]() {
        RcArgs a; a.ns = 1; a.nd = 1; a.preIdx = -1; a.post = false; a.opArg = 0;
        a.src[0] = cx.recvSlot(root, 1); a.dst[0] = uniformPtr(w.recv + off); a.nLocal = 1;
        copyWG<CP_END>(a, nb, cx.tid, cx.nthr);
      });
    }
  }
}

// Reduce to a root: every other rank pushes its input into the root's window; the root folds the n
// contributions in rank order (role of src/device/reduce.h, one hop).
template <typename T, int OP>
__device__ __forceinline__ void directReduce(Ctx& cx, const RaWork& w, int role) {
  const int n = uniformInt(cx.nRanks), r = uniformInt(cx.rank), root = uniformInt(w.root);
  role = uniformInt(role);
  const uint32_t peers = peersMask(n, r);
  const T* in = (const T*)w.send;
  T* out = (T*)w.recv;
  const size_t count = uniform64(w.count), chunk = uniform64(w.chunkElems);
  const size_t nRounds = (count + chunk - 1) / chunk;
  if ((role == 1) != (r == root)) return;
  for (size_t g = cx.ch; g < nRounds; g += (size_t)uniformInt(w.nChannels)) {
    const size_t off = g * chunk;
    const size_t ne = count - off < chunk ? count - off : chunk;
    cx.stepBytes = (uint32_t)(ne * sizeof(T));       // what a slot of this round carries (the host-staged proxy ships that much)
    if (role == 0) {
      cx.step(0, 0, 1u << root, 0, [&]() {
        RcArgs a; a.ns = 1; a.nd = 1; a.preIdx = 0; a.post = false; a.opArg = w.opArg;
        a.src[0] = uniformPtr((const char*)(in + off)); a.dst[0] = cx.sendSlot(root, 0);
        rcWG<T, OP, RC_SEND>(a, ne, cx.tid, cx.nthr);
      });
    } else {
      cx.step(peers, 0, 0, 0, [&]() {
        RcArgs a; a.ns = n; a.nd = 1; a.preIdx = r; a.post = true; a.opArg = w.opArg;
        RC_EACH(j, 0, n, a.src[j] = (j == r) ? uniformPtr((const char*)(in + off)) : cx.recvSlot(j, 0);)
        a.dst[0] = uniformPtr((char*)(out + off)); a.nLocal = 1;
        rcWG<T, OP, RC_FOLD_ONE>(a, ne, cx.tid, cx.nthr);
      });
    }
  }
}

// ------------------------------------------------------------------ ring schedules
// Same walk as the reference ring (src/device/all_reduce.h:71-199): in a round, the piece
// with ring index c is seeded at ring position c+1, gains one rank's input per hop and is
// finished at position c, which also starts its circulation.  One workgroup per channel.
struct RingPos {
  int ix, prev, next;
  const int8_t* order;
  __device__ __forceinline__ RingPos(const Ctx& cx) {
    order = cx.comm->ringRanks[cx.ch % cx.comm->nRings];
    ix = 0;
    for (int i = 0; i < cx.nRanks; i++) if (order[i] == cx.rank) ix = i;
    prev = uniformInt(order[(ix + cx.nRanks - 1) % cx.nRanks]);
    next = uniformInt(order[(ix + 1) % cx.nRanks]);
    ix = uniformInt(ix);
  }
};

// Registered receive buffers on the ring and the chain (zero-copy circulation phase; the reference's Direct = 1
// primitives, src/device/prims_simple.h:243-283, which it leaves off for its rings, src/device/all_reduce.h:62).
// The rank that writes into my recvbuff is the one I receive finished pieces from (`from`), and it has to learn where
// the buffer is after my launch has begun: for payloads of RA_AG_ANNOUNCE_BYTES and more (a rule every rank evaluates
// alike, so the step counters stay in agreement) each workgroup that receives finished pieces first sends its
// registration word to `from` on the kind-1 flags of that direction, which carry no data in these schedules, and each
// workgroup that sends finished pieces (to `to`) waits for that word.  A sender that has the registration mapped writes
// finished pieces straight into the peer's recvbuff and says so with the aux word of its first post; the receiver then
// forwards from its own recvbuff instead of the FIFO slot and has nothing to copy for itself.
// from < 0: this workgroup receives no finished pieces; to < 0: it sends none.  Sends before it waits: no cycle.
__device__ __forceinline__ void announceRecvBuff(Ctx& cx, const RaWork& w, int from, int to) {
  if (from >= 0) {
    if (cx.tid == from) cx.sh->auxOut[from] = w.regWord;
    cx.auxSend = 1;
    cx.step(0, 0, 1u << from, 1, []() {});
    cx.auxSend = 0;
  }
  if (to >= 0) {
    cx.auxRecv = 1;
    cx.step(1u << to, 1, 0, 0, [&]() {
      if (cx.tid == to) {
        const uint64_t base = regBaseOf(cx.comm, to, w.peerRegs ? cx.sh->auxIn[to] : 0);
        cx.sh->directBase[to] = base;
        cx.sh->auxOut[to] = base ? RA_AUX_DIRECT : 0;
      }
    });
    cx.auxRecv = 0;
  }
}
__device__ __forceinline__ void countPiece(const Ctx& cx, bool direct) {
  if (cx.tid == 0) __hip_atomic_fetch_add(&cx.comm->stats[direct ? 0 : 1], (uint64_t)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename T, int OP>
__device__ __forceinline__ void ringAllReduce(Ctx& cx, const RaWork& w) {
  const int n = uniformInt(cx.nRanks);
  const Rounds R(w, n);
  const RingPos rp(cx);
  const uint32_t pm = 1u << rp.prev, nm = 1u << rp.next;
  const T* in = (const T*)w.send;
  T* out = (T*)w.recv;
  const bool announce = !cx.comm->netMode && R.count * sizeof(T) >= RA_AG_ANNOUNCE_BYTES && (size_t)cx.ch < R.nRounds;
  if (announce) announceRecvBuff(cx, w, rp.prev, rp.next);
  for (size_t g = cx.ch; g < R.nRounds; g += (size_t)uniformInt(w.nChannels)) {
    const bool first = announce && g == (size_t)cx.ch;
    size_t off, ne;
    cx.stepBytes = (uint32_t)(R.pieceMax(g) * sizeof(T));
    // step 0: seed the piece behind me (its first post carries the "I write directly" word)
    R.piece(g, (rp.ix + n - 1) % n, off, ne);
    cx.auxSend = first;
    cx.step(0, 0, nm, 0, [&]() {
      RcArgs a; a.ns = 1; a.nd = 1; a.preIdx = 0; a.post = false; a.opArg = w.opArg;
      a.src[0] = uniformPtr((const char*)(in + off)); a.dst[0] = cx.sendSlot(rp.next, 0);
      rcWG<T, OP, RC_SEND>(a, ne, cx.tid, cx.nthr);
    });
    cx.auxSend = 0;
    cx.auxRecv = first;          // the first step that receives (always one of the reducing steps) reads that word
    // n-2 hops: local + received -> next
    for (int j = 2; j < n; j++) {
      R.piece(g, (rp.ix + n - j) % n, off, ne);
      cx.step(pm, 0, nm, 0, [&]() {
        RcArgs a; a.ns = 2; a.nd = 1; a.preIdx = 0; a.post = false; a.opArg = w.opArg;
        a.src[0] = uniformPtr((const char*)(in + off)); a.src[1] = cx.recvSlot(rp.prev, 0);
        a.dst[0] = cx.sendSlot(rp.next, 0);
        rcWG<T, OP, RC_HOP>(a, ne, cx.tid, cx.nthr);
      });
      cx.auxRecv = 0;
    }
    // my own piece: final value, kept and sent on (into the next rank's recvbuff when that is mapped here)
    R.piece(g, rp.ix, off, ne);
    cx.step(pm, 0, nm, 0, [&]() {
      const uint64_t nb = uniform64(cx.sh->directBase[rp.next]);
      RcArgs a; a.ns = 2; a.nd = 2; a.preIdx = 0; a.post = true; a.opArg = w.opArg;
      a.src[0] = uniformPtr((const char*)(in + off)); a.src[1] = cx.recvSlot(rp.prev, 0);
      a.dst[0] = uniformPtr((char*)(out + off)); a.nLocal = 1;
      a.dst[1] = nb ? (char*)nb + off * sizeof(T) : cx.sendSlot(rp.next, 0);
      rcWG<T, OP, RC_HOP_KEEP>(a, ne, cx.tid, cx.nthr);
      if (announce) countPiece(cx, nb != 0);
    });
    cx.auxRecv = 0;
    // n-2 hops: keep and forward finished pieces
    for (int j = 1; j < n - 1; j++) {
      R.piece(g, (rp.ix + n - j) % n, off, ne);
      cx.step(pm, 0, nm, 0, [&]() {
        const uint64_t nb = uniform64(cx.sh->directBase[rp.next]);
        const bool here = uniform64(cx.sh->auxIn[rp.prev]) == RA_AUX_DIRECT;    // the piece already lies in my recvbuff
        char* fwd = nb ? (char*)nb + off * sizeof(T) : cx.sendSlot(rp.next, 0);
        RcArgs a; a.ns = 1; a.preIdx = -1; a.post = false; a.opArg = 0;
        if (here) {
          a.src[0] = uniformPtr((const char*)(out + off)); a.dst[0] = fwd; a.nd = 1; a.nLocal = 0;
          rcWG<int8_t, RA_OP_SUM, RC_SEND>(a, ne * sizeof(T), cx.tid, cx.nthr);
        } else {
          a.src[0] = cx.recvSlot(rp.prev, 0);
          a.dst[0] = uniformPtr((char*)(out + off)); a.nLocal = 1; a.dst[1] = fwd; a.nd = 2;
          copyWG<CP_FWD>(a, ne * sizeof(T), cx.tid, cx.nthr);
        }
        if (announce) countPiece(cx, nb != 0);
      });
    }
    // last finished piece
    R.piece(g, (rp.ix + 1) % n, off, ne);
    cx.step(pm, 0, 0, 0, [&]() {
      if (uniform64(cx.sh->auxIn[rp.prev]) == RA_AUX_DIRECT) return;
      RcArgs a; a.ns = 1; a.nd = 1; a.preIdx = -1; a.post = false; a.opArg = 0;
      a.src[0] = cx.recvSlot(rp.prev, 0); a.dst[0] = uniformPtr((char*)(out + off)); a.nLocal = 1;
      copyWG<CP_END>(a, ne * sizeof(T), cx.tid, cx.nthr);
    });
  }
  cx.auxRecv = 0; cx.auxSend = 0;
}

// Ring reduce-scatter (src/device/reduce_scatter.h:36-57): blocks travel in the order
// order[n-1], ..., order[0] relative to each rank's rotated ring and finish at their owner.
template <typename T, int OP>
__device__ __forceinline__ void ringReduceScatter(Ctx& cx, const RaWork& w) {
  const int n = uniformInt(cx.nRanks);
  const RingPos rp(cx);
  const uint32_t pm = 1u << rp.prev, nm = 1u << rp.next;
  const T* in = (const T*)w.send;
  T* out = (T*)w.recv;
  const size_t count = uniform64(w.count), chunk = uniform64(w.chunkElems);
  const size_t nRounds = (count + chunk - 1) / chunk;
  for (size_t g = cx.ch; g < nRounds; g += (size_t)uniformInt(w.nChannels)) {
    const size_t off = g * chunk;
    const size_t ne = count - off < chunk ? count - off : chunk;
    cx.stepBytes = (uint32_t)(ne * sizeof(T));       // what a slot of this round carries (the host-staged proxy ships that much)
    int blk = uniformInt(rp.order[(rp.ix + n - 1) % n]);
    cx.step(0, 0, nm, 0, [&]() {
      RcArgs a; a.ns = 1; a.nd = 1; a.preIdx = 0; a.post = false; a.opArg = w.opArg;
      a.src[0] = uniformPtr((const char*)(in + (size_t)blk * count + off)); a.dst[0] = cx.sendSlot(rp.next, 0);
      rcWG<T, OP, RC_SEND>(a, ne, cx.tid, cx.nthr);
    });
    for (int j = 2; j < n; j++) {
      blk = uniformInt(rp.order[(rp.ix + n - j) % n]);
      cx.step(pm, 0, nm, 0, [&]() {
        RcArgs a; a.ns = 2; a.nd = 1; a.preIdx = 0; a.post = false; a.opArg = w.opArg;
        a.src[0] = uniformPtr((const char*)(in + (size_t)blk * count + off)); a.src[1] = cx.recvSlot(rp.prev, 0);
        a.dst[0] = cx.sendSlot(rp.next, 0);
        rcWG<T, OP, RC_HOP>(a, ne, cx.tid, cx.nthr);
      });
    }
    cx.step(pm, 0, 0, 0, [&]() {
      RcArgs a; a.ns = 2; a.nd = 1; a.preIdx = 0; a.post = true; a.opArg = w.opArg;
      a.src[0] = uniformPtr((const char*)(in + (size_t)cx.rank * count + off)); a.src[1] = cx.recvSlot(rp.prev, 0);
      a.dst[0] = uniformPtr((char*)(out + off)); a.nLocal = 1;
      rcWG<T, OP, RC_HOP_END>(a, ne, cx.tid, cx.nthr);
    });
  }
}

// Ring all-gather (src/device/all_gather.h:69-145), bytes.  Blocks of RA_AG_ANNOUNCE_BYTES and more go straight into
// registered receive buffers (announceRecvBuff).
__device__ __forceinline__ void ringAllGather(Ctx& cx, const RaWork& w) {
  const int n = uniformInt(cx.nRanks);
  const RingPos rp(cx);
  const uint32_t pm = 1u << rp.prev, nm = 1u << rp.next;
  const size_t nbytes = uniform64(w.count * w.elemSize);
  const size_t chunkB = uniform64(w.chunkElems * w.elemSize);
  const size_t nRounds = (nbytes + chunkB - 1) / chunkB;
  const bool announce = !cx.comm->netMode && nbytes >= RA_AG_ANNOUNCE_BYTES && (size_t)cx.ch < nRounds;
  if (announce) announceRecvBuff(cx, w, rp.prev, rp.next);
  for (size_t g = cx.ch; g < nRounds; g += (size_t)uniformInt(w.nChannels)) {
    const bool first = announce && g == (size_t)cx.ch;
    const size_t off = g * chunkB;
    const size_t nb = nbytes - off < chunkB ? nbytes - off : chunkB;
    cx.stepBytes = (uint32_t)nb;
    cx.auxSend = first;
    cx.step(0, 0, nm, 0, [&]() {
      const uint64_t nbase = uniform64(cx.sh->directBase[rp.next]);
      RcArgs a; a.ns = 1; a.preIdx = -1; a.post = false; a.opArg = 0;
      a.src[0] = uniformPtr(w.send + off);
      char* mine = uniformPtr(w.recv + (size_t)cx.rank * nbytes + off);
      char* fwd = nbase ? (char*)nbase + (size_t)cx.rank * nbytes + off : cx.sendSlot(rp.next, 0);
      if (mine != a.src[0]) { a.dst[0] = mine; a.dst[1] = fwd; a.nd = 2; a.nLocal = 1; }
      else { a.dst[0] = fwd; a.nd = 1; a.nLocal = 0; }
      copyWG<CP_SPREAD>(a, nb, cx.tid, cx.nthr);
      if (announce) countPiece(cx, nbase != 0);
    });
    cx.auxSend = 0;
    cx.auxRecv = first;          // the first step that receives reads the sender's "I write directly" word
    for (int j = 1; j < n - 1; j++) {
      const int blk = uniformInt(rp.order[(rp.ix + n - j) % n]);
      cx.step(pm, 0, nm, 0, [&]() {
        const uint64_t nbase = uniform64(cx.sh->directBase[rp.next]);
        const bool here = uniform64(cx.sh->auxIn[rp.prev]) == RA_AUX_DIRECT;    // the block already lies in my recvbuff
        char* fwd = nbase ? (char*)nbase + (size_t)blk * nbytes + off : cx.sendSlot(rp.next, 0);
        char* mine = uniformPtr(w.recv + (size_t)blk * nbytes + off);
        RcArgs a; a.ns = 1; a.preIdx = -1; a.post = false; a.opArg = 0;
        if (here) {
          a.src[0] = mine; a.dst[0] = fwd; a.nd = 1; a.nLocal = 0;
          rcWG<int8_t, RA_OP_SUM, RC_SEND>(a, nb, cx.tid, cx.nthr);
        } else {
          a.src[0] = cx.recvSlot(rp.prev, 0);
          a.dst[0] = mine; a.nLocal = 1; a.dst[1] = fwd; a.nd = 2;
          copyWG<CP_FWD>(a, nb, cx.tid, cx.nthr);
        }
        if (announce) countPiece(cx, nbase != 0);
      });
      cx.auxRecv = 0;
    }
    const int blk = uniformInt(rp.order[(rp.ix + 1) % n]);
    cx.step(pm, 0, 0, 0, [&]() {
      if (uniform64(cx.sh->auxIn[rp.prev]) == RA_AUX_DIRECT) return;
      RcArgs a; a.ns = 1; a.nd = 1; a.preIdx = -1; a.post = false; a.opArg = 0;
      a.src[0] = cx.recvSlot(rp.prev, 0); a.dst[0] = uniformPtr(w.recv + (size_t)blk * nbytes + off); a.nLocal = 1;
      copyWG<CP_END>(a, nb, cx.tid, cx.nthr);
    });
    cx.auxRecv = 0;
  }
  cx.auxRecv = 0; cx.auxSend = 0;
}

// ------------------------------------------------------------------ chain ("tree") all-reduce
// Inside a node the reference's tree is a chain in ring order (src/graph/connect.cc:61-62):
// order[0] is the root, order[n-1] the leaf.  Up (kind 0): leaf sends, middles compute
// local + received, the root finishes (src/device/all_reduce.h:269-289).  Down (kind 1): the
// finished chunk travels back (all_reduce.h:318-338).  Two roles per channel so that the up
// and down pipelines overlap: role 0 runs the up steps (the root's up step also starts the
// way down), role 1 runs the down steps.  Round g covers elements [g*chunk, (g+1)*chunk).
template <typename T, int OP>
__device__ __forceinline__ void chainAllReduce(Ctx& cx, const RaWork& w, int role) {
  const int n = uniformInt(cx.nRanks);
  role = uniformInt(role);
  const RingPos rp(cx);
  const bool root = rp.ix == 0, leaf = rp.ix == n - 1;
  const int parent = rp.prev, child = rp.next;
  const T* in = (const T*)w.send;
  T* out = (T*)w.recv;
  const size_t count = uniform64(w.count), chunk = uniform64(w.chunkElems);
  const size_t nRounds = (count + chunk - 1) / chunk;
  // Registered receive buffers: the finished chunk travels down from recvbuff to recvbuff.  The workgroups that send
  // it down are the root's up workgroup and every other rank's down workgroup; the ones that receive it are the down
  // workgroups (announceRecvBuff: word to the parent, then wait for the child's).
  const bool announce = !cx.comm->netMode && count * sizeof(T) >= RA_AG_ANNOUNCE_BYTES && (size_t)cx.ch < nRounds;
  if (announce) {
    if (role == 0) { if (root) announceRecvBuff(cx, w, -1, child); }
    else if (!root) announceRecvBuff(cx, w, parent, leaf ? -1 : child);
  }
  for (size_t g = cx.ch; g < nRounds; g += (size_t)uniformInt(w.nChannels)) {
    const bool first = announce && g == (size_t)cx.ch;
    const size_t off = g * chunk;
    const size_t ne = count - off < chunk ? count - off : chunk;
    cx.stepBytes = (uint32_t)(ne * sizeof(T));       // what a slot of this round carries (the host-staged proxy ships that much)
    if (role == 0) {
      if (leaf) {
        cx.step(0, 0, 1u << parent, 0, [&]() {
          RcArgs a; a.ns = 1; a.nd = 1; a.preIdx = 0; a.post = false; a.opArg = w.opArg;
          a.src[0] = uniformPtr((const char*)(in + off)); a.dst[0] = cx.sendSlot(parent, 0);
          rcWG<T, OP, RC_SEND>(a, ne, cx.tid, cx.nthr);
        });
      } else if (!root) {
        cx.step(1u << child, 0, 1u << parent, 0, [&]() {
          RcArgs a; a.ns = 2; a.nd = 1; a.preIdx = 0; a.post = false; a.opArg = w.opArg;
          a.src[0] = uniformPtr((const char*)(in + off)); a.src[1] = cx.recvSlot(child, 0);
          a.dst[0] = cx.sendSlot(parent, 0);
          rcWG<T, OP, RC_HOP>(a, ne, cx.tid, cx.nthr);
        });
      } else {
        cx.auxSend = first;
        cx.step(1u << child, 0, 1u << child, 1, [&]() {
          const uint64_t nb = uniform64(cx.sh->directBase[child]);
          RcArgs a; a.ns = 2; a.nd = 2; a.preIdx = 0; a.post = true; a.opArg = w.opArg;
          a.src[0] = uniformPtr((const char*)(in + off)); a.src[1] = cx.recvSlot(child, 0);
          a.dst[0] = uniformPtr((char*)(out + off)); a.nLocal = 1;
          a.dst[1] = nb ? (char*)nb + off * sizeof(T) : cx.sendSlot(child, 1);
          rcWG<T, OP, RC_HOP_KEEP>(a, ne, cx.tid, cx.nthr);
          if (announce) countPiece(cx, nb != 0);
        });
        cx.auxSend = 0;
      }
    } else {
      if (root) continue;
      cx.auxRecv = first;
      if (leaf) {
        cx.step(1u << parent, 1, 0, 0, [&]() {
          if (uniform64(cx.sh->auxIn[parent]) == RA_AUX_DIRECT) return;
          RcArgs a; a.ns = 1; a.nd = 1; a.preIdx = -1; a.post = false; a.opArg = 0;
          a.src[0] = cx.recvSlot(parent, 1); a.dst[0] = uniformPtr((char*)(out + off)); a.nLocal = 1;
          copyWG<CP_END>(a, ne * sizeof(T), cx.tid, cx.nthr);
        });
      } else {
        cx.auxSend = first;
        cx.step(1u << parent, 1, 1u << child, 1, [&]() {
          const uint64_t nb = uniform64(cx.sh->directBase[child]);
          const bool here = uniform64(cx.sh->auxIn[parent]) == RA_AUX_DIRECT;     // the chunk already lies in my recvbuff
          char* fwd = nb ? (char*)nb + off * sizeof(T) : cx.sendSlot(child, 1);
          RcArgs a; a.ns = 1; a.preIdx = -1; a.post = false; a.opArg = 0;
          if (here) {
            a.src[0] = uniformPtr((const char*)(out + off)); a.dst[0] = fwd; a.nd = 1; a.nLocal = 0;
            rcWG<int8_t, RA_OP_SUM, RC_SEND>(a, ne * sizeof(T), cx.tid, cx.nthr);
          } else {
            a.src[0] = cx.recvSlot(parent, 1);
            a.dst[0] = uniformPtr((char*)(out + off)); a.nLocal = 1; a.dst[1] = fwd; a.nd = 2;
            copyWG<CP_FWD>(a, ne * sizeof(T), cx.tid, cx.nthr);
          }
          if (announce) countPiece(cx, nb != 0);
        });
        cx.auxSend = 0;
      }
      cx.auxRecv = 0;
    }
  }
  cx.auxRecv = 0; cx.auxSend = 0;
}

// ------------------------------------------------------------------ low-latency one-shot all-reduce
// For small payloads: every rank stores its whole input, as 16-byte {lo, epoch, hi, epoch} lines,
// into every peer's line buffer, then folds the n contributions in rank order as they arrive.
// One network hop, no barriers, no flag words.  Two line buffers alternate by epoch parity; a rank
// cannot get two launches ahead of a peer because finishing a launch needs every peer's lines.
// Role of the reference's LL all-reduce for small sizes (src/device/prims_ll.h, selected below
// 256-512 KiB by src/graph/tuning.cc:632-671), on the all-pairs schedule.
template <typename T>
__device__ __forceinline__ uint64_t llLoadUnit(const T* in, size_t u, size_t count) {
  constexpr int EPU = 8 / sizeof(T);
  const size_t e0 = u * EPU;
  if (e0 + EPU <= count && ((uintptr_t)in & 7) == 0) return ((const uint64_t*)in)[u];
  union { uint64_t w; T e[EPU]; } x; x.w = 0;
  for (int i = 0; i < EPU; i++) if (e0 + i < count) x.e[i] = in[e0 + i];
  return x.w;
}
template <typename T>
__device__ __forceinline__ void llStoreUnit(T* out, size_t u, size_t count, uint64_t w) {
  constexpr int EPU = 8 / sizeof(T);
  const size_t e0 = u * EPU;
  if (e0 + EPU <= count && ((uintptr_t)out & 7) == 0) { ((uint64_t*)out)[u] = w; return; }
  union { uint64_t w; T e[EPU]; } x; x.w = w;
  for (int i = 0; i < EPU; i++) if (e0 + i < count) out[e0 + i] = x.e[i];
}
template <typename T, int OP>
__device__ __forceinline__ uint64_t llRed(uint64_t a, uint64_t b) {
  constexpr int R = (OP == RA_OP_PREMULSUM || OP == RA_OP_SUMPOSTDIV) ? RA_OP_SUM : OP;
  constexpr int EPU = 8 / sizeof(T);
  union { uint64_t w; T e[EPU]; } x, y; x.w = a; y.w = b;
#pragma unroll
  for (int i = 0; i < EPU; i++) x.e[i] = Elem<T, R>::red(x.e[i], y.e[i]);
  return x.w;
}
template <typename T, bool PRE>
__device__ __forceinline__ uint64_t llScale(uint64_t a, uint64_t arg) {
  constexpr int EPU = 8 / sizeof(T);
  union { uint64_t w; T e[EPU]; } x; x.w = a;
#pragma unroll
  for (int i = 0; i < EPU; i++) x.e[i] = PRE ? Scale<T>::pre(x.e[i], arg) : Scale<T>::post(x.e[i], arg);
  return x.w;
}
__device__ __forceinline__ u32x4 llLoadLine(const char* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// Four lines in flight at once (the fold below is latency-bound on these loads when a thread has several units).
__device__ __forceinline__ void llLoadLines4(const char* const (&p)[4], u32x4 (&v)[4]) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc0 sc1\n\t"
      "global_load_dwordx4 %1, %5, off sc0 sc1\n\t"
      "global_load_dwordx4 %2, %6, off sc0 sc1\n\t"
      "global_load_dwordx4 %3, %7, off sc0 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
      : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3])
      : "memory");
}

// Waits until the 16-byte line at lp carries `epoch` in both flag words; false once the communicator gave up.
__device__ __forceinline__ bool llWaitLine(const RaDevComm* comm, const char* lp, uint32_t epoch, bool& dead, u32x4& line) {
  line = llLoadLine(lp);
  if (!dead && (line.y != epoch || line.w != epoch)) {
    const uint64_t t0 = wall_clock64();
    uint32_t spins = 0;
    do {
      __builtin_amdgcn_s_sleep(1);
      line = llLoadLine(lp);
      if ((++spins & 0xff) == 0) {
        const bool asked = __hip_atomic_load(comm->abortFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
        if (asked || wall_clock64() - t0 > comm->timeoutTicks) {
          uint32_t expect = RA_ERR_NONE;
          __hip_atomic_compare_exchange_strong(comm->errorFlag, &expect, asked ? RA_ERR_ABORTED : RA_ERR_TIMEOUT,
                                               __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          dead = true;
          break;
        }
      }
    } while (line.y != epoch || line.w != epoch);
  }
  return !dead;
}

// The fold (or placement) of what arrives, U units per thread and pass with their lines loaded together.
template <typename T, int OP, int U>
__device__ __forceinline__ void llFold(const RaDevComm* comm, const RaWork& w, const T* mine, T* out, bool gather, int buf,
                                       uint32_t epoch, size_t nUnits, size_t gtid, size_t gsize, bool& dead) {
  const int n = comm->nRanks, r = comm->rank;
  for (size_t u0 = gtid; u0 < nUnits; u0 += gsize * U) {
    uint64_t acc[U];
#pragma unroll
    for (int j = 0; j < U; j++) acc[j] = 0;
    for (int q = 0; q < n; q++) {
      uint64_t x[U];
      if (q == r) {
#pragma unroll
        for (int j = 0; j < U; j++) {
          const size_t u = u0 + (size_t)j * gsize;
          x[j] = u < nUnits ? llLoadUnit<T>(mine, u, w.count) : 0;
          if (OP == RA_OP_PREMULSUM) x[j] = llScale<T, true>(x[j], w.opArg);
        }
      } else {
        const char* base = comm->win[r] + raLLOffset(comm->lay, buf, q);
        const char* lp[U];
        u32x4 line[U];
#pragma unroll
        for (int j = 0; j < U; j++) {
          const size_t u = u0 + (size_t)j * gsize;
          lp[j] = base + (u < nUnits ? u : u0) * 16;
        }
        if constexpr (U == 4) llLoadLines4(lp, line);
        else {
#pragma unroll
          for (int j = 0; j < U; j++) line[j] = llLoadLine(lp[j]);
        }
#pragma unroll
        for (int j = 0; j < U; j++) {
          const size_t u = u0 + (size_t)j * gsize;
          if (u < nUnits && (line[j].y != epoch || line[j].w != epoch)) llWaitLine(comm, lp[j], epoch, dead, line[j]);
          x[j] = (uint64_t)line[j].x | ((uint64_t)line[j].z << 32);
        }
      }
#pragma unroll
      for (int j = 0; j < U; j++) {
        const size_t u = u0 + (size_t)j * gsize;
        if (u >= nUnits) continue;
        if (gather) { if (!dead) llStoreUnit<T>(out + (size_t)q * w.count, u, w.count, x[j]); }
        else acc[j] = q == 0 ? x[j] : llRed<T, OP>(acc[j], x[j]);
      }
    }
    if (gather) continue;
#pragma unroll
    for (int j = 0; j < U; j++) {
      const size_t u = u0 + (size_t)j * gsize;
      if (u >= nUnits) continue;
      if (OP == RA_OP_SUMPOSTDIV) acc[j] = llScale<T, false>(acc[j], w.opArg);
      if (!dead) llStoreUnit<T>(out, u, w.count, acc[j]);
    }
  }
}

// All-reduce, reduce-scatter (rank p is sent block p) and all-gather (no fold: every sender's lines are copied
// to its block of recvbuff; launched with T = int8 and a byte count).  All three need every peer's lines to
// finish, which is what keeps any rank from running two launches ahead of another.
// Broadcast (T = int8, byte count) and reduce move data along the root's links only, so they add a presence
// token: every workgroup starts by writing one line to each peer (in the tail of its line buffer, which payloads
// of at most RA_LL_DEFAULT_BYTES never reach) and ends by waiting for each peer's.  A token of launch e says its
// sender has finished launch e-1, so "finishing a launch needs a line of that launch from every peer" holds for
// them too and the two alternating buffers stay safe under any mix of collectives.
template <typename T, int OP>
__global__ __launch_bounds__(RA_LL_THREADS) void raLLAllReduceKernel(const RaDevComm* __restrict__ comm, RaWork w) {
  const int n = comm->nRanks, r = comm->rank;
  if (OP == RA_OP_PREMULSUM && w.opArgPtr) w.opArg = loadScalarArg<T>(w.opArgPtr);
  uint32_t epoch = comm->llEpoch[blockIdx.x] + 1;
  if (epoch == 0) epoch = 1;
  const int buf = (int)(epoch & 1u);
  const T* in = (const T*)w.send;
  T* out = (T*)w.recv;
  const bool scatter = w.coll == RA_COLL_REDUCESCATTER, gather = w.coll == RA_COLL_ALLGATHER;
  const bool bcast = w.coll == RA_COLL_BROADCAST, toRoot = w.coll == RA_COLL_REDUCE;
  const bool rooted = bcast || toRoot;
  const int root = w.root;
  const size_t nUnits = (w.count * sizeof(T) + 7) / 8;
  const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gsize = (size_t)gridDim.x * blockDim.x;
  const size_t tokenLine = RA_LL_TOKEN_LINE + blockIdx.x;
  if (rooted && (int)threadIdx.x < n && (int)threadIdx.x != r) {
    u32x4 line; line.x = 0; line.y = epoch; line.z = 0; line.w = epoch;
    stSys((u32x4*)(comm->win[threadIdx.x] + raLLOffset(comm->lay, buf, r)) + tokenLine, line);
  }
  // 1. my contribution to everyone (broadcast: the root's only; reduce: to the root only)
  const bool sends = !rooted || (bcast ? r == root : r != root);
  for (size_t u = gtid; sends && u < nUnits; u += gsize) {
    uint64_t v = 0;
    if (!scatter) {
      v = llLoadUnit<T>(in, u, w.count);
      if (OP == RA_OP_PREMULSUM) v = llScale<T, true>(v, w.opArg);
    }
    for (int i = 1; i < n; i++) {
      const int p = (r + i) % n;
      if (toRoot && p != root) continue;
      if (scatter) {
        v = llLoadUnit<T>(in + (size_t)p * w.count, u, w.count);
        if (OP == RA_OP_PREMULSUM) v = llScale<T, true>(v, w.opArg);
      }
      u32x4 line; line.x = (uint32_t)v; line.y = epoch; line.z = (uint32_t)(v >> 32); line.w = epoch;
      stSys((u32x4*)(comm->win[p] + raLLOffset(comm->lay, buf, r)) + u, line);
    }
  }
  // 2. fold (or place) what arrives, in rank order
  bool dead = false;
  const T* mine = scatter ? in + (size_t)r * w.count : in;
  const bool folds = !rooted || (toRoot && r == root);
  // one unit per thread and pass while every thread has at most one (payloads up to 128 KiB), four beyond that: the
  // loop is bound by the latency of the line loads (1 MiB at 2 ranks: 23 us one at a time, 16 us four at a time)
  if (folds) {
    // (the scaled sum of the 8-bit floats two at a time: with four, its float temporaries do not fit the registers)
    if (nUnits > gsize) llFold<T, OP, (IsF8<T>::value && OP == RA_OP_PREMULSUM) ? 2 : 4>(comm, w, mine, out, gather, buf, epoch, nUnits, gtid, gsize, dead);
    else llFold<T, OP, 1>(comm, w, mine, out, gather, buf, epoch, nUnits, gtid, gsize, dead);
  }
  if (bcast) {
    for (size_t u = gtid; u < nUnits; u += gsize) {
      if (r == root) {
        if (out != in) llStoreUnit<T>(out, u, w.count, llLoadUnit<T>(in, u, w.count));
      } else {
        u32x4 line;
        llWaitLine(comm, comm->win[r] + raLLOffset(comm->lay, buf, root) + u * 16, epoch, dead, line);
        if (!dead) llStoreUnit<T>(out, u, w.count, (uint64_t)line.x | ((uint64_t)line.z << 32));
      }
    }
  }
  if (rooted && (int)threadIdx.x < n && (int)threadIdx.x != r) {
    u32x4 line;
    llWaitLine(comm, comm->win[r] + raLLOffset(comm->lay, buf, (int)threadIdx.x) + tokenLine * 16, epoch, dead, line);
  }
  // 3. this workgroup has run one more launch (every thread is past its reads of the old value)
  __syncthreads();
  if (threadIdx.x == 0) comm->llEpoch[blockIdx.x] = epoch;
}

// The line protocol along the chain (NCCL_ALGO=Tree with NCCL_PROTO=LL; role of the reference's LL tree all-reduce,
// src/device/all_reduce.h:358-551 runTreeSplit over prims_ll.h): lines travel up the chain in ring order, every rank
// folding its own input with what its child sent (local (op) received, the order of chainAllReduce), the root finishes
// and the result travels back down.  A thread owns the same units on every rank, so a unit's path is one dependent
// chain of 2(n-1) line hops with no barrier anywhere; the up pass of all of a thread's units runs before its down pass.
// Lines of sender q live where the one-hop kernel keeps them, and the launch counter is shared with it, so every
// workgroup exchanges a presence token with EVERY peer as the rooted one-hop collectives do: finishing a launch then
// needs a line of that launch from every peer, which is what keeps the two alternating buffers safe under any mix.
template <typename T, int OP>
__global__ __launch_bounds__(RA_LL_THREADS) void raLLChainKernel(const RaDevComm* __restrict__ comm, RaWork w) {
  const int n = comm->nRanks, r = comm->rank;
  if (OP == RA_OP_PREMULSUM && w.opArgPtr) w.opArg = loadScalarArg<T>(w.opArgPtr);
  uint32_t epoch = comm->llEpoch[blockIdx.x] + 1;
  if (epoch == 0) epoch = 1;
  const int buf = (int)(epoch & 1u);
  const T* in = (const T*)w.send;
  T* out = (T*)w.recv;
  const int8_t* order = comm->ringRanks[0];
  int ix = 0;
  for (int i = 0; i < n; i++) if (order[i] == r) ix = i;
  const int parent = ix > 0 ? order[ix - 1] : -1, child = ix < n - 1 ? order[ix + 1] : -1;
  const size_t nUnits = (w.count * sizeof(T) + 7) / 8;
  const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gsize = (size_t)gridDim.x * blockDim.x;
  const size_t tokenLine = RA_LL_TOKEN_LINE + blockIdx.x;
  if ((int)threadIdx.x < n && (int)threadIdx.x != r) {
    u32x4 line; line.x = 0; line.y = epoch; line.z = 0; line.w = epoch;
    stSys((u32x4*)(comm->win[threadIdx.x] + raLLOffset(comm->lay, buf, r)) + tokenLine, line);
  }
  bool dead = false;
  // up: own input, folded with the child's line, to the parent; the root finishes, keeps the result and starts the way down
  for (size_t u = gtid; u < nUnits; u += gsize) {
    uint64_t v = llLoadUnit<T>(in, u, w.count);
    if (OP == RA_OP_PREMULSUM) v = llScale<T, true>(v, w.opArg);
    if (child >= 0) {
      u32x4 line;
      llWaitLine(comm, comm->win[r] + raLLOffset(comm->lay, buf, child) + u * 16, epoch, dead, line);
      v = llRed<T, OP>(v, (uint64_t)line.x | ((uint64_t)line.z << 32));
    }
    if (parent < 0 && OP == RA_OP_SUMPOSTDIV) v = llScale<T, false>(v, w.opArg);
    const int to = parent >= 0 ? parent : child;
    if (to >= 0 && !dead) {
      u32x4 line; line.x = (uint32_t)v; line.y = epoch; line.z = (uint32_t)(v >> 32); line.w = epoch;
      stSys((u32x4*)(comm->win[to] + raLLOffset(comm->lay, buf, r)) + u, line);
    }
    if (parent < 0 && !dead) llStoreUnit<T>(out, u, w.count, v);
  }
  // down: the parent's line is the result; it goes to recvbuff and on to the child
  for (size_t u = gtid; parent >= 0 && u < nUnits; u += gsize) {
    u32x4 line;
    llWaitLine(comm, comm->win[r] + raLLOffset(comm->lay, buf, parent) + u * 16, epoch, dead, line);
    if (dead) continue;
    if (child >= 0) stSys((u32x4*)(comm->win[child] + raLLOffset(comm->lay, buf, r)) + u, line);
    llStoreUnit<T>(out, u, w.count, (uint64_t)line.x | ((uint64_t)line.z << 32));
  }
  if ((int)threadIdx.x < n && (int)threadIdx.x != r) {
    u32x4 line;
    llWaitLine(comm, comm->win[r] + raLLOffset(comm->lay, buf, (int)threadIdx.x) + tokenLine * 16, epoch, dead, line);
  }
  __syncthreads();
  if (threadIdx.x == 0) comm->llEpoch[blockIdx.x] = epoch;
}

// Waits until the line at lp carries the flag pair (fy, fw).
__device__ __forceinline__ void llWaitPair(const RaDevComm* comm, const char* lp, uint32_t fy, uint32_t fw, bool& dead, u32x4& line) {
  line = llLoadLine(lp);
  if (dead || (line.y == fy && line.w == fw)) return;
  const uint64_t t0 = wall_clock64();
  uint32_t spins = 0;
  do {
    __builtin_amdgcn_s_sleep(1);
    line = llLoadLine(lp);
    if ((++spins & 0xff) == 0) {
      const bool asked = __hip_atomic_load(comm->abortFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
      if (asked || wall_clock64() - t0 > comm->timeoutTicks) {
        uint32_t expect = RA_ERR_NONE;
        __hip_atomic_compare_exchange_strong(comm->errorFlag, &expect, asked ? RA_ERR_ABORTED : RA_ERR_TIMEOUT,
                                             __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        dead = true;
        return;
      }
    }
  } while (line.y != fy || line.w != fw);
}

// The line protocol around the ring (NCCL_ALGO=Ring with NCCL_PROTO=LL; role of the reference's LL ring all-reduce,
// src/device/all_reduce.h:17-199 over prims_ll.h): the payload is cut into n pieces of w.chunkElems elements; piece j is
// seeded at ring position j+1, every hop folds local (op) received, position j finishes it (the order of ringAllReduce,
// src/device/all_reduce.h:82-144), and the finished pieces go round once more.  A thread owns the same offsets of every piece on every
// rank, so there is no barrier: n-1 reducing hops and n-1 copying hops of lines between neighbours.
// A finished piece lands in the line slots its partial sum used earlier in the same launch (the partial was consumed
// before the piece could be finished), so finished lines carry the flag pair (epoch, ~epoch) instead of (epoch, epoch).
// Presence tokens to every peer as in raLLChainKernel.
template <typename T, int OP>
__global__ __launch_bounds__(RA_LL_THREADS) void raLLRingKernel(const RaDevComm* __restrict__ comm, RaWork w) {
  const int n = comm->nRanks, r = comm->rank;
  if (OP == RA_OP_PREMULSUM && w.opArgPtr) w.opArg = loadScalarArg<T>(w.opArgPtr);
  uint32_t epoch = comm->llEpoch[blockIdx.x] + 1;
  if (epoch == 0) epoch = 1;
  const int buf = (int)(epoch & 1u);
  const T* in = (const T*)w.send;
  T* out = (T*)w.recv;
  const int8_t* order = comm->ringRanks[0];
  int pos = 0;
  for (int i = 0; i < n; i++) if (order[i] == r) pos = i;
  const int next = order[(pos + 1) % n], prev = order[(pos + n - 1) % n];
  const size_t nUnits = (w.count * sizeof(T) + 7) / 8;
  const size_t pu = w.chunkElems * sizeof(T) / 8;            // units per piece (pieces are whole 16-byte packs)
  const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gsize = (size_t)gridDim.x * blockDim.x;
  const size_t tokenLine = RA_LL_TOKEN_LINE + blockIdx.x;
  if ((int)threadIdx.x < n && (int)threadIdx.x != r) {
    u32x4 line; line.x = 0; line.y = epoch; line.z = 0; line.w = epoch;
    stSys((u32x4*)(comm->win[threadIdx.x] + raLLOffset(comm->lay, buf, r)) + tokenLine, line);
  }
  bool dead = false;
  const char* mineLines = comm->win[r] + raLLOffset(comm->lay, buf, prev);        // what prev sends me
  u32x4* nextLines = (u32x4*)(comm->win[next] + raLLOffset(comm->lay, buf, r));   // what I send next
  auto local = [&](size_t u) {
    uint64_t v = llLoadUnit<T>(in, u, w.count);
    if (OP == RA_OP_PREMULSUM) v = llScale<T, true>(v, w.opArg);
    return v;
  };
  // reducing hops: step s sends piece pos-1-s (own input at s = 0, else what the last step folded) and folds piece pos-2-s
  for (int s = 0; s < n - 1; s++) {
    const size_t sendBase = (size_t)((pos - 1 - s + 2 * n) % n) * pu, recvBase = (size_t)((pos - 2 - s + 2 * n) % n) * pu;
    for (size_t o = gtid; o < pu; o += gsize) {
      if (s == 0 && sendBase + o < nUnits && !dead) {
        const uint64_t v = local(sendBase + o);
        u32x4 line; line.x = (uint32_t)v; line.y = epoch; line.z = (uint32_t)(v >> 32); line.w = epoch;
        stSys(nextLines + sendBase + o, line);
      }
      const size_t u = recvBase + o;
      if (u >= nUnits) continue;
      u32x4 line;
      llWaitPair(comm, mineLines + u * 16, epoch, epoch, dead, line);
      if (dead) continue;
      uint64_t v = llRed<T, OP>(local(u), (uint64_t)line.x | ((uint64_t)line.z << 32));
      const bool last = s == n - 2;                      // piece pos is finished here
      if (last && OP == RA_OP_SUMPOSTDIV) v = llScale<T, false>(v, w.opArg);
      u32x4 o2; o2.x = (uint32_t)v; o2.y = epoch; o2.z = (uint32_t)(v >> 32); o2.w = last ? ~epoch : epoch;
      stSys(nextLines + u, o2);
      if (last) llStoreUnit<T>(out, u, w.count, v);
    }
  }
  // copying hops: step t receives finished piece pos-1-t, keeps it and passes it on (the last one stays)
  for (int t = 0; t < n - 1; t++) {
    const size_t base = (size_t)((pos - 1 - t + 2 * n) % n) * pu;
    for (size_t o = gtid; o < pu; o += gsize) {
      const size_t u = base + o;
      if (u >= nUnits) continue;
      u32x4 line;
      llWaitPair(comm, mineLines + u * 16, epoch, ~epoch, dead, line);
      if (dead) continue;
      if (t < n - 2) stSys(nextLines + u, line);
      llStoreUnit<T>(out, u, w.count, (uint64_t)line.x | ((uint64_t)line.z << 32));
    }
  }
  if ((int)threadIdx.x < n && (int)threadIdx.x != r) {
    u32x4 line;
    llWaitLine(comm, comm->win[r] + raLLOffset(comm->lay, buf, (int)threadIdx.x) + tokenLine * 16, epoch, dead, line);
  }
  __syncthreads();
  if (threadIdx.x == 0) comm->llEpoch[blockIdx.x] = epoch;
}

// ------------------------------------------------------------------ the all-reduce that allocates nothing
// Same protocol as raLLAllReduceKernel, all state in the kernel arguments: a.lines[p] is rank p's line array, the
// lines of sender q for buffer b start at (b * RA_NA_MAX_RANKS + q) * RA_NA_SENDER_BYTES.  Ranks that share a GPU
// share one array; what sender q stores is the same for every receiver, so they may share q's region too.
// A wait that outlasts a.timeoutTicks, or that sees the call cancelled (a later launch of the same call failed on the
// host), gives up: the thread's outputs stay unwritten and the call's number goes into the GPU's status word, where
// rcclAmdNoAllocAsyncError finds it.
template <typename T, int OP>
__global__ __launch_bounds__(RA_LL_THREADS) void raNoAllocAllReduceKernel(RaNaArgs a) {
  const int n = a.nRanks, r = a.rank;
  const uint32_t epoch = a.epoch;
  const size_t bufOff = (size_t)(epoch & 1u) * RA_NA_MAX_RANKS * RA_NA_SENDER_BYTES;
  const T* in = (const T*)a.send;
  T* out = (T*)a.recv;
  const size_t nUnits = (a.count * sizeof(T) + 7) / 8;
  const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gsize = (size_t)gridDim.x * blockDim.x;
  for (size_t u = gtid; u < nUnits; u += gsize) {
    uint64_t v = llLoadUnit<T>(in, u, a.count);
    if (OP == RA_OP_PREMULSUM) v = llScale<T, true>(v, a.opArg);
    u32x4 line; line.x = (uint32_t)v; line.y = epoch; line.z = (uint32_t)(v >> 32); line.w = epoch;
    for (int i = 1; i < n; i++) {
      const int p = (r + i) % n;
      stSys((u32x4*)(a.lines[p] + bufOff + (size_t)r * RA_NA_SENDER_BYTES) + u, line);
    }
  }
  bool dead = false;
  for (size_t u = gtid; u < nUnits; u += gsize) {
    uint64_t acc = 0;
    for (int q = 0; q < n; q++) {
      uint64_t x;
      if (q == r) {
        x = llLoadUnit<T>(in, u, a.count);
        if (OP == RA_OP_PREMULSUM) x = llScale<T, true>(x, a.opArg);
      } else {
        const char* lp = a.lines[r] + bufOff + (size_t)q * RA_NA_SENDER_BYTES + u * 16;
        u32x4 line = llLoadLine(lp);
        if (!dead && (line.y != epoch || line.w != epoch)) {
          const uint64_t t0 = wall_clock64();
          uint32_t spins = 0;
          do {
            __builtin_amdgcn_s_sleep(1);
            line = llLoadLine(lp);
            if ((++spins & 0xff) == 0) {
              const bool cancelled = __hip_atomic_load(a.words + RA_NA_WORD_CANCEL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == epoch;
              if (cancelled || wall_clock64() - t0 > a.timeoutTicks) {
                uint32_t expect = 0;
                __hip_atomic_compare_exchange_strong(a.words + RA_NA_WORD_FAILED, &expect, epoch, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_SYSTEM);
                dead = true;
                break;
              }
            }
          } while (line.y != epoch || line.w != epoch);
        }
        x = (uint64_t)line.x | ((uint64_t)line.z << 32);
      }
      acc = q == 0 ? x : llRed<T, OP>(acc, x);
    }
    if (OP == RA_OP_SUMPOSTDIV) acc = llScale<T, false>(acc, a.opArg);
    if (!dead) llStoreUnit<T>(out, u, a.count, acc);
  }
}

// ------------------------------------------------------------------ all-reduce in two exchanges (bucket-size payloads)
// The bandwidth-optimal all-pairs all-reduce (2(n-1)/n x S link bytes per rank, like the three-phase FIFO schedule) on the
// machinery of the one-exchange kernel: no FIFOs, no credits, two flag exchanges per launch.  The payload is cut into n
// pieces (piece j is folded by rank j, whole 1 KiB rows), and workgroup b of every rank handles the same slice of every
// piece: (1) it pushes its input of piece p's slice into rank p's window, for every peer p, and raises the first flag;
// (2) when the peers' first flags are up it folds its own piece's slice in rank order, writes the result to recvbuff
// and into every peer's window, and raises the second flag; (3) when the peers' second flags are up it copies their
// finished slices into recvbuff.  A buffer of the window region holds n slots of one piece for the first exchange and
// n for the second.  In place: a slice of recvbuff is written only by the workgroup that has already read it.
template <typename T, int OP>
__device__ __forceinline__ void twoExchangeAllReduce(const RaDevComm* comm, const RaWork& w, uint64_t epoch, LdsInt* aborted) {
  const int n = uniformInt(comm->nRanks), r = uniformInt(comm->rank), tid = threadIdx.x, nthr = blockDim.x, b = blockIdx.x;
  const int buf = (int)(epoch & 1u);
  constexpr size_t ROW = 64 * (16 / sizeof(T));                      // elements of a 1 KiB row
  const size_t count = uniform64(w.count);
  const size_t pe = ((count + n - 1) / n + ROW - 1) / ROW * ROW;   // elements of a piece
  const size_t se = ((pe + gridDim.x - 1) / gridDim.x + ROW - 1) / ROW * ROW;   // ... of it one workgroup handles
  const size_t s0 = (size_t)b * se;
  if (s0 >= pe) return;                                               // the same on every rank: no flags either
  const size_t s1 = s0 + se < pe ? s0 + se : pe;
  auto slice = [&](int j, size_t& off, size_t& ne) {                  // my slice of piece j, clipped to the payload
    const size_t lo = (size_t)j * pe + s0, hi = (size_t)j * pe + s1;
    off = lo < count ? lo : count;
    ne = lo < count ? (hi < count ? hi : count) - lo : 0;
  };
  const T* in = (const T*)w.send;
  T* out = (T*)w.recv;
  const size_t peB = pe * sizeof(T), sB = s0 * sizeof(T);
  const size_t area1 = raOsDataOffset(comm->lay, buf, 0), area2 = area1 + (size_t)n * peB;
  // 1. my input of every peer's piece
  {
    RcArgs pa; pa.ns = n - 1; pa.nd = n - 1; pa.preIdx = -1; pa.post = false; pa.opArg = 0;
    size_t ne0 = 0; bool same = OP != RA_OP_PREMULSUM;
    RC_EACH(i, 0, n - 1,
      const int p = (r + 1 + (i + b) % (n - 1)) % n;                  // workgroups start on different links
      size_t off, ne; slice(p, off, ne);
      if (i == 0) ne0 = ne; else same = same && ne == ne0;
      pa.src[i] = uniformPtr((const char*)(in + off)); pa.dst[i] = uniformPtr(comm->win[p] + area1 + (size_t)r * peB + sB);
    )
    if (same) {
      pairCopyWG<true>(pa, ne0 * sizeof(T), tid, nthr);
    } else {
      for (int i = 0; i < n - 1; i++) {
        const int p = (r + 1 + (i + b) % (n - 1)) % n;
        size_t off, ne; slice(p, off, ne);
        RcArgs a; a.ns = 1; a.nd = 1; a.preIdx = 0; a.post = false; a.opArg = w.opArg;
        a.src[0] = uniformPtr((const char*)(in + off)); a.dst[0] = uniformPtr(comm->win[p] + area1 + (size_t)r * peB + sB);
        rcWG<T, OP, RC_SEND>(a, ne, tid, nthr);
      }
    }
  }
  drainVmem();
  __syncthreads();
  if (tid < n && tid != r) stFlag((uint64_t*)(comm->win[tid] + raOsFlagOffset(comm->lay, buf, b, r)), 2 * epoch - 1);
  // 2. my piece: fold in rank order, result to recvbuff and to everyone
  if (tid < n && tid != r) {
    raSpin(comm, aborted, (const uint64_t*)(comm->win[r] + raOsFlagOffset(comm->lay, buf, b, tid)), 2 * epoch - 1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  }
  if (tid < 64) drainVmem();
  __syncthreads();
  if (!uniformInt(*aborted)) {
    size_t off, ne; slice(r, off, ne);
    RcArgs a; a.ns = n; a.nd = n; a.preIdx = r; a.post = true; a.opArg = w.opArg;
    RC_EACH(j, 0, n, a.src[j] = j == r ? uniformPtr((const char*)(in + off)) : uniformPtr((const char*)comm->win[r] + area1 + (size_t)j * peB + sB);)
    a.dst[0] = uniformPtr((char*)(out + off)); a.nLocal = 1;
    RC_EACH(i, 1, n, a.dst[i] = uniformPtr(comm->win[(r + i) % n] + area2 + (size_t)r * peB + sB);)
    rcWG<T, OP, RC_FOLD_ALL>(a, ne, tid, nthr);
  }
  drainVmem();
  __syncthreads();
  if (tid < n && tid != r) stFlag((uint64_t*)(comm->win[tid] + raOsFlagOffset(comm->lay, buf, b, r)), 2 * epoch);
  // 3. the peers' finished pieces
  if (tid < n && tid != r) {
    raSpin(comm, aborted, (const uint64_t*)(comm->win[r] + raOsFlagOffset(comm->lay, buf, b, tid)), 2 * epoch);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  }
  if (tid < 64) drainVmem();
  __syncthreads();
  if (!uniformInt(*aborted)) {
    RcArgs pa; pa.ns = n - 1; pa.nd = n - 1; pa.preIdx = -1; pa.post = false; pa.opArg = 0; pa.nLocal = n - 1;
    size_t ne0 = 0; bool same = true;
    RC_EACH(i, 0, n - 1,
      const int p = (r + n - 1 - i) % n;
      size_t off, ne; slice(p, off, ne);
      if (i == 0) ne0 = ne; else same = same && ne == ne0;
      pa.src[i] = uniformPtr((const char*)comm->win[r] + area2 + (size_t)p * peB + sB); pa.dst[i] = uniformPtr((char*)(out + off));
    )
    if (same) {
      pairCopyWG<false>(pa, ne0 * sizeof(T), tid, nthr);
    } else {
      for (int i = 0; i < n - 1; i++) {
        const int p = (r + n - 1 - i) % n;
        size_t off, ne; slice(p, off, ne);
        RcArgs a; a.ns = 1; a.nd = 1; a.preIdx = -1; a.post = false; a.opArg = 0; a.nLocal = 1;
        a.src[0] = uniformPtr((const char*)comm->win[r] + area2 + (size_t)p * peB + sB); a.dst[0] = uniformPtr((char*)(out + off));
        copyWG<CP_END>(a, ne * sizeof(T), tid, nthr);
      }
    }
  }
}

template <typename T, int OP>
__global__ __launch_bounds__(1024) void raTwoExchangeKernel(const RaDevComm* __restrict__ comm, RaWork w) {
  __shared__ int aborted;
  const int tid = threadIdx.x, b = blockIdx.x;
  if (OP == RA_OP_PREMULSUM && w.opArgPtr) w.opArg = loadScalarArg<T>(w.opArgPtr);
  const uint64_t epoch = comm->osEpoch[b] + 1;
  if (tid == 0) aborted = 0;
  __syncthreads();
  twoExchangeAllReduce<T, OP>(comm, w, epoch, (LdsInt*)&aborted);
  __syncthreads();
  if (tid == 0) comm->osEpoch[b] = epoch;
}

// ------------------------------------------------------------------ all-reduce in one exchange (mid-size payloads)
// Every rank pushes its whole input into every peer's window and folds the n contributions itself, in rank order, so
// all ranks end with the same bits as the all-pairs schedule gives.  Workgroup b of every rank owns the same slice of
// the payload (whole 16-byte packs; the last one with work also takes the sub-pack tail): it pushes the slice (plain
// packs, write-through), raises one flag per peer, waits for the peers' flags of the same slice and folds.  Nothing
// is waited for before the push: two buffers alternate by launch parity, and a rank cannot get two launches ahead of a
// peer because finishing a launch needs that peer's flags of the launch (the flow control of the low-latency kernel).
// Every workgroup counts every launch (with or without work), so all of them agree on the parity.
// In place: a slice is overwritten by the workgroup that pushed it, after its push.
// Reduce-scatter (rank p is sent block p of sendbuff only; count is per rank) and all-gather (no fold: every sender's
// slice is copied to its block of recvbuff; launched with T = int8 and a byte count) use the same exchange, with the
// same link bytes as their FIFO schedules, so they take it at every size the window region has room for.
template <typename T, int OP>
__global__ __launch_bounds__(1024) void raOneShotKernel(const RaDevComm* __restrict__ comm, RaWork w) {
  __shared__ int abortedVar;
  LdsInt* aborted = (LdsInt*)&abortedVar;
  const int n = uniformInt(comm->nRanks), r = uniformInt(comm->rank), tid = threadIdx.x, nthr = blockDim.x, b = blockIdx.x;
  if (OP == RA_OP_PREMULSUM && w.opArgPtr) w.opArg = loadScalarArg<T>(w.opArgPtr);
  const uint64_t epoch = comm->osEpoch[b] + 1;
  const int buf = (int)(epoch & 1u);
  constexpr size_t EPP = 16 / sizeof(T);
  const size_t count = uniform64(w.count);
  const size_t npacks = count / EPP;
  size_t per = (npacks + gridDim.x - 1) / gridDim.x;
  per = (per + 63) / 64 * 64;                      // whole 1 KiB rows
  if (per == 0) per = 64;
  const size_t last = npacks == 0 ? 0 : (npacks - 1) / per;       // the workgroup that also owns the tail
  size_t e0 = 0, ne = 0;
  if ((size_t)b <= last) {
    const size_t p0 = (size_t)b * per, p1 = p0 + per < npacks ? p0 + per : npacks;
    e0 = p0 * EPP;
    ne = ((size_t)b == last ? count : p1 * EPP) - e0;
  }
  const bool scatter = w.coll == RA_COLL_REDUCESCATTER, gather = w.coll == RA_COLL_ALLGATHER;
  const T* in = (const T*)w.send + e0;
  T* out = (T*)w.recv + e0;
  const size_t boff = e0 * sizeof(T);
  if (tid == 0) *aborted = 0;
  if (ne) {
    // 1. my slice to everyone: entry i is peer (r + 1 + (i + b) % (n - 1)) % n, so that workgroups start on different links
    RcArgs rc; rc.ns = 1; rc.nd = n - 1; rc.preIdx = 0; rc.post = false; rc.nLocal = 0; rc.opArg = w.opArg;
    rc.src[0] = uniformPtr((const char*)in);
    RC_EACH(i, 0, n - 1,
      const int p = (r + 1 + (i + b) % (n - 1)) % n;
      rc.dst[i] = uniformPtr(comm->win[p] + raOsDataOffset(comm->lay, buf, r) + boff);
      if (scatter) rc.src[i] = uniformPtr((const char*)(in + (size_t)p * count));
    )
    __syncthreads();                               // `aborted` is initialised
    if (scatter && OP != RA_OP_PREMULSUM) {
      rc.ns = n - 1;                               // n-1 independent copies, block p to rank p
      pairCopyWG<true>(rc, ne * sizeof(T), tid, nthr);
    } else if (scatter) {
      for (int i = 0; i < n - 1; i++) {
        const int p = (r + 1 + (i + b) % (n - 1)) % n;
        RcArgs one; one.ns = 1; one.nd = 1; one.preIdx = 0; one.post = false; one.nLocal = 0; one.opArg = w.opArg;
        one.src[0] = uniformPtr((const char*)(in + (size_t)p * count));
        one.dst[0] = uniformPtr(comm->win[p] + raOsDataOffset(comm->lay, buf, r) + boff);
        rcWG<T, OP, RC_SEND>(one, ne, tid, nthr);
      }
    } else if (OP == RA_OP_PREMULSUM) {
      rcWG<T, OP, RC_SEND>(rc, ne, tid, nthr);     // (one source into n - 1 windows: the plain loop, or 1 -> 1 with two ranks)
    } else {
      copyWG<CP_SPREAD>(rc, ne * sizeof(T), tid, nthr);
    }
    drainVmem();                                   // every wave: its stores are at their destination
    __syncthreads();
    // (flag values: 2 x launch - 1 after the first exchange of a launch, 2 x launch after the second, which only the
    // two-exchange all-reduce has; a line is reused two launches later, so its value only ever grows)
    if (tid < n && tid != r) stFlag((uint64_t*)(comm->win[tid] + raOsFlagOffset(comm->lay, buf, b, r)), 2 * epoch - 1);
    // 2. the peers' slices
    if (tid < n && tid != r) {
      raSpin(comm, aborted, (const uint64_t*)(comm->win[r] + raOsFlagOffset(comm->lay, buf, b, tid)), 2 * epoch - 1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    if (tid < 64) drainVmem();
    __syncthreads();
    // 3. the fold, in rank order (all-gather: the copies)
    if (!uniformInt(*aborted)) {
      if (gather) {
        // every peer's slice to its block of recvbuff, then my own from sendbuff unless it is already there
        RcArgs pa; pa.ns = n - 1; pa.nd = n - 1; pa.preIdx = -1; pa.post = false; pa.nLocal = n - 1; pa.opArg = 0;
        RC_EACH(i, 0, n - 1,
          const int p = (r + 1 + i) % n;
          pa.src[i] = uniformPtr((const char*)comm->win[r] + raOsDataOffset(comm->lay, buf, p) + boff);
          pa.dst[i] = uniformPtr((char*)out + (size_t)p * count * sizeof(T));
        )
        pairCopyWG<false>(pa, ne * sizeof(T), tid, nthr);
        char* mine = uniformPtr((char*)out + (size_t)r * count * sizeof(T));
        if (mine != (const char*)in) {
          RcArgs a; a.ns = 1; a.nd = 1; a.preIdx = -1; a.post = false; a.nLocal = 1; a.opArg = 0;
          a.src[0] = uniformPtr((const char*)in); a.dst[0] = mine;
          copyWG<CP_END>(a, ne * sizeof(T), tid, nthr);
        }
      } else {
        RcArgs a; a.ns = n; a.nd = 1; a.preIdx = r; a.post = true; a.nLocal = 1; a.opArg = w.opArg;
        RC_EACH(j, 0, n,
          a.src[j] = j == r ? uniformPtr((const char*)(scatter ? in + (size_t)r * count : in))
                            : uniformPtr((const char*)comm->win[r] + raOsDataOffset(comm->lay, buf, j) + boff);
        )
        a.dst[0] = uniformPtr((char*)out);
        rcWG<T, OP, RC_FOLD_ONE>(a, ne, tid, nthr);
      }
    }
  }
  __syncthreads();
  if (tid == 0) comm->osEpoch[b] = epoch;
}

// ------------------------------------------------------------------ the collective kernels
// One kernel per schedule, everything inlined into it: no calls, no stack frame.
// grid = roles * nChannels workgroups: blockIdx.x = role * nChannels + channel.  Exception (w.pairRoles, the all-pairs
// all-reduce with two ranks): there a reducer moves twice the bytes of a scatter or gather step (two sources and two
// destinations against one pair), so a scatter or gather workgroup serves TWO channels and all workgroups stay equally
// busy: nChannels/2 scatter workgroups, nChannels reducers, nChannels/2 gather workgroups.
enum { RA_K_DIRECT_ALLREDUCE, RA_K_DIRECT_REDUCESCATTER, RA_K_DIRECT_REDUCE, RA_K_RING_ALLREDUCE, RA_K_RING_REDUCESCATTER,
       RA_K_CHAIN_ALLREDUCE };
template <typename T, int OP, int K>
__global__ __launch_bounds__(1024) void raCollKernel(const RaDevComm* __restrict__ comm, RaWork w) {
  __shared__ ChanShared shA, shB;
  LdsChan* const sh = RA_LDS_CHAN(shA);
  if (OP == RA_OP_PREMULSUM && w.opArgPtr) w.opArg = loadScalarArg<T>(w.opArgPtr);
  Ctx cx;
  if constexpr (K == RA_K_DIRECT_ALLREDUCE) {
    LdsChan* const sh1 = RA_LDS_CHAN(shB);
    int role = blockIdx.x / w.nChannels, ch0 = blockIdx.x % w.nChannels, nCtx = 1;
    if (w.pairRoles == 2) {
      // one workgroup per channel walks all three roles (directAllReduce, role 3)
      role = 3; ch0 = (int)blockIdx.x;
    } else if (w.pairRoles) {
      // all-pairs all-reduce, two ranks
      const int half = w.nChannels / 2, b = (int)blockIdx.x;
      if (b >= half && b < half + w.nChannels) { role = 1; ch0 = b - half; }
      else { role = b < half ? 0 : 2; ch0 = 2 * (b < half ? b : b - half - w.nChannels); nCtx = 2; }
    }
    role = uniformInt(role); ch0 = uniformInt(ch0); nCtx = uniformInt(nCtx);
    if (nCtx == 2) Ctx::loadChannel(comm, sh1, ch0 + 1);
    cx.init(comm, sh, ch0, w.stepBytes);
    directAllReduce<T, OP>(cx, sh, sh1, ch0, nCtx, w, role);
    __syncthreads();
    Ctx::storeChannel(comm, sh, ch0, role);
    if (nCtx == 2) Ctx::storeChannel(comm, sh1, ch0 + 1, role);
    return;
  }
  const int role = blockIdx.x / w.nChannels, ch = blockIdx.x % w.nChannels;
  cx.init(comm, sh, ch, w.stepBytes);
  if constexpr (K == RA_K_DIRECT_REDUCESCATTER) directReduceScatter<T, OP>(cx, w, role);
  else if constexpr (K == RA_K_DIRECT_REDUCE) directReduce<T, OP>(cx, w, role);
  else if constexpr (K == RA_K_RING_ALLREDUCE) ringAllReduce<T, OP>(cx, w);
  else if constexpr (K == RA_K_RING_REDUCESCATTER) ringReduceScatter<T, OP>(cx, w);
  else chainAllReduce<T, OP>(cx, w, role);
  cx.finish(role);
}
