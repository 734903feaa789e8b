// Shared device/host helpers for libmmd (gfx950 / CDNA4 only - no CUDA compat, no dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/mmd.h"   // the C ABI: every extern "C" definition is checked against its declaration

#define MMD_F32 0
#define MMD_BF16 1

#define MMD_OK 0
#define MMD_ERR_ARG (-1)
#define MMD_ERR_LAUNCH (-2)
#define MMD_ERR_UNSUPPORTED (-3)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// ----------------------------------------------------------------------------- error plumbing
int mmd_set_error(int code, const char* fmt, ...);
int mmd_check_launch(const char* what);

#define MMD_REQUIRE(cond, ...)                                  \
  do {                                                          \
    if (!(cond)) return mmd_set_error(MMD_ERR_ARG, __VA_ARGS__); \
  } while (0)

// ----------------------------------------------------------------------------- bf16 <-> f32
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t bits16) { return __uint_as_float(bits16 << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {   // round-to-nearest-even
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0u;            // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return f32_to_bf16_bits(lo) | (f32_to_bf16_bits(hi) << 16);
}

// Element traits: T = float or __bf16.  A "vec" is always 16 bytes.
template <typename T> struct Elt;
template <> struct Elt<float> {
  static constexpr int EPV = 4;      // elements per 16-byte vec
  static constexpr int DT = MMD_F32;
  __device__ static __forceinline__ void unpack(const u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v[i]);
  }
  __device__ static __forceinline__ u32x4 pack(const float* f) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __float_as_uint(f[i]);
    return v;
  }
  __device__ static __forceinline__ float ld(const void* p, int64_t i) { return ((const float*)p)[i]; }
  __device__ static __forceinline__ void st(void* p, int64_t i, float v) { ((float*)p)[i] = v; }
};
template <> struct Elt<__bf16> {
  static constexpr int EPV = 8;
  static constexpr int DT = MMD_BF16;
  __device__ static __forceinline__ void unpack(const u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(v[i] << 16);
      f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
  }
  __device__ static __forceinline__ u32x4 pack(const float* f) {   // RNE; lowers to v_cvt_pk_bf16_f32
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (__bf16)f[i];
    return __builtin_bit_cast(u32x4, t);
  }
  __device__ static __forceinline__ float ld(const void* p, int64_t i) {
    return bf16_bits_to_f32(((const uint16_t*)p)[i]);
  }
  __device__ static __forceinline__ void st(void* p, int64_t i, float v) {
    ((__bf16*)p)[i] = (__bf16)v;
  }
};

// SiLU with v_rcp_f32 (1 ulp) instead of an IEEE division: 5 VALU instructions per element where the division sequence took 14.  Every
// kernel that applies SiLU in the forward uses this one function, so fused and unfused paths stay bitwise equal.
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// 64-lane butterfly reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Sum over the 32 lanes of each half-wave without LDS traffic or selects: four DPP row rotations give every lane of a 16-lane row the
// row total, row_bcast:15 then adds the total of rows 0 / 2 into rows 1 / 3.  Valid in lanes 16-31 (half 0) and 48-63 (half 1).
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWS, 0xf, false));
}
__device__ __forceinline__ float halfwave_total(float v) {
  v = dpp_add<0x128, 0xf>(v);      // row_ror:8
  v = dpp_add<0x124, 0xf>(v);      // row_ror:4
  v = dpp_add<0x122, 0xf>(v);      // row_ror:2
  v = dpp_add<0x121, 0xf>(v);      // row_ror:1
  return dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
}


static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel: the launchers remember it per device slot
// (the only process-lifetime state of the library besides the read-only zero pages; a benign race sets it twice).
#define MMD_MAX_DEVICES 16
static inline int mmd_device_slot() {
  int d = 0;
  (void)hipGetDevice(&d);
  return d & (MMD_MAX_DEVICES - 1);
}
