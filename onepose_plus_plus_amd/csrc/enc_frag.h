// Shared device helpers of the fused encoder-layer kernels (enc_chain.hip, enc_layer64.hip): the exact bf16x3 operand split
// and the LDS operand-tile geometry.
#pragma once
#include "opp_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

// operand tile row stride for K values per row: 48 B per 8 k + 16 B (the 16 rows of a ds_read_b128 lane group fall on 16
// distinct 4-bank groups)
__host__ __device__ constexpr int a_stride_bytes(int K) { return K * 6 + 16; }

// one level of the exact bf16x3 split on a pair of values: p = (bf16_rne(a), bf16_rne(b)) packed, residuals exact
__device__ __forceinline__ unsigned split_lvl(float a, float b, float& ra, float& rb) {
  const f32x2 t = {a, b};
  const unsigned p = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2));
  ra = a - __uint_as_float(p << 16);
  rb = b - __uint_as_float(p & 0xffff0000u);
  return p;
}
// two values -> one packed dword per part
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  float r0, r1, s0, s1, u0, u1;
  hi = split_lvl(a, b, r0, r1);
  mid = split_lvl(r0, r1, s0, s1);
  lo = split_lvl(s0, s1, u0, u1);
}
__device__ __forceinline__ void split8(const float4 v0, const float4 v1, u32x4& hi, u32x4& mid, u32x4& lo) {
  unsigned h[4], m[4], l[4];
  split2(v0.x, v0.y, h[0], m[0], l[0]);
  split2(v0.z, v0.w, h[1], m[1], l[1]);
  split2(v1.x, v1.y, h[2], m[2], l[2]);
  split2(v1.z, v1.w, h[3], m[3], l[3]);
  hi = u32x4{h[0], h[1], h[2], h[3]};
  mid = u32x4{m[0], m[1], m[2], m[3]};
  lo = u32x4{l[0], l[1], l[2], l[3]};
}
// eight consecutive k of one row -> [hi x8 | mid x8 | lo x8] at dst (48 B, 16-byte aligned LDS)
__device__ __forceinline__ void split8_store(const float4 v0, const float4 v1, char* dst) {
  u32x4 hi, mid, lo;
  split8(v0, v1, hi, mid, lo);
  *reinterpret_cast<u32x4*>(dst) = hi;
  *reinterpret_cast<u32x4*>(dst + 16) = mid;
  *reinterpret_cast<u32x4*>(dst + 32) = lo;
}

}  // namespace
