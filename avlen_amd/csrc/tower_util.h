// Device helpers shared by the fused tower kernels (tower_head.hip: stem + layers 1-2; tower_tail.hip: layers 3-4): packed bf16
// register tiles, ReLU on packed pairs, LDS-only barriers, per-lane statistics, DPP row sums.
#pragma once
#include <hip/hip_runtime.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

// four bf16 values packed in two 32-bit registers (a bf16x4 vector may be kept one element per register: 2x the pressure)
struct P4 { unsigned lo, hi; };
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ unsigned pack2(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }   // one v_cvt_pk_bf16_f32
__device__ __forceinline__ P4 pack4(float a, float b, float c, float d) { return P4{pack2((f32x2){a, b}), pack2((f32x2){c, d})}; }
__device__ __forceinline__ f32x2 unlo(const P4& p) { return (f32x2){__uint_as_float(p.lo << 16), __uint_as_float(p.lo & 0xffff0000u)}; }
__device__ __forceinline__ f32x2 unhi(const P4& p) { return (f32x2){__uint_as_float(p.hi << 16), __uint_as_float(p.hi & 0xffff0000u)}; }
typedef __attribute__((ext_vector_type(2))) short i16x2;
// bf16 pair -> relu of both halves: as int16 a negative float is a negative integer (v_pk_max_i16 against 0; -0.0 -> +0.0)
__device__ __forceinline__ unsigned relu_pk(unsigned v) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, v), (i16x2){0, 0}));
}
// workgroup barrier that orders LDS traffic only: weight prefetches (global loads) stay in flight across it -- __syncthreads()
// would drain them (vmcnt(0)); the kernel's only global stores are its last statements
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// per-lane statistics of one accumulator tile, two channels per instruction (v_pk_add_f32 / v_pk_fma_f32)
__device__ __forceinline__ void stat16(const f32x4& v, f32x2& sA, f32x2& sB, f32x2& qA, f32x2& qB) {
  const f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
  sA += a; sB += b;
  qA = __builtin_elementwise_fma(a, a, qA); qB = __builtin_elementwise_fma(b, b, qB);
}
// 32-channel stages: channels (0, 1) and (2, 3) of a tile are the two GroupNorm groups of the lane -> horizontal pairs
__device__ __forceinline__ void stat32(const f32x4& v, float& g0, float& g1, float& h0, float& h1) {
  g0 += v[0] + v[1]; g1 += v[2] + v[3];
  h0 = __builtin_fmaf(v[1], v[1], __builtin_fmaf(v[0], v[0], h0)); h1 = __builtin_fmaf(v[3], v[3], __builtin_fmaf(v[2], v[2], h1));
}
// sum over the 16 lanes of a DPP row (the 16 pixels of an MFMA tile column group): 4 v_add_f32 with DPP operands, no LDS
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));   // row_mirror
  return v;
}


}  // namespace
