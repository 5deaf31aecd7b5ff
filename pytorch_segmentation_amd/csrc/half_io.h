// fp16 storage helpers for the half-precision (`-mp`) path: every kernel computes in fp32 and only the HBM / LDS images of
// activations are 2 bytes per element.  A tensor is [M pixels][C channels] of _Float16 with pixel stride ld (elements);
// C % 8 == 0, ld % 8 == 0 and a 16-byte aligned base, so a lane can always move 8 channels (16 bytes) at a time and the
// 4-channel units of the fp32 kernels (8 bytes here) stay naturally aligned.
#pragma once
#include "common.h"

namespace pseg {

#ifdef __HIPCC__

typedef _Float16 half_t;
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
typedef int i32x2v __attribute__((ext_vector_type(2)));

// 4 consecutive channels <-> f32x4 (conversion to half rounds to nearest even: v_cvt_f16_f32)
__device__ __forceinline__ f32x4 ldv4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ldv4(const half_t* p) {
  return __builtin_convertvector(*reinterpret_cast<const f16x4v*>(p), f32x4);
}
__device__ __forceinline__ void stv4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void stv4(half_t* p, f32x4 v) {
  *reinterpret_cast<f16x4v*>(p) = __builtin_convertvector(v, f16x4v);
}

// the same through a buffer resource (range-checked: out-of-range loads give 0, stores are dropped); offsets in BYTES
template <typename T>
__device__ __forceinline__ f32x4 buf_ldv4(__amdgpu_buffer_rsrc_t r, int voff, int soff);
template <>
__device__ __forceinline__ f32x4 buf_ldv4<float>(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
template <>
__device__ __forceinline__ f32x4 buf_ldv4<half_t>(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_convertvector(__builtin_bit_cast(f16x4v, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0)), f32x4);
}
template <typename T>
__device__ __forceinline__ void buf_stv4(f32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff);
// 128-bit buffer stores carry their whole offset in the VGPR (soffset = 0).  A store of more than 64 bits must not have its
// data registers overwritten by the next VALU instruction; the compiler inserts the wait state for that -- except when the
// store has an SGPR soffset, where its hazard model (GCNHazardRecognizer: "this hazard only exists if the instruction is not
// using a register in the soffset field") assumes the hardware needs none.  On gfx950 it does: `buffer_store_dwordx4 v[0:3],
// v85, s[24:27], s67 offen` followed by `v_sub_f32 v0, ...` wrote the NEW v0 (an fp32 bit pattern where two packed halves
// belonged) for lanes 12-15 of every 16, in ~10 % of the blocks of a BatchNorm forward pass at M = 65536, C = 64 (round 4,
// profiles/EXPERIMENTS.md 0.13).  The same instruction pair sat in bn_act_bwd_apply_kernel (fp32 and fp16), where no test and
// no bit-for-bit soak ever caught it misbehaving; it is gone from every kernel now (tools/scan_store_hazard.py).
template <>
__device__ __forceinline__ void buf_stv4<float>(f32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), r, voff + soff, 0, 0);
}
template <>
__device__ __forceinline__ void buf_stv4<half_t>(f32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2v, __builtin_convertvector(v, f16x4v)), r, voff, soff, 0);
}

// V consecutive channels (4 or 8) <-> V floats.  The fp16 passes move 16 bytes per lane with V = 8 (an 8-byte access runs at
// 0.54-0.70 of the 16-byte rate on this memory system); fp32 tensors keep V = 4.
template <int V>
using fvec = float __attribute__((ext_vector_type(V)));

template <int V, typename T>
__device__ __forceinline__ fvec<V> ldvec(const T* p) {
  if constexpr (V == 4) {
    return ldv4(p);
  } else if constexpr (sizeof(T) == 2) {
    return __builtin_convertvector(*reinterpret_cast<const f16x8v*>(p), fvec<8>);
  } else {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    return fvec<8>{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  }
}
template <int V, typename T>
__device__ __forceinline__ void stvec(T* p, fvec<V> v) {
  if constexpr (V == 4) {
    stv4(p, v);
  } else if constexpr (sizeof(T) == 2) {
    *reinterpret_cast<f16x8v*>(p) = __builtin_convertvector(v, f16x8v);
  } else {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
}
template <int V, typename T>
__device__ __forceinline__ fvec<V> buf_ldvec(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  if constexpr (V == 4) {
    return buf_ldv4<T>(r, voff, soff);
  } else {
    static_assert(sizeof(T) == 2, "eight channels per lane: fp16 tensors only");
    return __builtin_convertvector(__builtin_bit_cast(f16x8v, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)), fvec<8>);
  }
}
template <int V, typename T>
__device__ __forceinline__ void buf_stvec(fvec<V> v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  if constexpr (V == 4) {
    buf_stv4<T>(v, r, voff, soff);
  } else {
    static_assert(sizeof(T) == 2, "eight channels per lane: fp16 tensors only");
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, __builtin_convertvector(v, f16x8v)), r, voff + soff, 0, 0);   // (see buf_stv4<float>)
  }
}

#endif  // __HIPCC__

}  // namespace pseg
