// Shared device/host helpers for the gfx950 (CDNA4, MI355X) kernels.
// Wave = 64 lanes everywhere in this tree; nothing here is portable to 32-wide hardware.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pseg_amd.h"

namespace pseg {

// ---------------------------------------------------------------- error plumbing (host)
// error codes: PSEG_OK / PSEG_ERR_* from include/pseg_amd.h
void set_error(const char* fmt, ...);
const char* last_error();
int launch_col_reduce(const float* part, int rows, int C, float* out, int accumulate, hipStream_t st);

#define PSEG_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      ::pseg::set_error(__VA_ARGS__);      \
      return PSEG_ERR_ARG;         \
    }                                      \
  } while (0)

#define PSEG_LAUNCH_CHECK()                                                          \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) {                                                         \
      ::pseg::set_error("%s:%d launch failed: %s", __FILE__, __LINE__,               \
                        hipGetErrorString(e__));                                     \
      return PSEG_ERR_HIP;                                                   \
    }                                                                                \
  } while (0)

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Exact n / d for 0 <= n < 2^31, 1 <= d < 2^31 by one 32x32->64 multiply and a shift
// (mul = ceil(2^(31+s)/d), s = ceil(log2 d); error term e = mul*d - 2^(31+s) < 2^s, so n*e < 2^(31+s)).
struct FastDiv {
  uint32_t mul, shift, d;
#ifdef __HIPCC__
  __host__ __device__
#endif
  FastDiv() : mul(0), shift(0), d(1) {}
#ifdef __HIPCC__
  __host__ __device__
#endif
  explicit FastDiv(uint32_t dd) : d(dd) {
    uint32_t s = 0;
    while ((1ull << s) < dd) ++s;
    shift = 31 + s;
    mul = (uint32_t)(((1ull << shift) + dd - 1) / dd);
  }
#ifdef __HIPCC__
  __host__ __device__
#endif
  inline uint32_t div(uint32_t n) const {
    return (uint32_t)(((unsigned long long)n * mul) >> shift);
  }
};

// ---------------------------------------------------------------- device helpers
#ifdef __HIPCC__

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// Buffer resource (V#) over a raw byte range.  Out-of-range loads return 0 and
// out-of-range stores are dropped by the hardware range check: this is what makes the
// implicit-GEMM zero padding and the ragged tile edges free and fault-proof.
// The inputs must be wave-uniform (kernel arguments are).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}

// Byte offset that is out of range for every tensor we accept (host enforces < 2 GiB).
// Raise a device-wide max|x| scalar (bit pattern of a non-negative float: unsigned order == float order).  Thousands of
// blocks publish into ONE address; a device-scope atomic is resolved at the memory side and they serialise there (+15 us
// on a 10 us BatchNorm pass).  A relaxed agent-scope LOAD first: only a block that would actually raise the value pays
// for the atomic (a stale read can only be too low -- then the atomic runs and nothing is lost).
__device__ __forceinline__ void publish_amax(unsigned* amax, float m) {
  if (!(m > 0.f)) return;
  const unsigned bits = __builtin_bit_cast(unsigned, m);
  if (bits > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax, bits);
}

constexpr uint32_t kOOB = 0x80000000u;

// Wave priorities by role.  The instruction arbiter of a SIMD serves its OLDEST wave first, and in the two-stream backward
// pass every kernel of the serial chain starts as the youngest wave on CUs that hold blocks of a long-running weight
// gradient: a BatchNorm reduction took 83 us beside one against 18 alone.  Helper passes of the serial chain (BatchNorm,
// pooling / resize, depthwise convs, copies) raise their waves to priority 3 on entry; data gradients run at 1
// (set_wave_prio, conv_mfma.hip); weight gradients and the slab reductions on their stream stay at 0.
// -DPSEG_NO_PRIO=1 builds the library without any of it (A/B measurements: PSEG_LIB_PATH selects the library file).
#if defined(PSEG_NO_PRIO) && PSEG_NO_PRIO
#define PSEG_HELPER_PRIO() ((void)0)
#else
#define PSEG_HELPER_PRIO() __builtin_amdgcn_s_setprio(3)
#endif


__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
  return __builtin_bit_cast(f32x4, v);
}

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

#endif  // __HIPCC__

}  // namespace pseg
