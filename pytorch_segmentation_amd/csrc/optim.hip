// Fused optimiser steps over the flat parameter arena (one launch for the whole model) on gfx950.
// The reference delegates the update to pytorch_modules.utils.Trainer (train.py:61-72) with torch.optim
// semantics selected by the --adam flag (train.py:94); these kernels follow torch.optim.SGD / Adam(W)
// element for element.  grad_scale folds the data-parallel mean (1/world) and 1/accumulate into the same pass.
#include "common.h"

#include <math.h>

namespace pseg {

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mbuf,
                                                  long long n, float lr, float mu, float wd, int nesterov, float gscale,
                                                  int first) {
  const long long n4 = n / 4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    f32x4 w = *reinterpret_cast<f32x4*>(p + i * 4);
    f32x4 d = *reinterpret_cast<const f32x4*>(g + i * 4) * gscale;
    if (wd != 0.f) d += wd * w;
    if (mu != 0.f) {
      f32x4 b = first ? d : mu * *reinterpret_cast<f32x4*>(mbuf + i * 4) + d;
      *reinterpret_cast<f32x4*>(mbuf + i * 4) = b;
      d = nesterov ? d + mu * b : b;
    }
    *reinterpret_cast<f32x4*>(p + i * 4) = w - lr * d;
  }
  if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) {
    const long long i = n4 * 4 + threadIdx.x;
    float w = p[i];
    float d = g[i] * gscale;
    if (wd != 0.f) d += wd * w;
    if (mu != 0.f) {
      const float b = first ? d : mu * mbuf[i] + d;
      mbuf[i] = b;
      d = nesterov ? d + mu * b : b;
    }
    p[i] = w - lr * d;
  }
}

__device__ __forceinline__ void adam_one(float& w, float gr, float& m, float& v, float lr, float b1, float b2, float eps,
                                         float wd, int decoupled, float gscale, float step_size, float inv_sqrt_bc2) {
  float d = gr * gscale;
  if (wd != 0.f) {
    if (decoupled) w *= (1.f - lr * wd);
    else d += wd * w;
  }
  m = b1 * m + (1.f - b1) * d;
  v = b2 * v + (1.f - b2) * d * d;
  const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
  w -= step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float lr, float b1, float b2,
                                                   float eps, float wd, int decoupled, float gscale, float step_size,
                                                   float inv_sqrt_bc2) {
  const long long n4 = n / 4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    f32x4 w = *reinterpret_cast<f32x4*>(p + i * 4);
    const f32x4 gr = *reinterpret_cast<const f32x4*>(g + i * 4);
    f32x4 mm = *reinterpret_cast<f32x4*>(m + i * 4);
    f32x4 vv = *reinterpret_cast<f32x4*>(v + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float we = w[e], me = mm[e], ve = vv[e];
      adam_one(we, gr[e], me, ve, lr, b1, b2, eps, wd, decoupled, gscale, step_size, inv_sqrt_bc2);
      w[e] = we;
      mm[e] = me;
      vv[e] = ve;
    }
    *reinterpret_cast<f32x4*>(p + i * 4) = w;
    *reinterpret_cast<f32x4*>(m + i * 4) = mm;
    *reinterpret_cast<f32x4*>(v + i * 4) = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) {
    const long long i = n4 * 4 + threadIdx.x;
    adam_one(p[i], g[i], m[i], v[i], lr, b1, b2, eps, wd, decoupled, gscale, step_size, inv_sqrt_bc2);
  }
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ x, long long n, float value) {
  const long long n4 = n / 4;
  const f32x4 v4 = {value, value, value, value};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
    *reinterpret_cast<f32x4*>(x + i * 4) = v4;
  if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) x[n4 * 4 + threadIdx.x] = value;
}

// max |x| of an [M][C] block with row stride ld, published as the bit pattern of the (non-negative) float via atomicMax
// (unsigned order == float order for non-negative values): what the fp16-limb conv kernels scale their operands by.
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, long long ld, long long M, int C,
                                                   unsigned* __restrict__ out) {
  const long long total = M * C;
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / C;
    m = fmaxf(m, fabsf(x[r * ld + (i - r * C)]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) publish_amax(out, m);
}

// every filter of a model in ONE launch: jobs[j] = {address of the floats, count, first block of job j}; out[j] = max|x|
// (overwritten, not accumulated: each block first reduces to one value, block 0 of a job zeroes nothing -- the outputs
// are zeroed by the same launch's predecessor memset on the host side)
__global__ __launch_bounds__(256) void amax_batch_kernel(const long long* __restrict__ jobs, int n, unsigned* __restrict__ out) {
  const long long b = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid * 3 + 2] <= b) lo = mid;
    else hi = mid - 1;
  }
  const float* x = reinterpret_cast<const float*>(jobs[lo * 3]);
  const long long cnt = jobs[lo * 3 + 1];
  const long long first = jobs[lo * 3 + 2];
  const long long nblk = (lo + 1 < n ? jobs[(lo + 1) * 3 + 2] : (long long)gridDim.x) - first;
  float m = 0.f;
  for (long long i = (b - first) * 256 + threadIdx.x; i < cnt; i += nblk * 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) publish_amax(out + lo, m);
}

static int grid_for(long long n) {
  long long b = (n / 4 + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace pseg

using namespace pseg;

extern "C" {

int pseg_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum,
                  float weight_decay, int nesterov, float grad_scale, int first_step, void* stream) {
  PSEG_REQUIRE(param && grad && n > 0, "sgd_step: bad argument");
  PSEG_REQUIRE(momentum == 0.f || momentum_buf, "sgd_step: momentum needs a buffer");
  PSEG_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)momentum_buf) & 15) == 0, "sgd_step: 16-byte alignment");
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, momentum_buf,
                     (long long)n, lr, momentum, weight_decay, nesterov, grad_scale, first_step);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

int pseg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int decoupled, float grad_scale, int step, void* stream) {
  PSEG_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adam_step: bad argument");
  PSEG_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
               "adam_step: 16-byte alignment");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                     (long long)n, lr, beta1, beta2, eps, weight_decay, decoupled, grad_scale, (float)(lr / bc1),
                     (float)(1.0 / sqrt(bc2)));
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

int pseg_amax(const float* x, int64_t ld, int64_t M, int C, float* amax_inout, void* stream) {
  PSEG_REQUIRE(x && amax_inout && M > 0 && C > 0 && ld >= C, "amax: bad argument");
  long long b = (M * C + 1023) / 1024;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  hipLaunchKernelGGL(amax_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, x, (long long)ld, (long long)M, C,
                     (unsigned*)amax_inout);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

int pseg_amax_batch(const int64_t* jobs, int n, int64_t total_blocks, float* out, void* stream) {
  PSEG_REQUIRE(jobs && out && n > 0 && total_blocks > 0 && total_blocks < (1LL << 31), "amax_batch: bad argument");
  if (hipMemsetAsync(out, 0, (size_t)n * 4, (hipStream_t)stream) != hipSuccess) {
    set_error("amax_batch: hipMemsetAsync failed");
    return PSEG_ERR_HIP;
  }
  hipLaunchKernelGGL(amax_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(jobs), n, (unsigned*)out);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

int pseg_fill(float* x, int64_t n, float value, void* stream) {
  PSEG_REQUIRE(x && n > 0 && ((uintptr_t)x & 15) == 0, "fill: bad argument");
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, (long long)n, value);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

}  // extern "C"
