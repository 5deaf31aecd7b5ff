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

// ------------------------------------------------------------------------------------------------ mixed precision (`-mp`)
// Dynamic loss scaling as apex / torch.cuda.amp do it (the reference's -mp, train.py:102-105), with the whole protocol on
// the device so that a step never synchronises the host and stays graph-capturable:
//   state[0] loss scale S (multiplies the loss gradient on its way into fp16, pseg_convert2d)   state[1] 1 / S
//   state[2] optimiser steps since the scale last changed                                       state[3] found-inf flag of this step
//   state[4] optimiser steps applied so far (Adam bias correction, SGD first-step)              state[5] steps skipped so far
// pseg_mp_check raises state[3] when any gradient is inf / nan; the optimiser kernels read it -- a flagged step changes
// nothing -- and fold 1 / S into grad_scale; pseg_mp_update then halves S after a flagged step, doubles it after
// growth_interval clean ones, and clears the flag.
constexpr int kMpScale = 0, kMpInv = 1, kMpTracker = 2, kMpFound = 3, kMpSteps = 4, kMpSkipped = 5;

__global__ __launch_bounds__(256) void mp_check_kernel(const float* __restrict__ g, long long n, float* __restrict__ state) {
  const long long n4 = n / 4;
  bool bad = false;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(g + i * 4);
    // (x - x is 0 for finite x and nan for inf / nan)
    const float t = (v[0] - v[0]) + (v[1] - v[1]) + (v[2] - v[2]) + (v[3] - v[3]);
    bad = bad || !(t == 0.f);
  }
  if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) {
    const float v = g[n4 * 4 + threadIdx.x];
    bad = bad || !(v - v == 0.f);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) state[kMpFound] = 1.f;     // (every writer stores the same value)
}

__global__ void mp_update_kernel(float* __restrict__ state, float growth, float backoff, int interval, float min_scale,
                                 float max_scale) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s = state[kMpScale];
  if (state[kMpFound] != 0.f) {
    s = fmaxf(s * backoff, min_scale);
    state[kMpTracker] = 0.f;
    state[kMpSkipped] += 1.f;
  } else {
    state[kMpSteps] += 1.f;
    float t = state[kMpTracker] + 1.f;
    if (t >= (float)interval) {
      s = fminf(s * growth, max_scale);
      t = 0.f;
    }
    state[kMpTracker] = t;
  }
  state[kMpScale] = s;
  state[kMpInv] = 1.f / s;
  state[kMpFound] = 0.f;
}

__global__ void mp_init_kernel(float* __restrict__ state, float scale) {
  if (threadIdx.x < 8) {
    float v = 0.f;
    if (threadIdx.x == kMpScale) v = scale;
    if (threadIdx.x == kMpInv) v = 1.f / scale;
    state[threadIdx.x] = v;
  }
}

__global__ __launch_bounds__(256) void sgd_mp_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                     float* __restrict__ mbuf, long long n, float lr, float mu, float wd,
                                                     int nesterov, float gscale, const float* __restrict__ state) {
  if (state[kMpFound] != 0.f) return;            // overflow somewhere in this step's gradients: skip it
  gscale *= state[kMpInv];
  const bool first = state[kMpSteps] == 0.f;
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

__global__ __launch_bounds__(256) void adam_mp_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ v, long long n, float lr,
                                                      float b1, float b2, float eps, float wd, int decoupled, float gscale,
                                                      const float* __restrict__ state) {
  if (state[kMpFound] != 0.f) return;
  gscale *= state[kMpInv];
  const float t = state[kMpSteps] + 1.f;         // this is the t-th applied step
  const float step_size = lr / (1.f - powf(b1, t));
  const float inv_sqrt_bc2 = 1.f / sqrtf(1.f - powf(b2, t));
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

int pseg_mp_state_init(float* state, float init_scale, void* stream) {
  PSEG_REQUIRE(state && init_scale > 0.f, "mp_state_init: bad argument");
  hipLaunchKernelGGL(mp_init_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, init_scale);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

int pseg_mp_check(const float* grad, int64_t n, float* state, void* stream) {
  PSEG_REQUIRE(grad && state && n > 0 && ((uintptr_t)grad & 15) == 0, "mp_check: bad argument");
  hipLaunchKernelGGL(mp_check_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, grad, (long long)n, state);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

int pseg_mp_update(float* state, float growth_factor, float backoff_factor, int growth_interval, float min_scale,
                   float max_scale, void* stream) {
  PSEG_REQUIRE(state && growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f && growth_interval >= 1 &&
                   min_scale > 0.f && max_scale >= min_scale,
               "mp_update: bad argument");
  hipLaunchKernelGGL(mp_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, growth_factor, backoff_factor,
                     growth_interval, min_scale, max_scale);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

int pseg_sgd_step_mp(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum,
                     float weight_decay, int nesterov, float grad_scale, const float* state, void* stream) {
  PSEG_REQUIRE(param && grad && state && n > 0, "sgd_step_mp: bad argument");
  PSEG_REQUIRE(momentum == 0.f || momentum_buf, "sgd_step_mp: momentum needs a buffer");
  PSEG_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)momentum_buf) & 15) == 0, "sgd_step_mp: 16-byte alignment");
  hipLaunchKernelGGL(sgd_mp_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, momentum_buf,
                     (long long)n, lr, momentum, weight_decay, nesterov, grad_scale, state);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

int pseg_adam_step_mp(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                      float beta2, float eps, float weight_decay, int decoupled, float grad_scale, const float* state,
                      void* stream) {
  PSEG_REQUIRE(param && grad && exp_avg && exp_avg_sq && state && n > 0, "adam_step_mp: bad argument");
  PSEG_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
               "adam_step_mp: 16-byte alignment");
  hipLaunchKernelGGL(adam_mp_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                     (long long)n, lr, beta1, beta2, eps, weight_decay, decoupled, grad_scale, state);
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
