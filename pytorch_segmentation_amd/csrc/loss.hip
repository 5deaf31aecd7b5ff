// Per-pixel cross-entropy (fused forward + backward), argmax masks and per-class counts on gfx950.
//  - ce: nn.CrossEntropyLoss() defaults as called by the reference's compute_loss (utils/utils.py:12,17-24):
//    mean over non-ignored pixels, ignore_index = -100.  One pass over the NCHW logits produces the loss
//    partials AND dlogits = (softmax - onehot) / n_valid (algorithmic traffic: one read + one write of the
//    logits + one read of the targets), preceded by a targets-only count pass (8 B/pixel).
//  - argmax: outputs.max(1)[1] (reference test.py:31), first index wins ties.
//  - confusion: per-class tp / fn / fp (reference test.py:34-46) without the 3*C host syncs per batch.
// Lanes walk consecutive pixels of one class plane, so every load is a coalesced 256-B / 1-KiB segment.
#include "common.h"

#include <math.h>

namespace pseg {

struct CeHeader {
  int n_valid;   // pixels that enter the mean: target != ignore_index and 0 <= target < C
  int n_bad;     // targets outside [0, C) that are not ignore_index (torch raises on these; here they are ignored AND reported)
  int pad[2];
};

// The divisor of the mean must be the number of pixels whose loss term ce_fused_kernel / ce_generic_kernel add up:
// the SAME predicate (target != ignore_index && 0 <= target < C) is used here and there.
__device__ __forceinline__ bool ce_valid(long long t, long long ignore_index, int C) {
  return (t != ignore_index) && (t >= 0) && (t < C);
}

__global__ __launch_bounds__(256) void ce_count_kernel(const int64_t* __restrict__ target, long long n,
                                                       long long ignore_index, int C, CeHeader* __restrict__ hdr) {
  int cnt = 0, bad = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long t = target[i];
    const bool v = ce_valid(t, ignore_index, C);
    cnt += v ? 1 : 0;
    bad += (!v && t != ignore_index) ? 1 : 0;
  }
  // integer reduction: order-independent, so an atomic is still bit-reproducible
  for (int o = 32; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o, 64);
    bad += __shfl_xor(bad, o, 64);
  }
  __shared__ int sh[2][4];
  if ((threadIdx.x & 63) == 0) {
    sh[0][threadIdx.x >> 6] = cnt;
    sh[1][threadIdx.x >> 6] = bad;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    const int b = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    if (t) atomicAdd(&hdr->n_valid, t);
    if (b) atomicAdd(&hdr->n_bad, b);
  }
}

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
  v = wave_sum_d(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// CMAX: compile-time bound on the class count (values live in registers); VEC consecutive pixels per lane.
template <int CMAX, int VEC>
__global__ __launch_bounds__(256) void ce_fused_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                       int C, long long HW, long long groups, long long ignore_index,
                                                       float* __restrict__ dlogits, const CeHeader* __restrict__ hdr,
                                                       double* __restrict__ partial) {
  typedef float vecf __attribute__((ext_vector_type(VEC)));
  __shared__ double sh[4];
  const int nv = hdr->n_valid;
  const float inv_n = nv > 0 ? 1.f / (float)nv : 0.f;
  const long long gpi = HW / VEC;  // groups per image
  double lsum = 0.0;
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < groups; g += (long long)gridDim.x * 256) {
    const long long b = g / gpi;
    const long long p = (g - b * gpi) * VEC;
    const float* lp = logits + b * C * HW + p;
    vecf v[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) v[c] = *reinterpret_cast<const vecf*>(lp + (long long)c * HW);
    long long t[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) t[e] = target[b * HW + p + e];
    vecf m = v[0];
#pragma unroll
    for (int c = 1; c < CMAX; ++c)
      if (c < C) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) m[e] = fmaxf(m[e], v[c][e]);
      }
    // one exponential per logit: v[c] becomes exp(v[c] - m) (the target logit is picked up first), the softmax is that
    // times 1/sum.  __expf / __logf are the hardware exp2 / log2 paths (~1e-6 relative), an order of magnitude fewer
    // instructions than the correctly-rounded library calls this kernel used to be bound by.
    vecf s = 0.f, vt = 0.f;
    bool valid[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) valid[e] = ce_valid(t[e], ignore_index, C);
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          if (t[e] == c) vt[e] = v[c][e];
          v[c][e] = __expf(v[c][e] - m[e]);
          s[e] += v[c][e];
        }
      }
    vecf scale;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      if (valid[e]) lsum += (double)(m[e] + __logf(s[e]) - vt[e]);
      scale[e] = valid[e] ? inv_n / s[e] : 0.f;
    }
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) {
        vecf gvec;
#pragma unroll
        for (int e = 0; e < VEC; ++e) gvec[e] = v[c][e] * scale[e] - ((valid[e] && t[e] == c) ? inv_n : 0.f);
        if (dlogits) *reinterpret_cast<vecf*>(dlogits + b * C * HW + p + (long long)c * HW) = gvec;
      }
  }
  const double tot = block_sum_d(lsum, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// any class count: three sweeps over the class planes (the re-reads hit L2)
__global__ __launch_bounds__(256) void ce_generic_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                         int C, long long HW, long long npix, long long ignore_index,
                                                         float* __restrict__ dlogits, const CeHeader* __restrict__ hdr,
                                                         double* __restrict__ partial) {
  __shared__ double sh[4];
  const int nv = hdr->n_valid;
  const float inv_n = nv > 0 ? 1.f / (float)nv : 0.f;
  double lsum = 0.0;
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < npix; g += (long long)gridDim.x * 256) {
    const long long b = g / HW;
    const long long p = g - b * HW;
    const float* lp = logits + b * C * HW + p;
    const long long t = target[g];
    const bool valid = ce_valid(t, ignore_index, C);
    float m = lp[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, lp[(long long)c * HW]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(lp[(long long)c * HW] - m);
    const float lse = m + logf(s);
    if (valid) lsum += (double)(lse - lp[t * HW]);
    if (dlogits) {
      float* dp = dlogits + b * C * HW + p;
      for (int c = 0; c < C; ++c) {
        const float pr = expf(lp[(long long)c * HW] - lse);
        dp[(long long)c * HW] = valid ? (pr - ((t == c) ? 1.f : 0.f)) * inv_n : 0.f;
      }
    }
  }
  const double tot = block_sum_d(lsum, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ void ce_finish_kernel(const double* __restrict__ partial, int nblocks, const CeHeader* __restrict__ hdr,
                                 float* __restrict__ loss_out) {
  // one wave, fixed order
  double s = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 64) s += partial[i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) {
    const int nv = hdr->n_valid;
    loss_out[0] = nv > 0 ? (float)(s / (double)nv) : NAN;  // torch: mean over zero elements is NaN
    loss_out[1] = (float)nv;
    loss_out[2] = (float)hdr->n_bad;
  }
}

__global__ __launch_bounds__(256) void scale_inplace_kernel(float* __restrict__ x, long long n4, long long n,
                                                            const float* __restrict__ gscale) {
  const float g = *gscale;
  if (g == 1.0f) return;  // the usual case (loss.backward() seeds 1): nothing to do
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    f32x4 v = *reinterpret_cast<f32x4*>(x + i * 4);
    *reinterpret_cast<f32x4*>(x + i * 4) = v * g;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n - n4 * 4)) x[n4 * 4 + threadIdx.x] *= g;
}

template <int VEC>
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, int C, long long HW, long long groups,
                                                     int64_t* __restrict__ mask) {
  typedef float vecf __attribute__((ext_vector_type(VEC)));
  const long long gpi = HW / VEC;
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < groups; g += (long long)gridDim.x * 256) {
    const long long b = g / gpi;
    const long long p = (g - b * gpi) * VEC;
    const float* lp = logits + b * C * HW + p;
    vecf best = *reinterpret_cast<const vecf*>(lp);
    int idx[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) idx[e] = 0;
    for (int c = 1; c < C; ++c) {
      const vecf v = *reinterpret_cast<const vecf*>(lp + (long long)c * HW);
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        if (v[e] > best[e]) {  // strict: the first maximum wins
          best[e] = v[e];
          idx[e] = c;
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) mask[b * HW + p + e] = idx[e];
  }
}

constexpr int kMaxConfusionClasses = 1024;

__global__ __launch_bounds__(256) void confusion_kernel(const int64_t* __restrict__ pred, const int64_t* __restrict__ target,
                                                        long long n, int C, unsigned long long* __restrict__ counters) {
  __shared__ unsigned int h[3 * kMaxConfusionClasses];
  for (int i = threadIdx.x; i < 3 * C; i += 256) h[i] = 0;
  __syncthreads();
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long t = target[i], p = pred[i];
    const bool tin = t >= 0 && t < C, pin = p >= 0 && p < C;
    if (tin) atomicAdd(&h[(p == t ? 0 : 1) * C + (int)t], 1u);  // tp / fn
    if (pin && p != t) atomicAdd(&h[2 * C + (int)p], 1u);       // fp
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * C; i += 256)
    if (h[i]) atomicAdd(&counters[i], (unsigned long long)h[i]);
}

static int capped_blocks(long long work_items, int cap) {
  long long b = (work_items + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

constexpr int kCeMaxBlocks = 4096;

}  // namespace pseg

using namespace pseg;

extern "C" {

int64_t pseg_ce_workspace_bytes(int64_t npix) {
  (void)npix;
  return (int64_t)sizeof(CeHeader) + (int64_t)kCeMaxBlocks * 8;
}

int pseg_ce_fwd_bwd(const float* logits, const int64_t* target, int B, int C, int64_t HW, int64_t ignore_index,
                    float* dlogits, float* loss_out, void* workspace, int64_t workspace_bytes, void* stream) {
  PSEG_REQUIRE(logits && target && loss_out && workspace, "ce: null pointer");
  PSEG_REQUIRE(B > 0 && C > 0 && HW > 0, "ce: bad sizes");
  PSEG_REQUIRE(workspace_bytes >= pseg_ce_workspace_bytes((int64_t)B * HW), "ce: workspace too small");
  PSEG_REQUIRE(((uintptr_t)workspace & 15) == 0, "ce: workspace must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  CeHeader* hdr = (CeHeader*)workspace;
  double* partial = (double*)((char*)workspace + sizeof(CeHeader));
  const long long npix = (long long)B * HW;
  if (hipMemsetAsync(hdr, 0, sizeof(CeHeader), st) != hipSuccess) {
    set_error("ce: hipMemsetAsync failed");
    return PSEG_ERR_HIP;
  }
  hipLaunchKernelGGL(ce_count_kernel, dim3(capped_blocks(npix, 2048)), dim3(256), 0, st, target, npix,
                     (long long)ignore_index, C, hdr);
  PSEG_LAUNCH_CHECK();
  const bool vec4 = (HW % 4 == 0) && (((uintptr_t)logits & 15) == 0) && (!dlogits || ((uintptr_t)dlogits & 15) == 0);
  int blocks;
#define CE_LAUNCH(CMAX, VEC)                                                                                       \
  do {                                                                                                             \
    const long long groups = npix / VEC;                                                                           \
    blocks = capped_blocks(groups, kCeMaxBlocks);                                                                  \
    hipLaunchKernelGGL((ce_fused_kernel<CMAX, VEC>), dim3(blocks), dim3(256), 0, st, logits, target, C,            \
                       (long long)HW, groups, (long long)ignore_index, dlogits, (const CeHeader*)hdr, partial);    \
  } while (0)
  if (C <= 4 && vec4) CE_LAUNCH(4, 4);
  else if (C <= 4) CE_LAUNCH(4, 1);
  else if (C <= 8 && vec4) CE_LAUNCH(8, 4);
  else if (C <= 8) CE_LAUNCH(8, 1);
  else if (C <= 24 && vec4) CE_LAUNCH(24, 4);
  else if (C <= 24) CE_LAUNCH(24, 1);
  else if (C <= 32 && vec4) CE_LAUNCH(32, 2);
  else if (C <= 32) CE_LAUNCH(32, 1);
  else {
    blocks = capped_blocks(npix, kCeMaxBlocks);
    hipLaunchKernelGGL(ce_generic_kernel, dim3(blocks), dim3(256), 0, st, logits, target, C, (long long)HW, npix,
                       (long long)ignore_index, dlogits, (const CeHeader*)hdr, partial);
  }
#undef CE_LAUNCH
  PSEG_LAUNCH_CHECK();
  hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(64), 0, st, (const double*)partial, blocks, (const CeHeader*)hdr,
                     loss_out);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

int pseg_scale_inplace(float* x, int64_t n, const float* gscale, void* stream) {
  PSEG_REQUIRE(x && gscale && n > 0 && ((uintptr_t)x & 15) == 0, "scale_inplace: bad argument");
  const long long n4 = n / 4;
  hipLaunchKernelGGL(scale_inplace_kernel, dim3(capped_blocks(n4 ? n4 : 1, 2048)), dim3(256), 0, (hipStream_t)stream, x,
                     n4, (long long)n, gscale);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

int pseg_argmax(const float* logits, int B, int C, int64_t HW, int64_t* mask, void* stream) {
  PSEG_REQUIRE(logits && mask && B > 0 && C > 0 && HW > 0, "argmax: bad argument");
  const long long npix = (long long)B * HW;
  if (HW % 4 == 0 && ((uintptr_t)logits & 15) == 0) {
    hipLaunchKernelGGL(argmax_kernel<4>, dim3(capped_blocks(npix / 4, 4096)), dim3(256), 0, (hipStream_t)stream, logits, C,
                       (long long)HW, npix / 4, mask);
  } else {
    hipLaunchKernelGGL(argmax_kernel<1>, dim3(capped_blocks(npix, 4096)), dim3(256), 0, (hipStream_t)stream, logits, C,
                       (long long)HW, npix, mask);
  }
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

int pseg_confusion(const int64_t* pred, const int64_t* target, int64_t n, int C, int64_t* counters, void* stream) {
  PSEG_REQUIRE(pred && target && counters && n > 0, "confusion: bad argument");
  PSEG_REQUIRE(C > 0 && C <= kMaxConfusionClasses, "confusion: class count must be in [1, %d]", kMaxConfusionClasses);
  hipLaunchKernelGGL(confusion_kernel, dim3(capped_blocks(n, 1024)), dim3(256), 0, (hipStream_t)stream, pred, target,
                     (long long)n, C, (unsigned long long*)counters);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

}  // extern "C"
