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
#include <stdlib.h>

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

__global__ __launch_bounds__(256) void ce_finish_kernel(const double* __restrict__ partial, int nblocks,
                                                        const CeHeader* __restrict__ hdr, float* __restrict__ loss_out) {
  // one block, fixed order: lane t adds partials t, t + 256, ... (eight loads in flight -- the up-sampled loss leaves 16384
  // partials, one dependent round trip each was 64 us between the forward and the backward pass), then waves, then lanes
  __shared__ double sh[4];
  double s = 0.0;
  int i = threadIdx.x;
  for (; i + 7 * 256 < nblocks; i += 8 * 256) {
    double t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = partial[i + k * 256];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += t[k];
  }
  for (; i < nblocks; i += 256) s += partial[i];
  s = block_sum_d(s, sh);
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

// ------------------------------------------------------------------------------------------------
// Cross-entropy on bilinearly up-sampled logits WITHOUT the full-resolution tensor (reference: models/deeplabv3plus.py:40-43
// up-samples the class logits x4, utils/utils.py:18-21 takes nn.CrossEntropyLoss of them): the three passes of the plain
// path -- write the [B,C,H,W] logits, read them / write their gradient, read that gradient -- move 1.1 GB per DeepLabV3+
// step for a 25 MB tensor of low-resolution logits.  Here a block owns a kUpTY x kUpTX tile of LOW-resolution pixels:
//   A. the logits of the tile plus a one-pixel halo go to LDS;
//   B. every full-resolution pixel whose interpolation touches the tile is evaluated ONCE per block (interpolated logits,
//      softmax, loss term, gradient (softmax - onehot) / n_valid) and its gradient parked in LDS; the loss term is added by
//      the block that owns the pixel's top-left source pixel (exactly one);
//   C. every (tile pixel, class) gathers its gradient  sum_{Y,X} w_y(Y,i) w_x(X,j) g[Y,X,c]  over its support in a fixed
//      order -- no floating-point atomics anywhere, bit-reproducible.
// Interpolation weights are those of bilinear_fwd (src_index below is the same function as pool_resize.hip's).
constexpr int kUpTY = 2, kUpTX = 8, kUpCP = 24;   // tile (2 x 8: 55 KB of LDS, three blocks per CU), padded class count
constexpr int kUpThreads = 512;
constexpr int kUpRY = 16, kUpRX = 42;             // full-resolution region a tile can touch (host-checked against the scales)

struct UpAxis {
  float scale;
  int in, out, align;
};

__host__ __device__ __forceinline__ void up_src_index(const UpAxis& a, int o, int& i0, int& i1, float& l0, float& l1) {
  float s;
  if (a.align) {
    s = a.scale * (float)o;
  } else {
    s = a.scale * ((float)o + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
  }
  i0 = (int)s;
  if (i0 > a.in - 1) i0 = a.in - 1;
  i1 = i0 + ((i0 < a.in - 1) ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.f - l1;
}

// destination range [lo, hi] = exactly the indices whose first tap lies in [i_first - 1, i_last] (every destination
// index that touches a source index of [i_first, i_last] with either tap is inside; up_src_index is monotone in o)
__device__ __forceinline__ void up_region(const UpAxis& a, int i_first, int i_last, int& lo, int& hi) {
  const float inv = a.scale > 0.f ? 1.f / a.scale : 0.f;
  lo = (int)floorf((float)(i_first - 1) * inv) - 2;
  hi = (int)ceilf((float)(i_last + 1) * inv) + 2;
  if (lo < 0) lo = 0;
  if (hi > a.out - 1) hi = a.out - 1;
  int i0, i1;
  float l0, l1;
  while (lo < hi) {          // drop indices whose first tap is still left of i_first - 1
    up_src_index(a, lo, i0, i1, l0, l1);
    if (i0 >= i_first - 1) break;
    ++lo;
  }
  while (hi > lo) {          // ... and those whose first tap is right of i_last
    up_src_index(a, hi, i0, i1, l0, l1);
    if (i0 <= i_last) break;
    --hi;
  }
}

__global__ __launch_bounds__(kUpThreads) void ce_up_fused_kernel(const float* __restrict__ L, int ldl, int B, int C,
                                                          const int64_t* __restrict__ target, UpAxis ay, UpAxis ax,
                                                          long long ignore_index, float* __restrict__ dL, int ldd,
                                                          const CeHeader* __restrict__ hdr, double* __restrict__ partial,
                                                          int tiles_y, int tiles_x) {
  __shared__ float s_l[(kUpTY + 2) * (kUpTX + 2) * kUpCP];     // low-resolution logits, tile + halo
  __shared__ float s_g[kUpRY * kUpRX * kUpCP];                  // gradients of the full-resolution pixels of the region
  __shared__ float s_wy[kUpRY][kUpTY], s_wx[kUpRX][kUpTX];      // interpolation weights region row/col -> tile row/col
  __shared__ int s_rng[4][kUpTX];                                // support ranges: rows lo / hi per tile row, cols lo / hi per tile col
  __shared__ double sh[kUpThreads / 64];
  const int tid = threadIdx.x;
  int blk = blockIdx.x;
  const int tx_i = blk % tiles_x;
  blk /= tiles_x;
  const int ty_i = blk % tiles_y;
  const int b = blk / tiles_y;
  const int i0 = ty_i * kUpTY, j0 = tx_i * kUpTX;
  const int h = ay.in, w = ax.in, H = ay.out, W = ax.out;
  const int nv = hdr->n_valid;
  const float inv_n = nv > 0 ? 1.f / (float)nv : 0.f;

  // ---- A: logits of rows [i0 - 1, i0 + TY], cols [j0 - 1, j0 + TX] (clamped reads; out-of-image entries are never used)
  for (int e = tid; e < (kUpTY + 2) * (kUpTX + 2) * (kUpCP / 4); e += kUpThreads) {
    const int c4 = e % (kUpCP / 4);
    const int px = e / (kUpCP / 4);
    int yy = i0 - 1 + px / (kUpTX + 2), xx = j0 - 1 + px % (kUpTX + 2);
    yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
    xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c4 * 4 < ldl) v = *reinterpret_cast<const f32x4*>(L + ((long long)(b * h + yy) * w + xx) * ldl + c4 * 4);   // (ldl % 4 == 0)
    *reinterpret_cast<f32x4*>(&s_l[px * kUpCP + c4 * 4]) = v;
  }
  int Ya, Yb, Xa, Xb;
  up_region(ay, i0, i0 + kUpTY - 1, Ya, Yb);
  up_region(ax, j0, j0 + kUpTX - 1, Xa, Xb);
  const int ry = Yb - Ya + 1, rx = Xb - Xa + 1;     // <= kUpRY, kUpRX (host-checked)
  // weights of region rows / columns towards the tile's rows / columns
  for (int e = tid; e < kUpRY * kUpTY; e += kUpThreads) {
    const int r = e / kUpTY, ii = e % kUpTY;
    float wgt = 0.f;
    if (r < ry) {
      int a0, a1;
      float l0, l1;
      up_src_index(ay, Ya + r, a0, a1, l0, l1);
      if (a0 == i0 + ii) wgt += l0;
      if (a1 == i0 + ii) wgt += l1;
    }
    s_wy[r][ii] = wgt;
  }
  for (int e = tid; e < kUpRX * kUpTX; e += kUpThreads) {
    const int r = e / kUpTX, jj = e % kUpTX;
    float wgt = 0.f;
    if (r < rx) {
      int a0, a1;
      float l0, l1;
      up_src_index(ax, Xa + r, a0, a1, l0, l1);
      if (a0 == j0 + jj) wgt += l0;
      if (a1 == j0 + jj) wgt += l1;
    }
    s_wx[r][jj] = wgt;
  }
  __syncthreads();
  // first / last region row (column) with a non-zero weight towards each tile row (column); empty: first > last
  if (tid < kUpTY + kUpTX) {
    const bool is_row = tid < kUpTY;
    const int k = is_row ? tid : tid - kUpTY;
    const int n = is_row ? ry : rx;
    int lo = n, hi = -1;
    for (int r = 0; r < n; ++r) {
      const float wgt = is_row ? s_wy[r][k] : s_wx[r][k];
      if (wgt != 0.f) {
        lo = r < lo ? r : lo;
        hi = r;
      }
    }
    s_rng[is_row ? 0 : 2][k] = lo;
    s_rng[is_row ? 1 : 3][k] = hi;
  }
  __syncthreads();

  // ---- B: the full-resolution pixels of the region
  double lsum = 0.0;
  for (int e = tid; e < ry * rx; e += kUpThreads) {
    const int r = e / rx, q = e - r * rx;
    const int Y = Ya + r, X = Xa + q;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    up_src_index(ay, Y, y0, y1, ly0, ly1);
    up_src_index(ax, X, x0, x1, lx0, lx1);
    float* gp = &s_g[(r * kUpRX + q) * kUpCP];
    const bool touches = (y0 <= i0 + kUpTY - 1) && (y1 >= i0) && (x0 <= j0 + kUpTX - 1) && (x1 >= j0);
    if (!touches) {
#pragma unroll
      for (int c4 = 0; c4 < kUpCP / 4; ++c4) *reinterpret_cast<f32x4*>(gp + c4 * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      continue;
    }
    const float* p00 = &s_l[((y0 - (i0 - 1)) * (kUpTX + 2) + (x0 - (j0 - 1))) * kUpCP];
    const float* p01 = &s_l[((y0 - (i0 - 1)) * (kUpTX + 2) + (x1 - (j0 - 1))) * kUpCP];
    const float* p10 = &s_l[((y1 - (i0 - 1)) * (kUpTX + 2) + (x0 - (j0 - 1))) * kUpCP];
    const float* p11 = &s_l[((y1 - (i0 - 1)) * (kUpTX + 2) + (x1 - (j0 - 1))) * kUpCP];
    float v[kUpCP];
    float m = -INFINITY;
#pragma unroll
    for (int c4 = 0; c4 < kUpCP / 4; ++c4) {
      const f32x4 a00 = *reinterpret_cast<const f32x4*>(p00 + c4 * 4), a01 = *reinterpret_cast<const f32x4*>(p01 + c4 * 4),
                  a10 = *reinterpret_cast<const f32x4*>(p10 + c4 * 4), a11 = *reinterpret_cast<const f32x4*>(p11 + c4 * 4);
      // the arithmetic of bilinear_fwd: rows first, then columns
      const f32x4 u = ly0 * (lx0 * a00 + lx1 * a01) + ly1 * (lx0 * a10 + lx1 * a11);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[c4 * 4 + k] = u[k];
        if (c4 * 4 + k < C) m = fmaxf(m, u[k]);
      }
    }
    const long long t = target[((long long)b * H + Y) * W + X];
    const bool valid = ce_valid(t, ignore_index, C);
    float ssum = 0.f, vt = 0.f;
#pragma unroll
    for (int c = 0; c < kUpCP; ++c) {
      if (c < C) {
        if (t == c) vt = v[c];
        v[c] = __expf(v[c] - m);
        ssum += v[c];
      }
    }
    const bool owner = (y0 >= i0) && (y0 <= i0 + kUpTY - 1) && (x0 >= j0) && (x0 <= j0 + kUpTX - 1);
    if (valid && owner) lsum += (double)(m + __logf(ssum) - vt);
    const float scale = valid ? inv_n / ssum : 0.f;
#pragma unroll
    for (int c4 = 0; c4 < kUpCP / 4; ++c4) {
      f32x4 gv;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = c4 * 4 + k;
        gv[k] = c < C ? v[c] * scale - ((valid && t == c) ? inv_n : 0.f) : 0.f;
      }
      *reinterpret_cast<f32x4*>(gp + c4 * 4) = gv;
    }
  }
  __syncthreads();

  // ---- C: gather the gradient of every (tile pixel, class) over its support, separably and in a fixed order: first
  // along the columns (region row x tile column x class), then along the rows.  The column sums live in s_l's place? no:
  // s_l is 6 KB; they go over the FRONT of s_g's rows as they are consumed (row r of s_gx only reads row r of s_g).
  if (dL != nullptr) {
    float* s_gx = s_g;     // [ry][kUpTX][kUpCP] written over s_g[r][0 .. kUpTX) after a barrier per pass below
    // pass 1 into registers, barrier, then store: every thread owns (r, jj, c4) items
    constexpr int kC4 = kUpCP / 4;
    const int items = ry * kUpTX * kC4;
    f32x4 hold[(kUpRY * kUpTX * kC4 + kUpThreads - 1) / kUpThreads];
#pragma unroll
    for (int u = 0; u < (kUpRY * kUpTX * kC4 + kUpThreads - 1) / kUpThreads; ++u) {
      const int e = tid + u * kUpThreads;
      f32x4 row = {0.f, 0.f, 0.f, 0.f};
      if (e < items) {
        const int c4 = e % kC4, jj = (e / kC4) % kUpTX, r = e / (kC4 * kUpTX);
        const int q0 = s_rng[2][jj], q1 = s_rng[3][jj];
        for (int q = q0; q <= q1; ++q)
          row += s_wx[q][jj] * *reinterpret_cast<const f32x4*>(&s_g[(r * kUpRX + q) * kUpCP + c4 * 4]);
      }
      hold[u] = row;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < (kUpRY * kUpTX * kC4 + kUpThreads - 1) / kUpThreads; ++u) {
      const int e = tid + u * kUpThreads;
      if (e < items) {
        const int c4 = e % kC4, jj = (e / kC4) % kUpTX, r = e / (kC4 * kUpTX);
        *reinterpret_cast<f32x4*>(&s_gx[(r * kUpRX + jj) * kUpCP + c4 * 4]) = hold[u];
      }
    }
    __syncthreads();
    for (int o = tid; o < kUpTY * kUpTX * kC4; o += kUpThreads) {
      const int c4 = o % kC4, px = o / kC4;
      const int ii = px / kUpTX, jj = px % kUpTX;
      const int i = i0 + ii, j = j0 + jj;
      if (i >= h || j >= w) continue;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const int r0 = s_rng[0][ii], r1 = s_rng[1][ii];
      for (int r = r0; r <= r1; ++r) acc += s_wy[r][ii] * *reinterpret_cast<const f32x4*>(&s_gx[(r * kUpRX + jj) * kUpCP + c4 * 4]);
      float* dp = dL + ((long long)(b * h + i) * w + j) * ldd + c4 * 4;
      if (c4 * 4 + 3 < ldd) {
        *reinterpret_cast<f32x4*>(dp) = acc;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (c4 * 4 + k < ldd) dp[k] = acc[k];
      }
    }
  }
  lsum = wave_sum_d(lsum);
  if ((tid & 63) == 0) sh[tid >> 6] = lsum;
  __syncthreads();
  if (tid == 0) {
    double tot = 0.0;
    for (int i = 0; i < kUpThreads / 64; ++i) tot += sh[i];     // fixed order
    partial[blockIdx.x] = tot;
  }
}


// ------------------------------------------------------------------------------------------------
// Round 5: the SCATTER formulation of the same loss.  ce_up_fused_kernel gathers -- a block owns low-resolution
// pixels and evaluates every full-resolution pixel that touches them: 1.7 softmax evaluations per pixel on a 2 x 8 tile, in a
// kernel that is VALU-bound (1300 VALU instructions per wave, pipes 72 % busy).  Here a block owns FULL-resolution pixels: those
// whose first taps (y0, x0) fall into its kSY x kSX tile of low-resolution pixels -- every pixel of the image belongs to exactly
// one block and is evaluated ONCE.  Its gradient lands on the four low-resolution pixels (y0 | y1) x (x0 | x1), all inside the
// (kSY + 1) x (kSX + 1) patch at the tile's origin; the block sums its patch in LDS (the same separable, fixed-order gather as
// before, over owned pixels only) and writes it to a workspace; ce_up_combine_kernel then adds, for every low-resolution pixel,
// the up to four patches it appears in -- own tile, the tile above (its halo row), the tile to the left (halo column), the
// diagonal one -- in that order.  No atomics, bit-reproducible; 27 x 24 floats of patch per 2 x 8 tile (the second pass moves
// ~3 x the low-resolution gradient: 12 us at 16 x 128 x 128 x 21).
constexpr int kSY = 2, kSX = 8;                     // low-resolution tile: 8 x 32 owned pixels at x4 = one per thread
constexpr int kSThreads = 256;
constexpr int kSPY = kSY + 1, kSPX = kSX + 1;       // patch
constexpr int kSRY = 4 * kSY + 2, kSRX = 4 * kSX + 2;   // owned full-resolution pixels per axis: 10 x 34 -- 37 KB of LDS, four blocks per CU;
                                                        // host-checked against the EXACT ownership counts of the launch's axes

// destination indices whose FIRST tap lies in [i_first, i_last] (up_src_index is monotone): [lo, hi], empty when lo > hi
__device__ __forceinline__ void up_owned(const UpAxis& a, int i_first, int i_last, int& lo, int& hi) {
  const float inv = a.scale > 0.f ? 1.f / a.scale : 0.f;
  lo = (int)floorf((float)i_first * inv) - 2;
  hi = (int)ceilf((float)(i_last + 1) * inv) + 2;
  if (lo < 0) lo = 0;
  if (hi > a.out - 1) hi = a.out - 1;
  int i0, i1;
  float l0, l1;
  while (lo <= hi) {
    up_src_index(a, lo, i0, i1, l0, l1);
    if (i0 >= i_first) break;
    ++lo;
  }
  while (hi >= lo) {
    up_src_index(a, hi, i0, i1, l0, l1);
    if (i0 <= i_last) break;
    --hi;
  }
}

__global__ __launch_bounds__(kSThreads) void ce_up_scatter_kernel(const float* __restrict__ L, int ldl, int B, int C,
                                                            const int64_t* __restrict__ target, UpAxis ay, UpAxis ax,
                                                            long long ignore_index, float* __restrict__ patches,
                                                            double* __restrict__ partial, int tiles_y, int tiles_x) {
  __shared__ float s_l[kSPY * kSPX * kUpCP];                   // low-resolution logits of the patch
  __shared__ float s_g[kSRY * kSRX * kUpCP];                   // gradients of the owned full-resolution pixels
  __shared__ float s_wy[kSRY][kSPY], s_wx[kSRX][kSPX];         // interpolation weights owned row / column -> patch row / column
  __shared__ int s_rng[4][kSPX];                               // support ranges (rows lo / hi per patch row, columns per patch column)
  __shared__ double sh[kSThreads / 64];
  const int tid = threadIdx.x;
  int blk = blockIdx.x;
  const int tx_i = blk % tiles_x;
  blk /= tiles_x;
  const int ty_i = blk % tiles_y;
  const int b = blk / tiles_y;
  const int i0 = ty_i * kSY, j0 = tx_i * kSX;
  const int h = ay.in, w = ax.in, H = ay.out, W = ax.out;
  // (gradients leave this kernel UNNORMALISED -- softmax - onehot -- and the block counts its valid / out-of-range labels: the
  // 1 / n_valid of the mean is applied by ce_up_combine_kernel, after ce_up_finish_kernel has added the counts up.  The separate
  // pass over the targets that the other loss kernels need for n_valid (ce_count_kernel, 30 us at 16 x 512 x 512) is gone.)

  // ---- A: logits of the patch rows [i0, i0 + kSY], columns [j0, j0 + kSX] (clamped: entries past the image are never used)
  for (int e = tid; e < kSPY * kSPX * (kUpCP / 4); e += kSThreads) {
    const int c4 = e % (kUpCP / 4);
    const int px = e / (kUpCP / 4);
    int yy = i0 + px / kSPX, xx = j0 + px % kSPX;
    yy = yy > h - 1 ? h - 1 : yy;
    xx = xx > w - 1 ? w - 1 : xx;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c4 * 4 < ldl) v = *reinterpret_cast<const f32x4*>(L + ((long long)(b * h + yy) * w + xx) * ldl + c4 * 4);
    *reinterpret_cast<f32x4*>(&s_l[px * kUpCP + c4 * 4]) = v;
  }
  int Ya, Yb, Xa, Xb;
  const int i_last = i0 + kSY - 1 < h - 1 ? i0 + kSY - 1 : h - 1, j_last = j0 + kSX - 1 < w - 1 ? j0 + kSX - 1 : w - 1;
  up_owned(ay, i0, i_last, Ya, Yb);
  up_owned(ax, j0, j_last, Xa, Xb);
  // <= kSRY, kSRX: host-checked with the kernel's own index function (max_owned, + 2 slack where the device may contract the
  // expression to an fma) -- and ENFORCED here: a tile that still owned more is cut to what its LDS image holds and its loss term
  // is poisoned (NaN), so a host / device disagreement shows as a NaN loss instead of an LDS overrun.
  int ry = Yb - Ya + 1, rx = Xb - Xa + 1;           // > 0: every source index owns a destination
  const bool overrun = ry > kSRY || rx > kSRX;
  ry = ry > kSRY ? kSRY : ry;
  rx = rx > kSRX ? kSRX : rx;
  for (int e = tid; e < kSRY * kSPY; e += kSThreads) {
    const int r = e / kSPY, ii = e % kSPY;
    float wgt = 0.f;
    if (r < ry) {
      int a0, a1;
      float l0, l1;
      up_src_index(ay, Ya + r, a0, a1, l0, l1);
      if (a0 == i0 + ii) wgt += l0;
      if (a1 == i0 + ii) wgt += l1;
    }
    s_wy[r][ii] = wgt;
  }
  for (int e = tid; e < kSRX * kSPX; e += kSThreads) {
    const int r = e / kSPX, jj = e % kSPX;
    float wgt = 0.f;
    if (r < rx) {
      int a0, a1;
      float l0, l1;
      up_src_index(ax, Xa + r, a0, a1, l0, l1);
      if (a0 == j0 + jj) wgt += l0;
      if (a1 == j0 + jj) wgt += l1;
    }
    s_wx[r][jj] = wgt;
  }
  __syncthreads();
  if (tid < kSPY + kSPX) {
    const bool is_row = tid < kSPY;
    const int k = is_row ? tid : tid - kSPY;
    const int n = is_row ? ry : rx;
    int lo = n, hi = -1;
    for (int r = 0; r < n; ++r) {
      const float wgt = is_row ? s_wy[r][k] : s_wx[r][k];
      if (wgt != 0.f) {
        lo = r < lo ? r : lo;
        hi = r;
      }
    }
    s_rng[is_row ? 0 : 2][k] = lo;
    s_rng[is_row ? 1 : 3][k] = hi;
  }
  __syncthreads();

  // ---- B: every owned full-resolution pixel, once
  double lsum = 0.0;
  int n_ok = 0, n_bad = 0;
  for (int e = tid; e < ry * rx; e += kSThreads) {
    const int r = e / rx, q = e - r * rx;
    const int Y = Ya + r, X = Xa + q;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    up_src_index(ay, Y, y0, y1, ly0, ly1);
    up_src_index(ax, X, x0, x1, lx0, lx1);
    const float* p00 = &s_l[((y0 - i0) * kSPX + (x0 - j0)) * kUpCP];
    const float* p01 = &s_l[((y0 - i0) * kSPX + (x1 - j0)) * kUpCP];
    const float* p10 = &s_l[((y1 - i0) * kSPX + (x0 - j0)) * kUpCP];
    const float* p11 = &s_l[((y1 - i0) * kSPX + (x1 - j0)) * kUpCP];
    float v[kUpCP];
    float m = -INFINITY;
#pragma unroll
    for (int c4 = 0; c4 < kUpCP / 4; ++c4) {
      const f32x4 a00 = *reinterpret_cast<const f32x4*>(p00 + c4 * 4), a01 = *reinterpret_cast<const f32x4*>(p01 + c4 * 4),
                  a10 = *reinterpret_cast<const f32x4*>(p10 + c4 * 4), a11 = *reinterpret_cast<const f32x4*>(p11 + c4 * 4);
      const f32x4 u = ly0 * (lx0 * a00 + lx1 * a01) + ly1 * (lx0 * a10 + lx1 * a11);     // bilinear_fwd's arithmetic
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[c4 * 4 + k] = u[k];
        if (c4 * 4 + k < C) m = fmaxf(m, u[k]);
      }
    }
    const long long t = target[((long long)b * H + Y) * W + X];
    const bool valid = ce_valid(t, ignore_index, C);
    float ssum = 0.f, vt = 0.f;
#pragma unroll
    for (int c = 0; c < kUpCP; ++c) {
      if (c < C) {
        if (t == c) vt = v[c];
        v[c] = __expf(v[c] - m);
        ssum += v[c];
      }
    }
    if (valid) lsum += (double)(m + __logf(ssum) - vt);
    n_ok += valid ? 1 : 0;
    n_bad += (!valid && t != ignore_index) ? 1 : 0;
    const float scale = valid ? 1.f / ssum : 0.f;
    float* gp = &s_g[(r * kSRX + q) * kUpCP];
#pragma unroll
    for (int c4 = 0; c4 < kUpCP / 4; ++c4) {
      f32x4 gv;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = c4 * 4 + k;
        gv[k] = c < C ? v[c] * scale - ((valid && t == c) ? 1.f : 0.f) : 0.f;
      }
      *reinterpret_cast<f32x4*>(gp + c4 * 4) = gv;
    }
  }
  __syncthreads();

  // ---- C: the patch: every (patch pixel, class) gathers over the OWNED pixels that touch it -- columns first, then rows, fixed
  // order -- and goes to the block's slot of the workspace
  if (patches != nullptr) {
    float* s_gx = s_g;      // [ry][kSPX][kUpCP] over the front of s_g's rows (row r of the column sums only reads row r of s_g)
    constexpr int kC4 = kUpCP / 4;
    const int items = ry * kSPX * kC4;
    constexpr int kHold = (kSRY * kSPX * kC4 + kSThreads - 1) / kSThreads;
    f32x4 hold[kHold];
#pragma unroll
    for (int u = 0; u < kHold; ++u) {
      const int e = tid + u * kSThreads;
      f32x4 row = {0.f, 0.f, 0.f, 0.f};
      if (e < items) {
        const int c4 = e % kC4, jj = (e / kC4) % kSPX, r = e / (kC4 * kSPX);
        const int q0 = s_rng[2][jj], q1 = s_rng[3][jj];
        for (int q = q0; q <= q1; ++q)
          row += s_wx[q][jj] * *reinterpret_cast<const f32x4*>(&s_g[(r * kSRX + q) * kUpCP + c4 * 4]);
      }
      hold[u] = row;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kHold; ++u) {
      const int e = tid + u * kSThreads;
      if (e < items) {
        const int c4 = e % kC4, jj = (e / kC4) % kSPX, r = e / (kC4 * kSPX);
        *reinterpret_cast<f32x4*>(&s_gx[(r * kSRX + jj) * kUpCP + c4 * 4]) = hold[u];
      }
    }
    __syncthreads();
    float* out = patches + (long long)blockIdx.x * (kSPY * kSPX * kUpCP);
    for (int o = tid; o < kSPY * kSPX * kC4; o += kSThreads) {
      const int c4 = o % kC4, px = o / kC4;
      const int ii = px / kSPX, jj = px % kSPX;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const int r0 = s_rng[0][ii], r1 = s_rng[1][ii];
      for (int r = r0; r <= r1; ++r) acc += s_wy[r][ii] * *reinterpret_cast<const f32x4*>(&s_gx[(r * kSRX + jj) * kUpCP + c4 * 4]);
      *reinterpret_cast<f32x4*>(out + px * kUpCP + c4 * 4) = acc;
    }
  }
  lsum = wave_sum_d(lsum);
  for (int o = 32; o > 0; o >>= 1) {
    n_ok += __shfl_xor(n_ok, o, 64);
    n_bad += __shfl_xor(n_bad, o, 64);
  }
  __shared__ int shc[2][kSThreads / 64];
  if ((tid & 63) == 0) {
    sh[tid >> 6] = lsum;
    shc[0][tid >> 6] = n_ok;
    shc[1][tid >> 6] = n_bad;
  }
  __syncthreads();
  if (tid == 0) {
    double tot = 0.0;
    int ok = 0, bad = 0;
    for (int i = 0; i < kSThreads / 64; ++i) {     // fixed order
      tot += sh[i];
      ok += shc[0][i];
      bad += shc[1][i];
    }
    partial[3 * (long long)blockIdx.x] = overrun ? (double)NAN : tot;
    partial[3 * (long long)blockIdx.x + 1] = (double)ok;       // (exact: integers far below 2^53)
    partial[3 * (long long)blockIdx.x + 2] = (double)bad;
  }
}

// one block, fixed order: the loss terms and the label counts of the scatter kernel's blocks -> loss_out and the header (whose
// n_valid the combine pass divides by)
__global__ __launch_bounds__(256) void ce_up_finish_kernel(const double* __restrict__ partial, int nblocks, CeHeader* __restrict__ hdr,
                                                           float* __restrict__ loss_out) {
  __shared__ double sh[4];
  double s[3] = {0.0, 0.0, 0.0};
  int i = threadIdx.x;
  for (; i + 3 * 256 < nblocks; i += 4 * 256) {
    double t[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int q = 0; q < 3; ++q) t[k][q] = partial[3 * (long long)(i + k * 256) + q];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int q = 0; q < 3; ++q) s[q] += t[k][q];
  }
  for (; i < nblocks; i += 256)
#pragma unroll
    for (int q = 0; q < 3; ++q) s[q] += partial[3 * (long long)i + q];
  double tot[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    tot[q] = block_sum_d(s[q], sh);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int nv = (int)tot[1];
    hdr->n_valid = nv;
    hdr->n_bad = (int)tot[2];
    loss_out[0] = nv > 0 ? (float)(tot[0] / (double)nv) : NAN;
    loss_out[1] = (float)nv;
    loss_out[2] = (float)tot[2];
  }
}



// dL[b, i, j, :] = own tile's patch entry + halo entries of the tiles above / to the left / diagonal (fixed order)
__global__ __launch_bounds__(256) void ce_up_combine_kernel(const float* __restrict__ patches, float* __restrict__ dL, int ldd,
                                                            int B, int h, int w, int tiles_y, int tiles_x,
                                                            const CeHeader* __restrict__ hdr) {
  constexpr int kC4 = kUpCP / 4;
  const int nv = hdr->n_valid;
  const float inv_n = nv > 0 ? 1.f / (float)nv : 0.f;
  const long long total = (long long)B * h * w * kC4;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int c4 = (int)(e % kC4);
    long long px = e / kC4;
    const int j = (int)(px % w);
    px /= w;
    const int i = (int)(px % h);
    const int b = (int)(px / h);
    const int ti = i / kSY, li = i - ti * kSY, tj = j / kSX, lj = j - tj * kSX;
    auto entry = [&](int tyy, int txx, int pi, int pj) {
      const long long blk = ((long long)b * tiles_y + tyy) * tiles_x + txx;
      return *reinterpret_cast<const f32x4*>(patches + blk * (kSPY * kSPX * kUpCP) + (pi * kSPX + pj) * kUpCP + c4 * 4);
    };
    f32x4 acc = entry(ti, tj, li, lj);
    if (li == 0 && ti > 0) acc += entry(ti - 1, tj, kSY, lj);
    if (lj == 0 && tj > 0) acc += entry(ti, tj - 1, li, kSX);
    if (li == 0 && ti > 0 && lj == 0 && tj > 0) acc += entry(ti - 1, tj - 1, kSY, kSX);
    acc *= inv_n;
    float* dp = dL + ((long long)(b * h + i) * w + j) * ldd + c4 * 4;
    if (c4 * 4 + 3 < ldd) {
      *reinterpret_cast<f32x4*>(dp) = acc;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (c4 * 4 + k < ldd) dp[k] = acc[k];
    }
  }
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
  hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(256), 0, st, (const double*)partial, blocks, (const CeHeader*)hdr,
                     loss_out);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

// the scatter formulation (ce_up_scatter_kernel) is the default; PSEG_CE_SCATTER=0 goes back to the gather kernel
static bool ce_scatter_on() {      // (read per call -- once per training step -- so that a test can compare the two forms in one process)
  const char* e = getenv("PSEG_CE_SCATTER");
  return e == nullptr || atoi(e) != 0;
}

// header + one double per block (whichever kernel: the gather kernel has more blocks) + the scatter kernel's patches
static int64_t ce_up_partial_bytes(int B, int h, int w) {
  const int64_t blocks = (int64_t)B * cdiv(h, kUpTY) * cdiv(w, kUpTX);      // (kSY x kSX is the same tile)
  return (((int64_t)sizeof(CeHeader) + blocks * 3 * 8) + 255) / 256 * 256;
}

int64_t pseg_ce_upsampled_workspace_bytes(int B, int h, int w) {
  if (B <= 0 || h <= 0 || w <= 0) return 0;
  const int64_t tiles = (int64_t)B * cdiv(h, kSY) * cdiv(w, kSX);
  return ce_up_partial_bytes(B, h, w) + tiles * (kSPY * kSPX * kUpCP) * 4;
}

int pseg_ce_upsampled_ok(int h, int w, int C, int H, int W, int align_corners) {
  // the region of full-resolution pixels a 4 x 8 tile can touch must fit the kernel's LDS image (with the +-2 slack of
  // up_region): scale factors of about 4 or more in both directions do not... smaller ones do
  if (h <= 0 || w <= 0 || H < h || W < w || C <= 0 || C > kUpCP) return 0;
  const double sy = align_corners ? (H > 1 ? (double)(h - 1) / (H - 1) : 0.0) : (double)h / H;
  const double sx = align_corners ? (W > 1 ? (double)(w - 1) / (W - 1) : 0.0) : (double)w / W;
  if (sy <= 0.0 || sx <= 0.0) return 0;
  // (destination indices whose first tap is one of TY + 1 / TX + 1 consecutive source indices: at most that many / scale + 2)
  double ry = (kUpTY + 1) / sy + 2.5, rx = (kUpTX + 1) / sx + 2.5;
  ry = ry > H ? H : ry;
  rx = rx > W ? W : rx;
  return (ry <= kUpRY && rx <= kUpRX) ? 1 : 0;
}

int pseg_ce_upsampled_fwd_bwd(const float* logits_lr, int ld, int B, int h, int w, int C, const int64_t* target, int H, int W,
                              int align_corners, int64_t ignore_index, float* dlogits_lr, int ldd, float* loss_out,
                              void* workspace, int64_t workspace_bytes, void* stream) {
  PSEG_REQUIRE(logits_lr && target && loss_out && workspace, "ce_upsampled: null pointer");
  PSEG_REQUIRE(pseg_ce_upsampled_ok(h, w, C, H, W, align_corners), "ce_upsampled: scale / class count not covered (use "
               "pseg_bilinear_fwd + pseg_ce_fwd_bwd): h=%d w=%d C=%d H=%d W=%d", h, w, C, H, W);
  PSEG_REQUIRE(ld % 4 == 0 && ld >= C && ((uintptr_t)logits_lr & 15) == 0, "ce_upsampled: logits need ld %% 4 == 0, ld >= C, "
               "16-byte alignment");
  PSEG_REQUIRE(!dlogits_lr || ldd >= C, "ce_upsampled: ldd < C");
  PSEG_REQUIRE(workspace_bytes >= pseg_ce_upsampled_workspace_bytes(B, h, w) && ((uintptr_t)workspace & 15) == 0,
               "ce_upsampled: workspace too small / misaligned");
  hipStream_t st = (hipStream_t)stream;
  CeHeader* hdr = (CeHeader*)workspace;
  double* partial = (double*)((char*)workspace + sizeof(CeHeader));
  const long long npix = (long long)B * H * W;
  UpAxis ay, ax;
  ay.in = h; ay.out = H; ay.align = align_corners;
  ax.in = w; ax.out = W; ax.align = align_corners;
  ay.scale = align_corners ? (H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f) : (float)h / (float)H;
  ax.scale = align_corners ? (W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f) : (float)w / (float)W;
  int tiles_y = cdiv(h, kUpTY), tiles_x = cdiv(w, kUpTX);
  long long blocks = (long long)B * tiles_y * tiles_x;
  PSEG_REQUIRE(blocks < (1LL << 31), "ce_upsampled: too many tiles");
  // Scatter form: the owned pixels of a tile must fit its LDS image.  max_owned = the largest number of destination indices
  // whose first tap falls into one tile of `tile` source indices -- exact: the kernel's own index function, evaluated here
  // (+ 2 where its arithmetic could contract differently on the device -- one pixel at either end: align_corners = 0)
  auto max_owned = [](const UpAxis& a, int tile) {
    int best = 0, cur_tile = -1, cur = 0;
    for (int o = 0; o < a.out; ++o) {
      int i0, i1;
      float l0, l1;
      up_src_index(a, o, i0, i1, l0, l1);
      const int t = i0 / tile;
      if (t != cur_tile) {
        cur_tile = t;
        cur = 0;
      }
      best = ++cur > best ? cur : best;
    }
    return best + (a.align ? 0 : 2);
  };
  const bool scatter = ce_scatter_on() && max_owned(ay, kSY) <= kSRY && max_owned(ax, kSX) <= kSRX;
  if (scatter) {
    // three launches: scatter (loss terms, label counts, unnormalised gradient patches) -> finish (loss, n_valid) -> combine
    tiles_y = cdiv(h, kSY);
    tiles_x = cdiv(w, kSX);
    blocks = (long long)B * tiles_y * tiles_x;
    float* patches = dlogits_lr ? (float*)((char*)workspace + ce_up_partial_bytes(B, h, w)) : nullptr;
    hipLaunchKernelGGL(ce_up_scatter_kernel, dim3((unsigned)blocks), dim3(kSThreads), 0, st, logits_lr, ld, B, C, target, ay, ax,
                       (long long)ignore_index, patches, partial, tiles_y, tiles_x);
    PSEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(ce_up_finish_kernel, dim3(1), dim3(256), 0, st, (const double*)partial, (int)blocks, hdr, loss_out);
    PSEG_LAUNCH_CHECK();
    if (dlogits_lr) {
      const long long total = (long long)B * h * w * (kUpCP / 4);
      hipLaunchKernelGGL(ce_up_combine_kernel, dim3(capped_blocks(total, 4096)), dim3(256), 0, st, (const float*)patches,
                         dlogits_lr, ldd, B, h, w, tiles_y, tiles_x, (const CeHeader*)hdr);
      PSEG_LAUNCH_CHECK();
    }
    return PSEG_OK;
  }
  if (hipMemsetAsync(hdr, 0, sizeof(CeHeader), st) != hipSuccess) {
    set_error("ce_upsampled: hipMemsetAsync failed");
    return PSEG_ERR_HIP;
  }
  hipLaunchKernelGGL(ce_count_kernel, dim3(capped_blocks(npix, 2048)), dim3(256), 0, st, target, npix,
                     (long long)ignore_index, C, hdr);
  PSEG_LAUNCH_CHECK();
  hipLaunchKernelGGL(ce_up_fused_kernel, dim3((unsigned)blocks), dim3(kUpThreads), 0, st, logits_lr, ld, B, C, target, ay, ax,
                     (long long)ignore_index, dlogits_lr, ldd, (const CeHeader*)hdr, partial, tiles_y, tiles_x);
  PSEG_LAUNCH_CHECK();
  hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(256), 0, st, (const double*)partial, (int)blocks, (const CeHeader*)hdr,
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
