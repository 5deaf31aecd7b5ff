// Pooling, bilinear resize, max-pool and layout kernels for NHWC fp32 on gfx950 (all HBM-bound).
//  - pool_sum / broadcast: AdaptiveAvgPool2d(1) and the constant "bilinear from 1x1" broadcast of the
//    ASPP image-level branch (reference models/aspp.py:11,16-19) and their backward.
//  - bilinear fwd/bwd: F.interpolate(mode='bilinear') with align_corners True/False (reference
//    models/deeplabv3plus.py:34-37,40-43, models/unet.py:30-55, utils/utils.py:18-20).  Source index and
//    weights follow ATen's area_pixel_compute_source_index in fp32.  Backward is a GATHER over the output
//    pixels that touch an input pixel (fixed summation order, no float atomics).
//  - maxpool 3x3/s2 of the ResNet stem, backward via saved window positions (first maximum wins, as ATen).
#include "common.h"
#include "half_io.h"

#include <math.h>

namespace pseg {

// 4 consecutive channels of an fp32 or fp16 tensor <-> f32x4; the NHWC kernels are templates on the storage type T
template <typename T>
__device__ __forceinline__ f32x4 ld4(const T* p) { return ldv4(p); }
template <typename T>
__device__ __forceinline__ void st4(T* p, f32x4 v) { stv4(p, v); }

// ------------------------------------------------------------------ pool_sum: out[b][c] = scale * sum_p x[b][p][c]
template <typename T>
__global__ __launch_bounds__(256) void pool_sum_kernel(const T* __restrict__ x, int ldx, int HW, int C, float scale,
                                                       T* __restrict__ out, int ldo) {
  PSEG_HELPER_PRIO();
  __shared__ f32x4 sh[256];
  const int TX = blockDim.x, TY = blockDim.y;
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int c4 = blockIdx.y * TX + tx;
  const bool cok = c4 * 4 < C;
  const int b = blockIdx.x;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (cok) {
    const T* xb = x + (long long)b * HW * ldx + c4 * 4;
    for (int p = ty; p < HW; p += TY) s += ld4(xb + (long long)p * ldx);
  }
  sh[ty * TX + tx] = s;
  __syncthreads();
  if (ty == 0 && cok) {
    f32x4 t = sh[tx];
    for (int j = 1; j < TY; ++j) t += sh[j * TX + tx];
    st4(out + (long long)b * ldo + c4 * 4, t * scale);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void broadcast_kernel(const T* __restrict__ x, int ldx, float scale,
                                                        T* __restrict__ y, int ldy, int accumulate, uint32_t total,
                                                        FastDiv c4div, FastDiv hwdiv) {
  PSEG_HELPER_PRIO();
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t r = c4div.div(i);
    const uint32_t c = (i - r * c4div.d) * 4;
    const uint32_t b = hwdiv.div(r);
    const f32x4 v = ld4(x + (long long)b * ldx + c) * scale;
    T* yp = y + (long long)r * ldy + c;
    st4(yp, accumulate ? ld4(yp) + v : v);
  }
}

// ------------------------------------------------------------------ bilinear
struct Axis {
  float scale;  // source units per destination unit
  int in, out;
  int align;
};

__device__ __forceinline__ void src_index(const Axis& a, int o, int& i0, int& i1, float& l0, float& l1) {
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

// weight with which destination index o reads source index i
__device__ __forceinline__ float tap_weight(const Axis& a, int o, int i) {
  int i0, i1;
  float l0, l1;
  src_index(a, o, i0, i1, l0, l1);
  float w = 0.f;
  if (i0 == i) w += l0;
  if (i1 == i) w += l1;
  return w;
}

// conservative range of destination indices that can read source index i
__device__ __forceinline__ void dst_range(const Axis& a, int i, int& lo, int& hi) {
  if (a.scale <= 0.f) {
    lo = 0;
    hi = a.out - 1;
    return;
  }
  const float inv = 1.f / a.scale;
  float flo, fhi;
  if (a.align) {
    flo = ((float)i - 1.f) * inv;
    fhi = ((float)i + 1.f) * inv;
  } else {
    flo = ((float)i - 0.5f) * inv - 0.5f;
    fhi = ((float)i + 1.5f) * inv - 0.5f;
  }
  lo = (int)floorf(flo) - 1;
  hi = (int)ceilf(fhi) + 1;
  if (lo < 0) lo = 0;
  if (hi > a.out - 1) hi = a.out - 1;
}

struct ResizeParams {
  Axis h, w;
  int B, C;
  int ldx, ldy;
  FastDiv c4div, pixdiv, rowdiv;   // NHWC forms: /C4, /(Hd*Wd), /Wd of the thread-space pixel grid
  FastDiv chwdiv, hwdiv, wdiv;     // NCHW forms: /(C*H*W), /(H*W), /W of the thread-space grid
};

// NHWC -> NHWC (possibly a channel slice of a concat buffer)
template <typename T>
__global__ __launch_bounds__(256) void bilinear_fwd_nhwc_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                                ResizeParams p, uint32_t total) {
  PSEG_HELPER_PRIO();
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t pix = p.c4div.div(i);
    const uint32_t c = (i - pix * p.c4div.d) * 4;
    const uint32_t b = p.pixdiv.div(pix);
    const uint32_t rem = pix - b * p.pixdiv.d;
    const uint32_t ho = p.rowdiv.div(rem);
    const uint32_t wo = rem - ho * p.rowdiv.d;
    int h0, h1, w0, w1;
    float lh0, lh1, lw0, lw1;
    src_index(p.h, (int)ho, h0, h1, lh0, lh1);
    src_index(p.w, (int)wo, w0, w1, lw0, lw1);
    const T* xb = x + (long long)b * p.h.in * p.w.in * p.ldx + c;
    const f32x4 p00 = ld4(xb + (long long)(h0 * p.w.in + w0) * p.ldx);
    const f32x4 p01 = ld4(xb + (long long)(h0 * p.w.in + w1) * p.ldx);
    const f32x4 p10 = ld4(xb + (long long)(h1 * p.w.in + w0) * p.ldx);
    const f32x4 p11 = ld4(xb + (long long)(h1 * p.w.in + w1) * p.ldx);
    st4(y + (long long)pix * p.ldy + c, lh0 * (lw0 * p00 + lw1 * p01) + lh1 * (lw0 * p10 + lw1 * p11));
  }
}

// NHWC -> contiguous NCHW (the logits the loss / argmax consume): one thread per (b, 4-channel group, ho, wo), wo fastest.
// The four source pixels are read as 16-byte channel quads (every byte fetched is used) and the four planes are written
// with 4-byte stores that are contiguous across the wave.  x must be readable up to channel 4*ceil(C/4) (ld >= that).
__global__ __launch_bounds__(256) void bilinear_fwd_nchw_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                ResizeParams p, uint32_t total) {
  PSEG_HELPER_PRIO();
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t b = p.chwdiv.div(i);          // / (C4 * Ho * Wo)
    uint32_t rem = i - b * p.chwdiv.d;
    const uint32_t cg = p.hwdiv.div(rem);        // / (Ho * Wo)
    rem -= cg * p.hwdiv.d;
    const uint32_t ho = p.wdiv.div(rem);
    const uint32_t wo = rem - ho * p.wdiv.d;
    int h0, h1, w0, w1;
    float lh0, lh1, lw0, lw1;
    src_index(p.h, (int)ho, h0, h1, lh0, lh1);
    src_index(p.w, (int)wo, w0, w1, lw0, lw1);
    const float* xb = x + (long long)b * p.h.in * p.w.in * p.ldx + cg * 4;
    const f32x4 p00 = ld4(xb + (long long)(h0 * p.w.in + w0) * p.ldx);
    const f32x4 p01 = ld4(xb + (long long)(h0 * p.w.in + w1) * p.ldx);
    const f32x4 p10 = ld4(xb + (long long)(h1 * p.w.in + w0) * p.ldx);
    const f32x4 p11 = ld4(xb + (long long)(h1 * p.w.in + w1) * p.ldx);
    const f32x4 v = lh0 * (lw0 * p00 + lw1 * p01) + lh1 * (lw0 * p10 + lw1 * p11);
    const long long plane = (long long)p.h.out * p.w.out;
    float* yp = y + ((long long)b * p.C + cg * 4) * plane + (long long)ho * p.w.out + wo;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if ((int)(cg * 4) + e < p.C) yp[e * plane] = v[e];
  }
}

// backward, NHWC grads -> NHWC: one thread per (input pixel, 4 channels)
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_nhwc_kernel(const T* __restrict__ dy, T* __restrict__ dx,
                                                                ResizeParams p, int accumulate, uint32_t total) {
  PSEG_HELPER_PRIO();
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t pix = p.c4div.div(i);
    const uint32_t c = (i - pix * p.c4div.d) * 4;
    const uint32_t b = p.pixdiv.div(pix);
    const uint32_t rem = pix - b * p.pixdiv.d;
    const int hi = (int)p.rowdiv.div(rem);
    const int wi = (int)(rem - (uint32_t)hi * p.rowdiv.d);
    int hlo, hhi, wlo, whi;
    dst_range(p.h, hi, hlo, hhi);
    dst_range(p.w, wi, wlo, whi);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const T* db = dy + (long long)b * p.h.out * p.w.out * p.ldy + c;
    for (int ho = hlo; ho <= hhi; ++ho) {
      const float wh = tap_weight(p.h, ho, hi);
      if (wh == 0.f) continue;
      for (int wo = wlo; wo <= whi; ++wo) {
        const float ww = tap_weight(p.w, wo, wi);
        if (ww == 0.f) continue;
        acc += (wh * ww) * ld4(db + (long long)(ho * p.w.out + wo) * p.ldy);
      }
    }
    T* dp = dx + (long long)pix * p.ldx + c;
    st4(dp, accumulate ? ld4(dp) + acc : acc);
  }
}

// backward from a contiguous NCHW gradient (dlogits) into NHWC: one thread per (b, 4-channel group, hi, wi), wi fastest.
// The column weights of the source pixel are evaluated once into registers (the row weights once per row) and shared by
// the four channel planes; the result leaves as one 16-byte store.  Channels >= C of the group are written as zeros.
constexpr int kMaxTaps = 12;   // destination columns one source column can feed after trimming (scale factors up to ~5)
__global__ __launch_bounds__(256) void bilinear_bwd_nchw_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                                ResizeParams p, int accumulate, uint32_t total) {
  PSEG_HELPER_PRIO();
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t b = p.chwdiv.div(i);          // / (C4 * Hi * Wi)
    uint32_t rem = i - b * p.chwdiv.d;
    const uint32_t cg = p.hwdiv.div(rem);        // / (Hi * Wi)
    rem -= cg * p.hwdiv.d;
    const int hi = (int)p.wdiv.div(rem);
    const int wi = (int)(rem - (uint32_t)hi * p.wdiv.d);
    int hlo, hhi, wlo, whi;
    dst_range(p.h, hi, hlo, hhi);
    dst_range(p.w, wi, wlo, whi);
    // the ranges are conservative: trim the zero-weight ends (a third of the taps at scale 4)
    while (wlo < whi && tap_weight(p.w, wlo, wi) == 0.f) ++wlo;
    while (whi > wlo && tap_weight(p.w, whi, wi) == 0.f) --whi;
    while (hlo < hhi && tap_weight(p.h, hlo, hi) == 0.f) ++hlo;
    while (hhi > hlo && tap_weight(p.h, hhi, hi) == 0.f) --hhi;
    const int c0 = (int)cg * 4;
    const long long plane = (long long)p.h.out * p.w.out;
    const float* db = dy + ((long long)b * p.C + c0) * plane;
    const int nch = p.C - c0 < 4 ? p.C - c0 : 4;
    const int nw = whi - wlo + 1;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (nw <= kMaxTaps) {
      float ww[kMaxTaps];
#pragma unroll
      for (int t = 0; t < kMaxTaps; ++t) ww[t] = t < nw ? tap_weight(p.w, wlo + t, wi) : 0.f;
      for (int ho = hlo; ho <= hhi; ++ho) {
        const float wh = tap_weight(p.h, ho, hi);
        const float* row = db + (long long)ho * p.w.out + wlo;
        if (nch == 4) {
#pragma unroll
          for (int t = 0; t < kMaxTaps; ++t)
            if (t < nw) {
              const float w = wh * ww[t];
              acc[0] += w * row[t];
              acc[1] += w * row[plane + t];
              acc[2] += w * row[2 * plane + t];
              acc[3] += w * row[3 * plane + t];
            }
        } else {
#pragma unroll
          for (int t = 0; t < kMaxTaps; ++t)
            if (t < nw) {
              const float w = wh * ww[t];
              for (int e = 0; e < nch; ++e) acc[e] += w * row[e * plane + t];
            }
        }
      }
    } else {   // extreme scale factors: weights evaluated in place
      for (int ho = hlo; ho <= hhi; ++ho) {
        const float wh = tap_weight(p.h, ho, hi);
        for (int wo = wlo; wo <= whi; ++wo) {
          const float w = wh * tap_weight(p.w, wo, wi);
          for (int e = 0; e < nch; ++e) acc[e] += w * db[e * plane + (long long)ho * p.w.out + wo];
        }
      }
    }
    float* dp = dx + ((long long)(b * p.h.in + hi) * p.w.in + wi) * p.ldx + c0;
    st4(dp, accumulate ? ld4(dp) + acc : acc);
  }
}

// Separable form of the NCHW backward (needs B*C*Ho*Wi floats of scratch): the tent weights factor, so
//   tmp[b][c][ho][wi] = sum_wo ww(wo, wi) dy[b][c][ho][wo]          (pass W: one thread per tmp element, wi fastest)
//   dx[b][hi][wi][c]  = sum_ho wh(ho, hi) tmp[b][c][ho][wi]         (pass H: one thread per (b, 4-channel group, hi, wi))
// which reads every gradient element ~2x instead of (2*scale)^2/... times and does ~2*scale taps per output, not (2*scale)^2.
constexpr int kRowsPerThread = 16;
__global__ __launch_bounds__(256) void bilinear_bwd_nchw_w_kernel(const float* __restrict__ dy, float* __restrict__ tmp,
                                                                  ResizeParams p, uint32_t nrows, uint32_t total) {
  PSEG_HELPER_PRIO();
  // thread = (group of kRowsPerThread (b, c, ho) rows, wi), wi fastest: the column weights of wi are evaluated once into
  // registers and reused for every row of the group
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t grp = p.wdiv.div(i);                 // wdiv = / Wi
    const int wi = (int)(i - grp * p.wdiv.d);
    int wlo, whi;
    dst_range(p.w, wi, wlo, whi);
    while (wlo < whi && tap_weight(p.w, wlo, wi) == 0.f) ++wlo;
    while (whi > wlo && tap_weight(p.w, whi, wi) == 0.f) --whi;
    const int nw = whi - wlo + 1;
    const uint32_t r0 = grp * kRowsPerThread;
    const uint32_t r1 = r0 + kRowsPerThread < nrows ? r0 + kRowsPerThread : nrows;
    if (nw <= kMaxTaps) {
      float ww[kMaxTaps];
#pragma unroll
      for (int t = 0; t < kMaxTaps; ++t) ww[t] = t < nw ? tap_weight(p.w, wlo + t, wi) : 0.f;
      for (uint32_t row = r0; row < r1; ++row) {
        const float* src = dy + (long long)row * p.w.out + wlo;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < kMaxTaps; ++t)
          if (t < nw) acc += ww[t] * src[t];
        tmp[(long long)row * p.w.in + wi] = acc;
      }
    } else {
      for (uint32_t row = r0; row < r1; ++row) {
        const float* src = dy + (long long)row * p.w.out;
        float acc = 0.f;
        for (int wo = wlo; wo <= whi; ++wo) acc += tap_weight(p.w, wo, wi) * src[wo];
        tmp[(long long)row * p.w.in + wi] = acc;
      }
    }
  }
}

constexpr int kChunks = 4;
// Pass W with the gradient rows staged in LDS: a block fetches RB = 8192/Wo rows with 16-byte coalesced loads (one
// barrier), then every thread owns one output column (its ~2*scale weights live in registers) of every (256/Wi_pad)-th
// row and reads its taps from LDS.  (Wi <= 256, Wo <= 2048, Wo % 4 == 0, 16-byte aligned rows; other shapes take the
// register kernel above.)
__global__ __launch_bounds__(256) void bilinear_bwd_nchw_w_lds_kernel(const float* __restrict__ dy, float* __restrict__ tmp,
                                                                      ResizeParams p, uint32_t nrows, int RB, int wi_pad) {
  PSEG_HELPER_PRIO();
  __shared__ __attribute__((aligned(16))) float row_s[8192];
  const int wi = threadIdx.x % wi_pad;
  const int rsub = threadIdx.x / wi_pad, rstep = 256 / wi_pad;
  const bool act = wi < p.w.in;
  int wlo = 0, whi = 0, nw = 0;
  float ww[kMaxTaps];
#pragma unroll
  for (int t = 0; t < kMaxTaps; ++t) ww[t] = 0.f;
  if (act) {
    dst_range(p.w, wi, wlo, whi);
    while (wlo < whi && tap_weight(p.w, wlo, wi) == 0.f) ++wlo;
    while (whi > wlo && tap_weight(p.w, whi, wi) == 0.f) --whi;
    nw = whi - wlo + 1;
#pragma unroll
    for (int t = 0; t < kMaxTaps; ++t) ww[t] = t < nw ? tap_weight(p.w, wlo + t, wi) : 0.f;
  }
  const int q = p.w.out >> 2;                                   // float4 per row
  // kChunks chunks of RB rows per block: the column weights above are set up once for all of them
  for (int ch = 0; ch < kChunks; ++ch) {
    const uint32_t r0 = (blockIdx.x * kChunks + ch) * RB;
    if (r0 >= nrows) break;
    const int rb = (int)(r0 + RB <= nrows ? RB : nrows - r0);
    const f32x4* src4 = reinterpret_cast<const f32x4*>(dy + (long long)r0 * p.w.out);
    if (ch) __syncthreads();                                    // everyone is done reading the previous chunk
    for (int j = threadIdx.x; j < rb * q; j += 256) reinterpret_cast<f32x4*>(row_s)[j] = src4[j];
    __syncthreads();
    if (act) {
      for (int r = rsub; r < rb; r += rstep) {
        const float* src = row_s + r * p.w.out + wlo;
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < kMaxTaps; ++t)
          if (t < nw) acc += ww[t] * src[t];
        tmp[(long long)(r0 + r) * p.w.in + wi] = acc;
      }
    }
  }
}

__global__ __launch_bounds__(256) void bilinear_bwd_nchw_h_kernel(const float* __restrict__ tmp, float* __restrict__ dx,
                                                                  ResizeParams p, int accumulate, uint32_t total) {
  PSEG_HELPER_PRIO();
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t b = p.chwdiv.div(i);          // / (C4 * Hi * Wi)
    uint32_t rem = i - b * p.chwdiv.d;
    const uint32_t cg = p.hwdiv.div(rem);        // / (Hi * Wi)
    rem -= cg * p.hwdiv.d;
    const int hi = (int)p.wdiv.div(rem);
    const int wi = (int)(rem - (uint32_t)hi * p.wdiv.d);
    int hlo, hhi;
    dst_range(p.h, hi, hlo, hhi);
    const int c0 = (int)cg * 4;
    const int nch = p.C - c0 < 4 ? p.C - c0 : 4;
    const long long plane = (long long)p.h.out * p.w.in;      // one (b, c) slab of tmp
    const float* src = tmp + ((long long)b * p.C + c0) * plane + wi;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ho = hlo; ho <= hhi; ++ho) {
      const float wh = tap_weight(p.h, ho, hi);
      if (wh == 0.f) continue;
      const float* r = src + (long long)ho * p.w.in;
      for (int e = 0; e < nch; ++e) acc[e] += wh * r[e * plane];
    }
    float* dp = dx + ((long long)(b * p.h.in + hi) * p.w.in + wi) * p.ldx + c0;
    st4(dp, accumulate ? ld4(dp) + acc : acc);
  }
}

// ------------------------------------------------------------------ max pool
struct PoolParams {
  int B, H, W, C, Ho, Wo, k, stride, pad, ldx, ldy;
  FastDiv c4div, pixdiv, rowdiv;
};

// BN (the ResNet stem in training: conv -> BatchNorm -> ReLU -> max-pool, where nothing else reads the activated map): x is the
// conv's raw output and every tap is normalised + activated on the way in -- z = act((x - mean) * scale + shift), bn_act_fwd_kernel's
// expression, rounded to the storage type as that kernel stores it -- so the activated map (268 MB at the benchmark shape) is
// neither written nor read back; its BatchNorm backward recomputes the mask from x, the pooling backward needs only `arg`.
template <typename T, bool BN = false>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          uint8_t* __restrict__ arg, PoolParams p, uint32_t total,
                                                          const float* __restrict__ mean = nullptr,
                                                          const float* __restrict__ scale = nullptr,
                                                          const float* __restrict__ shift = nullptr, int act = 0) {
  PSEG_HELPER_PRIO();
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t pix = p.c4div.div(i);
    const uint32_t c = (i - pix * p.c4div.d) * 4;
    const uint32_t b = p.pixdiv.div(pix);
    const uint32_t rem = pix - b * p.pixdiv.d;
    const int ho = (int)p.rowdiv.div(rem);
    const int wo = (int)(rem - (uint32_t)ho * p.rowdiv.d);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    uint32_t pos[4] = {0, 0, 0, 0};
    bool any = false;
    f32x4 bmu = {0.f, 0.f, 0.f, 0.f}, bsc = bmu, bsh = bmu;
    if constexpr (BN) {
      bmu = *reinterpret_cast<const f32x4*>(mean + c);
      bsc = *reinterpret_cast<const f32x4*>(scale + c);
      bsh = *reinterpret_cast<const f32x4*>(shift + c);
    }
    for (int r = 0; r < p.k; ++r) {
      const int hi = ho * p.stride - p.pad + r;
      if ((unsigned)hi >= (unsigned)p.H) continue;
      for (int s = 0; s < p.k; ++s) {
        const int wi = wo * p.stride - p.pad + s;
        if ((unsigned)wi >= (unsigned)p.W) continue;
        f32x4 v = ld4(x + ((long long)(b * p.H + hi) * p.W + wi) * p.ldx + c);
        if constexpr (BN) {
          v = (v - bmu) * bsc + bsh;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (act == PSEG_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
            else if (act == PSEG_ACT_RELU6) v[e] = fminf(fmaxf(v[e], 0.f), 6.f);
            if constexpr (sizeof(T) == 2) v[e] = (float)(half_t)v[e];      // (the value as the separate pass would have stored it)
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (!any || v[e] > best[e]) {
            best[e] = v[e];
            pos[e] = (uint32_t)(r * p.k + s);
          }
        }
        any = true;
      }
    }
    st4(y + (long long)pix * p.ldy + c, best);
    if (arg) {
      const uint32_t packed = pos[0] | (pos[1] << 8) | (pos[2] << 16) | (pos[3] << 24);
      *reinterpret_cast<uint32_t*>(arg + (long long)pix * p.C + c) = packed;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ arg,
                                                          T* __restrict__ dx, PoolParams p, int accumulate,
                                                          uint32_t total) {
  PSEG_HELPER_PRIO();
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t pix = p.c4div.div(i);  // input pixel
    const uint32_t c = (i - pix * p.c4div.d) * 4;
    const uint32_t b = p.pixdiv.div(pix);
    const uint32_t rem = pix - b * p.pixdiv.d;
    const int hi = (int)p.rowdiv.div(rem);
    const int wi = (int)(rem - (uint32_t)hi * p.rowdiv.d);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // output rows / columns whose window covers this input pixel: ho in [ceil((hi+pad-k+1)/s), floor((hi+pad)/s)]
    const int hn = hi + p.pad, wn = wi + p.pad;
    int ho_hi = hn / p.stride, wo_hi = wn / p.stride;
    int ho_lo = hn - (p.k - 1) <= 0 ? 0 : (hn - (p.k - 1) + p.stride - 1) / p.stride;
    int wo_lo = wn - (p.k - 1) <= 0 ? 0 : (wn - (p.k - 1) + p.stride - 1) / p.stride;
    if (ho_hi > p.Ho - 1) ho_hi = p.Ho - 1;
    if (wo_hi > p.Wo - 1) wo_hi = p.Wo - 1;
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
      const int r = hn - ho * p.stride;
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        const int sx = wn - wo * p.stride;
        const long long opix = (long long)(b * p.Ho + ho) * p.Wo + wo;
        const uint32_t packed = *reinterpret_cast<const uint32_t*>(arg + opix * p.C + c);
        const f32x4 g = ld4(dy + opix * p.ldy + c);
        const uint32_t me = (uint32_t)(r * p.k + sx);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (((packed >> (8 * e)) & 0xFFu) == me) acc[e] += g[e];
      }
    }
    T* dp = dx + (long long)pix * p.ldx + c;
    st4(dp, accumulate ? ld4(dp) + acc : acc);
  }
}

// ------------------------------------------------------------------ layout
// x[b][c][hw] -> y[b][hw][ld] for tiny C (<= 4, padded to 4): one thread per pixel, 16-byte store
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float* __restrict__ x, float* __restrict__ y, int ldy,
                                                            int C, uint32_t HW, uint32_t total, FastDiv hwdiv) {
  PSEG_HELPER_PRIO();
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t b = hwdiv.div(i);
    const uint32_t p = i - b * HW;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const float* xb = x + (long long)b * C * HW + p;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < C) v[c] = xb[(long long)c * HW];
    st4(y + (long long)i * ldy, v);
  }
}

// the same into an fp16 tensor whose pixels carry 8 channels (the half-precision policy's image input): C <= 8, one 16-byte
// store per pixel -- instead of a generic transpose to 8 fp32 channels and a conversion pass (113 + 18 us at 16x3x512x512)
__global__ __launch_bounds__(256) void nchw_to_nhwc8h_kernel(const float* __restrict__ x, half_t* __restrict__ y, int ldy,
                                                             int C, uint32_t HW, uint32_t total, FastDiv hwdiv) {
  PSEG_HELPER_PRIO();
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t b = hwdiv.div(i);
    const uint32_t p = i - b * HW;
    fvec<8> v = {};
    const float* xb = x + (long long)b * C * HW + p;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < C) v[c] = xb[(long long)c * HW];
    stvec<8>(y + (long long)i * ldy, v);
  }
}

// generic batched transpose through a 32x33 LDS tile: in[b][R][Cc] (row stride ldi) -> out[b][Cc][R] (row stride ldo);
// columns of the output in [R, Rpad) are zero-filled (used for channel padding).
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, long long in_bstride, int ldi,
                                                        float* __restrict__ out, long long out_bstride, int ldo, int R,
                                                        int Cc, int Rpad) {
  PSEG_HELPER_PRIO();
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* ib = in + (long long)b * in_bstride;
  float* ob = out + (long long)b * out_bstride;
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < R && c < Cc) ? ib[(long long)r * ldi + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (c < Cc && r < Rpad) ob[(long long)c * ldo + r] = tile[tx][j];
  }
}

// ------------------------------------------------------------------ host
static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
static int ew_grid(long long total) {
  long long b = (total + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

static Axis make_axis(int in, int out, int align) {
  Axis a;
  a.in = in;
  a.out = out;
  a.align = align;
  if (align) a.scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  else a.scale = (float)in / (float)out;
  return a;
}

template <typename T>
static bool ld_ok(int ld) { return ld % (sizeof(T) == 2 ? 8 : 4) == 0; }

template <typename T>
static int pool_sum_impl(const T* x, int ldx, int B, int HW, int C, float scale, T* out, int ldo, void* stream) {
  PSEG_REQUIRE(x && out && B > 0 && HW > 0 && C > 0 && C % 4 == 0, "pool_sum: bad argument");
  PSEG_REQUIRE(ld_ok<T>(ldx) && ld_ok<T>(ldo) && al16(x) && al16(out), "pool_sum: alignment");
  const int c4 = C / 4;
  int tx = 64;
  while (tx > 16 && (long long)B * cdiv(c4, tx) < 512) tx >>= 1;
  hipLaunchKernelGGL(pool_sum_kernel<T>, dim3(B, cdiv(c4, tx)), dim3(tx, 256 / tx), 0, (hipStream_t)stream, x, ldx, HW, C,
                     scale, out, ldo);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

template <typename T>
static int broadcast_impl(const T* x, int ldx, int B, int HW, int C, float scale, T* y, int ldy, int accumulate,
                          void* stream) {
  PSEG_REQUIRE(x && y && B > 0 && HW > 0 && C > 0 && C % 4 == 0, "broadcast: bad argument");
  PSEG_REQUIRE(ld_ok<T>(ldx) && ld_ok<T>(ldy) && al16(x) && al16(y), "broadcast: alignment");
  const long long total = (long long)B * HW * (C / 4);
  PSEG_REQUIRE(total < (1LL << 31), "broadcast: tensor too large");
  hipLaunchKernelGGL(broadcast_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, scale, y, ldy,
                     accumulate, (uint32_t)total, FastDiv((uint32_t)(C / 4)), FastDiv((uint32_t)HW));
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

template <typename T>
static int bilinear_fwd_nhwc_impl(const T* x, int ldx, int B, int Hi, int Wi, int C, T* y, int ldy, int Ho, int Wo,
                                  int align_corners, void* stream) {
  PSEG_REQUIRE(x && y && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "bilinear_fwd: bad argument");
  ResizeParams p;
  p.h = make_axis(Hi, Ho, align_corners);
  p.w = make_axis(Wi, Wo, align_corners);
  p.B = B;
  p.C = C;
  p.ldx = ldx;
  p.ldy = ldy;
  PSEG_REQUIRE(C % 4 == 0 && ld_ok<T>(ldx) && ld_ok<T>(ldy) && al16(x) && al16(y), "bilinear_fwd: NHWC alignment");
  const long long total = (long long)B * Ho * Wo * (C / 4);
  PSEG_REQUIRE(total < (1LL << 31), "bilinear_fwd: tensor too large");
  p.c4div = FastDiv((uint32_t)(C / 4));
  p.pixdiv = FastDiv((uint32_t)(Ho * Wo));
  p.rowdiv = FastDiv((uint32_t)Wo);
  hipLaunchKernelGGL(bilinear_fwd_nhwc_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, p,
                     (uint32_t)total);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

template <typename T>
static int bilinear_bwd_nhwc_impl(const T* dy, int ldy, int B, int Hi, int Wi, int C, T* dx, int ldx, int Ho, int Wo,
                                  int align_corners, int accumulate, void* stream) {
  PSEG_REQUIRE(dy && dx && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "bilinear_bwd: bad argument");
  ResizeParams p;
  p.h = make_axis(Hi, Ho, align_corners);
  p.w = make_axis(Wi, Wo, align_corners);
  p.B = B;
  p.C = C;
  p.ldx = ldx;
  p.ldy = ldy;
  PSEG_REQUIRE(C % 4 == 0 && ld_ok<T>(ldx) && ld_ok<T>(ldy) && al16(dx) && al16(dy), "bilinear_bwd: NHWC alignment");
  const long long total = (long long)B * Hi * Wi * (C / 4);
  PSEG_REQUIRE(total < (1LL << 31), "bilinear_bwd: tensor too large");
  p.c4div = FastDiv((uint32_t)(C / 4));
  p.pixdiv = FastDiv((uint32_t)(Hi * Wi));
  p.rowdiv = FastDiv((uint32_t)Wi);
  hipLaunchKernelGGL(bilinear_bwd_nhwc_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, dx, p,
                     accumulate, (uint32_t)total);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

}  // namespace pseg

using namespace pseg;

#define HP(p) reinterpret_cast<const half_t*>(p)
#define HPM(p) reinterpret_cast<half_t*>(p)

extern "C" {

int pseg_pool_sum(const float* x, int ldx, int B, int HW, int C, float scale, float* out, int ldo, void* stream) {
  return pool_sum_impl<float>(x, ldx, B, HW, C, scale, out, ldo, stream);
}
int pseg_pool_sum_h(const pseg_half_t* x, int ldx, int B, int HW, int C, float scale, pseg_half_t* out, int ldo,
                    void* stream) {
  return pool_sum_impl<half_t>(HP(x), ldx, B, HW, C, scale, HPM(out), ldo, stream);
}

int pseg_broadcast(const float* x, int ldx, int B, int HW, int C, float scale, float* y, int ldy, int accumulate,
                   void* stream) {
  return broadcast_impl<float>(x, ldx, B, HW, C, scale, y, ldy, accumulate, stream);
}
int pseg_broadcast_h(const pseg_half_t* x, int ldx, int B, int HW, int C, float scale, pseg_half_t* y, int ldy,
                     int accumulate, void* stream) {
  return broadcast_impl<half_t>(HP(x), ldx, B, HW, C, scale, HPM(y), ldy, accumulate, stream);
}

int pseg_bilinear_fwd_h(const pseg_half_t* x, int ldx, int B, int Hi, int Wi, int C, pseg_half_t* y, int ldy, int Ho, int Wo,
                        int align_corners, void* stream) {
  return bilinear_fwd_nhwc_impl<half_t>(HP(x), ldx, B, Hi, Wi, C, HPM(y), ldy, Ho, Wo, align_corners, stream);
}
int pseg_bilinear_bwd_h(const pseg_half_t* dy, int ldy, int B, int Hi, int Wi, int C, pseg_half_t* dx, int ldx, int Ho,
                        int Wo, int align_corners, int accumulate, void* stream) {
  return bilinear_bwd_nhwc_impl<half_t>(HP(dy), ldy, B, Hi, Wi, C, HPM(dx), ldx, Ho, Wo, align_corners, accumulate, stream);
}

int pseg_bilinear_fwd(const float* x, int ldx, int B, int Hi, int Wi, int C, float* y, int ldy, int Ho, int Wo,
                      int align_corners, int out_nchw, void* stream) {
  if (!out_nchw) return bilinear_fwd_nhwc_impl<float>(x, ldx, B, Hi, Wi, C, y, ldy, Ho, Wo, align_corners, stream);
  PSEG_REQUIRE(x && y && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "bilinear_fwd: bad argument");
  ResizeParams p;
  p.h = make_axis(Hi, Ho, align_corners);
  p.w = make_axis(Wi, Wo, align_corners);
  p.B = B;
  p.C = C;
  p.ldx = ldx;
  p.ldy = ldy;
  const int C4 = (C + 3) / 4;
  PSEG_REQUIRE(ldx % 4 == 0 && ldx >= 4 * C4 && al16(x), "bilinear_fwd: NHWC source must be 16-byte aligned with ld >= C rounded up to 4");
  const long long total = (long long)B * C4 * Ho * Wo;
  PSEG_REQUIRE((long long)B * C * Ho * Wo < (1LL << 31), "bilinear_fwd: tensor too large");
  p.chwdiv = FastDiv((uint32_t)((long long)C4 * Ho * Wo));
  p.hwdiv = FastDiv((uint32_t)(Ho * Wo));
  p.wdiv = FastDiv((uint32_t)Wo);
  hipLaunchKernelGGL(bilinear_fwd_nchw_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, p,
                     (uint32_t)total);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

int64_t pseg_bilinear_bwd_workspace_bytes(int B, int Hi, int Wi, int C, int Ho, int Wo, int dy_nchw) {
  (void)Hi;
  (void)Wo;
  return dy_nchw ? (int64_t)B * C * Ho * Wi * 4 : 0;
}

int pseg_bilinear_bwd(const float* dy, int ldy, int B, int Hi, int Wi, int C, float* dx, int ldx, int Ho, int Wo,
                      int align_corners, int dy_nchw, int accumulate, void* workspace, int64_t workspace_bytes,
                      void* stream) {
  PSEG_REQUIRE(dy && dx && B > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "bilinear_bwd: bad argument");
  ResizeParams p;
  p.h = make_axis(Hi, Ho, align_corners);
  p.w = make_axis(Wi, Wo, align_corners);
  p.B = B;
  p.C = C;
  p.ldx = ldx;
  p.ldy = ldy;
  if (dy_nchw) {
    const int C4 = (C + 3) / 4;
    PSEG_REQUIRE(ldx % 4 == 0 && ldx >= 4 * C4 && al16(dx), "bilinear_bwd: NHWC gradient must be 16-byte aligned with ld >= C rounded up to 4");
    const long long total = (long long)B * C4 * Hi * Wi;
    PSEG_REQUIRE((long long)B * C * Ho * Wo < (1LL << 31), "bilinear_bwd: tensor too large");
    p.chwdiv = FastDiv((uint32_t)((long long)C4 * Hi * Wi));
    p.hwdiv = FastDiv((uint32_t)(Hi * Wi));
    p.wdiv = FastDiv((uint32_t)Wi);
    const long long tmp_elems = (long long)B * C * Ho * Wi;
    if (workspace != nullptr && workspace_bytes >= tmp_elems * 4 && tmp_elems < (1LL << 31) && al16(workspace)) {
      // separable two-pass form (scratch: pseg_bilinear_bwd_workspace_bytes)
      const long long nrows = (long long)B * C * Ho;
      // taps per output column after trimming: <= 2*ceil(Wo/Wi) + 1
      const int taps = 2 * ((Wo + Wi - 1) / Wi) + 1;
      if (Wi <= 256 && Wo <= 2048 && Wo % 4 == 0 && al16(dy) && taps <= kMaxTaps) {
        const int RB = 8192 / Wo;                                  // rows per block (32 KB of LDS)
        int wi_pad = 1;
        while (wi_pad < Wi) wi_pad <<= 1;                          // columns padded to a divisor of 256
        hipLaunchKernelGGL(bilinear_bwd_nchw_w_lds_kernel, dim3((unsigned)((nrows + RB * kChunks - 1) / (RB * kChunks))),
                           dim3(256), 0, (hipStream_t)stream, dy, (float*)workspace, p, (uint32_t)nrows, RB, wi_pad);
      } else {
        const long long wthreads = ((nrows + kRowsPerThread - 1) / kRowsPerThread) * Wi;
        hipLaunchKernelGGL(bilinear_bwd_nchw_w_kernel, dim3(ew_grid(wthreads)), dim3(256), 0, (hipStream_t)stream, dy,
                           (float*)workspace, p, (uint32_t)nrows, (uint32_t)wthreads);
      }
      PSEG_LAUNCH_CHECK();
      hipLaunchKernelGGL(bilinear_bwd_nchw_h_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                         (const float*)workspace, dx, p, accumulate, (uint32_t)total);
    } else {
      hipLaunchKernelGGL(bilinear_bwd_nchw_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, dx, p,
                         accumulate, (uint32_t)total);
    }
  } else {
    return bilinear_bwd_nhwc_impl<float>(dy, ldy, B, Hi, Wi, C, dx, ldx, Ho, Wo, align_corners, accumulate, stream);
  }
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

static int fill_pool(PoolParams& p, int B, int H, int W, int C, int Ho, int Wo, int k, int stride, int pad, int ldx,
                     int ldy) {
  PSEG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && k >= 1 && k <= 15 && stride >= 1 && pad >= 0 && pad < k,
               "maxpool: bad argument");
  PSEG_REQUIRE(Ho == (H + 2 * pad - k) / stride + 1 && Wo == (W + 2 * pad - k) / stride + 1, "maxpool: Ho/Wo mismatch");
  p.B = B; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.k = k; p.stride = stride; p.pad = pad;
  p.ldx = ldx; p.ldy = ldy;
  p.c4div = FastDiv((uint32_t)(C / 4));
  return PSEG_OK;
}

extern "C++" {
template <typename T>
static int maxpool_fwd_impl(const T* x, int ldx, int B, int H, int W, int C, T* y, int ldy, uint8_t* argmax, int Ho,
                            int Wo, int k, int stride, int pad, void* stream, const float* mean = nullptr,
                            const float* scale = nullptr, const float* shift = nullptr, int act = 0) {
  PSEG_REQUIRE(x && y && ld_ok<T>(ldx) && ld_ok<T>(ldy) && al16(x) && al16(y) && ((uintptr_t)argmax & 3) == 0,
               "maxpool_fwd: alignment / null");
  PoolParams p;
  int rc = fill_pool(p, B, H, W, C, Ho, Wo, k, stride, pad, ldx, ldy);
  if (rc) return rc;
  p.pixdiv = FastDiv((uint32_t)(Ho * Wo));
  p.rowdiv = FastDiv((uint32_t)Wo);
  const long long total = (long long)B * Ho * Wo * (C / 4);
  PSEG_REQUIRE(total < (1LL << 31), "maxpool_fwd: tensor too large");
  if (mean != nullptr)
    hipLaunchKernelGGL((maxpool_fwd_kernel<T, true>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, argmax, p,
                       (uint32_t)total, mean, scale, shift, act);
  else
    hipLaunchKernelGGL((maxpool_fwd_kernel<T, false>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, argmax, p,
                       (uint32_t)total, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}
}  // extern "C++"
int pseg_bn_act_maxpool_fwd(const float* x, int ldx, const float* mean, const float* scale, const float* shift, int act, int B,
                            int H, int W, int C, float* y, int ldy, uint8_t* argmax, int Ho, int Wo, int k, int stride, int pad,
                            void* stream) {
  PSEG_REQUIRE(mean && scale && shift && (((uintptr_t)mean | (uintptr_t)scale | (uintptr_t)shift) & 15) == 0,
               "bn_act_maxpool_fwd: coefficient rows must be non-null and 16-byte aligned");
  PSEG_REQUIRE(act == PSEG_ACT_NONE || act == PSEG_ACT_RELU || act == PSEG_ACT_RELU6, "bn_act_maxpool_fwd: unknown activation");
  return maxpool_fwd_impl<float>(x, ldx, B, H, W, C, y, ldy, argmax, Ho, Wo, k, stride, pad, stream, mean, scale, shift, act);
}
int pseg_bn_act_maxpool_fwd_h(const pseg_half_t* x, int ldx, const float* mean, const float* scale, const float* shift, int act,
                              int B, int H, int W, int C, pseg_half_t* y, int ldy, uint8_t* argmax, int Ho, int Wo, int k,
                              int stride, int pad, void* stream) {
  PSEG_REQUIRE(mean && scale && shift && (((uintptr_t)mean | (uintptr_t)scale | (uintptr_t)shift) & 15) == 0,
               "bn_act_maxpool_fwd_h: coefficient rows must be non-null and 16-byte aligned");
  PSEG_REQUIRE(act == PSEG_ACT_NONE || act == PSEG_ACT_RELU || act == PSEG_ACT_RELU6, "bn_act_maxpool_fwd_h: unknown activation");
  return maxpool_fwd_impl<half_t>(HP(x), ldx, B, H, W, C, HPM(y), ldy, argmax, Ho, Wo, k, stride, pad, stream, mean, scale, shift,
                                  act);
}
int pseg_maxpool_fwd(const float* x, int ldx, int B, int H, int W, int C, float* y, int ldy, uint8_t* argmax, int Ho,
                     int Wo, int k, int stride, int pad, void* stream) {
  return maxpool_fwd_impl<float>(x, ldx, B, H, W, C, y, ldy, argmax, Ho, Wo, k, stride, pad, stream);
}
int pseg_maxpool_fwd_h(const pseg_half_t* x, int ldx, int B, int H, int W, int C, pseg_half_t* y, int ldy, uint8_t* argmax,
                       int Ho, int Wo, int k, int stride, int pad, void* stream) {
  return maxpool_fwd_impl<half_t>(HP(x), ldx, B, H, W, C, HPM(y), ldy, argmax, Ho, Wo, k, stride, pad, stream);
}

extern "C++" {
template <typename T>
static int maxpool_bwd_impl(const T* dy, int ldy, const uint8_t* argmax, int B, int H, int W, int C, T* dx, int ldx,
                            int Ho, int Wo, int k, int stride, int pad, int accumulate, void* stream) {
  PSEG_REQUIRE(dy && dx && argmax && ld_ok<T>(ldx) && ld_ok<T>(ldy) && al16(dx) && al16(dy) && ((uintptr_t)argmax & 3) == 0,
               "maxpool_bwd: alignment / null");
  PoolParams p;
  int rc = fill_pool(p, B, H, W, C, Ho, Wo, k, stride, pad, ldx, ldy);
  if (rc) return rc;
  p.pixdiv = FastDiv((uint32_t)(H * W));
  p.rowdiv = FastDiv((uint32_t)W);
  const long long total = (long long)B * H * W * (C / 4);
  PSEG_REQUIRE(total < (1LL << 31), "maxpool_bwd: tensor too large");
  hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, argmax, dx, p,
                     accumulate, (uint32_t)total);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}
}  // extern "C++"
int pseg_maxpool_bwd(const float* dy, int ldy, const uint8_t* argmax, int B, int H, int W, int C, float* dx, int ldx,
                     int Ho, int Wo, int k, int stride, int pad, int accumulate, void* stream) {
  return maxpool_bwd_impl<float>(dy, ldy, argmax, B, H, W, C, dx, ldx, Ho, Wo, k, stride, pad, accumulate, stream);
}
int pseg_maxpool_bwd_h(const pseg_half_t* dy, int ldy, const uint8_t* argmax, int B, int H, int W, int C, pseg_half_t* dx,
                       int ldx, int Ho, int Wo, int k, int stride, int pad, int accumulate, void* stream) {
  return maxpool_bwd_impl<half_t>(HP(dy), ldy, argmax, B, H, W, C, HPM(dx), ldx, Ho, Wo, k, stride, pad, accumulate, stream);
}

int pseg_nchw_to_nhwc(const float* x, float* y, int ldy, int B, int C, int HW, int Cpad, void* stream) {
  PSEG_REQUIRE(x && y && B > 0 && C > 0 && HW > 0 && Cpad >= C && Cpad <= ldy && ldy % 4 == 0 && al16(y),
               "nchw_to_nhwc: bad argument");
  if (C <= 4 && Cpad == 4) {
    const long long total = (long long)B * HW;
    PSEG_REQUIRE(total < (1LL << 31), "nchw_to_nhwc: tensor too large");
    hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, y, ldy, C,
                       (uint32_t)HW, (uint32_t)total, FastDiv((uint32_t)HW));
  } else {
    PSEG_REQUIRE(B <= 65535, "nchw_to_nhwc: batch too large");
    // in[b][R=C][Cc=HW] -> out[b][HW][C..Cpad)
    dim3 grid((unsigned)cdiv(HW, 32), (unsigned)cdiv(Cpad, 32), (unsigned)B);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, (long long)C * HW, HW, y,
                       (long long)HW * ldy, ldy, C, HW, Cpad);
  }
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

int pseg_nchw_to_nhwc_h(const float* x, pseg_half_t* y, int ldy, int B, int C, int HW, void* stream) {
  PSEG_REQUIRE(x && y && B > 0 && C > 0 && C <= 8 && HW > 0 && ldy >= 8 && ldy % 8 == 0 && al16(y),
               "nchw_to_nhwc_h: bad argument (C <= 8 channels into 8-channel fp16 pixels)");
  const long long total = (long long)B * HW;
  PSEG_REQUIRE(total < (1LL << 31), "nchw_to_nhwc_h: tensor too large");
  hipLaunchKernelGGL(nchw_to_nhwc8h_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x,
                     reinterpret_cast<half_t*>(y), ldy, C, (uint32_t)HW, (uint32_t)total, FastDiv((uint32_t)HW));
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

int pseg_nhwc_to_nchw(const float* x, int ldx, float* y, int B, int C, int HW, void* stream) {
  PSEG_REQUIRE(x && y && B > 0 && B <= 65535 && C > 0 && HW > 0 && ldx >= C, "nhwc_to_nchw: bad argument");
  // in[b][R=HW][Cc=C] -> out[b][C][HW]
  dim3 grid((unsigned)cdiv(C, 32), (unsigned)cdiv(HW, 32), (unsigned)B);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, (long long)HW * ldx, ldx, y,
                     (long long)C * HW, HW, HW, C, HW);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

}  // extern "C"
