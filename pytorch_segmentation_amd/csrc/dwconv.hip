// Depthwise k x k convolution (NHWC fp32) for the MobileNetV2 encoder behind the reference's UNet
// (models/unet.py:16-17,28).  Direct kernels: 2*k*k FLOP per 4 bytes moved -> HBM-bound, one lane per
// (pixel, 4 channels), 16-byte accesses.  Filter layout [k][k][C].  wgrad reduces through fixed-order
// per-row-group partials (bit-reproducible).
#include "common.h"
#include "half_io.h"

namespace pseg {

// activations: fp32 or fp16 (template parameter T); the filter and its gradient are always fp32
template <typename T>
__device__ __forceinline__ f32x4 ld4(const T* p) { return ldv4(p); }
template <typename T>
__device__ __forceinline__ void st4(T* p, f32x4 v) { stv4(p, v); }

constexpr int kDwMaxTaps = 9;

struct DwParams {
  int B, H, W, C, Ho, Wo, k, stride, pad, ldx, ldy;
  FastDiv c4div, pixdiv, rowdiv;
};

template <typename T>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                     T* __restrict__ y, DwParams p, uint32_t total) {
  PSEG_HELPER_PRIO();
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t pix = p.c4div.div(i);
    const uint32_t c = (i - pix * p.c4div.d) * 4;
    const uint32_t b = p.pixdiv.div(pix);
    const uint32_t rem = pix - b * p.pixdiv.d;
    const int ho = (int)p.rowdiv.div(rem);
    const int wo = (int)(rem - (uint32_t)ho * p.rowdiv.d);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < p.k; ++r) {
      const int hi = ho * p.stride - p.pad + r;
      if ((unsigned)hi >= (unsigned)p.H) continue;
      for (int s = 0; s < p.k; ++s) {
        const int wi = wo * p.stride - p.pad + s;
        if ((unsigned)wi >= (unsigned)p.W) continue;
        acc += ld4(x + ((long long)(b * p.H + hi) * p.W + wi) * p.ldx + c) * ld4(w + (r * p.k + s) * p.C + c);
      }
    }
    st4(y + (long long)pix * p.ldy + c, acc);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const T* __restrict__ dy, const float* __restrict__ w,
                                                       T* __restrict__ dx, DwParams p, uint32_t total) {
  PSEG_HELPER_PRIO();
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t pix = p.c4div.div(i);  // input pixel
    const uint32_t c = (i - pix * p.c4div.d) * 4;
    const uint32_t b = p.pixdiv.div(pix);
    const uint32_t rem = pix - b * p.pixdiv.d;
    const int hi = (int)p.rowdiv.div(rem);
    const int wi = (int)(rem - (uint32_t)hi * p.rowdiv.d);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < p.k; ++r) {
      const int hn = hi + p.pad - r;
      if (hn < 0 || hn % p.stride != 0) continue;
      const int ho = hn / p.stride;
      if (ho >= p.Ho) continue;
      for (int s = 0; s < p.k; ++s) {
        const int wn = wi + p.pad - s;
        if (wn < 0 || wn % p.stride != 0) continue;
        const int wo = wn / p.stride;
        if (wo >= p.Wo) continue;
        acc += ld4(dy + ((long long)(b * p.Ho + ho) * p.Wo + wo) * p.ldy + c) * ld4(w + (r * p.k + s) * p.C + c);
      }
    }
    st4(dx + (long long)pix * p.ldx + c, acc);
  }
}

// partial[rowgroup][tap][C]: blockDim = (TX chunk-columns, TY pixel lanes), grid = (row groups, column groups).
// The operands are tiny next to the chip (16-50 MB per layer of the UNet encoder) and every lane owns nine independent
// accumulators, so the kernel lives on memory-level parallelism: enough row groups to put several blocks on every CU
// (dw_rows_per_block), buffer loads whose out-of-image taps return zero (no branch between the loads of a pixel: ten
// 16-byte loads per pixel issue back to back), and two pixels in flight per lane.  (The round-1 form -- 512 pixels per
// block, a branch per tap -- ran one block per CU at 285 us per layer: 4.9 of UNet's 11 ms per step.)
template <typename T>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                       float* __restrict__ part, DwParams p, long long P, int rows_per_block,
                                                       uint32_t x_bytes, uint32_t dy_bytes) {
  constexpr int ES = (int)sizeof(T);
  PSEG_HELPER_PRIO();
  __shared__ f32x4 sh[256];
  const int TX = blockDim.x, TY = blockDim.y;
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int c4 = blockIdx.y * TX + tx;
  const bool cok = c4 * 4 < p.C;
  const int c = c4 * 4;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > P) r1 = P;
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(x, x_bytes);
  const __amdgpu_buffer_rsrc_t dr = make_rsrc(dy, dy_bytes);
  const int taps = p.k * p.k;
  f32x4 acc[kDwMaxTaps];
#pragma unroll
  for (int t = 0; t < kDwMaxTaps; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto offsets = [&](long long pix, uint32_t& goff, uint32_t (&xoff)[kDwMaxTaps]) {
    const bool live = cok && pix < r1;
    const uint32_t b = p.pixdiv.div((uint32_t)pix);
    const uint32_t rem = (uint32_t)pix - b * p.pixdiv.d;
    const int ho = (int)p.rowdiv.div(rem);
    const int wo = (int)(rem - (uint32_t)ho * p.rowdiv.d);
    goff = live ? (uint32_t)((pix * p.ldy + c) * ES) : kOOB;
#pragma unroll
    for (int t = 0; t < kDwMaxTaps; ++t) {
      const int r = t / p.k, s = t - r * p.k;
      const int hi = ho * p.stride - p.pad + r, wi = wo * p.stride - p.pad + s;
      const bool ok = live && t < taps && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      xoff[t] = ok ? (uint32_t)((((long long)(b * p.H + hi) * p.W + wi) * p.ldx + c) * ES) : kOOB;
    }
  };
  for (long long pix = r0 + ty; pix < r1; pix += 2 * TY) {
    uint32_t g0o, g1o, x0o[kDwMaxTaps], x1o[kDwMaxTaps];
    offsets(pix, g0o, x0o);
    offsets(pix + TY, g1o, x1o);
    const f32x4 g0 = buf_ldv4<T>(dr, (int)g0o, 0), g1 = buf_ldv4<T>(dr, (int)g1o, 0);
    f32x4 v0[kDwMaxTaps], v1[kDwMaxTaps];
#pragma unroll
    for (int t = 0; t < kDwMaxTaps; ++t) {
      v0[t] = buf_ldv4<T>(xr, (int)x0o[t], 0);
      v1[t] = buf_ldv4<T>(xr, (int)x1o[t], 0);
    }
#pragma unroll
    for (int t = 0; t < kDwMaxTaps; ++t) acc[t] += g0 * v0[t] + g1 * v1[t];
  }
#pragma unroll
  for (int t = 0; t < kDwMaxTaps; ++t) {
    if (t < taps) {
      __syncthreads();
      sh[ty * TX + tx] = acc[t];
      __syncthreads();
      if (ty == 0 && cok) {
        f32x4 s = sh[tx];
        for (int j = 1; j < TY; ++j) s += sh[j * TX + tx];
        st4(part + ((long long)blockIdx.x * taps + t) * p.C + c, s);
      }
    }
  }
}

// lanes across the channel chunks of a block, and output pixels per block: at most 1024 partial rows, at least two
// pixels per pixel lane
static int dw_tx(int C) {
  const int c4 = C / 4;
  return c4 > 32 ? 64 : (c4 > 16 ? 32 : (c4 > 8 ? 16 : 8));
}
static int dw_rows_per_block(long long P, int C) {
  const int ty = 256 / dw_tx(C);
  long long r = (P + 1023) / 1024;
  r = (r + 2 * ty - 1) / (2 * ty) * (2 * ty);
  if (r < 2 * ty) r = 2 * ty;
  return (int)r;
}

static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
static int ew_grid(long long total) {
  long long b = (total + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

static int fill_dw(DwParams& p, int B, int H, int W, int C, int Ho, int Wo, int k, int stride, int pad, int ldx, int ldy,
                   int ldq = 4) {
  PSEG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && k >= 1 && k * k <= kDwMaxTaps && stride >= 1 && pad >= 0,
               "dwconv: bad argument (C %% 4 == 0, k <= 3)");
  PSEG_REQUIRE(Ho == (H + 2 * pad - k) / stride + 1 && Wo == (W + 2 * pad - k) / stride + 1, "dwconv: Ho/Wo mismatch");
  PSEG_REQUIRE(ldx % ldq == 0 && ldy % ldq == 0, "dwconv: ld must be a multiple of %d", ldq);
  p.B = B; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.k = k; p.stride = stride; p.pad = pad;
  p.ldx = ldx; p.ldy = ldy;
  p.c4div = FastDiv((uint32_t)(C / 4));
  return PSEG_OK;
}

template <typename T>
static int dw_fwd_impl(const T* x, int ldx, const float* w, T* y, int ldy, int B, int H, int W, int C, int Ho, int Wo,
                       int k, int stride, int pad, void* stream) {
  PSEG_REQUIRE(x && w && y && al16(x) && al16(w) && al16(y), "dwconv_fwd: null / alignment");
  DwParams p;
  int rc = fill_dw(p, B, H, W, C, Ho, Wo, k, stride, pad, ldx, ldy, sizeof(T) == 2 ? 8 : 4);
  if (rc) return rc;
  p.pixdiv = FastDiv((uint32_t)(Ho * Wo));
  p.rowdiv = FastDiv((uint32_t)Wo);
  const long long total = (long long)B * Ho * Wo * (C / 4);
  PSEG_REQUIRE(total < (1LL << 31), "dwconv_fwd: tensor too large");
  hipLaunchKernelGGL(dw_fwd_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, w, y, p, (uint32_t)total);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

template <typename T>
static int dw_dgrad_impl(const T* dy, int ldy, const float* w, T* dx, int ldx, int B, int H, int W, int C, int Ho, int Wo,
                         int k, int stride, int pad, void* stream) {
  PSEG_REQUIRE(dy && w && dx && al16(dy) && al16(w) && al16(dx), "dwconv_dgrad: null / alignment");
  DwParams p;
  int rc = fill_dw(p, B, H, W, C, Ho, Wo, k, stride, pad, ldx, ldy, sizeof(T) == 2 ? 8 : 4);
  if (rc) return rc;
  p.pixdiv = FastDiv((uint32_t)(H * W));
  p.rowdiv = FastDiv((uint32_t)W);
  const long long total = (long long)B * H * W * (C / 4);
  PSEG_REQUIRE(total < (1LL << 31), "dwconv_dgrad: tensor too large");
  hipLaunchKernelGGL(dw_dgrad_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, w, dx, p,
                     (uint32_t)total);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

static int64_t dw_wgrad_ws(int B, int Ho, int Wo, int C, int k) {
  const long long P = (long long)B * Ho * Wo;
  if (P <= 0 || C <= 0) return 0;
  return (int64_t)cdiv(P, dw_rows_per_block(P, C)) * k * k * C * 4;
}

template <typename T>
static int dw_wgrad_impl(const T* x, int ldx, const T* dy, int ldy, float* dw, int B, int H, int W, int C, int Ho, int Wo,
                         int k, int stride, int pad, int accumulate, void* workspace, int64_t workspace_bytes,
                         void* stream) {
  PSEG_REQUIRE(x && dy && dw && al16(x) && al16(dy) && al16(workspace), "dwconv_wgrad: null / alignment");
  DwParams p;
  int rc = fill_dw(p, B, H, W, C, Ho, Wo, k, stride, pad, ldx, ldy, sizeof(T) == 2 ? 8 : 4);
  if (rc) return rc;
  p.pixdiv = FastDiv((uint32_t)(Ho * Wo));
  p.rowdiv = FastDiv((uint32_t)Wo);
  const long long P = (long long)B * Ho * Wo;
  PSEG_REQUIRE(P < (1LL << 31), "dwconv_wgrad: tensor too large");
  const int64_t need = dw_wgrad_ws(B, Ho, Wo, C, k);
  if (!workspace || workspace_bytes < need) {
    set_error("dwconv_wgrad: needs %lld workspace bytes, got %lld", (long long)need, (long long)workspace_bytes);
    return PSEG_ERR_WORKSPACE;
  }
  const int c4 = C / 4;
  const int tx = dw_tx(C);
  const int rpb = dw_rows_per_block(P, C);
  const int rows = cdiv(P, rpb);
  const long long xb = (((long long)B * H * W - 1) * ldx + C) * (long long)sizeof(T), db = ((P - 1) * ldy + C) * (long long)sizeof(T);
  PSEG_REQUIRE(xb < (1LL << 31) && db < (1LL << 31), "dwconv_wgrad: tensor exceeds 2 GiB");
  hipLaunchKernelGGL(dw_wgrad_kernel<T>, dim3(rows, cdiv(c4, tx)), dim3(tx, 256 / tx), 0, (hipStream_t)stream, x, dy,
                     (float*)workspace, p, P, rpb, (uint32_t)xb, (uint32_t)db);
  PSEG_LAUNCH_CHECK();
  return launch_col_reduce((const float*)workspace, rows, k * k * C, dw, accumulate, (hipStream_t)stream);
}

}  // namespace pseg

using namespace pseg;

#define HP(p) reinterpret_cast<const half_t*>(p)
#define HPM(p) reinterpret_cast<half_t*>(p)

extern "C" {

int pseg_dwconv_fwd(const float* x, int ldx, const float* w, float* y, int ldy, int B, int H, int W, int C, int Ho, int Wo,
                    int k, int stride, int pad, void* stream) {
  return dw_fwd_impl<float>(x, ldx, w, y, ldy, B, H, W, C, Ho, Wo, k, stride, pad, stream);
}
int pseg_dwconv_fwd_h(const pseg_half_t* x, int ldx, const float* w, pseg_half_t* y, int ldy, int B, int H, int W, int C,
                      int Ho, int Wo, int k, int stride, int pad, void* stream) {
  return dw_fwd_impl<half_t>(HP(x), ldx, w, HPM(y), ldy, B, H, W, C, Ho, Wo, k, stride, pad, stream);
}

int pseg_dwconv_dgrad(const float* dy, int ldy, const float* w, float* dx, int ldx, int B, int H, int W, int C, int Ho,
                      int Wo, int k, int stride, int pad, void* stream) {
  return dw_dgrad_impl<float>(dy, ldy, w, dx, ldx, B, H, W, C, Ho, Wo, k, stride, pad, stream);
}
int pseg_dwconv_dgrad_h(const pseg_half_t* dy, int ldy, const float* w, pseg_half_t* dx, int ldx, int B, int H, int W, int C,
                        int Ho, int Wo, int k, int stride, int pad, void* stream) {
  return dw_dgrad_impl<half_t>(HP(dy), ldy, w, HPM(dx), ldx, B, H, W, C, Ho, Wo, k, stride, pad, stream);
}

int64_t pseg_dwconv_wgrad_workspace_bytes(int B, int Ho, int Wo, int C, int k) { return dw_wgrad_ws(B, Ho, Wo, C, k); }

int pseg_dwconv_wgrad(const float* x, int ldx, const float* dy, int ldy, float* dw, int B, int H, int W, int C, int Ho,
                      int Wo, int k, int stride, int pad, int accumulate, void* workspace, int64_t workspace_bytes,
                      void* stream) {
  return dw_wgrad_impl<float>(x, ldx, dy, ldy, dw, B, H, W, C, Ho, Wo, k, stride, pad, accumulate, workspace,
                              workspace_bytes, stream);
}
int pseg_dwconv_wgrad_h(const pseg_half_t* x, int ldx, const pseg_half_t* dy, int ldy, float* dw, int B, int H, int W, int C,
                        int Ho, int Wo, int k, int stride, int pad, int accumulate, void* workspace, int64_t workspace_bytes,
                        void* stream) {
  return dw_wgrad_impl<half_t>(HP(x), ldx, HP(dy), ldy, dw, B, H, W, C, Ho, Wo, k, stride, pad, accumulate, workspace,
                               workspace_bytes, stream);
}

}  // extern "C"
