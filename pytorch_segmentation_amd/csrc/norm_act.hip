// BatchNorm (training / eval) + activation + residual for NHWC fp32 on gfx950.
// Replaces nn.BatchNorm2d + nn.ReLU inside ConvNormAct (reference models/aspp.py:12,27-30,
// models/deeplabv3plus.py:20, models/unet.py:19-21) and the bottleneck tails of the backbones.
//
// All kernels are HBM-bound: 16-byte accesses per lane, channel = fastest dimension so a wave reads
// whole 1-KiB row segments; reductions go through fixed-order partials (bit-reproducible, no float atomics).
// A tensor is [M pixels][C channels] with pixel stride ld; C % 4 == 0, ld % 4 == 0.
#include "common.h"
#include "half_io.h"

#include <stdlib.h>

namespace pseg {

constexpr int kMaxStatRows = 512;  // upper bound of pixel rows per statistics group

// rows per group: as large as possible (fewer partials) while the (row groups x column groups) grid still fills
// the chip a few times over
static int stat_group(long long M, int C) {
  const int c4 = C / 4;
  const int tx = c4 >= 64 ? 64 : (c4 > 16 ? 32 : 16);
  const long long colblocks = (c4 + tx - 1) / tx;
  int r = kMaxStatRows;
  while (r > 32 && ((M + r - 1) / r) * colblocks < 1024) r >>= 1;
  return r;
}

// 4 consecutive channels of an fp32 or fp16 tensor <-> f32x4 (half_io.h); every kernel below that touches activations is
// a template on the storage type T (float, or half_t under the half-precision `-mp` policy): arithmetic, statistics and
// per-channel vectors stay fp32.
template <typename T>
__device__ __forceinline__ f32x4 ld4(const T* p) { return ldv4(p); }
template <typename T>
__device__ __forceinline__ void st4(T* p, f32x4 v) { stv4(p, v); }

template <int V>
__device__ __forceinline__ fvec<V> act_mask_v(fvec<V> z, int act) {
  fvec<V> m;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    bool on = true;
    if (act == PSEG_ACT_RELU) on = z[i] > 0.f;
    else if (act == PSEG_ACT_RELU6) on = (z[i] > 0.f) && (z[i] < 6.f);
    m[i] = on ? 1.f : 0.f;
  }
  return m;
}
__device__ __forceinline__ f32x4 act_mask(f32x4 z, int act) { return act_mask_v<4>(z, act); }

// bits c % 32 .. c % 32 + V - 1 of a mask word as V floats (0 / 1)
template <int V>
__device__ __forceinline__ fvec<V> bits_to_mask(uint32_t w, int c) {
  const uint32_t b = w >> (c & 31);
  fvec<V> m;
#pragma unroll
  for (int i = 0; i < V; ++i) m[i] = (float)((b >> i) & 1u);
  return m;
}

// Activation bitmask of a residual layer (bn_act_fwd with mask_out): row r holds C/32 words, bit (c % 32) of word c / 32
// is set when act'(z[r][c]) = 1.  The backward passes of such a layer read 1 bit instead of 32 per element for the mask
// (z = act(BN(y) + residual) cannot be recomputed from y alone): 4.3 GB less HBM traffic per DeepLabV3+ step.
__device__ __forceinline__ f32x4 mask_from_bits(const uint32_t* __restrict__ mask, long long r, int c, int words) {
  const uint32_t w = mask[r * words + (c >> 5)];
  const uint32_t nib = (w >> (c & 31)) & 0xFu;
  return f32x4{(float)(nib & 1u), (float)((nib >> 1) & 1u), (float)((nib >> 2) & 1u), (float)((nib >> 3) & 1u)};
}

// ---- column statistics per row group: blockDim = (TX chunk-columns, TY row lanes); grid = (row groups, column groups)
// SHIFTED = true: writes [K, sum(v-K), sum((v-K)^2)] with K = the group's first row (BatchNorm statistics);
// SHIFTED = false: plain column sums (bias gradients).
template <bool SHIFTED, typename T>
__global__ __launch_bounds__(256) void col_stats_kernel(const T* __restrict__ y, int ld, long long M, int C, int R,
                                                        float* __restrict__ out, long long plane) {
  PSEG_HELPER_PRIO();
  __shared__ f32x4 sh[2][256];
  const int TX = blockDim.x, TY = blockDim.y;
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int c4 = blockIdx.y * TX + tx;
  const bool cok = c4 * 4 < C;
  const long long r0 = (long long)blockIdx.x * R;
  long long r1 = r0 + R;
  if (r1 > M) r1 = M;
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f}, k = {0.f, 0.f, 0.f, 0.f};
  if (cok) {
    const T* yp = y + c4 * 4;
    if (SHIFTED) k = ld4(yp + r0 * ld);
    long long r = r0 + ty;
    for (; r + 3 * TY < r1; r += 4 * TY) {  // four independent 16-byte loads in flight per lane
      const f32x4 v0 = ld4(yp + r * ld) - k, v1 = ld4(yp + (r + TY) * ld) - k, v2 = ld4(yp + (r + 2 * TY) * ld) - k,
                  v3 = ld4(yp + (r + 3 * TY) * ld) - k;
      s += (v0 + v1) + (v2 + v3);
      if (SHIFTED) q += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
    }
    for (; r < r1; r += TY) {
      const f32x4 v = ld4(yp + r * ld) - k;
      s += v;
      if (SHIFTED) q += v * v;
    }
  }
  sh[0][ty * TX + tx] = s;
  sh[1][ty * TX + tx] = q;
  __syncthreads();
  if (ty == 0 && cok) {
    f32x4 ts = sh[0][tx], tq = sh[1][tx];
    for (int j = 1; j < TY; ++j) {
      ts += sh[0][j * TX + tx];
      tq += sh[1][j * TX + tx];
    }
    const long long o = (long long)blockIdx.x * C + c4 * 4;
    if (SHIFTED) {
      st4(out + o, k);
      st4(out + plane + o, ts);
      st4(out + 2 * plane + o, tq);
    } else {
      st4(out + o, ts);
    }
  }
}

// ---- backward partials: dyh = dz * act'(z); sums of dyh and dyh * xhat
// MODE: where act'(z) comes from -- 0 no activation, 1 bitmask, 2 the stored z, 3 recomputed from y exactly as the forward
// pass did (z == nullptr, no residual: saves re-reading the whole output tensor).
// In the training step this pass runs BESIDE the weight gradient of the layer above (second stream), whose two resident
// blocks leave a CU 96 VGPRs per SIMD and 16 KB of LDS: one wave of this kernel per SIMD.  Its rate there is bytes in
// flight per wave, so: NR rows in flight per lane, and addressing that costs no registers -- buffer loads from a
// block-uniform resource (rows [r0, r1) of the tensor), ONE per-lane byte offset per tensor, the row advance in the scalar
// offset.  Rows past r1 are out of the resource's range and read as zeros, which contribute nothing: no tail loop.
// The add order is the same in every mode (mask-fed == z-fed, bit for bit).
template <int MODE, int NR, typename T, int V>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dz, int lddz,
                                                            const T* __restrict__ z, int ldz,
                                                            const T* __restrict__ y, int ldy,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int act, long long M, int C,
                                                            int R, float* __restrict__ pdb, float* __restrict__ pdg,
                                                            const uint32_t* __restrict__ mask) {
  PSEG_HELPER_PRIO();
  typedef fvec<V> vf;
  __shared__ vf sh[2][256];
  const int TX = blockDim.x, TY = blockDim.y;
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int cv = blockIdx.y * TX + tx;
  const bool cok = cv * V < C;
  const long long r0 = (long long)blockIdx.x * R;
  const int nrows = (int)((r0 + R > M ? M : r0 + R) - r0);
  const int words = C >> 5;
  constexpr int ES = (int)sizeof(T);
  const __amdgpu_buffer_rsrc_t dzr = make_rsrc(dz + r0 * lddz, (uint32_t)(((long long)(nrows - 1) * lddz + C) * ES));
  const __amdgpu_buffer_rsrc_t yr = make_rsrc(y + r0 * ldy, (uint32_t)(((long long)(nrows - 1) * ldy + C) * ES));
  const __amdgpu_buffer_rsrc_t zr =
      make_rsrc(MODE == 2 ? z + r0 * ldz : y, MODE == 2 ? (uint32_t)(((long long)(nrows - 1) * ldz + C) * ES) : 0u);
  const __amdgpu_buffer_rsrc_t mr =
      make_rsrc(MODE == 1 ? mask + r0 * words : (const uint32_t*)y, MODE == 1 ? (uint32_t)(nrows * words * 4) : 0u);
  vf s = {}, q = {};
  if (cok) {
    const int c = cv * V;
    const vf mu = ldvec<V>(mean + c), is = ldvec<V>(invstd + c);
    vf sc = {}, sh4 = {};
    if (MODE == 3) {
      sc = ldvec<V>(scale + c);
      sh4 = ldvec<V>(shift + c);
    }
    const int vdz = (ty * lddz + c) * ES, vy = (ty * ldy + c) * ES, vz = (ty * ldz + c) * ES, vm = (ty * words + (c >> 5)) * 4;
    for (int base = 0; base < nrows; base += NR * TY) {   // block-uniform trip count
      vf g[NR], yv[NR];
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const int row = base + k * TY;                    // (+ ty: in the lane offset)
        g[k] = buf_ldvec<V, T>(dzr, vdz, row * lddz * ES);
        yv[k] = buf_ldvec<V, T>(yr, vy, row * ldy * ES);
      }
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        const int row = base + k * TY;
        if (MODE == 1) {
          const uint32_t w = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(mr, vm, row * words * 4, 0);
          g[k] *= bits_to_mask<V>(w, c);
        } else if (MODE == 2) {
          g[k] *= act_mask_v<V>(buf_ldvec<V, T>(zr, vz, row * ldz * ES), act);
        } else if (MODE == 3) {
          g[k] *= act_mask_v<V>((yv[k] - mu) * sc + sh4, act);
        }
      }
      vf ts = g[0], tq = g[0] * ((yv[0] - mu) * is);
#pragma unroll
      for (int k = 1; k < NR; ++k) {
        ts += g[k];
        tq += g[k] * ((yv[k] - mu) * is);
      }
      s += ts;
      q += tq;
    }
  }
  sh[0][ty * TX + tx] = s;
  sh[1][ty * TX + tx] = q;
  __syncthreads();
  if (ty == 0 && cok) {
    vf ts = sh[0][tx], tq = sh[1][tx];
    for (int j = 1; j < TY; ++j) {
      ts += sh[0][j * TX + tx];
      tq += sh[1][j * TX + tx];
    }
    stvec<V>(pdb + (long long)blockIdx.x * C + cv * V, ts);
    stvec<V>(pdg + (long long)blockIdx.x * C + cv * V, tq);
  }
}

// ---- reduce [rows][C] partials over rows in double.  Block = kFinCh channels x LANES row lanes.  These launches sit on
// the critical path of every BatchNorm layer and are pure latency: a lane's chain of dependent load batches, then the
// cross-lane sum.  So: as many lanes as a block holds (128 -> 1024 threads) once there are enough rows, four rows in
// flight per lane, and the cross-lane sum in two levels (16 + LANES/16 terms instead of LANES).  The order of the
// additions is fixed by (rows, LANES): bit-reproducible.
constexpr int kFinCh = 8, kFinLanes = 32, kFinLanesWide = 128;
constexpr int kFinWideRows = 128;     // use the wide block from this many partial rows

// sum of one value per lane, lanes in order (two levels); valid in lane 0.  sh: [LANES][kFinCh] doubles per quantity.
template <int LANES, int NQ>
__device__ __forceinline__ void lane_sum(double (*sh)[LANES][kFinCh], int ty, int lx, double (&v)[NQ]) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) sh[q][ty][lx] = v[q];
  __syncthreads();
  constexpr int SEG = LANES >= 64 ? 16 : LANES;      // lanes per first-level sum
  if (ty < LANES / SEG) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      double t = sh[q][ty * SEG][lx];
      for (int j = 1; j < SEG; ++j) t += sh[q][ty * SEG + j][lx];
      v[q] = t;
    }
  }
  if (LANES / SEG > 1) {
    __syncthreads();
    if (ty < LANES / SEG) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) sh[q][ty][lx] = v[q];
    }
    __syncthreads();
    if (ty == 0) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        double t = sh[q][0][lx];
        for (int j = 1; j < LANES / SEG; ++j) t += sh[q][j][lx];
        v[q] = t;
      }
    }
  }
}

template <int LANES>
__device__ __forceinline__ void reduce_pair(const float* __restrict__ pa, const float* __restrict__ pb, int rows, int C,
                                            int c, int ty, double (*sh)[LANES][kFinCh], double& a, double& b) {
  const int lx = threadIdx.x % kFinCh;
  double v[2] = {0.0, 0.0};
  if (c < C) {
    // four rows in flight per lane, added in the original order
    int g = ty;
    for (; g + 3 * LANES < rows; g += 4 * LANES) {
      const float a0 = pa[(long long)g * C + c], a1 = pa[(long long)(g + LANES) * C + c],
                  a2 = pa[(long long)(g + 2 * LANES) * C + c], a3 = pa[(long long)(g + 3 * LANES) * C + c];
      float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
      if (pb) {
        b0 = pb[(long long)g * C + c];
        b1 = pb[(long long)(g + LANES) * C + c];
        b2 = pb[(long long)(g + 2 * LANES) * C + c];
        b3 = pb[(long long)(g + 3 * LANES) * C + c];
      }
      v[0] += (double)a0;
      v[0] += (double)a1;
      v[0] += (double)a2;
      v[0] += (double)a3;
      if (pb) {
        v[1] += (double)b0;
        v[1] += (double)b1;
        v[1] += (double)b2;
        v[1] += (double)b3;
      }
    }
    for (; g < rows; g += LANES) {
      v[0] += (double)pa[(long long)g * C + c];
      if (pb) v[1] += (double)pb[(long long)g * C + c];
    }
  }
  lane_sum<LANES, 2>(sh, ty, lx, v);
  a = v[0];
  b = v[1];
}

// Stage A of a two-stage finalize (many groups, few channels: e.g. 16384 groups x 64 channels for the stem):
// merge `per` consecutive groups into one, re-centred on the first group's pivot, exact in double.
__global__ __launch_bounds__(256) void stat_merge_kernel(const float* __restrict__ stat, int rows, int group,
                                                         long long count, int C, int per, float* __restrict__ out,
                                                         int out_rows) {
  PSEG_HELPER_PRIO();
  __shared__ double sh[2][kFinLanes][kFinCh];
  const int lx = threadIdx.x % kFinCh;
  const int c = blockIdx.x * kFinCh + lx;
  const int ty = threadIdx.x / kFinCh;
  const int g0 = blockIdx.y * per;
  int g1 = g0 + per;
  if (g1 > rows) g1 = rows;
  const long long plane = (long long)rows * C, oplane = (long long)out_rows * C;
  double s1 = 0.0, s2 = 0.0;
  float k0 = 0.f;
  if (c < C) {
    k0 = stat[(long long)g0 * C + c];
    for (int g = g0 + ty; g < g1; g += kFinLanes) {
      long long n = count - (long long)g * group;
      if (n > group) n = group;
      if (n <= 0) continue;
      const double d = (double)stat[(long long)g * C + c] - (double)k0;
      const double a = stat[plane + (long long)g * C + c], b = stat[2 * plane + (long long)g * C + c];
      s1 += a + (double)n * d;
      s2 += b + 2.0 * d * a + (double)n * d * d;
    }
  }
  sh[0][ty][lx] = s1;
  sh[1][ty][lx] = s2;
  __syncthreads();
  if (ty == 0 && c < C) {
    for (int j = 1; j < kFinLanes; ++j) {
      s1 += sh[0][j][lx];
      s2 += sh[1][j][lx];
    }
    const long long o = (long long)blockIdx.y * C + c;
    out[o] = k0;
    out[oplane + o] = (float)s1;
    out[2 * oplane + o] = (float)s2;
  }
}

// stat = [3][rows][C]: pivot K, S1 = sum(v-K), S2 = sum((v-K)^2) of row group g = rows [g*group, min(M, (g+1)*group)).
// ONE pass: every group is re-centred on the FIRST group's pivot K0 (d = K_g - K0, exact in double:
// S1' = S1 + n d, S2' = S2 + 2 d S1 + n d^2) and summed; mean = K0 + S1'/N, M2 = S2' - S1'^2/N.  K0 is a sample of the
// channel itself, so |K0 - mean| is of the order of the standard deviation and the last subtraction loses nothing in
// double (test_batchnorm_statistics_are_cancellation_safe: mean 1e3, std 1e-2).  (Rounds 1-2 walked the groups twice --
// mean first, then M2 about it: twice the latency chain for the same digits.)
template <int LANES>
__global__ __launch_bounds__(kFinCh * LANES) void bn_finalize_kernel(
    const float* __restrict__ stat, int rows, int group, long long count, int C, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
    float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift) {
  PSEG_HELPER_PRIO();
  __shared__ double sh[2][LANES][kFinCh];
  const int lx = threadIdx.x % kFinCh;
  const int c = blockIdx.x * kFinCh + lx;
  const int ty = threadIdx.x / kFinCh;
  const long long plane = (long long)rows * C;
  auto rows_of = [&](int g) -> long long {
    long long n = count - (long long)g * group;
    return n > group ? group : n;
  };
  double v[2] = {0.0, 0.0};
  double k0 = 0.0;
  if (c < C) {
    k0 = (double)stat[c];
    auto add = [&](int g, float kf, float s1f, float s2f) {
      const long long n = rows_of(g);
      if (n <= 0) return;
      const double d = (double)kf - k0, s1 = s1f;
      v[0] += s1 + (double)n * d;
      v[1] += (double)s2f + 2.0 * d * s1 + (double)n * d * d;
    };
    int g = ty;
    for (; g + 3 * LANES < rows; g += 4 * LANES) {   // four row groups in flight per lane, added in the original order
      float kk[4], ss1[4], ss2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long o = (long long)(g + u * LANES) * C + c;
        kk[u] = stat[o];
        ss1[u] = stat[plane + o];
        ss2[u] = stat[2 * plane + o];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) add(g + u * LANES, kk[u], ss1[u], ss2[u]);
    }
    for (; g < rows; g += LANES) {
      const long long o = (long long)g * C + c;
      add(g, stat[o], stat[plane + o], stat[2 * plane + o]);
    }
  }
  lane_sum<LANES, 2>(sh, ty, lx, v);
  if (ty == 0 && c < C) {
    const double n = (double)count;
    const double mu = k0 + v[0] / n;
    double M2 = v[1] - v[0] * v[0] / n;
    if (M2 < 0.0) M2 = 0.0;
    const double var = M2 / n;  // biased (normalisation) variance
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float fm = (float)mu;
    mean[c] = fm;
    invstd[c] = is;
    const float sc = g * is;
    scale[c] = sc;
    shift[c] = b;  // applied as (y - mean) * scale + beta: subtracting first keeps |mean| >> std channels exact
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * fm;
    if (rvar) {
      const double unbiased = count > 1 ? M2 / (n - 1.0) : var;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
    }
  }
}

template <int LANES>
__global__ __launch_bounds__(kFinCh * LANES) void bn_bwd_finalize_kernel(
    const float* __restrict__ pdb, const float* __restrict__ pdg, int rows, long long count, int C,
    float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate, int frozen, float* __restrict__ c1,
    float* __restrict__ c2) {
  PSEG_HELPER_PRIO();
  __shared__ double sh[2][LANES][kFinCh];
  const int c = blockIdx.x * kFinCh + (threadIdx.x % kFinCh);
  const int ty = threadIdx.x / kFinCh;
  double db, dg;
  reduce_pair<LANES>(pdb, pdg, rows, C, c, ty, sh, db, dg);
  if (ty == 0 && c < C) {
    const float fdb = (float)db, fdg = (float)dg;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + fdb : fdb;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + fdg : fdg;
    c1[c] = frozen ? 0.f : (float)(db / (double)count);
    c2[c] = frozen ? 0.f : (float)(dg / (double)count);
  }
}

template <int LANES>
__global__ __launch_bounds__(kFinCh * LANES) void col_reduce_kernel(const float* __restrict__ part, int rows, int C,
                                                                    float* __restrict__ out, int accumulate) {
  PSEG_HELPER_PRIO();
  __shared__ double sh[2][LANES][kFinCh];
  const int c = blockIdx.x * kFinCh + (threadIdx.x % kFinCh);
  const int ty = threadIdx.x / kFinCh;
  double s, unused;
  reduce_pair<LANES>(part, nullptr, rows, C, c, ty, sh, s, unused);
  if (ty == 0 && c < C) out[c] = accumulate ? out[c] + (float)s : (float)s;
}

__global__ void bn_eval_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rmean, const float* __restrict__ rvar, float eps, int C,
                                      float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale,
                                      float* __restrict__ shift) {
  PSEG_HELPER_PRIO();
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float is = 1.f / sqrtf(rvar[c] + eps);
    mean[c] = rmean[c];
    invstd[c] = is;
    scale[c] = (gamma ? gamma[c] : 1.f) * is;
    shift[c] = beta ? beta[c] : 0.f;
  }
}

// ---- elementwise passes over [M][C/4] float4 elements
template <typename T, int V>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ y, int ldy,
                                                         const float* __restrict__ mean, const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         const T* __restrict__ res, int ldr, int act,
                                                         T* __restrict__ z, int ldz, uint32_t total, FastDiv cvdiv,
                                                         unsigned* __restrict__ amax, uint32_t* __restrict__ maskout) {
  PSEG_HELPER_PRIO();
  typedef fvec<V> vf;
  float vmax = 0.f;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t r = cvdiv.div(i);
    const uint32_t c = (i - r * cvdiv.d) * V;
    vf v = ldvec<V>(y + (long long)r * ldy + c);
    if (scale) v = (v - ldvec<V>(mean + c)) * ldvec<V>(scale + c) + ldvec<V>(shift + c);
    if (res) v += ldvec<V>(res + (long long)r * ldr + c);
    if (maskout != nullptr) {
      // (C % 32 == 0, host-checked: 32 / V consecutive lanes hold the bit groups of one word; total % (32 / V) == 0, and
      // the grid stride is a multiple of 256, so the lanes of a word are always active together)
      // the mask of the value AS STORED: under fp16 storage a pre-activation below half the smallest fp16 subnormal is stored
      // as 0 (and one just under 6 as 6): backward passes that read z instead of this bitmask (the fused small-tensor kernels)
      // must see the same mask -- a replayed step (bitmask) and an eager one (z) parted at step 133 of an HRNet -mp run before
      vf vs = v;
      if constexpr (sizeof(T) == 2) vs = __builtin_convertvector(__builtin_convertvector(v, f16x8v), vf);
      const vf m = act_mask_v<V>(vs, act);
      uint32_t bits = 0;
#pragma unroll
      for (int k = 0; k < V; ++k) bits |= (m[k] != 0.f ? 1u : 0u) << k;
      bits <<= (c & 31);
#pragma unroll
      for (int o = 1; o < 32 / V; o <<= 1) bits |= __shfl_xor(bits, o, 64);
      if ((threadIdx.x & (32 / V - 1)) == 0) maskout[(long long)r * ((cvdiv.d * V) >> 5) + (c >> 5)] = bits;
    }
    if (act == PSEG_ACT_RELU) {
#pragma unroll
      for (int k = 0; k < V; ++k) v[k] = fmaxf(v[k], 0.f);
    } else if (act == PSEG_ACT_RELU6) {
#pragma unroll
      for (int k = 0; k < V; ++k) v[k] = fminf(fmaxf(v[k], 0.f), 6.f);
    }
    stvec<V>(z + (long long)r * ldz + c, v);
#pragma unroll
    for (int k = 0; k < V; ++k) vmax = fmaxf(vmax, fabsf(v[k]));
  }
  if (amax != nullptr) {  // publish max|z| (bit pattern of a non-negative float: unsigned order == float order)
    __shared__ float shm[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float m = fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]));
      publish_amax(amax, m);
    }
  }
}

// The forward pass laid out like the backward passes below (fp16 tensors): a thread owns eight channels -- mean / scale /
// shift are loaded ONCE, not per element -- and streams NR rows at a time.  In bn_act_fwd_kernel every 16 bytes of payload
// come with 96 bytes of per-channel vector loads through the L1: 10-20 % of the pass at the DeepLabV3+ shapes (45.5 against
// 40.8 us without them at 16 x 128 x 128 x 256; profiles/EXPERIMENTS.md 0.13).  Same arithmetic in the same order: the
// results are bit-identical to bn_act_fwd_kernel's.
template <int NR, typename T, int V>
__global__ __launch_bounds__(256) void bn_act_fwd_rows_kernel(const T* __restrict__ y, int ldy,
                                                              const float* __restrict__ mean, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const T* __restrict__ res,
                                                              int ldr, int act, T* __restrict__ z, int ldz, long long M, int C,
                                                              int RB, uint32_t* __restrict__ maskout) {
  PSEG_HELPER_PRIO();
  typedef fvec<V> vf;
  const int TX = blockDim.x, TY = blockDim.y;
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int cv = blockIdx.y * TX + tx;
  if (cv * V >= C) return;       // (C % 32 == 0 when a mask is written: the four lanes of a mask word leave together)
  const int c = cv * V;
  const long long r0 = (long long)blockIdx.x * RB;
  const int nrows = (int)((r0 + RB > M ? M : r0 + RB) - r0);
  const int words = C >> 5;
  constexpr int ES = (int)sizeof(T);
  auto span = [&](int ld) { return (uint32_t)(((long long)(nrows - 1) * ld + C) * ES); };
  const __amdgpu_buffer_rsrc_t yr = make_rsrc(y + r0 * ldy, span(ldy));
  const __amdgpu_buffer_rsrc_t zr = make_rsrc(z + r0 * ldz, span(ldz));
  const bool has_res = res != nullptr, has_mask = maskout != nullptr;
  const __amdgpu_buffer_rsrc_t rr = make_rsrc(has_res ? res + r0 * ldr : y, has_res ? span(ldr) : 0u);
  const __amdgpu_buffer_rsrc_t mr =
      make_rsrc(has_mask ? maskout + r0 * words : (uint32_t*)z, has_mask ? (uint32_t)(nrows * words * 4) : 0u);
  const vf mu = ldvec<V>(mean + c), sc = ldvec<V>(scale + c), sh = ldvec<V>(shift + c);
  const int vy = (ty * ldy + c) * ES, vz = (ty * ldz + c) * ES, vr = (ty * ldr + c) * ES, vm = (ty * words + (c >> 5)) * 4;
  for (int base = 0; base < nrows; base += NR * TY) {   // block-uniform trip count
    vf v[NR], rs[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int row = base + k * TY;                     // (+ ty: in the lane offset)
      v[k] = buf_ldvec<V, T>(yr, vy, row * ldy * ES);
      if (has_res) rs[k] = buf_ldvec<V, T>(rr, vr, row * ldr * ES);
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int row = base + k * TY;
      vf w = (v[k] - mu) * sc + sh;
      if (has_res) w += rs[k];
      if (has_mask) {
        vf ws = w;      // the mask of the value as stored (see bn_act_fwd_kernel)
        if constexpr (sizeof(T) == 2) ws = __builtin_convertvector(__builtin_convertvector(w, f16x8v), vf);
        const vf m = act_mask_v<V>(ws, act);
        uint32_t bits = 0;
#pragma unroll
        for (int j = 0; j < V; ++j) bits |= (m[j] != 0.f ? 1u : 0u) << j;
        bits <<= (c & 31);
#pragma unroll
        for (int o = 1; o < 32 / V; o <<= 1) bits |= __shfl_xor(bits, o, 64);
        if ((tx & (32 / V - 1)) == 0) __builtin_amdgcn_raw_buffer_store_b32((int)bits, mr, vm, row * words * 4, 0);
      }
      if (act == PSEG_ACT_RELU) {
#pragma unroll
        for (int j = 0; j < V; ++j) w[j] = fmaxf(w[j], 0.f);
      } else if (act == PSEG_ACT_RELU6) {
#pragma unroll
        for (int j = 0; j < V; ++j) w[j] = fminf(fmaxf(w[j], 0.f), 6.f);
      }
      buf_stvec<V, T>(w, zr, vz, row * ldz * ES);
    }
  }
}

// dy = scale * (dyh - c1 - xhat * c2), dyh = dz * act'(z); optional dres (+)= dyh; optional bf16 limb planes of dy.
// Laid out like bn_bwd_reduce_kernel and for the same reason (it runs beside a weight gradient that leaves it 96 VGPRs per
// SIMD): a thread owns four channels -- the per-channel vectors are loaded once, not per element -- and streams NR rows at
// a time through buffer loads / stores with one lane offset per tensor and the row advance in the scalar offset; rows past
// the block's range read as zeros and their stores are dropped by the range check.
template <int MODE, int NR, typename T, int V>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(
    const T* __restrict__ dz, int lddz, const T* __restrict__ z, int ldz, const T* __restrict__ y, int ldy,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ c1, const float* __restrict__ c2, int act,
    T* __restrict__ dy, int lddy, T* __restrict__ dres, int lddres, int res_acc, long long M, int C, int RB,
    const uint32_t* __restrict__ mask, uint16_t* __restrict__ dy_hi, uint16_t* __restrict__ dy_lo, int ldp) {
  PSEG_HELPER_PRIO();
  typedef fvec<V> vf;
  const int TX = blockDim.x, TY = blockDim.y;
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int cv = blockIdx.y * TX + tx;
  if (cv * V >= C) return;
  const int c = cv * V;
  const long long r0 = (long long)blockIdx.x * RB;
  const int nrows = (int)((r0 + RB > M ? M : r0 + RB) - r0);
  const int words = C >> 5;
  constexpr int ES = (int)sizeof(T);
  auto span = [&](int ld) { return (uint32_t)(((long long)(nrows - 1) * ld + C) * ES); };
  const __amdgpu_buffer_rsrc_t dzr = make_rsrc(dz + r0 * lddz, span(lddz));
  const __amdgpu_buffer_rsrc_t yr = make_rsrc(y + r0 * ldy, span(ldy));
  const __amdgpu_buffer_rsrc_t dyr = make_rsrc(dy + r0 * lddy, span(lddy));
  const __amdgpu_buffer_rsrc_t zr = make_rsrc(MODE == 2 ? z + r0 * ldz : y, MODE == 2 ? span(ldz) : 0u);
  const __amdgpu_buffer_rsrc_t mr =
      make_rsrc(MODE == 1 ? mask + r0 * words : (const uint32_t*)y, MODE == 1 ? (uint32_t)(nrows * words * 4) : 0u);
  const bool has_res = dres != nullptr, has_pl = V == 4 && dy_hi != nullptr;
  const __amdgpu_buffer_rsrc_t rr = make_rsrc(has_res ? dres + r0 * lddres : dy, has_res ? span(lddres) : 0u);
  const __amdgpu_buffer_rsrc_t hr =
      make_rsrc(has_pl ? (const void*)(dy_hi + r0 * ldp) : (const void*)dy, has_pl ? (uint32_t)(((long long)(nrows - 1) * ldp + C) * 2) : 0u);
  const __amdgpu_buffer_rsrc_t lr =
      make_rsrc(has_pl ? (const void*)(dy_lo + r0 * ldp) : (const void*)dy, has_pl ? (uint32_t)(((long long)(nrows - 1) * ldp + C) * 2) : 0u);
  const vf mu = ldvec<V>(mean + c), is = ldvec<V>(invstd + c), sc = ldvec<V>(scale + c), k1 = ldvec<V>(c1 + c),
           k2 = ldvec<V>(c2 + c);
  vf sh4 = {};
  if (MODE == 3) sh4 = ldvec<V>(shift + c);
  const int vdz = (ty * lddz + c) * ES, vy = (ty * ldy + c) * ES, vz = (ty * ldz + c) * ES, vdy = (ty * lddy + c) * ES,
            vr = (ty * lddres + c) * ES, vm = (ty * words + (c >> 5)) * 4, vp = (ty * ldp + c) * 2;
  for (int base = 0; base < nrows; base += NR * TY) {   // block-uniform trip count
    vf g[NR], yv[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int row = base + k * TY;                     // (+ ty: in the lane offset)
      g[k] = buf_ldvec<V, T>(dzr, vdz, row * lddz * ES);
      yv[k] = buf_ldvec<V, T>(yr, vy, row * ldy * ES);
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      const int row = base + k * TY;
      if (MODE == 1) {
        const uint32_t w = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(mr, vm, row * words * 4, 0);
        g[k] *= bits_to_mask<V>(w, c);
      } else if (MODE == 2) {
        g[k] *= act_mask_v<V>(buf_ldvec<V, T>(zr, vz, row * ldz * ES), act);
      } else if (MODE == 3) {
        g[k] *= act_mask_v<V>((yv[k] - mu) * sc + sh4, act);
      }
      if (has_res) {
        vf rv = g[k];
        if (res_acc) rv = buf_ldvec<V, T>(rr, vr, row * lddres * ES) + g[k];
        buf_stvec<V, T>(rv, rr, vr, row * lddres * ES);
      }
      const vf xh = (yv[k] - mu) * is;
      const vf out = sc * (g[k] - k1 - xh * k2);
      buf_stvec<V, T>(out, dyr, vdy, row * lddy * ES);
      if constexpr (V == 4) {
        if (has_pl) {
          // bf16 limb planes of dy for the pre-split LDS-DMA data gradient (pseg_conv2d_dgrad_planes): hi = bf16(x),
          // lo = bf16(x - hi) -- the residual is exact in fp32, so these are the limbs pseg_split_planes would write
          typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
          typedef int i32x2 __attribute__((ext_vector_type(2)));
          bf16x4 hi, lo;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            hi[j] = (__bf16)out[j];
            lo[j] = (__bf16)(out[j] - (float)hi[j]);
          }
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, hi), hr, vp, row * ldp * 2, 0);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, lo), lr, vp, row * ldp * 2, 0);
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ dz, int lddz, const T* __restrict__ z,
                                                      int ldz, const float* __restrict__ scale, int act,
                                                      T* __restrict__ dy, int lddy, T* __restrict__ dres,
                                                      int lddres, int res_acc, uint32_t total, FastDiv c4div) {
  PSEG_HELPER_PRIO();
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t r = c4div.div(i);
    const uint32_t c = (i - r * c4div.d) * 4;
    f32x4 g = ld4(dz + (long long)r * lddz + c);
    if (act != PSEG_ACT_NONE) g *= act_mask(ld4(z + (long long)r * ldz + c), act);
    if (dres) {
      T* dp = dres + (long long)r * lddres + c;
      st4(dp, res_acc ? ld4(dp) + g : g);
    }
    if (dy) st4(dy + (long long)r * lddy + c, scale ? g * ld4(scale + c) : g);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void copy2d_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy,
                                                     int accumulate, uint32_t total, FastDiv c4div) {
  PSEG_HELPER_PRIO();
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t r = c4div.div(i);
    const uint32_t c = (i - r * c4div.d) * 4;
    T* yp = y + (long long)r * ldy + c;
    const f32x4 v = ld4(x + (long long)r * ldx + c);
    st4(yp, accumulate ? ld4(yp) + v : v);
  }
}

// ------------------------------------------------------------------------------------------------
// Small tensors (the launch-bound configurations: UNet at 256x256, HRNet's low-resolution branches): a training step is
// ~500-1000 dependent launches of a few microseconds each, so the two tiny finalize launches per BatchNorm layer cost as
// much as the passes they serve.  Here every block of the apply pass re-derives the per-channel coefficients of ITS 64
// channels from the (few) partial rows -- same arithmetic, same order, same double accumulators as bn_finalize_kernel /
// bn_bwd_finalize_kernel, hence bit-identical coefficients -- and the blocks of row chunk 0 publish them (coefficients for
// backward, running statistics, dgamma / dbeta).  One launch instead of two, forward and backward.
constexpr int kSmallCh = 64;        // channels per block (16 float4 lanes)
constexpr int kSmallRowsPerBlock = 64;
constexpr int kSmallMaxPartials = 64;   // use the fused kernels up to this many partial rows

template <typename T>
__global__ __launch_bounds__(256) void bn_fwd_small_kernel(
    const float* __restrict__ stat, int rows, int group, long long count, int C, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
    float* __restrict__ mean_o, float* __restrict__ invstd_o, float* __restrict__ scale_o, float* __restrict__ shift_o,
    const T* __restrict__ y, int ldy, const T* __restrict__ res, int ldr, int act, T* __restrict__ z, int ldz,
    long long M, unsigned* __restrict__ amax) {
  PSEG_HELPER_PRIO();
  __shared__ __attribute__((aligned(16))) float s_mean[kSmallCh], s_scale[kSmallCh], s_shift[kSmallCh];
  __shared__ float shm[4];
  __shared__ double s_part[2][kSmallCh][kFinLanes + 1];
  const int c0 = blockIdx.y * kSmallCh;
  {
    // Four threads per channel share the 32 strided lane sums of bn_finalize_kernel<kFinLanes> (thread q takes lanes q,
    // q + 4, ...; each lane is <= 2 row groups here), one of them adds the 32 in lane order: the same terms in the same
    // order -- bit-identical coefficients -- without one thread walking all the row groups.
    const int ch = threadIdx.x & (kSmallCh - 1), q = threadIdx.x / kSmallCh;
    const int c = c0 + ch;
    const bool cok = c < C;
    const long long plane = (long long)rows * C;
    auto rows_of = [&](int g) -> long long {
      long long n = count - (long long)g * group;
      return n > group ? group : n;
    };
    double k0 = 0.0;
    if (cok) {
      k0 = (double)stat[c];
      for (int j = q; j < kFinLanes; j += 256 / kSmallCh) {
        double v0 = 0.0, v1 = 0.0;
        for (int g = j; g < rows; g += kFinLanes) {
          const long long n = rows_of(g);
          if (n <= 0) continue;
          const long long o = (long long)g * C + c;
          const double d = (double)stat[o] - k0, s1 = stat[plane + o];
          v0 += s1 + (double)n * d;
          v1 += (double)stat[2 * plane + o] + 2.0 * d * s1 + (double)n * d * d;
        }
        s_part[0][ch][j] = v0;
        s_part[1][ch][j] = v1;
      }
    }
    __syncthreads();
    if (cok && q == 0) {
      double v0 = s_part[0][ch][0], v1 = s_part[1][ch][0];
      for (int j = 1; j < kFinLanes; ++j) {
        v0 += s_part[0][ch][j];
        v1 += s_part[1][ch][j];
      }
      const double n = (double)count;
      const double mu = k0 + v0 / n;
      double M2 = v1 - v0 * v0 / n;
      if (M2 < 0.0) M2 = 0.0;
      const double var = M2 / n;
      const float is = (float)(1.0 / sqrt(var + (double)eps));
      const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
      const float fm = (float)mu;
      s_mean[ch] = fm;
      s_scale[ch] = g * is;
      s_shift[ch] = b;
      if (blockIdx.x == 0) {
        mean_o[c] = fm;
        invstd_o[c] = is;
        scale_o[c] = g * is;
        shift_o[c] = b;
        if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * fm;
        if (rvar) {
          const double unbiased = count > 1 ? M2 / (n - 1.0) : var;
          rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
        }
      }
    }
  }
  __syncthreads();
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = c0 + tx * 4;
  float vmax = 0.f;
  if (c < C) {
    const f32x4 mu = ld4(&s_mean[tx * 4]), sc = ld4(&s_scale[tx * 4]), sh = ld4(&s_shift[tx * 4]);
    const long long r0 = (long long)blockIdx.x * kSmallRowsPerBlock;
    long long r1 = r0 + kSmallRowsPerBlock;
    if (r1 > M) r1 = M;
    for (long long r = r0 + ty; r < r1; r += 16) {
      f32x4 v = (ld4(y + r * ldy + c) - mu) * sc + sh;
      if (res) v += ld4(res + r * ldr + c);
      if (act == PSEG_ACT_RELU) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
      } else if (act == PSEG_ACT_RELU6) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fminf(fmaxf(v[k], 0.f), 6.f);
      }
      st4(z + r * ldz + c, v);
      vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
  }
  if (amax != nullptr) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float m = fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]));
      publish_amax(amax, m);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_small_kernel(
    const float* __restrict__ pdb, const float* __restrict__ pdg, int rows, long long count, int C,
    float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate, int frozen, const T* __restrict__ dz,
    int lddz, const T* __restrict__ z, int ldz, const T* __restrict__ y, int ldy, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ scale, const float* __restrict__ shift, int act,
    T* __restrict__ dy, int lddy, T* __restrict__ dres, int lddres, int res_acc, long long M) {
  PSEG_HELPER_PRIO();
  __shared__ __attribute__((aligned(16))) float s_c1[kSmallCh], s_c2[kSmallCh];
  __shared__ double s_pa[kSmallCh][kFinLanes + 1], s_pb[kSmallCh][kFinLanes + 1];
  const int c0 = blockIdx.y * kSmallCh;
  {
    // (four threads per channel over the 32 strided partial sums of bn_bwd_finalize_kernel, summed in lane order by one:
    // see bn_fwd_small_kernel)
    const int ch = threadIdx.x & (kSmallCh - 1), q = threadIdx.x / kSmallCh;
    const int c = c0 + ch;
    const bool cok = c < C;
    if (cok)
      for (int j = q; j < kFinLanes; j += 256 / kSmallCh) {
        double a = 0.0, b = 0.0;
        for (int g = j; g < rows; g += kFinLanes) {
          a += (double)pdb[(long long)g * C + c];
          b += (double)pdg[(long long)g * C + c];
        }
        s_pa[ch][j] = a;
        s_pb[ch][j] = b;
      }
    __syncthreads();
    if (q == 0) {
      float c1 = 0.f, c2 = 0.f;
      if (cok) {
        double db = 0.0, dg = 0.0;
        for (int j = 0; j < kFinLanes; ++j) {
          db += s_pa[ch][j];
          dg += s_pb[ch][j];
        }
        const float fdb = (float)db, fdg = (float)dg;
        if (blockIdx.x == 0) {
          if (dbeta) dbeta[c] = accumulate ? dbeta[c] + fdb : fdb;
          if (dgamma) dgamma[c] = accumulate ? dgamma[c] + fdg : fdg;
        }
        c1 = frozen ? 0.f : (float)(db / (double)count);
        c2 = frozen ? 0.f : (float)(dg / (double)count);
      }
      s_c1[ch] = c1;
      s_c2[ch] = c2;
    }
  }
  __syncthreads();
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = c0 + tx * 4;
  if (c >= C) return;
  const f32x4 c1 = ld4(&s_c1[tx * 4]), c2 = ld4(&s_c2[tx * 4]);
  const f32x4 mu = ld4(mean + c), is = ld4(invstd + c), sc = ld4(scale + c);
  f32x4 sh = {0.f, 0.f, 0.f, 0.f};
  if (shift) sh = ld4(shift + c);
  const long long r0 = (long long)blockIdx.x * kSmallRowsPerBlock;
  long long r1 = r0 + kSmallRowsPerBlock;
  if (r1 > M) r1 = M;
  for (long long r = r0 + ty; r < r1; r += 16) {
    f32x4 g = ld4(dz + r * lddz + c);
    const f32x4 yv = ld4(y + r * ldy + c);
    if (act != PSEG_ACT_NONE) g *= act_mask(z != nullptr ? ld4(z + r * ldz + c) : (yv - mu) * sc + sh, act);
    if (dres) {
      T* dp = dres + r * lddres + c;
      st4(dp, res_acc ? ld4(dp) + g : g);
    }
    const f32x4 xh = (yv - mu) * is;
    st4(dy + r * lddy + c, sc * (g - c1 - xh * c2));
  }
}

// ------------------------------------------------------------------------------------------------ host
static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// V: channels per lane (4; 8 for the fp16 streaming passes)
static void stat_block(int C, dim3& block, dim3& grid, long long M, int R, int V = 4) {
  const int cv = C / V;
  // (narrow tensors: eight column lanes, so that at most half a block idles on HRNet's 32-channel branches)
  const int tx = cv >= 64 ? 64 : (cv > 16 ? 32 : (cv > 8 ? 16 : 8));
  block = dim3(tx, 256 / tx);
  grid = dim3((unsigned)cdiv(M, R), (unsigned)cdiv(cv, tx));
}

// channels per lane of the streaming BatchNorm passes: 16 bytes either way
template <typename T>
constexpr int lane_channels() { return sizeof(T) == 2 ? 8 : 4; }

static int ew_grid(long long total) {
  long long b = (total + 255) / 256;
  if (b > 256 * 8) b = 256 * 8;  // 8 blocks per CU, grid-stride the rest
  if (b < 1) b = 1;
  return (int)b;
}

#define EW_COMMON_CHECKS(name, M, C)                                                              \
  PSEG_REQUIRE((M) > 0 && (C) > 0 && (C) % 4 == 0, name ": need M > 0, C %% 4 == 0 (M=%lld C=%d)", \
               (long long)(M), (int)(C));                                                         \
  PSEG_REQUIRE((long long)(M) * ((C) / 4) < (1LL << 31), name ": tensor too large")

// used by dwconv.hip as well
int launch_col_reduce(const float* part, int rows, int C, float* out, int accumulate, hipStream_t st) {
  if (rows >= kFinWideRows)
    hipLaunchKernelGGL(col_reduce_kernel<kFinLanesWide>, dim3(cdiv(C, kFinCh)), dim3(kFinCh * kFinLanesWide), 0, st, part,
                       rows, C, out, accumulate);
  else
    hipLaunchKernelGGL(col_reduce_kernel<kFinLanes>, dim3(cdiv(C, kFinCh)), dim3(kFinCh * kFinLanes), 0, st, part, rows, C,
                       out, accumulate);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}


// ------------------------------------------------------------------------------------------------ entry points, by storage type
template <typename T>
static bool ld_ok(int ld) { return ld % (sizeof(T) == 2 ? 8 : 4) == 0; }

template <typename T>
static int col_stats_impl(const T* y, int ldy, int64_t M, int C, float* stat, void* stream) {
  PSEG_REQUIRE(y && stat, "col_stats: null pointer");
  EW_COMMON_CHECKS("col_stats", M, C);
  PSEG_REQUIRE(ld_ok<T>(ldy) && al16(y) && al16(stat), "col_stats: alignment");
  dim3 block, grid;
  const int R = stat_group(M, C);
  stat_block(C, block, grid, M, R);
  hipLaunchKernelGGL((col_stats_kernel<true, T>), grid, block, 0, (hipStream_t)stream, y, ldy, (long long)M, C, R, stat,
                     (long long)cdiv(M, R) * C);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

template <typename T>
static int bn_fwd_fused_impl(const float* stat, int rows, int group, int64_t count, int C, const float* gamma,
                             const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                             float* mean, float* invstd, float* scale, float* shift, const T* y, int ldy, const T* residual,
                             int ldr, int act, T* z, int ldz, int64_t M, float* amax_z, void* stream) {
  PSEG_REQUIRE(stat && mean && invstd && scale && shift && y && z, "bn_fwd_fused: null pointer");
  PSEG_REQUIRE(rows > 0 && rows <= kSmallMaxPartials && group > 0 && count > 0, "bn_fwd_fused: bad sizes (rows %d)", rows);
  PSEG_REQUIRE((long long)rows * group >= count, "bn_fwd_fused: groups do not cover the rows");
  EW_COMMON_CHECKS("bn_fwd_fused", M, C);
  PSEG_REQUIRE(ld_ok<T>(ldy) && ld_ok<T>(ldz) && (!residual || ld_ok<T>(ldr)) && al16(y) && al16(z) && al16(residual),
               "bn_fwd_fused: alignment");
  const dim3 grid((unsigned)cdiv(M, kSmallRowsPerBlock), (unsigned)cdiv(C, kSmallCh));
  hipLaunchKernelGGL(bn_fwd_small_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, stat, rows, group, (long long)count,
                     C, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, y, ldy, residual,
                     ldr, act, z, ldz, (long long)M, (unsigned*)amax_z);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

template <typename T>
static int bn_bwd_fused_impl(const float* part_db, const float* part_dg, int rows, int64_t count, int C, float* dgamma,
                             float* dbeta, int accumulate, int frozen, const T* dz, int lddz, const T* z, int ldz,
                             const T* y, int ldy, const float* mean, const float* invstd, const float* scale,
                             const float* shift, int act, T* dy, int lddy, T* dres, int lddres, int res_accumulate,
                             int64_t M, void* stream) {
  PSEG_REQUIRE(part_db && part_dg && dz && y && mean && invstd && scale && dy, "bn_bwd_fused: null pointer");
  PSEG_REQUIRE(rows > 0 && count > 0, "bn_bwd_fused: bad sizes");
  PSEG_REQUIRE(act == PSEG_ACT_NONE || z || shift, "bn_bwd_fused: activation needs z or shift");
  EW_COMMON_CHECKS("bn_bwd_fused", M, C);
  PSEG_REQUIRE(ld_ok<T>(lddz) && ld_ok<T>(ldy) && ld_ok<T>(lddy) && (!z || ld_ok<T>(ldz)) && (!dres || ld_ok<T>(lddres)) &&
                   al16(dz) && al16(z) && al16(y) && al16(dy) && al16(dres),
               "bn_bwd_fused: alignment");
  const dim3 grid((unsigned)cdiv(M, kSmallRowsPerBlock), (unsigned)cdiv(C, kSmallCh));
  hipLaunchKernelGGL(bn_bwd_small_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, part_db, part_dg, rows,
                     (long long)count, C, dgamma, dbeta, accumulate, frozen, dz, lddz, z, ldz, y, ldy, mean, invstd, scale,
                     shift, act, dy, lddy, dres, lddres, res_accumulate, (long long)M);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

template <typename T>
static int bn_act_fwd_impl(const T* y, int ldy, const float* mean, const float* scale, const float* shift, const T* residual,
                           int ldr, int act, T* z, int ldz, int64_t M, int C, float* amax_z, uint32_t* mask_out,
                           void* stream) {
  PSEG_REQUIRE(y && z, "bn_act_fwd: null pointer");
  PSEG_REQUIRE(mask_out == nullptr || (C % 32 == 0 && act != PSEG_ACT_NONE), "bn_act_fwd: mask_out needs C %% 32 == 0 and an activation");
  PSEG_REQUIRE((scale == nullptr) == (shift == nullptr) && (scale == nullptr) == (mean == nullptr),
               "bn_act_fwd: mean/scale/shift must come together");
  EW_COMMON_CHECKS("bn_act_fwd", M, C);
  PSEG_REQUIRE(ld_ok<T>(ldy) && ld_ok<T>(ldz) && (!residual || ld_ok<T>(ldr)) && al16(y) && al16(z) && al16(residual) &&
                   al16(mean) && al16(scale) && al16(shift),
               "bn_act_fwd: alignment");
  constexpr int V = lane_channels<T>();
  const uint32_t total = (uint32_t)(M * (C / V));
  static const int rows_mode = [] {
    const char* e = getenv("PSEG_BN_FWD_ROWS");      // 0: the element-streaming kernel everywhere (A/B)
    return e ? atoi(e) : 1;
  }();
  if constexpr (sizeof(T) == 2) {
    // (without a residual: there it wins 5-12 % at every DeepLabV3+ shape; with one the two kernels are within +-3 %)
    if (rows_mode != 0 && scale != nullptr && amax_z == nullptr && C % 8 == 0 && M >= 256 && (residual == nullptr || rows_mode == 2)) {
      dim3 block, grid;
      stat_block(C, block, grid, M, 1, V);
      constexpr int kNR = 4;
      const int sweep = kNR * (int)block.y;
      long long sweeps = (M * (long long)grid.y) / ((long long)sweep * 2048);
      sweeps = sweeps < 1 ? 1 : (sweeps > 16 ? 16 : sweeps);
      const int RB = (int)sweeps * sweep;
      grid.x = (unsigned)cdiv(M, RB);
      hipLaunchKernelGGL((bn_act_fwd_rows_kernel<kNR, T, V>), grid, block, 0, (hipStream_t)stream, y, ldy, mean, scale, shift,
                         residual, ldr, act, z, ldz, (long long)M, C, RB, mask_out);
      PSEG_LAUNCH_CHECK();
      return PSEG_OK;
    }
  }
  hipLaunchKernelGGL((bn_act_fwd_kernel<T, V>), dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, y, ldy, mean, scale,
                     shift, residual, ldr, act, z, ldz, total, FastDiv((uint32_t)(C / V)), (unsigned*)amax_z, mask_out);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

template <typename T>
static int bn_act_bwd_reduce_impl(const T* dz, int lddz, const T* z, int ldz, const T* y, int ldy, const float* mean,
                                  const float* invstd, const float* scale, const float* shift, int act, int64_t M, int C,
                                  float* part_db, float* part_dg, const uint32_t* mask, void* stream) {
  PSEG_REQUIRE(dz && y && mean && invstd && part_db && part_dg, "bn_act_bwd_reduce: null pointer");
  PSEG_REQUIRE(act == PSEG_ACT_NONE || z || mask || (scale && shift), "bn_act_bwd_reduce: activation needs z, mask or scale/shift");
  PSEG_REQUIRE(mask == nullptr || C % 32 == 0, "bn_act_bwd_reduce: mask needs C %% 32 == 0");
  EW_COMMON_CHECKS("bn_act_bwd_reduce", M, C);
  PSEG_REQUIRE(ld_ok<T>(lddz) && ld_ok<T>(ldy) && (!z || ld_ok<T>(ldz)) && al16(dz) && al16(z) && al16(y), "bn_act_bwd_reduce: alignment");
  dim3 block, grid;
  const int R = stat_group(M, C);
  constexpr int V = lane_channels<T>();
  stat_block(C, block, grid, M, R, V);
  const int mode = act == PSEG_ACT_NONE ? 0 : (mask != nullptr ? 1 : (z != nullptr ? 2 : 3));
  // four rows in flight: 58-86 VGPRs (six: 102 -- past the 96 a resident weight gradient leaves, and then no faster than
  // the old two-row kernel: DeepLabV3+ step 348.1 / 469.8 images/s fp32 / mixed with four, 345.5 / 464.6 with six)
#define PSEG_BWD_REDUCE(MODE)                                                                                              \
  hipLaunchKernelGGL((bn_bwd_reduce_kernel<MODE, (V == 8 ? 2 : 4), T, V>), grid, block, 0, (hipStream_t)stream, dz, lddz, z, ldz, y, ldy, mean, \
                     invstd, scale, shift, act, (long long)M, C, R, part_db, part_dg, mask)
  switch (mode) {
    case 0: PSEG_BWD_REDUCE(0); break;
    case 1: PSEG_BWD_REDUCE(1); break;
    case 2: PSEG_BWD_REDUCE(2); break;
    default: PSEG_BWD_REDUCE(3); break;
  }
#undef PSEG_BWD_REDUCE
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

template <typename T>
static int bn_act_bwd_apply_impl(const T* dz, int lddz, const T* z, int ldz, const T* y, int ldy, const float* mean,
                                 const float* invstd, const float* scale, const float* shift, const float* c1,
                                 const float* c2, int act, T* dy, int lddy, T* dres, int lddres, int res_accumulate,
                                 int64_t M, int C, const uint32_t* mask, uint16_t* dy_hi, uint16_t* dy_lo, int ldp,
                                 void* stream) {
  PSEG_REQUIRE(dz && y && mean && invstd && scale && c1 && c2 && dy, "bn_act_bwd_apply: null pointer");
  PSEG_REQUIRE((dy_hi == nullptr) == (dy_lo == nullptr), "bn_act_bwd_apply: dy_hi / dy_lo come together");
  PSEG_REQUIRE(dy_hi == nullptr || (ldp == C && C % 8 == 0 && (((uintptr_t)dy_hi | (uintptr_t)dy_lo) & 15) == 0),
               "bn_act_bwd_apply: limb planes need ldp == C, C %% 8 == 0, 16-byte alignment");
  PSEG_REQUIRE(act == PSEG_ACT_NONE || z || mask || shift, "bn_act_bwd_apply: activation needs z, mask or shift");
  PSEG_REQUIRE(mask == nullptr || C % 32 == 0, "bn_act_bwd_apply: mask needs C %% 32 == 0");
  EW_COMMON_CHECKS("bn_act_bwd_apply", M, C);
  PSEG_REQUIRE(ld_ok<T>(lddz) && ld_ok<T>(ldy) && ld_ok<T>(lddy) && (!z || ld_ok<T>(ldz)) && (!dres || ld_ok<T>(lddres)) &&
                   al16(dz) && al16(z) && al16(y) && al16(dy) && al16(dres),
               "bn_act_bwd_apply: alignment");
  // rows per block: whole sweeps of NR x TY rows, as many as keep >= ~2048 blocks in the launch
  dim3 block, grid;
  constexpr int V = lane_channels<T>();
  stat_block(C, block, grid, M, 1, V);
  // rows in flight per lane: the same bytes in flight for 8- and 4-channel lanes (four rows at V = 8: no faster, 98-130 VGPRs)
  constexpr int kNR = V == 8 ? 2 : 4;
  const int sweep = kNR * (int)block.y;
  long long sweeps = (M * (long long)grid.y) / ((long long)sweep * 2048);
  sweeps = sweeps < 1 ? 1 : (sweeps > 16 ? 16 : sweeps);
  const int RB = (int)sweeps * sweep;
  grid.x = (unsigned)cdiv(M, RB);
  const int mode = act == PSEG_ACT_NONE ? 0 : (mask != nullptr ? 1 : (z != nullptr ? 2 : 3));
#define PSEG_BWD_APPLY(MODE)                                                                                                 \
  hipLaunchKernelGGL((bn_act_bwd_apply_kernel<MODE, kNR, T, V>), grid, block, 0, (hipStream_t)stream, dz, lddz, z, ldz, y, ldy, \
                     mean, invstd, scale, shift, c1, c2, act, dy, lddy, dres, lddres, res_accumulate, (long long)M, C, RB,    \
                     mask, dy_hi, dy_lo, ldp)
  switch (mode) {
    case 0: PSEG_BWD_APPLY(0); break;
    case 1: PSEG_BWD_APPLY(1); break;
    case 2: PSEG_BWD_APPLY(2); break;
    default: PSEG_BWD_APPLY(3); break;
  }
#undef PSEG_BWD_APPLY
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

template <typename T>
static int act_bwd_impl(const T* dz, int lddz, const T* z, int ldz, const float* scale, int act, T* dy, int lddy, T* dres,
                        int lddres, int res_accumulate, int64_t M, int C, void* stream) {
  PSEG_REQUIRE(dz && (dy || dres), "act_bwd: null pointer");
  PSEG_REQUIRE(act == PSEG_ACT_NONE || z, "act_bwd: activation needs z");
  EW_COMMON_CHECKS("act_bwd", M, C);
  PSEG_REQUIRE(ld_ok<T>(lddz) && (!dy || ld_ok<T>(lddy)) && (!z || ld_ok<T>(ldz)) && (!dres || ld_ok<T>(lddres)) && al16(dz) &&
                   al16(z) && al16(dy) && al16(dres) && al16(scale),
               "act_bwd: alignment");
  const uint32_t total = (uint32_t)(M * (C / 4));
  hipLaunchKernelGGL(act_bwd_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dz, lddz, z, ldz, scale, act,
                     dy, lddy, dres, lddres, res_accumulate, total, FastDiv((uint32_t)(C / 4)));
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

template <typename T>
static int col_sum_impl(const T* dy, int ldy, int64_t M, int C, float* out, int accumulate, void* workspace,
                        int64_t workspace_bytes, void* stream) {
  PSEG_REQUIRE(dy && out, "col_sum: null pointer");
  EW_COMMON_CHECKS("col_sum", M, C);
  PSEG_REQUIRE(ld_ok<T>(ldy) && al16(dy) && al16(workspace), "col_sum: alignment");
  const int R = stat_group(M, C);
  const int rows = cdiv(M, R);
  const long long need = (long long)rows * C * 4;
  if (!workspace || workspace_bytes < need) {
    set_error("col_sum: needs %lld workspace bytes, got %lld", need, (long long)workspace_bytes);
    return PSEG_ERR_WORKSPACE;
  }
  dim3 block, grid;
  stat_block(C, block, grid, M, R);
  hipLaunchKernelGGL((col_stats_kernel<false, T>), grid, block, 0, (hipStream_t)stream, dy, ldy, (long long)M, C, R,
                     (float*)workspace, 0LL);
  PSEG_LAUNCH_CHECK();
  return launch_col_reduce((const float*)workspace, rows, C, out, accumulate, (hipStream_t)stream);
}

template <typename T>
static int copy2d_impl(const T* x, int ldx, T* y, int ldy, int64_t M, int C, int accumulate, void* stream) {
  PSEG_REQUIRE(x && y, "copy2d: null pointer");
  EW_COMMON_CHECKS("copy2d", M, C);
  PSEG_REQUIRE(ld_ok<T>(ldx) && ld_ok<T>(ldy) && al16(x) && al16(y), "copy2d: alignment");
  const uint32_t total = (uint32_t)(M * (C / 4));
  hipLaunchKernelGGL(copy2d_kernel<T>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, accumulate,
                     total, FastDiv((uint32_t)(C / 4)));
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

}  // namespace pseg

using namespace pseg;

#define HP(p) reinterpret_cast<const half_t*>(p)
#define HPM(p) reinterpret_cast<half_t*>(p)

extern "C" {

int pseg_col_stats_rows(int64_t M, int C) { return cdiv(M, stat_group(M, C)); }

int pseg_col_stats_group(int64_t M, int C) { return stat_group(M, C); }

int pseg_col_stats(const float* y, int ldy, int64_t M, int C, float* stat, void* stream) {
  return col_stats_impl<float>(y, ldy, M, C, stat, stream);
}
int pseg_col_stats_h(const pseg_half_t* y, int ldy, int64_t M, int C, float* stat, void* stream) {
  return col_stats_impl<half_t>(HP(y), ldy, M, C, stat, stream);
}

constexpr int kMergePer = 64;      // groups merged per stage-A block
constexpr int kTwoStageRows = 2048;  // use two stages above this many groups (one wide block handles 2048 in 4 batches)

int64_t pseg_bn_finalize_workspace_bytes(int rows, int C) {
  return rows > kTwoStageRows ? (int64_t)3 * cdiv(rows, kMergePer) * C * 4 : 0;
}

int pseg_bn_finalize(const float* stat, int rows, int group, int64_t count, int C, const float* gamma,
                     const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* mean,
                     float* invstd, float* scale, float* shift, void* workspace, int64_t workspace_bytes, void* stream) {
  PSEG_REQUIRE(stat && mean && invstd && scale && shift, "bn_finalize: null pointer");
  PSEG_REQUIRE(rows > 0 && group > 0 && count > 0 && C > 0, "bn_finalize: bad sizes");
  PSEG_REQUIRE((long long)rows * group >= count, "bn_finalize: %d groups of %d rows do not cover %lld rows", rows, group,
               (long long)count);
  if (rows > kTwoStageRows) {
    const int64_t need = pseg_bn_finalize_workspace_bytes(rows, C);
    if (!workspace || workspace_bytes < need) {
      set_error("bn_finalize: needs %lld workspace bytes, got %lld", (long long)need, (long long)workspace_bytes);
      return PSEG_ERR_WORKSPACE;
    }
    const int out_rows = cdiv(rows, kMergePer);
    hipLaunchKernelGGL(stat_merge_kernel, dim3(cdiv(C, kFinCh), out_rows), dim3(256), 0, (hipStream_t)stream, stat, rows,
                       group, (long long)count, C, kMergePer, (float*)workspace, out_rows);
    PSEG_LAUNCH_CHECK();
    stat = (const float*)workspace;
    rows = out_rows;
    group *= kMergePer;
  }
  if (rows >= kFinWideRows)
    hipLaunchKernelGGL(bn_finalize_kernel<kFinLanesWide>, dim3(cdiv(C, kFinCh)), dim3(kFinCh * kFinLanesWide), 0,
                       (hipStream_t)stream, stat, rows, group, (long long)count, C, gamma, beta, running_mean, running_var,
                       momentum, eps, mean, invstd, scale, shift);
  else
    hipLaunchKernelGGL(bn_finalize_kernel<kFinLanes>, dim3(cdiv(C, kFinCh)), dim3(kFinCh * kFinLanes), 0,
                       (hipStream_t)stream, stat, rows, group, (long long)count, C, gamma, beta, running_mean, running_var,
                       momentum, eps, mean, invstd, scale, shift);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

int pseg_bn_small_path(int rows, int64_t M, int C) {
  // fused finalize + apply (one launch) pays while the per-block coefficient recomputation is cheap and the apply grid
  // is small anyway: few partial rows and at most a few MB of activations
  static const int mode = [] {
    const char* e = getenv("PSEG_BN_SMALL");     // 0: never, 1: as below (default)
    return e ? atoi(e) : 1;
  }();
  if (mode == 0) return 0;
  return (rows > 0 && rows <= kSmallMaxPartials && M * (int64_t)C <= (int64_t)(8 << 20)) ? 1 : 0;
}

int pseg_bn_fwd_fused(const float* stat, int rows, int group, int64_t count, int C, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                      float* scale, float* shift, const float* y, int ldy, const float* residual, int ldr, int act,
                      float* z, int ldz, int64_t M, float* amax_z, void* stream) {
  return bn_fwd_fused_impl<float>(stat, rows, group, count, C, gamma, beta, running_mean, running_var, momentum, eps, mean,
                                  invstd, scale, shift, y, ldy, residual, ldr, act, z, ldz, M, amax_z, stream);
}
int pseg_bn_fwd_fused_h(const float* stat, int rows, int group, int64_t count, int C, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                        float* scale, float* shift, const pseg_half_t* y, int ldy, const pseg_half_t* residual, int ldr,
                        int act, pseg_half_t* z, int ldz, int64_t M, void* stream) {
  return bn_fwd_fused_impl<half_t>(stat, rows, group, count, C, gamma, beta, running_mean, running_var, momentum, eps, mean,
                                   invstd, scale, shift, HP(y), ldy, HP(residual), ldr, act, HPM(z), ldz, M, nullptr, stream);
}

int pseg_bn_bwd_fused(const float* part_db, const float* part_dg, int rows, int64_t count, int C, float* dgamma,
                      float* dbeta, int accumulate, int frozen, const float* dz, int lddz, const float* z, int ldz,
                      const float* y, int ldy, const float* mean, const float* invstd, const float* scale,
                      const float* shift, int act, float* dy, int lddy, float* dres, int lddres, int res_accumulate,
                      int64_t M, void* stream) {
  return bn_bwd_fused_impl<float>(part_db, part_dg, rows, count, C, dgamma, dbeta, accumulate, frozen, dz, lddz, z, ldz, y,
                                  ldy, mean, invstd, scale, shift, act, dy, lddy, dres, lddres, res_accumulate, M, stream);
}
int pseg_bn_bwd_fused_h(const float* part_db, const float* part_dg, int rows, int64_t count, int C, float* dgamma,
                        float* dbeta, int accumulate, int frozen, const pseg_half_t* dz, int lddz, const pseg_half_t* z,
                        int ldz, const pseg_half_t* y, int ldy, const float* mean, const float* invstd, const float* scale,
                        const float* shift, int act, pseg_half_t* dy, int lddy, pseg_half_t* dres, int lddres,
                        int res_accumulate, int64_t M, void* stream) {
  return bn_bwd_fused_impl<half_t>(part_db, part_dg, rows, count, C, dgamma, dbeta, accumulate, frozen, HP(dz), lddz, HP(z),
                                   ldz, HP(y), ldy, mean, invstd, scale, shift, act, HPM(dy), lddy, HPM(dres), lddres,
                                   res_accumulate, M, stream);
}

int pseg_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                        float eps, int C, float* mean, float* invstd, float* scale, float* shift, void* stream) {
  PSEG_REQUIRE(running_mean && running_var && mean && invstd && scale && shift && C > 0, "bn_eval_coeffs: bad argument");
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, eps, C, mean, invstd, scale, shift);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

int pseg_bn_act_fwd(const float* y, int ldy, const float* mean, const float* scale, const float* shift,
                    const float* residual, int ldr, int act, float* z, int ldz, int64_t M, int C, float* amax_z,
                    uint32_t* mask_out, void* stream) {
  return bn_act_fwd_impl<float>(y, ldy, mean, scale, shift, residual, ldr, act, z, ldz, M, C, amax_z, mask_out, stream);
}
int pseg_bn_act_fwd_h(const pseg_half_t* y, int ldy, const float* mean, const float* scale, const float* shift,
                      const pseg_half_t* residual, int ldr, int act, pseg_half_t* z, int ldz, int64_t M, int C,
                      uint32_t* mask_out, void* stream) {
  return bn_act_fwd_impl<half_t>(HP(y), ldy, mean, scale, shift, HP(residual), ldr, act, HPM(z), ldz, M, C, nullptr,
                                 mask_out, stream);
}

int pseg_bn_act_bwd_reduce(const float* dz, int lddz, const float* z, int ldz, const float* y, int ldy, const float* mean,
                           const float* invstd, const float* scale, const float* shift, int act, int64_t M, int C,
                           float* part_db, float* part_dg, const uint32_t* mask, void* stream) {
  return bn_act_bwd_reduce_impl<float>(dz, lddz, z, ldz, y, ldy, mean, invstd, scale, shift, act, M, C, part_db, part_dg,
                                       mask, stream);
}
int pseg_bn_act_bwd_reduce_h(const pseg_half_t* dz, int lddz, const pseg_half_t* z, int ldz, const pseg_half_t* y, int ldy,
                             const float* mean, const float* invstd, const float* scale, const float* shift, int act,
                             int64_t M, int C, float* part_db, float* part_dg, const uint32_t* mask, void* stream) {
  return bn_act_bwd_reduce_impl<half_t>(HP(dz), lddz, HP(z), ldz, HP(y), ldy, mean, invstd, scale, shift, act, M, C, part_db,
                                        part_dg, mask, stream);
}

int pseg_bn_bwd_finalize(const float* part_db, const float* part_dg, int rows, int64_t count, int C, float* dgamma,
                         float* dbeta, int accumulate, int frozen, float* c1, float* c2, void* stream) {
  PSEG_REQUIRE(part_db && part_dg && c1 && c2 && rows > 0 && count > 0 && C > 0, "bn_bwd_finalize: bad argument");
  if (rows >= kFinWideRows)
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<kFinLanesWide>, dim3(cdiv(C, kFinCh)), dim3(kFinCh * kFinLanesWide), 0,
                       (hipStream_t)stream, part_db, part_dg, rows, (long long)count, C, dgamma, dbeta, accumulate, frozen, c1,
                       c2);
  else
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<kFinLanes>, dim3(cdiv(C, kFinCh)), dim3(kFinCh * kFinLanes), 0,
                       (hipStream_t)stream, part_db, part_dg, rows, (long long)count, C, dgamma, dbeta, accumulate, frozen, c1,
                       c2);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

int pseg_bn_act_bwd_apply(const float* dz, int lddz, const float* z, int ldz, const float* y, int ldy, const float* mean,
                          const float* invstd, const float* scale, const float* shift, const float* c1, const float* c2,
                          int act, float* dy, int lddy, float* dres, int lddres, int res_accumulate, int64_t M, int C,
                          const uint32_t* mask, uint16_t* dy_hi, uint16_t* dy_lo, int ldp, void* stream) {
  return bn_act_bwd_apply_impl<float>(dz, lddz, z, ldz, y, ldy, mean, invstd, scale, shift, c1, c2, act, dy, lddy, dres,
                                      lddres, res_accumulate, M, C, mask, dy_hi, dy_lo, ldp, stream);
}
int pseg_bn_act_bwd_apply_h(const pseg_half_t* dz, int lddz, const pseg_half_t* z, int ldz, const pseg_half_t* y, int ldy,
                            const float* mean, const float* invstd, const float* scale, const float* shift, const float* c1,
                            const float* c2, int act, pseg_half_t* dy, int lddy, pseg_half_t* dres, int lddres,
                            int res_accumulate, int64_t M, int C, const uint32_t* mask, void* stream) {
  return bn_act_bwd_apply_impl<half_t>(HP(dz), lddz, HP(z), ldz, HP(y), ldy, mean, invstd, scale, shift, c1, c2, act, HPM(dy),
                                       lddy, HPM(dres), lddres, res_accumulate, M, C, mask, nullptr, nullptr, 0, stream);
}

int pseg_act_bwd(const float* dz, int lddz, const float* z, int ldz, const float* scale, int act, float* dy, int lddy,
                 float* dres, int lddres, int res_accumulate, int64_t M, int C, void* stream) {
  return act_bwd_impl<float>(dz, lddz, z, ldz, scale, act, dy, lddy, dres, lddres, res_accumulate, M, C, stream);
}
int pseg_act_bwd_h(const pseg_half_t* dz, int lddz, const pseg_half_t* z, int ldz, const float* scale, int act,
                   pseg_half_t* dy, int lddy, pseg_half_t* dres, int lddres, int res_accumulate, int64_t M, int C,
                   void* stream) {
  return act_bwd_impl<half_t>(HP(dz), lddz, HP(z), ldz, scale, act, HPM(dy), lddy, HPM(dres), lddres, res_accumulate, M, C,
                              stream);
}

int pseg_col_sum(const float* dy, int ldy, int64_t M, int C, float* out, int accumulate, void* workspace,
                 int64_t workspace_bytes, void* stream) {
  return col_sum_impl<float>(dy, ldy, M, C, out, accumulate, workspace, workspace_bytes, stream);
}
int pseg_col_sum_h(const pseg_half_t* dy, int ldy, int64_t M, int C, float* out, int accumulate, void* workspace,
                   int64_t workspace_bytes, void* stream) {
  return col_sum_impl<half_t>(HP(dy), ldy, M, C, out, accumulate, workspace, workspace_bytes, stream);
}

int pseg_copy2d(const float* x, int ldx, float* y, int ldy, int64_t M, int C, int accumulate, void* stream) {
  return copy2d_impl<float>(x, ldx, y, ldy, M, C, accumulate, stream);
}
int pseg_copy2d_h(const pseg_half_t* x, int ldx, pseg_half_t* y, int ldy, int64_t M, int C, int accumulate, void* stream) {
  return copy2d_impl<half_t>(HP(x), ldx, HPM(y), ldy, M, C, accumulate, stream);
}

}  // extern "C"
