// Shared by the conv translation units (conv_mfma.hip: fp32 / limb arithmetic; conv_half.hip: fp16 storage + single-pass
// fp16 MFMA): the parameter blocks of the gather / weight-gradient kernels, GEMM-row orders (patch / parity / liveness-class
// sorted), XCD-aware tile order, the LDS-staged epilogue, and the host-side planners (tile choice, split choice, dilated-conv
// tap-skipping schedules).  Everything here is `static` / `inline`: each translation unit gets its own copy of the code, the
// planning overrides read from the environment (cfg()) are shared (inline variables).
#pragma once
#include "common.h"

#include <stdio.h>
#include <stdlib.h>

namespace pseg {

constexpr int BK = 32;        // K-step (floats): 8*TM*TN MFMAs of 64 cycles per wave between two barriers
constexpr int LDT = BK + 4;   // LDS row stride of the K-contiguous images (144 B = 9 x 16-B slots, 9 coprime to 16)
constexpr int CPR = BK / 4;   // 16-byte chunks per K-contiguous row
constexpr int RPP = 256 / CPR; // rows covered by one pass of the 256 threads

// ------------------------------------------------------------------------------------------------
// Epilogue shared by both kernels: the wave's TM x TN accumulator tiles go through a wave-private LDS patch
// ([WTM][WTN+4] floats) so that global memory is touched in 16-byte, row-contiguous accesses (a half-wave covers
// 256 contiguous bytes of one output row) instead of 4 bytes per lane.  The memory-bound launches -- 1x1 convs with
// few input channels, and every dgrad that ACCUMULATES into dx (residual merges: read + write of the whole tensor)
// -- were running at 1.5 TB/s with the per-lane form.
// IL (interleaved tiles, fp32 weight gradient): MFMA tile (i, j) holds wave-local rows TM*t + i and columns TN*c + j
// instead of rows 32*i + t / columns 32*j + c (its operands then come out of LDS TM / TN at a time).
// BNS (data gradient feeding a BatchNorm backward, see GatherConvParams::bns_y): while the rows go out, the same lanes read the
// producing layer's y at the same addresses (16 bytes per lane, coalesced like the stores) and keep two running sums per
// column; the lanes that share a column chunk are folded at the end (fixed order) and lane rr == 0 writes the wave's partial.
// which kernel the calling thread's last convolution entry point enqueued (pseg_debug_last_conv_kernel; codes: pseg_amd.h)
extern thread_local int g_last_conv_kernel;

struct BnsEpilogue {
  const float* y;
  long long ldy;
  const float* mean;
  const float* invstd;
  const float* scale;
  const float* shift;
  int act;
  float* db;       // [groups][N]
  float* dg;
  long long out_off;   // group * N
};

template <int TM, int TN, bool IL = false, bool BNS = false, typename RowMap>
__device__ __forceinline__ void store_tiles(const f32x16 (&acc)[TM][TN], float* patch, float* out, long long ld,
                                            int row0, int col0, int rows_valid, int cols_valid,
                                            const float* bias, bool accumulate, int lane, RowMap&& out_row,
                                            const BnsEpilogue* bns = nullptr) {
  constexpr int WTM = TM * 32, WTN = TN * 32, LDW = WTN + 4;
  const int col_l = lane & 31;
  const int row_h = (lane >> 5) * 4;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int tr = (r & 3) + 8 * (r >> 2) + row_h;
        if constexpr (IL) patch[(TM * tr + i) * LDW + TN * col_l + j] = acc[i][j][r];
        else patch[(i * 32 + tr) * LDW + j * 32 + col_l] = acc[i][j][r];
      }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  constexpr int C4 = WTN / 4;          // 16-byte chunks per row
  constexpr int RPI = 64 / C4;         // rows per wave-instruction
  const int c4 = lane % C4, rr = lane / C4;
  const int col = c4 * 4;
  const bool cok = col < cols_valid;   // cols_valid is a multiple of 4
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias != nullptr && cok) bv = *reinterpret_cast<const f32x4*>(bias + col0 + col);
  f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = mu, sc = mu, sh = mu, s1 = mu, s2 = mu;
  if constexpr (BNS) {
    if (cok) {
      mu = *reinterpret_cast<const f32x4*>(bns->mean + col0 + col);
      is = *reinterpret_cast<const f32x4*>(bns->invstd + col0 + col);
      sc = *reinterpret_cast<const f32x4*>(bns->scale + col0 + col);
      sh = *reinterpret_cast<const f32x4*>(bns->shift + col0 + col);
    }
  }
#pragma unroll
  for (int it = 0; it < WTM / RPI; ++it) {
    const int row = it * RPI + rr;
    if (cok && row < rows_valid) {
      f32x4 v = *reinterpret_cast<const f32x4*>(&patch[row * LDW + col]) + bv;
      const long long orow = (long long)out_row(row0 + row);
      float* gp = out + orow * ld + col0 + col;
      if (accumulate) v += *reinterpret_cast<const f32x4*>(gp);
      *reinterpret_cast<f32x4*>(gp) = v;
      if constexpr (BNS) {
        const f32x4 yv = *reinterpret_cast<const f32x4*>(bns->y + orow * bns->ldy + col0 + col);
        const f32x4 d = yv - mu;
        const f32x4 pre = d * sc + sh;          // the forward pass's own expression: the mask is the one it applied
        f32x4 g = v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          bool on = true;
          if (bns->act == PSEG_ACT_RELU) on = pre[e] > 0.f;
          else if (bns->act == PSEG_ACT_RELU6) on = (pre[e] > 0.f) && (pre[e] < 6.f);
          g[e] = on ? g[e] : 0.f;
        }
        s1 += g;
        s2 += g * (d * is);
      }
    }
  }
  if constexpr (BNS) {
#pragma unroll
    for (int o = C4; o < 64; o <<= 1)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s1[e] += __shfl_xor(s1[e], o, 64);
        s2[e] += __shfl_xor(s2[e], o, 64);
      }
    if (rr == 0 && cok) {
      *reinterpret_cast<f32x4*>(bns->db + bns->out_off + col0 + col) = s1;
      *reinterpret_cast<f32x4*>(bns->dg + bns->out_off + col0 + col) = s2;
    }
  }
}

// The same epilogue out of a patch of EIGHT rows per wave ([8][WTN + 4] floats, 1.2 KB for 32 columns), one pass per eight
// rows of the wave tile: accumulator registers 4q .. 4q + 3 of a 32-row MFMA tile are its rows 8q .. 8q + 7.  For kernels whose
// LDS is the operand ring and must stay that while a tile is stored (the persistent pointwise kernel: the next tile's operands
// are landing).  The patch is private to the wave and LDS executes a wave's instructions in order: no fence -- which would also
// wait for the operand DMAs in flight -- only the counted wait for the pass's own writes.
template <int TM, int TN, bool BNS = false, typename RowMap>
__device__ __forceinline__ void store_tiles_rows8(const f32x16 (&acc)[TM][TN], float* patch8, float* out, long long ld, int row0,
                                                  int col0, int rows_valid, int cols_valid, const float* bias, bool accumulate,
                                                  int lane, RowMap&& out_row, const BnsEpilogue* bns = nullptr) {
  constexpr int WTN = TN * 32, LDW = WTN + 4;
  constexpr int C4 = WTN / 4;          // 16-byte chunks per row
  constexpr int RPI = 64 / C4;         // rows per wave-instruction of the read-back (8 for 32 columns, 4 for 64)
  static_assert(RPI >= 1 && 8 % RPI == 0, "patch rows per read");
  const int col_l = lane & 31, hh = lane >> 5;
  const int c4 = lane % C4, rr = lane / C4;
  const int col = c4 * 4;
  const bool cok = col < cols_valid;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias != nullptr && cok) bv = *reinterpret_cast<const f32x4*>(bias + col0 + col);
  f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = mu, sc = mu, sh = mu, s1 = mu, s2 = mu;
  if constexpr (BNS) {
    if (cok) {
      mu = *reinterpret_cast<const f32x4*>(bns->mean + col0 + col);
      is = *reinterpret_cast<const f32x4*>(bns->invstd + col0 + col);
      sc = *reinterpret_cast<const f32x4*>(bns->scale + col0 + col);
      sh = *reinterpret_cast<const f32x4*>(bns->shift + col0 + col);
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) patch8[(e + 4 * hh) * LDW + j * 32 + col_l] = acc[i][j][4 * q + e];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 8 / RPI; ++it) {
        const int prow = it * RPI + rr;                     // row of the patch
        const int row = i * 32 + 8 * q + prow;              // row of the wave tile
        if (cok && row < rows_valid) {
          f32x4 v = *reinterpret_cast<const f32x4*>(&patch8[prow * LDW + col]) + bv;
          const long long orow = (long long)out_row(row0 + row);
          float* gp = out + orow * ld + col0 + col;
          if (accumulate) v += *reinterpret_cast<const f32x4*>(gp);
          *reinterpret_cast<f32x4*>(gp) = v;
          if constexpr (BNS) {
            const f32x4 yv = *reinterpret_cast<const f32x4*>(bns->y + orow * bns->ldy + col0 + col);
            const f32x4 d = yv - mu;
            const f32x4 pre = d * sc + sh;
            f32x4 g = v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              bool on = true;
              if (bns->act == PSEG_ACT_RELU) on = pre[e] > 0.f;
              else if (bns->act == PSEG_ACT_RELU6) on = (pre[e] > 0.f) && (pre[e] < 6.f);
              g[e] = on ? g[e] : 0.f;
            }
            s1 += g;
            s2 += g * (d * is);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the patch is rewritten by the next pass
      __builtin_amdgcn_wave_barrier();
    }
  if constexpr (BNS) {
#pragma unroll
    for (int o = C4; o < 64; o <<= 1)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s1[e] += __shfl_xor(s1[e], o, 64);
        s2[e] += __shfl_xor(s2[e], o, 64);
      }
    if (rr == 0 && cok) {
      *reinterpret_cast<f32x4*>(bns->db + bns->out_off + col0 + col) = s1;
      *reinterpret_cast<f32x4*>(bns->dg + bns->out_off + col0 + col) = s2;
    }
  }
}

// Block id -> tile id.  The dispatcher deals consecutive workgroups round-robin over the 8 XCDs (each with its own L2) and,
// inside an XCD, over its 32 CUs: block i of a launch starts on XCD i % 8, CU slot (i / 8) % 32, and blocks i, i + 256,
// i + 512 share a CU while they all fit (probed: tools/micro/dispatch_probe.hip).
// mode 1: XCD x works on one CONTIGUOUS range of tiles -- the column tiles that share a gathered A tile, and neighbouring
//         row tiles that share halo pixels, hit the same L2.
// mode 2: the same inside every round of 256 blocks (XCD x takes tiles [256 r + 32 x, + 32) of round r), so that the CU
//         which got tile t of round 0 gets tile t + 256 of round 1: with tiles sorted by descending cost (class-sorted
//         dilated convs) every CU pairs an expensive tile with a cheap one.
__host__ __device__ __forceinline__ int remap_tile(int mode, int bid, int nblocks) {
  if (mode == 1) {
    const int full = (nblocks / 8) * 8;
    return bid < full ? (bid & 7) * (full >> 3) + (bid >> 3) : bid;
  }
  if (mode == 2) {
    const int full = (nblocks / 256) * 256;
    return bid < full ? (bid & ~255) + (bid & 7) * 32 + ((bid & 255) >> 3) : bid;
  }
  return bid;
}

// ------------------------------------------------------------------------------------------------
// row_perm == 4: GEMM rows sorted by LIVENESS CLASS.  The taps of a dilated conv cut each axis of the map into at most
// three bands (rate 12 on 32 rows: [0,12) sees taps {0,+}, [12,20) all three, [20,32) {-,0}); a class is a (row band,
// column band) rectangle, inside which every pixel has the same set of in-range taps.  Rows run class by class, inside
// a class image by image, inside an image in raster order of the rectangle: an M tile that lies inside one class
// executes exactly the taps that are live for its pixels -- no padding is multiplied at all (only the few tiles that
// straddle a class boundary run the union of two sets).
struct BandMap {
  int start[10];                       // first GEMM row of class c (start[9] = M); empty classes have equal starts
  int h0[9], w0[9], cw[9], area[9];    // top-left pixel, width and pixel count of the class rectangle
};

struct GatherConvParams {
  const float* x;
  const float* w;
  float* y;
  const float* bias;
  float* stat;     // [3][stat_rows][N] shifted column statistics, or null
  int stat_rows;
  uint32_t x_bytes, w_bytes;
  int ldx, ldy;
  int Hi, Wi, Cin;   // gather source
  int Ho, Wo, HoWo;  // GEMM row space
  int M, N, K;
  int kw;
  int s_out, s_in, dstep, off0;
  int accumulate;
  int kt_total, kt_per_split;
  long long slab_stride;  // elements between split-K slabs (0 when gridDim.z == 1)
  int skip_taps;          // dilated convs: skip the K-steps of taps that are zero padding for the whole M tile
  int ntaps, ktiles_per_tap;
  int xcd_remap;          // tile order: contiguous tile ranges per XCD (see the kernel)
  int prio;               // wave priority (s_setprio) of this launch: the data gradients sit on the step's critical path
  int row_perm;           // 3: pointwise conv, rows ARE pixels (no index arithmetic; the host passes a 1 x M image);
                          // 1: stride-2 dgrad, GEMM rows ordered (b, parity class, h/2, w/2) -> parity-homogeneous tiles
                          // 2: dilated convs, GEMM rows ordered in patch_h x patch_w pixel patches (one M tile = one patch)
                          // 4: dilated convs, GEMM rows sorted by liveness class (BandMap)
  int patch_w, patch_hw, patches_per_row;   // row_perm == 2
  BandMap band;                             // row_perm == 4
  unsigned long long* trace;                // debug (PSEG_CONV_TRACE): 4 timestamps per block of gather_f32_dma_kernel
  int precision;          // 0 exact fp32 MFMA, 1/2 split-bf16 (3/6 products), 3 split-fp16 (3 products, scaled)
  const unsigned* amax_a;  // PREC 3: per-tensor max|x| bit patterns of the gathered tensor and of the filter
  const unsigned* amax_b;
  // pre-split bf16 limb planes (gather_limb_dma_kernel): hi / lo images of the gathered tensor [pixels][ldxp] and of the
  // filter [N][K], 2 bytes per element
  const uint16_t* xh;
  const uint16_t* xl;
  const uint16_t* wh;
  const uint16_t* wl;
  uint32_t xp_bytes, wp_bytes;
  int ldxp;
  // Data gradient with the BatchNorm-backward partial sums of the layer that PRODUCED the conv's input (bns_y != null; exact
  // fp32 LDS-DMA kernel, no accumulate): the tile this block writes is dz of that layer -- its epilogue reads the layer's saved
  // conv output y at the same pixels, recomputes the activation mask from it exactly as the forward pass did, and writes
  // sum(dz * act') and sum(dz * act' * xhat) per (row group, channel): what bn_bwd_reduce_kernel<MODE 3> would re-read dz and y
  // for.  Row group = (M tile, wave row): bns_rows = gridM * waves_m.
  const float* bns_y;
  int bns_ldy;
  const float* bns_mean;
  const float* bns_invstd;
  const float* bns_scale;
  const float* bns_shift;
  int bns_act;
  float* bns_db;
  float* bns_dg;
  // GENERIC form of the exact-fp32 LDS-DMA kernel (channels % 32 != 0): 16-byte k-slots per tap, taps per filter row
  FastDiv gen_spt, gen_kw;
};

// GEMM row -> output pixel index.  Identity normally.  With row_perm (Ho, Wo even) row m = ((b*4 + cls)*H2 + h2)*W2 + w2
// maps to pixel (b, 2*h2 + cls/2, 2*w2 + cls%2): every 128-row tile then holds pixels of ONE parity class, and for a
// stride-2 data gradient only the taps whose parity matches that class can ever be in range, so the tap-skipping
// variant drops the other 3/4 of the K-steps instead of multiplying zeros.
// With row_perm == 2 (Ho % patch_h == 0, Wo % patch_w == 0) the rows of one image run patch by patch: a 64- / 128- /
// 256-row M tile is then a patch_h x patch_w rectangle of pixels instead of a few full image rows, so a dilated tap is
// dead for the whole tile when EITHER its rows or its columns fall into the zero padding (rate 18 on a 32x32 map:
// 44 % of the (tile, tap) pairs stay live with 4x16 / 8x16 patches against 67 % with full rows).
__device__ __forceinline__ void row_to_pixel(const GatherConvParams& p, int m, int& b, int& ho, int& wo) {
  if (p.row_perm == 4) {
    int st = 0, h0 = p.band.h0[0], w0 = p.band.w0[0], cw = p.band.cw[0], area = p.band.area[0];
#pragma unroll
    for (int c = 1; c < 9; ++c) {
      const bool in = m >= p.band.start[c];
      st = in ? p.band.start[c] : st;
      h0 = in ? p.band.h0[c] : h0;
      w0 = in ? p.band.w0[c] : w0;
      cw = in ? p.band.cw[c] : cw;
      area = in ? p.band.area[c] : area;
    }
    const int rem = m - st;
    b = rem / area;
    const int r2 = rem - b * area;
    const int y = r2 / cw;
    ho = h0 + y;
    wo = w0 + (r2 - y * cw);
  } else if (p.row_perm == 3) {   // pointwise (1x1, unit stride, no padding): the tensor is one long row of M pixels
    b = 0;
    ho = 0;
    wo = m;
  } else if (p.row_perm == 2) {
    b = m / p.HoWo;
    const int rem = m - b * p.HoWo;
    const int patch = rem / p.patch_hw;
    const int in = rem - patch * p.patch_hw;
    const int ph = patch / p.patches_per_row, pw = patch - ph * p.patches_per_row;
    const int ih = in / p.patch_w, iw = in - ih * p.patch_w;
    ho = ph * (p.patch_hw / p.patch_w) + ih;
    wo = pw * p.patch_w + iw;
  } else if (p.row_perm) {
    const int W2 = p.Wo >> 1, H2 = p.Ho >> 1;
    const int q = H2 * W2;
    b = m / (4 * q);
    int rem = m - b * 4 * q;
    const int cls = rem / q;
    rem -= cls * q;
    const int h2 = rem / W2;
    const int w2 = rem - h2 * W2;
    ho = 2 * h2 + (cls >> 1);
    wo = 2 * w2 + (cls & 1);
  } else {
    b = m / p.HoWo;
    const int rem = m - b * p.HoWo;
    ho = rem / p.Wo;
    wo = rem - ho * p.Wo;
  }
}

// Wave priority of a launch (0..3).  The instruction arbiter of a SIMD favours the OLDEST wave; a kernel that starts beside
// a resident kernel of another stream is the youngest everywhere.  s_setprio takes an immediate.
__device__ __forceinline__ void set_wave_prio(int prio) {
#if defined(PSEG_NO_PRIO) && PSEG_NO_PRIO
  return;
#endif
  if (prio == 3) __builtin_amdgcn_s_setprio(3);
  else if (prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (prio == 1) __builtin_amdgcn_s_setprio(1);
}

// ------------------------------------------------------------------------------------------------
struct WgradParams {
  const float* x;
  const float* dy;
  float* dw;
  uint32_t x_bytes, dy_bytes;
  int ldx, ldy;
  int Hi, Wi, Cin;
  int Ho, Wo, HoWo;
  int Cout, K, P;
  int kw;
  int stride, pad, dil;
  int pix_per_split;
  int accumulate;
  long long slab_stride;
  int skip_rows;  // dilated convs: skip pixel K-steps that are zero padding for this block's tap.
                  // 1: pixels in row-major order, a K-step (32 pixels) is dead when all its image ROWS are out of range;
                  // 2: pixels in patch order -- K-step q is the patch_h x patch_w (= 32 pixels, patch_w >= 8) patch q of
                  //    the map, dead when its rows OR its columns are out of range for the tap (unit stride, P % 32 == 0)
  int patch_mode;     // pixels of the contraction run in patch order (set whenever the map tiles into such patches, also
                      // without skipping: K-step addresses are then a block-uniform origin + a thread-constant offset)
  int patch_h, patch_w;
  FastDiv ppr, ppi;   // patches per patch-row (Wo / patch_w) and per image
  // skip_rows == 3 (exact-fp32 LDS-DMA kernel, unit stride, a column tile inside one tap): PACKED contraction.  The pixels a
  // block multiplies are exactly the live rectangle of its tap -- output rows [r0, r1) x columns [c0, c1) for which the tap's
  // source pixel lies inside the image -- of every image, numbered densely; K-step s of the block holds packed pixels
  // [32 s, 32 s + 32), each lane derives its pixel by two divisions per DMA piece.  No K-step multiplies padding (a 32-pixel
  // patch that straddles the rectangle's edge did: rate 6 on a 32x32 map ran 0.92 of the dense work for 0.77 live).  The pixel
  // split divides each block's OWN step count (pix_per_split is unused).
  // lpt_per > 0: column tiles in LONGEST-FIRST order inside every group of taps x lpt_per tiles (one group = the range one XCD's
  // blocks walk in order): tile position -> (group, rank, c) -> tap tap_order[rank], column tile group * lpt_per + c of that tap.
  int lpt_per;
  int tap_order[9];
  FastDiv rm_howo, rm_wo;   // skip_rows == 4 (row-major pixels on the LDS-DMA kernel): pixel -> (image, row, column)
};

// patch mode: image index and top-left pixel of K-step `pt` (a multiple of 32)
__device__ __forceinline__ void wg_patch_origin(const WgradParams& p, int pt, int& b, int& h0, int& w0) {
  const uint32_t q = (uint32_t)pt >> 5;
  const uint32_t bb = p.ppi.div(q);
  const uint32_t r = q - bb * p.ppi.d;
  const uint32_t ph = p.ppr.div(r);
  b = (int)bb;
  h0 = (int)ph * p.patch_h;
  w0 = (int)(r - ph * p.ppr.d) * p.patch_w;
}

// is K-step `pt` pure padding for the tap at offset (t_dh, t_dw)?  (block-uniform)
__device__ __forceinline__ bool wg_step_dead(const WgradParams& p, int pt, int p_end, int t_dh, int t_dw) {
  if (p.skip_rows == 2) {
    int b, h0, w0;
    wg_patch_origin(p, pt, b, h0, w0);
    return ((h0 + p.patch_h - 1) * p.stride + t_dh < 0) || (h0 * p.stride + t_dh >= p.Hi) ||
           ((w0 + p.patch_w - 1) * p.stride + t_dw < 0) || (w0 * p.stride + t_dw >= p.Wi);
  }
  int pl = pt + BK;
  if (pl > p_end) pl = p_end;
  pl -= 1;
  const int bf = pt / p.HoWo, bl = pl / p.HoWo;
  if (bf != bl) return false;
  const int hf = (pt - bf * p.HoWo) / p.Wo, hl = (pl - bl * p.HoWo) / p.Wo;
  return (hl * p.stride + t_dh < 0) || (hf * p.stride + t_dh >= p.Hi);
}

// Weight-gradient blocks of one pixel split (blockIdx.z) all read the same slab of dy / x pixels, each a different
// (Cout tile, K tile).  The dispatcher deals consecutive workgroups round-robin over the 8 XCDs, so by default every
// XCD's L2 ends up fetching every slab.  Remap the linear id so that XCD x owns a contiguous range of (split, tile)
// pairs: all tiles of a split then share one L2 and the slab crosses the fabric once instead of up to eight times
// (measured before: 26.8 GB of L2-fabric reads per step for 8.4 GB of operands).
__device__ __forceinline__ void wgrad_block(int tiles, int& tile, int& split) {
  const long long total = (long long)gridDim.x * gridDim.z;
  long long lid = (long long)blockIdx.z * gridDim.x + blockIdx.x;
  const long long full = (total / 8) * 8;
  if (lid < full) lid = (lid & 7) * (full >> 3) + (lid >> 3);
  split = (int)(lid / tiles);
  tile = (int)(lid - (long long)split * tiles);
}

// ------------------------------------------------------------------------------------------------ host side
struct TileCfg {
  int bm, bn;
};

static TileCfg pick_tile(long long rows, long long cols) {
  TileCfg c;
  c.bn = cols > 64 ? 128 : (cols > 32 ? 64 : 32);
  c.bm = rows > 64 ? 128 : (rows > 32 ? 64 : 32);
  if (c.bm == 32 && c.bn == 32) c.bm = 64;  // smallest instantiated block is 64x32 / 32x64
  if (c.bm == 64 && c.bn == 64) c.bm = 128; // 64x64 not instantiated
  if (c.bm == 32 && c.bn == 64) c.bn = 128; // ditto
  if (c.bm == 64 && c.bn == 32) c.bm = 128;
  return c;
}

// Planning overrides (testing / tuning knobs).  Read from the environment ONCE (first launch) -- getenv is a linear scan
// of the process environment and used to run 6-10 times per conv launch; pseg_config_reload() re-reads them (the tests
// that change PSEG_* at run time call it through _lib.clear_query_cache()).
struct EnvCfg {
  int conv_nobig, conv_forcebig, conv_bm, conv_bn, conv_splitk, conv_noskip, conv_noband, plan_debug, wgrad_bpc, conv_dma32, conv_narrow, conv_noxcd, conv_nodma, conv_f32dma, wgrad_f32dma, conv_big, dgrad_prio;
  int wgrad_big, wgrad_bm, wgrad_bn, wgrad_splits;
  int hconv_persist, hconv_persist_kt, hconv_tile;
  int conv_pw, conv_pw_kt, conv_pw_resident, conv_halo, wgrad_halo;
};
inline EnvCfg g_cfg;
inline volatile int g_cfg_ready = 0;
static int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return s ? atoi(s) : dflt;
}
inline void cfg_load() {
  EnvCfg c;
  c.conv_nobig = env_int("PSEG_CONV_NOBIG", 0);
  c.conv_forcebig = env_int("PSEG_CONV_FORCEBIG", 0);
  c.conv_bm = env_int("PSEG_CONV_BM", 0);
  c.conv_bn = env_int("PSEG_CONV_BN", 0);
  c.conv_splitk = env_int("PSEG_CONV_SPLITK", 0);
  c.conv_noskip = env_int("PSEG_CONV_NOSKIP", 0);
  c.conv_noband = env_int("PSEG_CONV_NOBAND", 0);
  c.conv_narrow = env_int("PSEG_CONV_NARROW", 1);
  c.dgrad_prio = env_int("PSEG_DGRAD_PRIO", 1);
  c.conv_dma32 = env_int("PSEG_CONV_DMA32", 1);
  c.wgrad_bpc = env_int("PSEG_WGRAD_BPC", 0);
  c.plan_debug = env_int("PSEG_PLAN_DEBUG", 0);
  c.conv_noxcd = env_int("PSEG_CONV_NOXCD", 0);
  c.conv_nodma = env_int("PSEG_CONV_NODMA", 0);
  c.conv_f32dma = env_int("PSEG_CONV_F32DMA", 3);
  c.wgrad_f32dma = env_int("PSEG_WGRAD_F32DMA", 1);
  c.conv_big = env_int("PSEG_CONV_BIG", 0);
  c.wgrad_big = env_int("PSEG_WGRAD_BIG", 0);
  c.wgrad_bm = env_int("PSEG_WGRAD_BM", 0);
  c.wgrad_bn = env_int("PSEG_WGRAD_BN", 0);
  c.wgrad_splits = env_int("PSEG_WGRAD_SPLITS", 0);
  c.hconv_persist = env_int("PSEG_HCONV_PERSIST", 1);
  c.hconv_persist_kt = env_int("PSEG_HCONV_PERSIST_KT", 24);
  c.hconv_tile = env_int("PSEG_HCONV_TILE", 0);
  c.conv_pw = env_int("PSEG_CONV_PW", 1);                   // persistent pointwise kernel of the exact-fp32 path (0: off)
  c.conv_pw_kt = env_int("PSEG_CONV_PW_KT", 32);            // ... for contractions of at most this many K-steps
  c.conv_halo = env_int("PSEG_CONV_HALO", 2);               // halo-staged narrow 3x3 of the exact-fp32 path (0: off, 1: 128x32 plan tiles, 2: 128x64 too)
  c.wgrad_halo = env_int("PSEG_WGRAD_HALO", 1);             // halo-staged weight gradient of narrow 3x3 filters (0: off)
  c.conv_pw_resident = env_int("PSEG_CONV_PW_RESIDENT", 0); // ... grid size override (tests: several tiles per block on small problems)
  g_cfg = c;                   // (racing first calls write identical values)
  __atomic_store_n(&g_cfg_ready, 1, __ATOMIC_RELEASE);
}
static inline const EnvCfg& cfg() {
  if (!__atomic_load_n(&g_cfg_ready, __ATOMIC_ACQUIRE)) cfg_load();
  return g_cfg;
}

static const long long kMaxBytes = (1LL << 31) - 64;

static long long nhwc_bytes(int B, int H, int W, int C, int ld) {
  return (((long long)B * H * W - 1) * ld + C) * 4;
}

// Split the reduction dimension into `s` slices so that tiles*s blocks spread evenly over the 256 CUs: every CU
// runs ceil(blocks/256) blocks, so the efficiency of a launch is blocks / (256 * ceil(blocks/256)) -- 288 tiles x 2
// splits = 576 blocks is only 75 % (some CUs get 3 blocks, most 2), x7 = 2016 blocks is 98 %.  Each extra slice costs
// one more slab write + read of the output, hence the small per-slice penalty.
static int pick_splits(long long tiles, long long units, long long min_units, int max_splits, int blocks_per_cu = 2,
                       double split_cost = 0.0005) {
  long long cap = units / min_units;
  if (cap < 1) cap = 1;
  if (cap > max_splits) cap = max_splits;
  const double ncu = 256.0;
  int best = 1;
  double best_score = -1.0;
  for (long long s = 1; s <= cap; ++s) {
    const double blocks = (double)(tiles * s);
    const double rounds = (double)((tiles * s + 255) / 256);
    // the slab cost of a split is only worth arguing about once the chip is full: below `want` resident blocks every
    // extra split is parallelism the launch does not have otherwise
    const long long s_fill = (long long)((blocks_per_cu * ncu + (double)tiles - 1.0) / (double)tiles);
    const double extra = s > s_fill ? (double)(s - s_fill) : 0.0;
    double score = blocks / (ncu * rounds) - 0.0005 * (double)(s - 1) - (split_cost - 0.0005) * extra;
    const double want = blocks_per_cu * ncu;   // resident blocks per CU of this tile shape (latency hiding)
    if (blocks < want) score -= 0.15 * (want - blocks) / want;
    if (score > best_score + 1e-9) {
      best_score = score;
      best = (int)s;
    }
    if (s >= 64 && blocks >= 16 * ncu) break;
  }
  return best;
}

struct FwdPlan {
  TileCfg tile;
  int gridM, gridN, splits, kt_total, kt_per_split;
  int patch_h, patch_w;   // > 0: GEMM rows run in patch_h x patch_w pixel patches (dilated convs), one M tile per patch
  bool banded;            // GEMM rows sorted by liveness class (takes precedence over the patch order)
  BandMap band;
  int hwaves = 0;         // fp16 kernels: waves per block of the chosen instantiation (0: the tile's default)
};

// Geometry of a gather problem, for the dilated-conv planning below (unit strides only).
struct DilGeom {
  int Ho, Wo;      // GEMM row space (output pixels of fwd, input pixels of dgrad)
  int Hi, Wi;      // gather source
  int taps_h, taps_w, dstep, off0;
};

// fraction of the (M tile, tap) pairs that are live (touch at least one in-range source pixel) when an M tile is a
// PH x PW patch of the row space; rows and columns separate
static double live_fraction(const DilGeom& g, int PH, int PW) {
  auto axis = [&](int n_out, int n_in, int P, int taps) {
    int live = 0, tot = 0;
    for (int p0 = 0; p0 < n_out; p0 += P)
      for (int t = 0; t < taps; ++t) {
        const int lo = p0 + g.off0 + t * g.dstep, hi = lo + P - 1;
        ++tot;
        if (hi >= 0 && lo < n_in) ++live;
      }
    return (double)live / (double)tot;
  };
  return axis(g.Ho, g.Hi, PH, g.taps_h) * axis(g.Wo, g.Wi, PW, g.taps_w);
}

// Liveness classes of a dilated gather problem (see BandMap).  Returns false when an axis needs more than three bands.
// tapmask[c]: bit (th * taps_w + tw) set when tap (th, tw) is in range for the pixels of class c.
static bool band_classes(const DilGeom& g, int B, BandMap& bm, unsigned (&tapmask)[9]) {
  if (g.taps_h * g.taps_w > 32) return false;
  int cut[2][4], nb[2];
  for (int ax = 0; ax < 2; ++ax) {
    const int n_out = ax ? g.Wo : g.Ho, n_in = ax ? g.Wi : g.Hi, taps = ax ? g.taps_w : g.taps_h;
    int pts[2 * 32 + 2], n = 0;
    pts[n++] = 0;
    pts[n++] = n_out;
    for (int t = 0; t < taps; ++t) {
      int lo = -g.off0 - t * g.dstep, hi = n_in - g.off0 - t * g.dstep;
      lo = lo < 0 ? 0 : (lo > n_out ? n_out : lo);
      hi = hi < 0 ? 0 : (hi > n_out ? n_out : hi);
      pts[n++] = lo;
      pts[n++] = hi;
    }
    for (int i = 1; i < n; ++i)      // insertion sort, then unique
      for (int j = i; j > 0 && pts[j] < pts[j - 1]; --j) {
        const int t = pts[j];
        pts[j] = pts[j - 1];
        pts[j - 1] = t;
      }
    int u = 0;
    for (int i = 0; i < n; ++i)
      if (u == 0 || pts[i] != pts[u - 1]) pts[u++] = pts[i];
    if (u > 4 || u < 2) return false;
    nb[ax] = u - 1;
    for (int i = 0; i < u; ++i) cut[ax][i] = pts[i];
  }
  auto axis_live = [&](int ax, int band, int t) {   // is tap t in range on the pixels [cut[band], cut[band+1]) of axis ax?
    const int n_in = ax ? g.Wi : g.Hi;
    const int lo = cut[ax][band] + g.off0 + t * g.dstep, hi = cut[ax][band + 1] - 1 + g.off0 + t * g.dstep;
    return lo >= 0 && hi < n_in;     // a band never straddles a tap's boundary: all in or all out
  };
  // classes in order of DESCENDING live-tap count: tile ids then run from the most to the least expensive tile (see
  // band_makespan for why)
  int n_cls = 0, c_h0[9], c_w0[9], c_cw[9], c_area[9], order[9];
  unsigned c_mask[9];
  for (int rb = 0; rb < nb[0]; ++rb)
    for (int cb = 0; cb < nb[1]; ++cb, ++n_cls) {
      c_h0[n_cls] = cut[0][rb];
      c_w0[n_cls] = cut[1][cb];
      c_cw[n_cls] = cut[1][cb + 1] - cut[1][cb];
      c_area[n_cls] = (cut[0][rb + 1] - cut[0][rb]) * c_cw[n_cls];
      unsigned mask = 0;
      for (int th = 0; th < g.taps_h; ++th)
        for (int tw = 0; tw < g.taps_w; ++tw)
          if (axis_live(0, rb, th) && axis_live(1, cb, tw)) mask |= 1u << (th * g.taps_w + tw);
      c_mask[n_cls] = mask;
      order[n_cls] = n_cls;
    }
  for (int i = 1; i < n_cls; ++i)
    for (int j = i; j > 0 && __builtin_popcount(c_mask[order[j]]) > __builtin_popcount(c_mask[order[j - 1]]); --j) {
      const int t = order[j];
      order[j] = order[j - 1];
      order[j - 1] = t;
    }
  int c = 0, row = 0;
  for (; c < n_cls; ++c) {
    const int o = order[c];
    bm.start[c] = row;
    bm.h0[c] = c_h0[o];
    bm.w0[c] = c_w0[o];
    bm.cw[c] = c_cw[o];
    bm.area[c] = c_area[o];
    tapmask[c] = c_mask[o];
    row += B * c_area[o];
  }
  for (; c < 9; ++c) {     // unused classes: empty, at the end
    bm.start[c] = row;
    bm.h0[c] = bm.w0[c] = 0;
    bm.cw[c] = bm.area[c] = 1;
    tapmask[c] = 0;
  }
  bm.start[9] = row;
  return true;
}

// Expected duration of a tap-skipping launch, as a fraction of the same launch with every tap live.  The tiles of such
// a launch differ in cost by up to 9 : 4 (live taps), so the mean live fraction says little: the launch is over when the
// busiest CU is.  Blocks are dealt round-robin -- 8 XCDs, then the CUs of an XCD -- so CU j works on blocks j, j + 256,
// j + 512, ...; the estimate is the largest such per-CU sum of tile costs.  `cost(tile_m)` = live taps of that row tile.
template <typename Cost>
static double skip_makespan(int tiles_m, int grid_n, int taps, int remap, Cost&& cost) {
  const int kCus = 256;
  const long long tiles = (long long)tiles_m * grid_n;
  if (tiles > (1 << 16)) return -1.0;     // many rounds: the mean is the estimate (caller falls back)
  long long cu_load[kCus];
  for (int j = 0; j < kCus; ++j) cu_load[j] = 0;
  for (int bid = 0; bid < (int)tiles; ++bid) cu_load[bid % kCus] += cost(remap_tile(remap, bid, (int)tiles) / grid_n);
  long long worst = 0;
  for (int j = 0; j < kCus; ++j) worst = cu_load[j] > worst ? cu_load[j] : worst;
  const long long rounds = (tiles + kCus - 1) / kCus;
  return (double)worst / (double)(rounds * taps);
}

// class-sorted rows, remap_tile mode 2: sorted by descending cost, the round-robin deal pairs the expensive tiles with
// the cheap ones
static double band_makespan(const DilGeom& g, const BandMap& bm, const unsigned (&tapmask)[9], int bm_rows, int grid_n) {
  const int M = bm.start[9];
  const double r = skip_makespan(cdiv(M, bm_rows), grid_n, g.taps_h * g.taps_w, 2, [&](int tm) {
    const int m0 = tm * bm_rows, m1 = m0 + bm_rows < M ? m0 + bm_rows : M;
    unsigned mask = 0;
    for (int c = 0; c < 9; ++c)
      if (bm.start[c] < m1 && bm.start[c + 1] > m0) mask |= tapmask[c];
    return __builtin_popcount(mask);
  });
  return r < 0.0 ? 1.0 : r;
}

// PH x PW pixel patches (row_perm 2; PW == Wo is the plain row-major order), XCD-remapped tile order
static double patch_makespan(const DilGeom& g, int B, int PH, int PW, int grid_n, int remap) {
  const int pr = g.Wo / PW, pc = g.Ho / PH, ppi = pr * pc;
  auto axis_live = [&](int p0, int P, int n_in, int taps) {
    int live = 0;
    for (int t = 0; t < taps; ++t) {
      const int lo = p0 + g.off0 + t * g.dstep, hi = lo + P - 1;
      if (hi >= 0 && lo < n_in) ++live;
    }
    return live;
  };
  const double r = skip_makespan(B * ppi, grid_n, g.taps_h * g.taps_w, remap, [&](int tm) {
    const int q = tm % ppi;
    return axis_live((q / pr) * PH, PH, g.Hi, g.taps_h) * axis_live((q % pr) * PW, PW, g.Wi, g.taps_w);
  });
  return r < 0.0 ? live_fraction(g, PH, PW) : r;
}

// relative cost per executed MAC of a tile shape in the tap-skipping launches (operand bytes staged per MAC; measured on the
// ASPP shapes)
static double band_shape_cost(TileCfg t) {
  static const int c32 = env_int("PSEG_CONV_BAND32_COST", 104);
  if (t.bn == 32) return c32 / 100.0;
  return (t.bm == 64 || t.bn == 64) ? 1.0 : (t.bm == 256 ? 0.90 : 0.92);
}

static FwdPlan plan_gather(long long M, int N, int K, bool allow_big = false, const DilGeom* geom = nullptr) {
  FwdPlan pl;
  pl.patch_h = pl.patch_w = 0;
  pl.banded = false;
  pl.tile = pick_tile(M, N);
  bool big = false;
  // staging-bound limb kernels: a 256x128 tile halves... (256+128)/(256*128) vs (128+128)/(128*128): 25 % less split +
  // LDS-write work per MAC.  One block (8 waves) per CU, so take it only when it still fills the chip in whole rounds.
  if (allow_big && cfg().conv_nobig == 0 && ((M >= 256 && N >= 128) || cfg().conv_forcebig != 0)) {
    const long long t = (long long)cdiv(M, 256) * cdiv(N, 128);
    const long long rounds = (t + 255) / 256;
    if ((t >= 256 && (double)t / (256.0 * rounds) >= 0.85) || cfg().conv_forcebig != 0) {
      pl.tile = TileCfg{256, 128};
      big = true;
    }
  }
  // fewer than two blocks per CU with the big tile: halve the N tile first (keeps the gathered A rows shared),
  // and only split K when even that leaves CUs idle
  if (!big && pl.tile.bm == 128 && pl.tile.bn == 128 && (long long)cdiv(M, 128) * cdiv(N, 128) < 512) pl.tile.bn = 64;
  // still under two blocks per CU (HRNet's 64-channel branch: 256 tiles of 128x64, one 4-wave block per CU, 22 % of the
  // matrix pipe): 32-column tiles -- the gathered rows are fetched once more per column tile, from L2
  if (!big && cfg().conv_narrow != 0 && pl.tile.bm == 128 && pl.tile.bn == 64 && N >= 64 && N % 32 == 0 &&
      (long long)cdiv(M, 128) * cdiv(N, 64) < 512 && (long long)cdiv(M, 128) * cdiv(N, 64) >= 128)
    pl.tile.bn = 32;
  const int force_bm = cfg().conv_bm, force_bn = cfg().conv_bn;
  const bool forced = force_bm && force_bn;
  if (forced) pl.tile = TileCfg{force_bm, force_bn};
  // Dilated 3x3 convs (the ASPP branches): pick the (tile, patch shape) with the least live (tile, tap) work.  Candidates:
  // the tile chosen above and, in its place, a 64-row tile with 128 columns (same operand traffic per MAC as 128x64);
  // patch shapes = every PH x PW = tile rows that tiles the map.  The current row-major order is the PW == Wo candidate.
  if (geom != nullptr && cfg().conv_noskip == 0 && !forced && (long long)geom->Ho * geom->Wo > 0 &&
      M % ((long long)geom->Ho * geom->Wo) == 0) {
    double best = 2.0;
    TileCfg best_tile = pl.tile;
    int best_ph = 0, best_pw = 0;
    TileCfg cands[3] = {pl.tile, TileCfg{64, 128}, TileCfg{128, 32}};
    int ncand = (!big && N >= 128 && pl.tile.bm == 128 && (long long)cdiv(M, 64) * cdiv(N, 128) >= 512) ? 2 : 1;
    // 128x32 tiles (round 5): the tap-skipping launches are bound by their BUSIEST CU -- rate 12 on a 32x32 map leaves tiles of
    // 9, 6 and 4 live taps, two of the 512 128x64 tiles per CU, worst pair 9 + 4 against a mean of 10.1 (0.72 of the dense time
    // for 0.56 live) -- and twice as many tiles of half the cost deal out more evenly (0.64).  The gathered rows are then
    // fetched once per 32 output columns (from L2): shape_cost below.
    static const int band32 = env_int("PSEG_CONV_BAND32", 1);
    if (ncand == 2 && band32 != 0 && N % 32 == 0 && cfg().conv_dma32 != 0) ncand = 3;
    for (int c = 0; c < ncand; ++c) {
      const int bm = cands[c].bm;
      // relative cost per executed MAC of the tile shape (operand bytes staged per MAC; measured on the ASPP shapes)
      const double shape_cost = band_shape_cost(cands[c]);
      if (cands[c].bn == 32) continue;      // (the narrow tile only with class-sorted rows, below)
      for (int pw = 1; pw <= geom->Wo && pw <= bm; pw *= 2) {
        if (bm % pw != 0 || geom->Wo % pw != 0) continue;
        const int ph = bm / pw;
        if (ph > geom->Ho || geom->Ho % ph != 0) continue;
        double score = patch_makespan(*geom, (int)(M / ((long long)geom->Ho * geom->Wo)), ph, pw, cdiv(N, cands[c].bn),
                                      cfg().conv_noxcd == 0 ? 1 : 0) * shape_cost;
        if (pw == geom->Wo) score -= 1e-6;   // ties: keep the row-major order
        if (cfg().plan_debug != 0)
          fprintf(stderr, "[pseg plan]   %dx%d patch %dx%d: %.3f\n", cands[c].bm, cands[c].bn, ph, pw, score);
        if (score < best - 1e-9) {
          best = score;
          best_tile = cands[c];
          best_ph = ph;
          best_pw = pw;
        }
      }
    }
    if (best_ph > 0) {
      pl.tile = best_tile;
      if (best_pw != geom->Wo) {
        pl.patch_h = best_ph;
        pl.patch_w = best_pw;
      }
    }
    // rows sorted by liveness class: no rectangle constraint at all -- taken when it runs >= 2 % fewer K-steps
    unsigned tapmask[9];
    BandMap bmap;
    if (cfg().conv_noband == 0 && band_classes(*geom, (int)(M / ((long long)geom->Ho * geom->Wo)), bmap, tapmask)) {
      for (int c = 0; c < ncand; ++c) {
        const double shape_cost = band_shape_cost(cands[c]);
        const double score = band_makespan(*geom, bmap, tapmask, cands[c].bm, cdiv(N, cands[c].bn)) * shape_cost;
        if (cfg().plan_debug != 0)
          fprintf(stderr, "[pseg plan]   %dx%d class-sorted: %.3f\n", cands[c].bm, cands[c].bn, score);
        if (score < best - 0.02) {
          best = score;
          pl.tile = cands[c];
          pl.banded = true;
          pl.band = bmap;
          pl.patch_h = pl.patch_w = 0;
        }
      }
    }
  }
  if (cfg().plan_debug != 0 && geom != nullptr)
    fprintf(stderr, "[pseg plan] M=%lld N=%d K=%d dstep=%d off0=%d: tile %dx%d %s (patch %dx%d)\n", M, N, K, geom->dstep,
            geom->off0, pl.tile.bm, pl.tile.bn, pl.banded ? "class-sorted" : (pl.patch_w ? "patches" : "row-major"),
            pl.patch_h, pl.patch_w);
  pl.gridM = cdiv(M, pl.tile.bm);
  pl.gridN = cdiv(N, pl.tile.bn);
  pl.kt_total = cdiv(K, BK);
  const long long tiles = (long long)pl.gridM * pl.gridN;
  int splits = (tiles < 256 && !big) ? pick_splits(tiles, pl.kt_total, 16, 64) : 1;
  const int force_s = cfg().conv_splitk;
  if (force_s > 0) splits = force_s < pl.kt_total ? force_s : pl.kt_total;
  pl.kt_per_split = cdiv(pl.kt_total, splits);
  pl.splits = cdiv(pl.kt_total, pl.kt_per_split);
  if (pl.splits > 1) {
    pl.patch_h = pl.patch_w = 0;
    pl.banded = false;
  }
  return pl;
}

// the planner sees the dilated geometry only for unit-stride convs with more than one tap and a rate >= 4
static bool dil_geom(DilGeom& g, int Ho, int Wo, int Hi, int Wi, int taps_h, int taps_w, int Cin, int s_out, int s_in,
                     int dstep, int off0) {
  const int adil = dstep < 0 ? -dstep : dstep;
  if (!(adil >= 4 && taps_h * taps_w > 1 && taps_h * taps_w <= 32 && Cin % BK == 0 && s_out == 1 && s_in == 1)) return false;
  g = DilGeom{Ho, Wo, Hi, Wi, taps_h, taps_w, dstep, off0};
  return true;
}

struct WgradPlan {
  TileCfg tile;
  int gridM, gridN, splits, pix_per_split;
};

// concurrent: the launch shares the chip with another stream's kernels (the training step's weight-gradient stream beside the
// data gradients); alone it is planned for two resident blocks per CU (exact-fp32 kernels, see below)
static WgradPlan plan_wgrad(long long P, int Cout, int K, bool allow_big = false, bool limb = false, int want_bpc = 0,
                            bool concurrent = true) {
  WgradPlan pl;
  pl.tile = pick_tile(Cout, K);
  // limb kernels are bound by the split + LDS-write work per staged element: a 256(Cout) x 128 tile (8 waves, one
  // block per CU) does 25 % less of it per MAC.
  // Measured slower than two 128x128 blocks per CU (aspp d6 0.57 -> 0.67 ms): the gather side of the loader sits in
  // two of the eight waves and becomes the critical path.  Kept selectable (PSEG_WGRAD_BIG=1, and the forced parity
  // test) until the loader roles are spread over all waves.
  // (round 4, measured no: putting the 32-output-channel weight gradients of HRNet's fine branch on the LDS-DMA kernel through a
  // 64-row tile whose upper half is out of range -- instead of the register-staged wgrad_kernel<32,128> -- made the replayed
  // HRNet fp32 step slower, 16.53 -> 17.02 ms; PSEG_WGRAD_NARROW64=1 re-enables it)
  static const int narrow64 = env_int("PSEG_WGRAD_NARROW64", 0);
  if (!limb && narrow64 != 0 && pl.tile.bm == 32 && pl.tile.bn == 128 && Cout > 16) pl.tile.bm = 64;
  // Exact-fp32, at most 32 output channels (round 5): a 32 x 256 tile.  The 32 x 128 tile of wgrad_kernel gave every wave ONE
  // accumulator fed by register-staged loads (41 TF on HRNet's 32-channel branch, 68 on the 21-class classifier): eight waves
  // on 256 columns with the operands by LDS-DMA (wgrad_f32_dma_kernel<32, 256>), or four waves with two accumulators each where
  // the map does not tile into 32-pixel patches.  PSEG_WGRAD_NARROW256=0 goes back.
  static const int narrow256 = env_int("PSEG_WGRAD_NARROW256", 1);
  // (only where 256-column tiles pad K by at most 5 % more than 128-column ones: 3 x 3 on 32 channels is K = 288 -- two
  // 256-column tiles waste 44 % and measured 62 against 55 us; the classifier's K = 3456 wastes 3.7 %: 649 -> 581 us)
  if (!limb && narrow256 != 0 && pl.tile.bm == 32 && pl.tile.bn == 128 && K >= 256 &&
      (long long)cdiv(K, 256) * 256 * 100 <= (long long)cdiv(K, 128) * 128 * 105)
    pl.tile.bn = 256;
  // ... and K == 288 -- 3x3 on 32 channels, every conv of HRNet's fine branch -- as ONE 32 x 288 tile on nine waves
  // (wgrad_f32_dma_kernel<32, 288, 1, 9>): dy is fetched once per K-step, not once per 128 columns, and no column is padding
  static const int narrow288 = env_int("PSEG_WGRAD_NARROW288", 1);
  if (!limb && narrow288 != 0 && cfg().wgrad_f32dma != 0 && pl.tile.bm == 32 && K == 288) pl.tile.bn = 288;
  const bool big = allow_big && cfg().conv_nobig == 0 &&
                   ((cfg().wgrad_big != 0 && Cout >= 256 && Cout % 256 == 0 && K >= 128) ||
                    cfg().conv_forcebig != 0);
  if (big) pl.tile = TileCfg{256, 128};
  const int force_bm = cfg().wgrad_bm, force_bn = cfg().wgrad_bn;
  if (force_bm && force_bn) pl.tile = TileCfg{force_bm, force_bn};
  pl.gridM = cdiv(Cout, pl.tile.bm);
  pl.gridN = cdiv(K, pl.tile.bn);
  const long long ptiles = cdiv(P, BK);
  // every extra pixel split writes and re-reads one more [Cout][K] slab: relative to the kernel's own time
  // (2*P*Cout*K flop at F flop/s against 8*Cout*K bytes at BW) that is 4*F / (BW * P) per split -- F/BW ~ 27 flop/byte
  // for the exact-fp32 kernel, ~3x that for the limb kernels
  const double split_cost = (limb ? 330.0 : 110.0) / (double)P;
  // resident blocks per CU the split count aims for: two, or one 8-wave block of the 256-row tile -- and one as well for
  // a SMALL exact-fp32 problem (8-wave LDS-DMA blocks; under 2 GMAC and 128k pixels: the HRNet / UNet layers) that could only fill
  // two per CU with blocks of fewer than 32 K-steps (1024 pixels): there the prologue / slab epilogue of a block costs
  // more than the second resident block hides (HRNet 512x512 B=8 replayed: 18.45 -> 18.0 ms).  The DeepLabV3+ layers keep
  // two: one per CU is 10-15 % faster for their 1x1 weight gradients in isolation (tools/shortk_sweep.py) but 0.3 ms
  // slower in the step, where they share the CUs with the data gradients.  PSEG_WGRAD_BPC overrides.
  // Exact-fp32 kernels: ONE (round 5).  Every split writes and re-reads a [Cout][K] slab -- at two blocks per CU 2.23 GB written
  // + 2.23 GB re-read per DeepLabV3+ step for 157 MB of gradients -- and in the two-stream step the weight gradients share the
  // CUs with the data gradients anyway: PSEG_WGRAD_BPC=1 against 2 on one box, 44.49 against 44.93 ms (round 2 had measured
  // the opposite, on the register-staged kernels).
  // ... for launches that run BESIDE the data gradients (`concurrent`); a launch that has the chip to itself keeps two: one
  // 8-wave block per CU does not fill the matrix pipe alone (the one-stream step's weight-gradient class 12.96 -> 13.04 ms with
  // one, despite 0.4 ms less work in it).
  int bpc = (pl.tile.bm == 256 || (!limb && concurrent)) ? 1 : 2;
  {
    const long long tiles = (long long)pl.gridM * pl.gridN;
    const long long s_two = (512 + tiles - 1) / tiles;
    if (!limb && ptiles / s_two < 32 && (double)P * Cout * K < 2e9 && P <= (1 << 17)) bpc = 1;
  }
  // 32-row tiles of the exact-fp32 kernel (HRNet's 32-channel branch): every wave owns ONE 32x32 accumulator, its MFMAs form
  // one dependent chain and the wave cannot issue its gather / staging VALU work under them -- a second resident block per
  // CU does (stand-alone 68 -> 59 us at 8x128x128x32 -> 32 3x3).  PSEG_WGRAD_BPC_NARROW=1 goes back to one.
  static const int narrow_bpc = env_int("PSEG_WGRAD_BPC_NARROW", 2);
  if (!limb && pl.tile.bm == 32 && narrow_bpc > 0) bpc = narrow_bpc;
  if (want_bpc > 0) bpc = want_bpc;
  if (cfg().wgrad_bpc > 0) bpc = cfg().wgrad_bpc;
  int splits = pick_splits((long long)pl.gridM * pl.gridN, ptiles, 8, 1024, bpc,
                           split_cost > 0.0005 ? split_cost : 0.0005);
  const int force_s = cfg().wgrad_splits;
  if (force_s > 0) splits = force_s < ptiles ? force_s : (int)ptiles;
  const long long tiles_per = cdiv(ptiles, splits);
  pl.pix_per_split = (int)(tiles_per * BK);
  pl.splits = cdiv(P, pl.pix_per_split);
  return pl;
}

// fixed-order slab reductions (kernels in conv_mfma.hip)
int launch_slab_reduce(const float* slabs, long long slab_stride, int nslab, float* out, int ld, long long M, int N,
                       const float* bias, int accumulate, hipStream_t st);

}  // namespace pseg
