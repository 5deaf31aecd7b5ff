// Implicit-GEMM convolution for gfx950 on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate).
//
// Two kernels cover forward, data-gradient and weight-gradient of every dense conv on the hot path
// (reference models/aspp.py:12,27-30, models/deeplabv3plus.py:20,22, models/unet.py:19-23):
//
//  gather_conv_kernel  C[M=pixels][N] = A[M][K] * Bw[N][K]^T,  both operands K-contiguous.
//     A is never materialised: row m = (b,ho,wo), column k = (r,s,c) is gathered from the NHWC tensor at
//     ((ho*s_out + off0 + r*dstep)/s_in, (wo*s_out + off0 + s*dstep)/s_in).  fwd: s_out=stride, s_in=1,
//     dstep=+dil, off0=-pad.  dgrad: source = dy, s_out=1, s_in=stride, dstep=-dil, off0=+pad, Bw = the
//     transposed filter.  Taps that land in the zero padding (most of them for the rate-18 ASPP branch)
//     are fetched through a buffer descriptor with an out-of-range offset: the hardware returns 0.
//  wgrad_kernel        dW[Cout][K] = dY[P][Cout]^T * A[P][K],  both operands K(=pixel)-strided.
//
// Tile: 256 threads = 4 waves, each wave owns a (BM/WARPS_M)x(BN/WARPS_N) block of 32x32 MFMA tiles.
// K-step 32, register-staged global->LDS double buffer (loads for step t+1 are issued before the MFMAs of
// step t and written to LDS after them), one barrier per step.  The fp32 MFMA issues once per 64 cycles per
// SIMD, so a step carries 16*TM*TN*64 cycles of matrix work per wave (4096 for the 128x128 tile) against
// 4-8 16-byte loads per lane; measured: the step's fixed costs (barrier, LDS write, first-read latency, address
// arithmetic) are what separates the kernel from the MFMA peak, hence the long step.
//
// LDS images:
//  gather_conv: [rows][32+4] floats (row stride 144 B = 9 x 16-B slots, 9 coprime to 16 -> the 16-lane groups
//     of ds_read_b128 hit 16 different slots; the 8 lanes of a ds_write_b128 group write one contiguous row).
//     Lane (i=l&31, h=l>>5) reads 4 consecutive k's {g*8+4h .. g*8+4h+3}; MFMA step j of group g therefore
//     contracts k = g*8+j (h=0 lanes) and g*8+4+j (h=1 lanes).  A and B use the same permutation, so the sum
//     over k is unchanged.
//  wgrad: [32 pixels][cols] floats, read with ds_read_b32 (32 consecutive floats per half-wave); all fragments of
//     half a step are fetched before its MFMAs so LDS latency is paid once per 16*TM*TN... MFMAs, not per 4.
#include "conv_common.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/pseg_amd.h"

// Per-block phase timestamps in gather_f32_dma_kernel: compiled in only with -DPSEG_CONV_TRACE=1 (PSEG_BUILD_TRACE=1 python -m
// pytorch_segmentation_amd.csrc.build --force) -- even dormant, the four extra branches cost 0.45 ms of the 46.3 ms step.
#ifndef PSEG_CONV_TRACE
#define PSEG_CONV_TRACE 0
#endif
constexpr bool kConvTrace = PSEG_CONV_TRACE != 0;

namespace pseg {

static thread_local char g_err[512] = "";
thread_local int g_last_conv_kernel = 0;      // pseg_debug_last_conv_kernel (conv_common.h)
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* last_error() { return g_err; }


// ------------------------------------------------------------------------------------------------
// Split-bf16 ("bf16x3") arithmetic: every fp32 operand x is split into hi = bf16_rne(x), lo = bf16_rne(x - hi)
// (x - hi is exact in fp32), and a product a*b is evaluated as ah*bh + ah*bl + al*bh on v_mfma_f32_32x32x16_bf16 with
// fp32 accumulation.  The dropped al*bl term and the rounding of lo are ~2^-17 of |a*b| (rms), i.e. the result carries
// ~17 significant bits per product -- two orders of magnitude inside the 1e-3 contract and of the same size as the
// summation-order noise of a K = 18432 fp32 contraction -- at 3/16 of the fp32-MFMA instruction time per MAC.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {   // v_cvt_pk_bf16_f32: round-to-nearest-even
  bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, v);
}

// N-limb split of 4 consecutive-k fp32 values: limb[0] = bf16(x), limb[1] = bf16(x - limb0), limb[2] = bf16(x - l0 - l1);
// every subtraction is exact in fp32, so three limbs carry all 24 mantissa bits.
// x - float(bf16 half of `packed`) in ONE instruction: v_dot2c_f32_bf16 computes p.lo*m.lo + p.hi*m.hi + acc with the
// products and the sum exact here (one factor is -1 or 0, and x - bf16(x) is representable).  The selector constants
// {-1, 0} / {0, -1} travel through an opaque SGPR: written as literals the compiler folds {-1, 0} into the inline
// constant "-1.0", which the hardware expands to 0xBF800000 = {0, -1} (tools/micro/dot2c_split.hip).
// (A non-finite partner in the same pair would turn 0 * inf into NaN; finite training tensors never reach bf16's range.)
struct ResidualSel {
  bf16x2 lo, hi;
  __device__ __forceinline__ ResidualSel() {
    unsigned c0 = 0x0000BF80u, c1 = 0xBF800000u;
    asm volatile("" : "+s"(c0), "+s"(c1));
    lo = __builtin_bit_cast(bf16x2, c0);
    hi = __builtin_bit_cast(bf16x2, c1);
  }
};
__device__ __forceinline__ float resid_lo(unsigned packed, float x, const ResidualSel& rs) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, packed), rs.lo, x, false);
}
__device__ __forceinline__ float resid_hi(unsigned packed, float x, const ResidualSel& rs) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, packed), rs.hi, x, false);
}

template <int NL>
__device__ __forceinline__ void split4n(f32x4 v, u32x2 (&limb)[NL], const ResidualSel& rs) {
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const unsigned p01 = pack_bf16(v[0], v[1]), p23 = pack_bf16(v[2], v[3]);
    limb[l] = u32x2{p01, p23};
    if (l + 1 < NL) {
      v[0] = resid_lo(p01, v[0], rs);
      v[1] = resid_hi(p01, v[1], rs);
      v[2] = resid_lo(p23, v[2], rs);
      v[3] = resid_hi(p23, v[3], rs);
    }
  }
}

// fp16 limbs: x*2^e = hi + lo with two 11-bit limbs (22-bit operands, products accurate to ~2^-22).  fp16 has only a
// 5-bit exponent, so the operand is first scaled by an exact power of two chosen from its per-tensor max |x| (published
// by the producer kernels as the bit pattern of a non-negative float, see amax_update): amax * 2^e lies in [2^14, 2^15).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_f16(float a, float b) {  // v_cvt_f16_f32 (RNE) x2 + pack
  f16x2 v = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float f16_lo(unsigned p) { return (float)__builtin_bit_cast(f16x2, p)[0]; }
__device__ __forceinline__ float f16_hi(unsigned p) { return (float)__builtin_bit_cast(f16x2, p)[1]; }

__device__ __forceinline__ void split4h(f32x4 v, float scale, u32x2 (&limb)[2]) {
  v *= scale;
  const unsigned h01 = pack_f16(v[0], v[1]), h23 = pack_f16(v[2], v[3]);
  limb[0] = u32x2{h01, h23};
  limb[1] = u32x2{pack_f16(v[0] - f16_lo(h01), v[1] - f16_hi(h01)), pack_f16(v[2] - f16_lo(h23), v[3] - f16_hi(h23))};
}

// exponent e such that amax * 2^e is in [2^14, 2^15) (e = 0 for an all-zero / non-finite tensor), as a float 2^e
__device__ __forceinline__ float pow2_scale_for(unsigned amax_bits, int& e_out) {
  const int ex = (int)((amax_bits >> 23) & 0xFFu);
  int e = 0;
  if (ex != 0 && ex != 255) e = 14 - (ex - 127);
  if (e > 100) e = 100;
  if (e < -100) e = -100;
  e_out = e;
  return __builtin_bit_cast(float, (unsigned)(e + 127) << 23);
}

// LDS image of a split tile: [row][32 bf16 = 64 bytes = four 16-byte k-slots], slot XOR-swizzled with (row>>2)&3 so the
// 16-lane groups of ds_read_b128 (rows {0-3,12-15,20-27} of a 32-row fragment, same slot) fall on 16 different 16-byte
// bank slots.  Returns the dword offset of k-slot `slot` of `row`.
__device__ __forceinline__ int swz(int row, int slot) { return row * 16 + ((slot ^ ((row >> 2) & 3)) << 2); }



// PREC: 0 = exact fp32 (v_mfma_f32_32x32x2_f32); 1 = two bf16 limbs, 3 partial products (~17 bits per product);
//       2 = three bf16 limbs, 6 partial products down to 2^-16 (error ~2^-23 per product: fp32-equivalent);
//       3 = two fp16 limbs of the power-of-two-scaled operand, 3 partial products (~2^-22 per product)
template <int BM, int BN, int WARPS_M, int WARPS_N, bool SKIP, int PREC>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N) void gather_conv_kernel(const GatherConvParams p) {
  set_wave_prio(p.prio);
  static_assert(WARPS_M * WARPS_N == 4 || WARPS_M * WARPS_N == 8, "4 waves, or 8 for the 256-row tile");
  constexpr int NT = 64 * WARPS_M * WARPS_N;   // threads
  constexpr int RPP = NT / CPR;                // rows covered by one pass of the block's threads (shadows the 256-thread global)
  constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1 && WTM % 32 == 0 && WTN % 32 == 0, "wave tile");
  constexpr int AR = (BM + RPP - 1) / RPP, BR = (BN + RPP - 1) / RPP;

  // staging: fp32 -> 2 x [rows][BK+4] floats; split-bf16 -> 2 x NL limb images x [rows][16 dwords]
  constexpr int NL = PREC == 0 ? 1 : (PREC == 3 ? 2 : PREC + 1);
  constexpr int kStage = PREC == 0 ? 2 * (BM + BN) * LDT : 2 * (BM + BN) * 16 * NL;
  constexpr int kPatch = (NT / 64) * WTM * (WTN + 4);
  __shared__ __attribute__((aligned(16))) float lds[kStage > kPatch ? kStage : kPatch];
  float* As = lds;
  float* Bs = lds + 2 * BM * LDT;
  unsigned* ldsw = reinterpret_cast<unsigned*>(lds);
  // split-bf16 image bases (dwords): A limb l at kAl(l), B limb l at kBl(l), each [2 buffers][rows][16]
  constexpr int kAsz = 2 * BM * 16, kBsz = 2 * BN * 16, kBbase = NL * kAsz;

  const int tid = threadIdx.x;
  __builtin_assume(tid >= 0 && tid < NT);
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;
  const int gridN = (p.N + BN - 1) / BN;
  // XCD-aware tile order: the dispatcher deals consecutive workgroups round-robin over the 8 XCDs, each with its own
  // L2.  Remap so that XCD x works on a CONTIGUOUS range of tiles: the gridN column tiles that share one gathered A
  // tile (and neighbouring row tiles that share halo pixels) then hit the same L2 instead of fetching A once per XCD.
  int bid = blockIdx.x;
  bid = remap_tile(p.xcd_remap, bid, (int)gridDim.x);
  const int tile_n = bid % gridN;
  const int tile_m = bid / gridN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.w, p.w_bytes);
  const ResidualSel rsel;

  float scale_a = 1.f, scale_b = 1.f, unscale = 1.f;
  if constexpr (PREC == 3) {
    int ea, eb;
    scale_a = pow2_scale_for(*p.amax_a, ea);
    scale_b = pow2_scale_for(*p.amax_b, eb);
    unscale = __builtin_bit_cast(float, (unsigned)(127 - (ea + eb)) << 23);  // |ea + eb| <= 200 < 127? clamp below
    if (ea + eb > 126 || ea + eb < -126) unscale = exp2f(-(float)(ea + eb));
  }

  // ---- per-thread load assignment: 16-byte chunk cc of rows r0 + RPP*i
  const int cc = tid % CPR;
  const int r0 = tid / CPR;
  int a_bh[AR], a_bw[AR], a_img[AR];
  bool a_ok[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int row = r0 + RPP * i;
    const int m = m0 + row;
    const bool ok = (row < BM) && (m < p.M);
    const int mm = ok ? m : 0;
    int b, ho, wo;
    row_to_pixel(p, mm, b, ho, wo);
    a_ok[i] = ok;
    a_bh[i] = ho * p.s_out + p.off0;
    a_bw[i] = wo * p.s_out + p.off0;
    a_img[i] = b * p.Hi * p.Wi;
  }
  uint32_t b_off[BR];
  bool b_ok[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) {
    const int row = r0 + RPP * i;
    const int n = n0 + row;
    b_ok[i] = (row < BN) && (n < p.N);
    b_off[i] = (uint32_t)n * (uint32_t)p.K;
  }

  // ---- K range of this block (split-K slice)
  const int kt_begin = blockIdx.z * p.kt_per_split;
  int kt_end = kt_begin + p.kt_per_split;
  if (kt_end > p.kt_total) kt_end = p.kt_total;

  // validity of tap (r, s) for load row i (the same test the loader applies)
  auto row_tap_ok = [&](int i, int dh, int dw, int& hn, int& wn_) -> bool {
    hn = a_bh[i] + dh;
    wn_ = a_bw[i] + dw;
    bool ok = a_ok[i];
    if (p.s_in != 1) {
      ok = ok && (hn % p.s_in == 0) && (wn_ % p.s_in == 0);
      hn /= p.s_in;
      wn_ /= p.s_in;
    }
    return ok && ((unsigned)hn < (unsigned)p.Hi) && ((unsigned)wn_ < (unsigned)p.Wi);
  };

  // ---- dilated convs: which taps touch at least one in-bounds pixel of this M tile?  (For the rate-18 ASPP branch
  // at 32x32 most (tile, tap) pairs are pure zero padding; their K-steps are never loaded nor multiplied.)
  unsigned tapmask = 0xFFFFFFFFu;
  if (SKIP) {
    unsigned mine = 0;
    for (int t = 0; t < p.ntaps; ++t) {
      const int r = t / p.kw, sx = t - r * p.kw;
      bool any = false;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        int hn, wn_;
        any = any || row_tap_ok(i, r * p.dstep, sx * p.dstep, hn, wn_);
      }
      if (any) mine |= 1u << t;
    }
    unsigned* sm = reinterpret_cast<unsigned*>(lds);
    if (tid == 0) sm[0] = 0u;
    __syncthreads();
    if (mine) atomicOr(&sm[0], mine);
    __syncthreads();
    tapmask = sm[0];
    __syncthreads();  // the staging buffers are written next
  }
  auto next_valid = [&](int kt) -> int {
    if (SKIP) {
      while (kt < kt_end) {
        const int tap = kt / p.ktiles_per_tap;
        if ((tapmask >> tap) & 1u) break;
        kt = (tap + 1) * p.ktiles_per_tap;
      }
      if (kt > kt_end) kt = kt_end;
    }
    return kt;
  };
  // K-step after `kt`: inside a live tap the successor is kt + 1 (no division); only a tap boundary rescans the mask
  int nv_tap_end = 0;   // first K-step of the tap after the one `kt` walks in (SKIP only)
  auto next_after = [&](int kt) -> int {
    if (!SKIP) return kt + 1;
    if (kt + 1 < nv_tap_end) return kt + 1;
    const int n = next_valid(kt + 1);
    nv_tap_end = (n / p.ktiles_per_tap + 1) * p.ktiles_per_tap;
    return n;
  };

  f32x4 areg[AR], breg[BR];

  // K-step kt: this thread's chunk is k = kt*BK + cc*4 = (r*kw + s)*Cin + c.  Everything that depends on the tap
  // (r, s) -- the in-range test of each gathered row and its pixel offset -- is computed when the tap CHANGES
  // (every Cin/BK steps for Cin >= BK) and kept as one byte offset per row (kOOB = zero padding); a plain K-step then
  // costs one add per load.  (Measured before this hoist: 2 VALU instructions per MFMA and ~500 instructions of
  // branchy address arithmetic in front of every 64-MFMA burst.)
  int kt_cur = -1, k_cur = 0, kr_cur = 0, ks_cur = 0, kc_cur = 0;
  uint32_t a_off[AR];
  uint32_t b_offb[BR];
#pragma unroll
  for (int i = 0; i < BR; ++i) b_offb[i] = b_ok[i] ? b_off[i] * 4u : kOOB;
  auto retap = [&]() {
    const int dh = kr_cur * p.dstep, dw = ks_cur * p.dstep;
    const bool tap_ok = kr_cur * p.kw + ks_cur < p.ntaps;   // K tail (K % BK != 0): chunks beyond the last tap are zero
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      int hn, wn_;
      const bool ok = row_tap_ok(i, dh, dw, hn, wn_) && tap_ok;
      a_off[i] = ok ? (uint32_t)((a_img[i] + hn * p.Wi + wn_) * p.ldx) * 4u : kOOB;
    }
  };
  auto seek = [&](int kt) {
    k_cur = kt * BK + cc * 4;
    const int tap = k_cur / p.Cin;
    kc_cur = k_cur - tap * p.Cin;
    kr_cur = tap / p.kw;
    ks_cur = tap - kr_cur * p.kw;
    kt_cur = kt;
    retap();
  };
  auto load_tile = [&](int kt) {
    if (SKIP) {
      if (kt != kt_cur) seek(kt);  // the tap-skipping walk jumps; plain steps stay incremental
    } else if (kt_cur < 0) {
      seek(kt);
    }
    const uint32_t kcb = (uint32_t)kc_cur * 4u;
#pragma unroll
    for (int i = 0; i < AR; ++i) areg[i] = buf_load4(xr, a_off[i] + kcb);   // kOOB + kcb stays out of range
    const bool kvalid = k_cur < p.K;
    const uint32_t kb = (uint32_t)k_cur * 4u;
#pragma unroll
    for (int i = 0; i < BR; ++i) breg[i] = buf_load4(wr, kvalid ? b_offb[i] + kb : kOOB);
    // advance to the next K-step
    k_cur += BK;
    kc_cur += BK;
    if (kc_cur >= p.Cin) {
      do {
        kc_cur -= p.Cin;
        if (++ks_cur == p.kw) {
          ks_cur = 0;
          ++kr_cur;
        }
      } while (kc_cur >= p.Cin);
      retap();
    }
    kt_cur = kt + 1;
  };

  auto store_tile = [&](int buf) {
    if constexpr (PREC == 0) {
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const int row = r0 + RPP * i;
        if (row < BM) *reinterpret_cast<f32x4*>(&As[(buf * BM + row) * LDT + cc * 4]) = areg[i];
      }
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        const int row = r0 + RPP * i;
        if (row < BN) *reinterpret_cast<f32x4*>(&Bs[(buf * BN + row) * LDT + cc * 4]) = breg[i];
      }
    } else {
      // chunk cc = k 4cc..4cc+3 -> 8 bytes at half (cc&1) of k-slot (cc>>1)
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const int row = r0 + RPP * i;
        if (row < BM) {
          u32x2 limb[NL];
          if constexpr (PREC == 3) split4h(areg[i], scale_a, limb);
          else split4n<NL>(areg[i], limb, rsel);
          const int o = buf * BM * 16 + swz(row, cc >> 1) + (cc & 1) * 2;
#pragma unroll
          for (int l = 0; l < NL; ++l) *reinterpret_cast<u32x2*>(&ldsw[l * kAsz + o]) = limb[l];
        }
      }
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        const int row = r0 + RPP * i;
        if (row < BN) {
          u32x2 limb[NL];
          if constexpr (PREC == 3) split4h(breg[i], scale_b, limb);
          else split4n<NL>(breg[i], limb, rsel);
          const int o = buf * BN * 16 + swz(row, cc >> 1) + (cc & 1) * 2;
#pragma unroll
          for (int l = 0; l < NL; ++l) *reinterpret_cast<u32x2*>(&ldsw[kBbase + l * kBsz + o]) = limb[l];
        }
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frag_row = lane & 31;
  const int frag_k = (lane >> 5) * 4;

  // ---- main loop.  Two fragment sets alternate between the two 16-deep halves of a K-step; three tiles are in flight:
  // `cur` in LDS buffer `buf` (being multiplied), `nxt` complete in buffer buf^1, `stg` in the staging registers (its
  // global loads were issued one whole K-step earlier).  Per K-step:
  //   A  fetch the fragments of half 1 of `cur`
  //   B  MFMAs on half 0 (fragments fetched during the previous burst)                 -- hides A
  //   C  barrier: every wave is done reading `buf`; buf^1 (written a K-step ago) is visible
  //   D  fetch the fragments of half 0 of `nxt` from buf^1
  //   E  split + write `stg` into `buf`, issue the global loads of the tile after it,
  //      MFMAs on half 1 (already in registers)                                        -- hides D and the traffic
  // so nothing the matrix pipe waits for is issued right in front of it, and the barrier has no data to wait for.
  // The tap bookkeeping of the load stream (carry into the next tap, tap-skipping jumps, end of stream) runs at the TOP
  // of an iteration, which leaves block E straight-line code the scheduler can spread over the MFMAs.  An exhausted
  // stream keeps "loading" with out-of-range offsets (zeros) into a buffer nobody reads again.
  static_assert(BK == 32, "two half-steps per K-step");
  constexpr int FRA = PREC == 0 ? 2 * TM : NL * TM, FRB = PREC == 0 ? 2 * TN : NL * TN;
  f32x4 fa[2][FRA], fb[2][FRB];   // PREC 0: [gg][tile] fp32 k-quads; limb variants: [limb][tile] 8 x 16-bit
  auto read_frags = [&](int set, int buf, int half) {
    if constexpr (PREC == 0) {
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        const int g = half * 2 + gg;
#pragma unroll
        for (int i = 0; i < TM; ++i)
          fa[set][gg * TM + i] = *reinterpret_cast<const f32x4*>(
              &As[(buf * BM + wm * WTM + i * 32 + frag_row) * LDT + g * 8 + frag_k]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          fb[set][gg * TN + j] = *reinterpret_cast<const f32x4*>(
              &Bs[(buf * BN + wn * WTN + j * 32 + frag_row) * LDT + g * 8 + frag_k]);
      }
    } else {
      // one 16-deep k-slice: lane (row = l&31, h = l>>5) holds k = 16*half + 8h .. +7 (k-slot 2*half + h) of its row
      const int slot = half * 2 + (lane >> 5);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int o = buf * BM * 16 + swz(wm * WTM + i * 32 + frag_row, slot);
#pragma unroll
        for (int l = 0; l < NL; ++l) fa[set][l * TM + i] = *reinterpret_cast<const f32x4*>(&ldsw[l * kAsz + o]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int o = buf * BN * 16 + swz(wn * WTN + j * 32 + frag_row, slot);
#pragma unroll
        for (int l = 0; l < NL; ++l) fb[set][l * TN + j] = *reinterpret_cast<const f32x4*>(&ldsw[kBbase + l * kBsz + o]);
      }
    }
  };
  auto mfmas = [&](int set) {
    if constexpr (PREC == 0) {
#pragma unroll
      for (int gg = 0; gg < 2; ++gg)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][gg * TM + i][e], fb[set][gg * TN + j][e], acc[i][j],
                                                               0, 0, 0);
    } else {
      // partial products la + lb <= NL - 1 (orders 2^0, 2^-8, 2^-16), smallest first
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int ord = NL - 1; ord >= 0; --ord)
#pragma unroll
            for (int la = 0; la <= ord; ++la) {
              if constexpr (PREC == 3)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[set][la * TM + i]),
                                                                   __builtin_bit_cast(f16x8, fb[set][(ord - la) * TN + j]),
                                                                   acc[i][j], 0, 0, 0);
              else
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[set][la * TM + i]),
                                                                    __builtin_bit_cast(bf16x8, fb[set][(ord - la) * TN + j]),
                                                                    acc[i][j], 0, 0, 0);
            }
    }
  };
  auto issue_load = [&]() {
    const uint32_t kcb = (uint32_t)kc_cur * 4u;
#pragma unroll
    for (int i = 0; i < AR; ++i) areg[i] = buf_load4(xr, a_off[i] + kcb);
    const bool kvalid = k_cur < p.K;
    const uint32_t kb = (uint32_t)k_cur * 4u;
#pragma unroll
    for (int i = 0; i < BR; ++i) breg[i] = buf_load4(wr, kvalid ? b_offb[i] + kb : kOOB);
    k_cur += BK;
    kc_cur += BK;
    kt_cur += 1;
  };
  {
    const int t0 = next_valid(kt_begin);
    if (t0 < kt_end) {
      load_tile(t0);
      store_tile(0);
      int cur = t0, nxt = next_after(t0), stg = kt_end;
      if (nxt < kt_end) {
        load_tile(nxt);
        store_tile(1);
        stg = next_after(nxt);
        if (stg < kt_end) load_tile(stg);
      } else {
        nxt = kt_end;
      }
      __syncthreads();
      read_frags(0, 0, 0);
      int buf = 0;
      bool dead = false;
      while (cur < kt_end) {
        const int after = stg < kt_end ? next_after(stg) : kt_end;
        if (after < kt_end) {
          if (SKIP && after != kt_cur) {
            seek(after);
          } else if (kc_cur >= p.Cin) {
            do {
              kc_cur -= p.Cin;
              if (++ks_cur == p.kw) {
                ks_cur = 0;
                ++kr_cur;
              }
            } while (kc_cur >= p.Cin);
            retap();
          }
        } else if (!dead) {
          dead = true;
#pragma unroll
          for (int i = 0; i < AR; ++i) a_off[i] = kOOB;
          k_cur = p.K;
          kc_cur = 0;
        }
        read_frags(1, buf, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        read_frags(0, buf ^ 1, 0);   // (garbage on the last step: never multiplied)
        __builtin_amdgcn_sched_barrier(0);
        store_tile(buf);
        issue_load();
        mfmas(1);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
        nxt = stg;
        stg = after;
        buf ^= 1;
      }
      __syncthreads();   // the epilogue reuses the staging buffers as output patches
    }
  }

  if constexpr (PREC == 3) {  // undo the operand scaling (exact: a power of two)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] *= unscale;
  }

  // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  // (the main loop ended with a barrier: the staging buffers are free, each wave takes a private patch)
  float* yout = p.y + (long long)blockIdx.z * p.slab_stride;
  const int col_l = lane & 31;
  const int row_h = (lane >> 5) * 4;
  {
    float* patch = lds + wave * (WTM * (WTN + 4));
    const int row0 = m0 + wm * WTM, col0 = n0 + wn * WTN;
    int rv = p.M - row0, cv = p.N - col0;
    rv = rv < 0 ? 0 : (rv > WTM ? WTM : rv);
    cv = cv < 0 ? 0 : (cv > WTN ? WTN : cv);
    store_tiles<TM, TN>(acc, patch, yout, p.ldy, row0, col0, rv, cv, p.bias, p.accumulate != 0, lane, [&](int m) {
      if (!p.row_perm) return m;
      int b, ho, wo;
      row_to_pixel(p, m, b, ho, wo);
      return (b * p.Ho + ho) * p.Wo + wo;
    });
  }

  // ---- fused BatchNorm batch statistics: one row group per (M tile, wave row), shifted by the group's first
  // sample so that sum((y-K)^2) does not cancel when |mean| >> std.  No LDS, no barrier.
  if (p.stat != nullptr) {
    const int group = tile_m * WARPS_M + wm;
    const long long gsz = (long long)p.stat_rows * p.N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * WTN + j * 32 + col_l;
      // row 0 of this wave's sub-tile lives in register 0 of the lanes with (lane>>5) == 0
      const float k0 = __shfl(acc[0][j][0], lane & 31, 64);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
          if (row < p.M) {
            const float d = acc[i][j][r] - k0;
            s1 += d;
            s2 += d * d;
          }
        }
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (lane < 32 && col < p.N) {
        const long long o = (long long)group * p.N + col;
        p.stat[o] = k0;
        p.stat[gsz + o] = s1;
        p.stat[2 * gsz + o] = s2;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Split-bf16 gather GEMM on PRE-SPLIT operands, staged by LDS-DMA.
//
// The limb variants of gather_conv_kernel spend 4-7 VALU instructions and one ds_write per MFMA on splitting fp32
// operands into bf16 limbs in every consumer (measured: MFMA pipe busy 0.25-0.5, issue-bound).  Here the producers split
// ONCE (pseg_split_planes / the BatchNorm backward pass emit bf16 hi and lo planes, 4 bytes per element like the fp32
// tensor) and the tiles go global -> LDS with `buffer_load_dwordx4 ... lds`: no staging registers, no VALU, no ds_write.
// The swizzled LDS image (16-byte k-slot XOR (row>>2)&3, conflict-free ds_read_b128) is produced on the SOURCE side: a
// wave-instruction fills 16 rows x 64 bytes of LDS linearly, lane l lands at (row l/4, physical slot l%4) and therefore
// fetches logical slot (l%4) ^ ((row>>2)&3) of its row.  Zero padding = out-of-range buffer offset = zeros in LDS.
//
// 256x128 tile, 8 waves (4x2), K-step 32, THREE LDS stages of 48 KB (A hi/lo 256 rows + B hi/lo 128 rows, 64 B per row):
// tile i is multiplied while i+1 and i+2 are in flight; per K-step one counted s_waitcnt vmcnt(6) (leaves the 6 DMAs of
// tile i+2 outstanding), one raw s_barrier, then the 6 DMAs of tile i+3 go into the stage that barrier freed.  An
// exhausted stream keeps issuing out-of-range DMAs so that the count stays exact.
// Requirements (host-checked): channels of the gathered tensor % 32 == 0 (a K-step never straddles a tap), no split-K.
constexpr int kDmaBM = 256, kDmaBN = 128;
constexpr int kDmaStageDw = (2 * kDmaBM + 2 * kDmaBN) * 16;   // dwords per stage: A_hi, A_lo, B_hi, B_lo

template <bool SKIP>
__global__ __launch_bounds__(512) void gather_limb_dma_kernel(const GatherConvParams p) {
  set_wave_prio(p.prio);
  constexpr int BM = kDmaBM, BN = kDmaBN, WARPS_N = 2;
  constexpr int WTM = 64, WTN = 64, TM = 2, TN = 2, NL = 2;
  constexpr int kPatch = 8 * WTM * (WTN + 4);
  constexpr int kLds = 3 * kDmaStageDw > kPatch ? 3 * kDmaStageDw : kPatch;
  __shared__ __attribute__((aligned(16))) float lds[kLds];
  unsigned* ldsw = reinterpret_cast<unsigned*>(lds);
  // plane bases inside a stage (dwords)
  constexpr int kAhi = 0, kAlo = BM * 16, kBhi = 2 * BM * 16, kBlo = 2 * BM * 16 + BN * 16;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;
  const int gridN = (p.N + BN - 1) / BN;
  int bid = blockIdx.x;
  bid = remap_tile(p.xcd_remap, bid, (int)gridDim.x);
  const int tile_n = bid % gridN;
  const int tile_m = bid / gridN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t xhr = make_rsrc(p.xh, p.xp_bytes), xlr = make_rsrc(p.xl, p.xp_bytes);
  const __amdgpu_buffer_rsrc_t whr = make_rsrc(p.wh, p.wp_bytes), wlr = make_rsrc(p.wl, p.wp_bytes);

  // ---- loader roles: wave w fills A rows [32w, 32w+32) (two 16-row groups) and B rows [16w, 16w+16), hi and lo
  const int lrow = lane >> 2, lslot = lane & 3;
  int a_bh[2], a_bw[2], a_img[2];
  bool a_ok[2];
  int a_slot_b[2];   // byte offset of this lane's LOGICAL k-slot inside a 32-deep K-step (8 bf16 = 16 B per slot)
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int row = 32 * wave + 16 * g + lrow;
    const int m = m0 + row;
    const bool ok = m < p.M;
    int b, ho, wo;
    row_to_pixel(p, ok ? m : 0, b, ho, wo);
    a_ok[g] = ok;
    a_bh[g] = ho * p.s_out + p.off0;
    a_bw[g] = wo * p.s_out + p.off0;
    a_img[g] = b * p.Hi * p.Wi;
    a_slot_b[g] = (lslot ^ ((row >> 2) & 3)) * 16;
  }
  const int b_row = 16 * wave + lrow;
  const bool b_ok = (n0 + b_row) < p.N;
  const uint32_t b_rowoff = b_ok ? (uint32_t)(n0 + b_row) * (uint32_t)p.K * 2u + (uint32_t)((lslot ^ ((b_row >> 2) & 3)) * 16) : kOOB;

  auto row_tap_ok = [&](int g, int dh, int dw, int& hn, int& wn_) -> bool {
    hn = a_bh[g] + dh;
    wn_ = a_bw[g] + dw;
    bool ok = a_ok[g];
    if (p.s_in != 1) {
      ok = ok && (hn % p.s_in == 0) && (wn_ % p.s_in == 0);
      hn /= p.s_in;
      wn_ /= p.s_in;
    }
    return ok && ((unsigned)hn < (unsigned)p.Hi) && ((unsigned)wn_ < (unsigned)p.Wi);
  };

  const int kt_end = p.kt_total;
  unsigned tapmask = 0xFFFFFFFFu;
  if (SKIP) {
    unsigned mine = 0;
    for (int t = 0; t < p.ntaps; ++t) {
      const int r = t / p.kw, sx = t - r * p.kw;
      int hn, wn_;
      const bool any = row_tap_ok(0, r * p.dstep, sx * p.dstep, hn, wn_) || row_tap_ok(1, r * p.dstep, sx * p.dstep, hn, wn_);
      if (any) mine |= 1u << t;
    }
    if (tid == 0) ldsw[0] = 0u;
    __syncthreads();
    if (mine) atomicOr(&ldsw[0], mine);
    __syncthreads();
    tapmask = ldsw[0];
    __syncthreads();
  }
  // ---- the stream of live K-steps, TAP major (all 32-channel chunks of a live tap, then the next live tap).
  // (Channel-chunk-major order -- the taps of one chunk back to back, so that the shifted re-reads of a tile's pixels hit
  // L1 / L2 -- was measured SLOWER, 370 -> 348 TF on the 512->512 3x3 layer: consecutive K-steps of a filter row then lie
  // Cout*2 bytes apart and every 64-byte k-slot costs its own half-used 128-byte line.)
  tapmask &= (p.ntaps >= 32) ? 0xFFFFFFFFu : ((1u << p.ntaps) - 1u);
  const int nlive = __builtin_popcount(tapmask);
  const int n_steps = nlive * p.ktiles_per_tap;
  int s_chunk = 0;
  unsigned s_tm = tapmask;
  auto next_kt = [&]() -> int {     // next K-step of the stream, kt_end when exhausted
    if (s_tm == 0u) return kt_end;
    const int kt = __builtin_ctz(s_tm) * p.ktiles_per_tap + s_chunk;
    if (++s_chunk == p.ktiles_per_tap) {
      s_chunk = 0;
      s_tm &= s_tm - 1u;
    }
    return kt;
  };

  // ---- DMA issue of K-step kt into stage st (kt >= kt_end: all-zero dummy tile, keeps the vmcnt arithmetic exact)
  int tap_cur = -1;
  uint32_t a_off[2] = {kOOB, kOOB};
  auto issue = [&](int kt, int st) {
    uint32_t ao[2], bo;
    if (kt < kt_end) {
      const int tap = kt / p.ktiles_per_tap;
      if (tap != tap_cur) {
        tap_cur = tap;
        const int kr = tap / p.kw, ks = tap - kr * p.kw;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          int hn, wn_;
          const bool ok = row_tap_ok(g, kr * p.dstep, ks * p.dstep, hn, wn_);
          a_off[g] = ok ? (uint32_t)((a_img[g] + hn * p.Wi + wn_) * p.ldxp) * 2u + (uint32_t)a_slot_b[g] : kOOB;
        }
      }
      const uint32_t kc_b = (uint32_t)((kt - tap * p.ktiles_per_tap) * BK) * 2u;
      ao[0] = a_off[0] + kc_b;     // kOOB + kc_b stays out of range
      ao[1] = a_off[1] + kc_b;
      bo = b_rowoff + (uint32_t)kt * (uint32_t)(BK * 2);
    } else {
      ao[0] = ao[1] = bo = kOOB;
    }
    unsigned* sb = ldsw + st * kDmaStageDw;
    typedef __attribute__((address_space(3))) void* lds_ptr;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int ro = (32 * wave + 16 * g) * 16;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xhr, (lds_ptr)(sb + kAhi + ro), 16, (int)ao[g], 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xlr, (lds_ptr)(sb + kAlo + ro), 16, (int)ao[g], 0, 0, 0);
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(whr, (lds_ptr)(sb + kBhi + 16 * wave * 16), 16, (int)bo, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wlr, (lds_ptr)(sb + kBlo + 16 * wave * 16), 16, (int)bo, 0, 0, 0);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frag_row = lane & 31;
  f32x4 fa[2][NL * TM], fb[2][NL * TN];
  auto read_frags = [&](int set, int st, int half) {
    const int slot = half * 2 + (lane >> 5);
    const unsigned* sb = ldsw + st * kDmaStageDw;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int o = swz(wm * WTM + i * 32 + frag_row, slot);
      fa[set][0 * TM + i] = *reinterpret_cast<const f32x4*>(&sb[kAhi + o]);
      fa[set][1 * TM + i] = *reinterpret_cast<const f32x4*>(&sb[kAlo + o]);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int o = swz(wn * WTN + j * 32 + frag_row, slot);
      fb[set][0 * TN + j] = *reinterpret_cast<const f32x4*>(&sb[kBhi + o]);
      fb[set][1 * TN + j] = *reinterpret_cast<const f32x4*>(&sb[kBlo + o]);
    }
  };
  auto mfmas = [&](int set) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int ord = NL - 1; ord >= 0; --ord)
#pragma unroll
          for (int la = 0; la <= ord; ++la)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[set][la * TM + i]),
                                                                __builtin_bit_cast(bf16x8, fb[set][(ord - la) * TN + j]),
                                                                acc[i][j], 0, 0, 0);
  };

  {
    // tile i is multiplied while i+1 and i+2 are in flight and i+3 is issued
    if (n_steps > 0) {
      issue(next_kt(), 0);
      issue(next_kt(), 1);
      issue(next_kt(), 2);
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // tile 0 has landed (this wave's share)
      __builtin_amdgcn_s_barrier();                        // ... and everybody's
      read_frags(0, 0, 0);
      int st = 0;
      for (int it = 0; it < n_steps; ++it) {
        const int st1 = st == 2 ? 0 : st + 1;
        read_frags(1, st, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // the tile after this one has landed; one more stays in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wave is done reading stage `st`
        __builtin_amdgcn_s_barrier();
        read_frags(0, st1, 0);      // (zeros on the last step: never multiplied)
        __builtin_amdgcn_sched_barrier(0);
        issue(next_kt(), st);       // stage `st` is free now
        mfmas(1);
        __builtin_amdgcn_sched_barrier(0);
        st = st1;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // dummy DMAs must not land in the output patches
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }

  // ---- epilogue (the stages are free: each wave takes a private patch)
  float* yout = p.y;
  {
    float* patch = lds + wave * (WTM * (WTN + 4));
    const int row0 = m0 + wm * WTM, col0 = n0 + wn * WTN;
    int rv = p.M - row0, cv = p.N - col0;
    rv = rv < 0 ? 0 : (rv > WTM ? WTM : rv);
    cv = cv < 0 ? 0 : (cv > WTN ? WTN : cv);
    store_tiles<TM, TN>(acc, patch, yout, p.ldy, row0, col0, rv, cv, nullptr, p.accumulate != 0, lane, [&](int m) {
      if (!p.row_perm) return m;
      int b, ho, wo;
      row_to_pixel(p, m, b, ho, wo);
      return (b * p.Ho + ho) * p.Wo + wo;
    });
  }
}

// ------------------------------------------------------------------------------------------------
// Exact-fp32 gather GEMM staged by LDS-DMA: the same K-step stream, three-stage ring and barrier / vmcnt protocol as
// gather_limb_dma_kernel, on the fp32 tensors themselves (no planes needed) and v_mfma_f32_32x32x2_f32.
// LDS image: [rows][32 floats = 128 B = eight 16-byte k-slots], slot XOR ((row >> 1) & 7): the 16-lane groups of the
// fragment ds_read_b128 (rows {0-3,12-15,20-27} of a 32-row fragment, one slot) then cover all sixteen (row & 1, slot)
// bank groups once.  A wave-instruction of the DMA fills 8 rows x 128 B linearly: lane l = (row l/8, physical slot l%8)
// fetches logical slot (l%8) ^ ((row>>1)&7) of its row.  Compared with gather_conv_kernel<..., 0> there are no staging
// registers, no ds_write_b128, no per-K-step address VALU, and the prefetch is two tiles deep.
// Tiles: 256x128 / 128x128 with 8 waves (one block per CU), 128x64 / 64x128 with 4 waves (two blocks per CU).
// Requirements (host-checked): channels of the gathered tensor % 32 == 0, no split-K.
// GENERIC (round 5): channel counts that are NOT a multiple of the K-step (the 7x7 stem on 4 channels, the classifier's data
// gradient on 24): a K-step then straddles taps, and every lane derives (tap, channel) of its own 16-byte k-slot -- K-slot index
// / slots per tap -- once per K-step and row group (two divisions by multiply-shift; ~40 VALU instructions per wave and K-step
// beside 2048 matrix cycles).  Every tap is issued (no skipping); slots past K read nothing.
template <int BM, int BN, int WARPS_M, int WARPS_N, bool SKIP, int STAGES = 3, bool GENERIC = false>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N) void gather_f32_dma_kernel(const GatherConvParams p) {
  static_assert(!(SKIP && GENERIC), "tap skipping needs whole K-steps per tap");
  set_wave_prio(p.prio);
  static_assert(STAGES == 2 || STAGES == 3, "ring depth");
  constexpr int NW = WARPS_M * WARPS_N, NT = 64 * NW;
  static_assert(NW == 8 || NW == 4, "8 or 4 waves");
  constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N, TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1 && WTM % 32 == 0 && WTN % 32 == 0, "wave tile");
  constexpr int kStageDw = (BM + BN) * 32;
  constexpr int kPatch = NW * WTM * (WTN + 4);
  constexpr int kLds = STAGES * kStageDw > kPatch ? STAGES * kStageDw : kPatch;
  __shared__ __attribute__((aligned(16))) float lds[kLds];
  unsigned* ldsw = reinterpret_cast<unsigned*>(lds);
  constexpr int kA = 0, kB = BM * 32;
  // DMA row groups (8 rows each): A groups [0, BM/8), then B groups; wave w owns groups w, w + NW, ...
  constexpr int GA = BM / 8 / NW, GB = BN / 8 / NW, NG = GA + GB;
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "whole row groups per wave");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;
  const int gridN = (p.N + BN - 1) / BN;
  int bid = blockIdx.x;
  bid = remap_tile(p.xcd_remap, bid, (int)gridDim.x);
  const int tile_n = bid % gridN;
  const int tile_m = bid / gridN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.w, p.w_bytes);

  const int lrow = lane >> 3, lslot = lane & 7;
  int a_bh[GA], a_bw[GA], a_img[GA], a_slot_b[GA];
  bool a_ok[GA];
#pragma unroll
  for (int g = 0; g < GA; ++g) {
    const int row = 8 * (wave + NW * g) + lrow;
    const int m = m0 + row;
    const bool ok = m < p.M;
    int b, ho, wo;
    row_to_pixel(p, ok ? m : 0, b, ho, wo);
    a_ok[g] = ok;
    a_bh[g] = ho * p.s_out + p.off0;
    a_bw[g] = wo * p.s_out + p.off0;
    a_img[g] = b * p.Hi * p.Wi;
    a_slot_b[g] = (lslot ^ ((row >> 1) & 7)) * 16;
  }
  uint32_t b_rowoff[GB];
  int b_slot[GB];      // GENERIC: logical k-slot of this lane in row group g (the filter row ends at K: slots past it read nothing)
#pragma unroll
  for (int g = 0; g < GB; ++g) {
    const int row = 8 * (wave + NW * g) + lrow;
    b_slot[g] = lslot ^ ((row >> 1) & 7);
    b_rowoff[g] = (n0 + row) < p.N ? (uint32_t)(n0 + row) * (uint32_t)p.K * 4u + (uint32_t)(b_slot[g] * 16) : kOOB;
  }

  auto row_tap_ok = [&](int g, int dh, int dw, int& hn, int& wn_) -> bool {
    hn = a_bh[g] + dh;
    wn_ = a_bw[g] + dw;
    bool ok = a_ok[g];
    if (p.s_in != 1) {
      ok = ok && (hn % p.s_in == 0) && (wn_ % p.s_in == 0);
      hn /= p.s_in;
      wn_ /= p.s_in;
    }
    return ok && ((unsigned)hn < (unsigned)p.Hi) && ((unsigned)wn_ < (unsigned)p.Wi);
  };

  const int kt_end = p.kt_total;
  unsigned tapmask = 0xFFFFFFFFu;
  if (SKIP) {
    unsigned mine = 0;
    for (int t = 0; t < p.ntaps; ++t) {
      const int r = t / p.kw, sx = t - r * p.kw;
      bool any = false;
#pragma unroll
      for (int g = 0; g < GA; ++g) {
        int hn, wn_;
        any = any || row_tap_ok(g, r * p.dstep, sx * p.dstep, hn, wn_);
      }
      if (any) mine |= 1u << t;
    }
    if (tid == 0) ldsw[0] = 0u;
    __syncthreads();
    if (mine) atomicOr(&ldsw[0], mine);
    __syncthreads();
    tapmask = ldsw[0];
    __syncthreads();
  }
  tapmask &= (p.ntaps >= 32) ? 0xFFFFFFFFu : ((1u << p.ntaps) - 1u);
  const int n_steps = GENERIC ? kt_end : __builtin_popcount(tapmask) * p.ktiles_per_tap;
  int s_chunk = 0, g_next = 0;
  unsigned s_tm = tapmask;
  auto next_kt = [&]() -> int {     // next live K-step (tap major), kt_end when exhausted
    if (GENERIC) return g_next < kt_end ? g_next++ : kt_end;      // (K-steps in order: they straddle taps)
    if (s_tm == 0u) return kt_end;
    const int kt = __builtin_ctz(s_tm) * p.ktiles_per_tap + s_chunk;
    if (++s_chunk == p.ktiles_per_tap) {
      s_chunk = 0;
      s_tm &= s_tm - 1u;
    }
    return kt;
  };

  int tap_cur = -1;
  uint32_t a_off[GA];
#pragma unroll
  for (int g = 0; g < GA; ++g) a_off[g] = kOOB;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  auto issue = [&](int kt, int st) {
    uint32_t ao[GA], bo[GB];
    if (GENERIC && kt < kt_end) {
      const int spt = p.Cin >> 2;                     // 16-byte slots per tap
#pragma unroll
      for (int g = 0; g < GA; ++g) {
        const uint32_t s = (uint32_t)kt * 8u + (uint32_t)(a_slot_b[g] >> 4);      // logical k-slot of this lane
        const uint32_t tap = p.gen_spt.div(s);
        const uint32_t ch = (s - tap * (uint32_t)spt) * 4u;
        const uint32_t kr = p.gen_kw.div(tap), ks = tap - kr * (uint32_t)p.kw;
        int hn, wn_;
        const bool ok = (int)tap < p.ntaps && row_tap_ok(g, (int)kr * p.dstep, (int)ks * p.dstep, hn, wn_);
        ao[g] = ok ? (uint32_t)((a_img[g] + hn * p.Wi + wn_) * p.ldx + (int)ch) * 4u : kOOB;
      }
#pragma unroll
      for (int g = 0; g < GB; ++g) {
        const bool in_k = (kt * 8 + b_slot[g]) * 4 < p.K;
        bo[g] = in_k ? b_rowoff[g] + (uint32_t)kt * (uint32_t)(BK * 4) : kOOB;
      }
    } else if (kt < kt_end) {
      const int tap = kt / p.ktiles_per_tap;
      if (tap != tap_cur) {
        tap_cur = tap;
        const int kr = tap / p.kw, ks = tap - kr * p.kw;
#pragma unroll
        for (int g = 0; g < GA; ++g) {
          int hn, wn_;
          const bool ok = row_tap_ok(g, kr * p.dstep, ks * p.dstep, hn, wn_);
          a_off[g] = ok ? (uint32_t)((a_img[g] + hn * p.Wi + wn_) * p.ldx) * 4u + (uint32_t)a_slot_b[g] : kOOB;
        }
      }
      const uint32_t kc_b = (uint32_t)((kt - tap * p.ktiles_per_tap) * BK) * 4u;
#pragma unroll
      for (int g = 0; g < GA; ++g) ao[g] = a_off[g] + kc_b;     // kOOB + kc_b stays out of range
#pragma unroll
      for (int g = 0; g < GB; ++g) bo[g] = b_rowoff[g] + (uint32_t)kt * (uint32_t)(BK * 4);
    } else {
#pragma unroll
      for (int g = 0; g < GA; ++g) ao[g] = kOOB;
#pragma unroll
      for (int g = 0; g < GB; ++g) bo[g] = kOOB;
    }
    unsigned* sb = ldsw + st * kStageDw;
#pragma unroll
    for (int g = 0; g < GA; ++g)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr)(sb + kA + 8 * (wave + NW * g) * 32), 16, (int)ao[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < GB; ++g)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr)(sb + kB + 8 * (wave + NW * g) * 32), 16, (int)bo[g], 0, 0, 0);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frag_row = lane & 31;
  const int frag_h = lane >> 5;
  auto swz32 = [](int row, int slot) -> int { return row * 32 + ((slot ^ ((row >> 1) & 7)) << 2); };
  f32x4 fa[2][2 * TM], fb[2][2 * TN];   // [set][gg * T + tile]: k-quads gg of a 16-deep half-step
  auto read_frags = [&](int set, int st, int half) {
    const float* sb = lds + st * kStageDw;
#pragma unroll
    for (int gg = 0; gg < 2; ++gg) {
      const int slot = 2 * (half * 2 + gg) + frag_h;
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[set][gg * TM + i] = *reinterpret_cast<const f32x4*>(&sb[kA + swz32(wm * WTM + i * 32 + frag_row, slot)]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[set][gg * TN + j] = *reinterpret_cast<const f32x4*>(&sb[kB + swz32(wn * WTN + j * 32 + frag_row, slot)]);
    }
  };
  auto mfmas = [&](int set) {
#pragma unroll
    for (int gg = 0; gg < 2; ++gg)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][gg * TM + i][e], fb[set][gg * TN + j][e], acc[i][j], 0, 0,
                                                             0);
  };
  // counted waits: NG DMAs per tile and wave
  // (STAGES - 1) tiles stay in flight after the prologue wait, (STAGES - 2) inside the loop
  auto wait_two_left = [&]() {
    constexpr int left = (STAGES - 1) * NG;
    if constexpr (left == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (left == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (left == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (left == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (left == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (left == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (left == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto wait_one_left = [&]() {
    constexpr int left = (STAGES - 2) * NG;
    if constexpr (left == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (left == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (left == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (left == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  static_assert(NG == 6 || NG == 5 || NG == 4 || NG == 3, "vmcnt immediates");

  unsigned long long tr0 = 0, tr1 = 0, tr2 = 0;
  if (kConvTrace && p.trace != nullptr) tr0 = wall_clock64();
  if (n_steps > 0) {
    issue(next_kt(), 0);
    issue(next_kt(), 1);
    if constexpr (STAGES == 3) issue(next_kt(), 2);
    wait_two_left();                   // tile 0 has landed (this wave's share)
    __builtin_amdgcn_s_barrier();      // ... and everybody's
    if (kConvTrace && p.trace != nullptr) tr1 = wall_clock64();
    read_frags(0, 0, 0);
    int st = 0;
    for (int it = 0; it < n_steps; ++it) {
      const int st1 = st == STAGES - 1 ? 0 : st + 1;
      read_frags(1, st, 1);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(0);
      __builtin_amdgcn_sched_barrier(0);
      wait_one_left();                                     // the next tile has landed; one more stays in flight
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave is done reading stage `st`
      __builtin_amdgcn_s_barrier();
      read_frags(0, st1, 0);      // (zeros on the last step: never multiplied)
      __builtin_amdgcn_sched_barrier(0);
      issue(next_kt(), st);       // stage `st` is free now
      mfmas(1);
      __builtin_amdgcn_sched_barrier(0);
      st = st1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // dummy DMAs must not land in the output patches
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  if (kConvTrace && p.trace != nullptr) tr2 = wall_clock64();
  // ---- epilogue: as gather_conv_kernel (bias / accumulate / row map, fused BatchNorm statistics)
  const int col_l = lane & 31;
  const int row_h = (lane >> 5) * 4;
  {
    float* patch = lds + wave * (WTM * (WTN + 4));
    const int row0 = m0 + wm * WTM, col0 = n0 + wn * WTN;
    int rv = p.M - row0, cv = p.N - col0;
    rv = rv < 0 ? 0 : (rv > WTM ? WTM : rv);
    cv = cv < 0 ? 0 : (cv > WTN ? WTN : cv);
    auto out_row = [&](int m) {
      if (!p.row_perm) return m;
      int b, ho, wo;
      row_to_pixel(p, m, b, ho, wo);
      return (b * p.Ho + ho) * p.Wo + wo;
    };
    if (p.bns_y != nullptr) {
      // data gradient that is the dz of a BatchNorm layer: its backward partial sums ride on the store loop
      const BnsEpilogue be{p.bns_y, p.bns_ldy, p.bns_mean, p.bns_invstd, p.bns_scale, p.bns_shift, p.bns_act, p.bns_db, p.bns_dg,
                           (long long)(tile_m * WARPS_M + wm) * p.N};
      store_tiles<TM, TN, false, true>(acc, patch, p.y, p.ldy, row0, col0, rv, cv, p.bias, p.accumulate != 0, lane, out_row, &be);
    } else {
      store_tiles<TM, TN>(acc, patch, p.y, p.ldy, row0, col0, rv, cv, p.bias, p.accumulate != 0, lane, out_row);
    }
  }
  if (p.stat != nullptr) {
    const int group = tile_m * WARPS_M + wm;
    const long long gsz = (long long)p.stat_rows * p.N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * WTN + j * 32 + col_l;
      const float k0 = __shfl(acc[0][j][0], lane & 31, 64);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
          if (row < p.M) {
            const float d = acc[i][j][r] - k0;
            s1 += d;
            s2 += d * d;
          }
        }
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (lane < 32 && col < p.N) {
        const long long o = (long long)group * p.N + col;
        p.stat[o] = k0;
        p.stat[gsz + o] = s1;
        p.stat[2 * gsz + o] = s2;
      }
    }
  }
  if (kConvTrace && p.trace != nullptr && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the stores of this wave have left
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned long long* t = p.trace + (long long)blockIdx.x * 5;
    t[0] = tr0;
    t[1] = tr1;
    t[2] = tr2;
    t[3] = wall_clock64();
    t[4] = hwid;
  }
}

// ------------------------------------------------------------------------------------------------
// HALO-STAGED 3x3 for NARROW outputs, exact fp32 (round 5; VERDICT r4 item 4 / item 2 "classifier kernel").  A 128x32 tile of
// gather_f32_dma_kernel pulls its A operand once per TAP: 16 KB of pixels + 4 KB of filter per 32-channel K-step for 1024 matrix
// cycles per SIMD -- 20 bytes per clock and CU through an LDS-DMA path that sustains ~27: with a 32-column tile the operand
// STREAM, not the matrix pipe, is the bound (HRNet's 32 -> 32 3x3 on 128x128 maps: 32 us for 18 us of matrix work; the classifier).
// Here an M tile is an 8 x 16 PATCH of output pixels and a 32-channel chunk of the A operand is DMA'd ONCE, as the 10 x 18 halo patch
// (23 KB; out-of-image pixels read as zeros = the conv's padding); the nine taps read their A fragments from it at shifted pixel
// rows.  The filter slice of the chunk runs through a three-stage ring of single taps ([BN columns][32 channels], 4 / 8 KB): one
// barrier per tap, the A fragments of the next tap are read before it (the halo image does not change inside a chunk).
// LDS 35 KB (32 columns: FOUR 4-wave blocks per CU) / 47 KB (64 columns: three).  A first form kept the chunk's whole filter slice
// resident (60 KB, two blocks per CU): faster alone, SLOWER in the steps, whose narrow layers run beside other kernels (HRNet's five
// replay lanes) -- residency is what they compete for (profiles/EXPERIMENTS.md 5.9).
//   halo image: row = halo pixel (hr * 18 + hc), 128 B = eight 16-byte k-slots, slot XOR ((hc >> 1) & 7).  A fragment ds_read_b128
//   serves lanes in groups {0-3, 12-15, 20-27} ...: 16 pixels of consecutive columns (two patch rows), i.e. 16 consecutive hc at
//   every tap, and 18 is even, so (LDS row parity, slot) = (hc & 1, slot ^ (hc >> 1) & 7) takes all sixteen values: conflict-free.
//   filter ring: row = stage * BN + column, the ring kernels' swizzle.
// Wave layouts = the ring kernel's for the same plan tile (so the statistics groups are the plan's): 128x32 on 4 x 1 waves (a wave:
// 2 patch rows x 32 columns), 128x64 on 2 x 2 (4 patch rows x 32 columns, two accumulators).
// Covers: 3x3, unit stride, dilation 1, padding 1 (forward and data gradient: the same gather with the taps reversed), channels of
// the gathered tensor % 32 == 0, maps of whole 8 x 16 patches (rows in patch order: row_perm 2); bias / accumulate / fused
// BatchNorm statistics / BatchNorm-backward sums as the ring kernel.  PSEG_CONV_HALO=0: off.
constexpr int kHaloPH = 8, kHaloPW = 16, kHaloHC = kHaloPW + 2, kHaloRows = (kHaloPH + 2) * kHaloHC;      // 180
constexpr int kHaloRowsPad = (kHaloRows + 7) / 8 * 8;                                                     // 184
template <int WARPS_M, int WARPS_N>
__global__ __launch_bounds__(256, WARPS_N == 1 ? 4 : 3) void gather_f32_halo_kernel(const GatherConvParams p) {
  set_wave_prio(p.prio);
  constexpr int NW = WARPS_M * WARPS_N;
  static_assert(NW == 4, "four waves");
  constexpr int BN = 32 * WARPS_N, WTM = 128 / WARPS_M, TM = WTM / 32;
  constexpr int kEpiDw = NW * WTM * 36;
  constexpr int kA = 0, kB = kHaloRowsPad * 32, kRing = kB + 3 * BN * 32, kLdsDw = kRing > kEpiDw ? kRing : kEpiDw;
  constexpr int GA = (kHaloRowsPad / 8 + NW - 1) / NW;      // 6 row groups of the halo image per wave (the last wave: 5)
  constexpr int GB = BN / 8 / NW;                            // row groups of one tap of the filter ring per wave (1 / 2)
  __shared__ __attribute__((aligned(16))) float lds[kLdsDw];
  unsigned* ldsw = reinterpret_cast<unsigned*>(lds);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;
  const int gridN = (p.N + BN - 1) / BN;
  int bid = blockIdx.x;
  bid = remap_tile(p.xcd_remap, bid, (int)gridDim.x);
  const int tile_n = bid % gridN;
  const int tile_m = bid / gridN;
  const int m0 = tile_m * 128, n0 = tile_n * BN;
  // the patch of this tile (rows are in patch order: row_perm 2 with 8 x 16 patches)
  const int img = m0 / p.HoWo;
  const int patch = (m0 - img * p.HoWo) / (kHaloPH * kHaloPW);
  const int ph = patch / p.patches_per_row, pw = patch - ph * p.patches_per_row;
  const int h0 = ph * kHaloPH - 1, w0 = pw * kHaloPW - 1;     // image position of halo pixel (0, 0)

  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.w, p.w_bytes);
  const int lrow = lane >> 3, lslot = lane & 7;
  uint32_t a_off[GA], b_off[GB];
#pragma unroll
  for (int g = 0; g < GA; ++g) {
    const int idx = 8 * (wave + NW * g) + lrow;
    const int hr = idx / kHaloHC, hc = idx - hr * kHaloHC;
    const int y = h0 + hr, x = w0 + hc;
    const bool ok = idx < kHaloRows && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
    const int slot = lslot ^ ((hc >> 1) & 7);
    a_off[g] = ok ? (uint32_t)(((img * p.Hi + y) * p.Wi + x) * p.ldx) * 4u + (uint32_t)(slot * 16) : kOOB;
  }
#pragma unroll
  for (int g = 0; g < GB; ++g) {
    const int n = 8 * (wave + NW * g) + lrow;         // column of the tile; its row of whichever tap is issued
    const int slot = lslot ^ ((n >> 1) & 7);
    b_off[g] = (n0 + n) < p.N ? (uint32_t)(n0 + n) * (uint32_t)p.K * 4u + (uint32_t)(slot * 16) : kOOB;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr;
  auto issue_a = [&](int chunk) {
    const uint32_t kc = (uint32_t)chunk * 128u;
#pragma unroll
    for (int g = 0; g < GA; ++g)
      if (wave + NW * g < kHaloRowsPad / 8)      // (wave-uniform)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr)(ldsw + kA + 8 * (wave + NW * g) * 32), 16, (int)(a_off[g] + kc), 0, 0, 0);
  };
  auto issue_b = [&](int chunk, int t) {     // tap t of the chunk into stage t % 3
    const uint32_t kc = (uint32_t)chunk * 128u + (uint32_t)(t * p.Cin) * 4u;
#pragma unroll
    for (int g = 0; g < GB; ++g)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr)(ldsw + kB + ((t % 3) * BN + 8 * (wave + NW * g)) * 32), 16,
                                               (int)(b_off[g] == kOOB ? kOOB : b_off[g] + kc), 0, 0, 0);
  };

  f32x16 acc[TM][1];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

  // fragment addressing: this lane's pixel in each of the wave's 2 x 16 strips, its halo row at tap offset (0, 0), the swizzle
  // terms of the three column offsets
  const int frag_row = lane & 31, frag_h = lane >> 5;
  const int pcol = frag_row & 15;
  const int hbase = (2 * wm * TM + (frag_row >> 4) + 1) * kHaloHC + (pcol + 1);      // strip i: + 2 i rows
  int a_col_dw[3], a_swz[3];
#pragma unroll
  for (int ts = 0; ts < 3; ++ts) {
    const int os = p.off0 + ts * p.dstep;          // -1, 0, +1 (forward) or +1, 0, -1 (data gradient)
    a_col_dw[ts] = os * 32;
    a_swz[ts] = ((pcol + 1 + os) >> 1) & 7;
  }
  const int b_frag = (wn * 32 + frag_row) * 32, b_swz = (frag_row >> 1) & 7;
  f32x4 fa[2][TM][4], fb[4];
  auto read_a = [&](int set, int t) {
    const int tr = t / 3, ts = t - tr * 3;
    const int orow = p.off0 + tr * p.dstep;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int arow = kA + (hbase + (2 * i + orow) * kHaloHC) * 32 + a_col_dw[ts];
#pragma unroll
      for (int q = 0; q < 4; ++q) fa[set][i][q] = *reinterpret_cast<const f32x4*>(&lds[arow + (((2 * q + frag_h) ^ a_swz[ts]) << 2)]);
    }
  };
  auto read_b = [&](int t) {
    const int brow = kB + (t % 3) * BN * 32 + b_frag;
#pragma unroll
    for (int q = 0; q < 4; ++q) fb[q] = *reinterpret_cast<const f32x4*>(&lds[brow + (((2 * q + frag_h) ^ b_swz) << 2)]);
  };
  auto mfmas = [&](int set) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i][q][e], fb[q][e], acc[i][0], 0, 0, 0);
  };

  const int nchunks = p.Cin >> 5;
  for (int c = 0; c < nchunks; ++c) {
    issue_a(c);
    issue_b(c, 0);
    issue_b(c, 1);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      // in order: the halo image and taps <= t have landed when at most one younger tap (t + 1: GB DMAs) is outstanding
      if (t < 8) {
        if constexpr (GB == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();               // ... everybody's share; and everybody is done with tap t - 1's ring stage
      if (t + 2 < 9) issue_b(c, t + 2);           // into the stage tap t - 1 was read from
      if (t == 0) read_a(0, 0);
      read_b(t);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < 9) read_a((t + 1) & 1, t + 1);
      mfmas(t & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // everybody is done reading: the next chunk (or the output patches) may land
  }

  // ---- epilogue: as gather_f32_dma_kernel (bias / accumulate / row map, fused BatchNorm statistics / BatchNorm-backward sums)
  const int col_l = lane & 31;
  const int row_h = (lane >> 5) * 4;
  {
    float* patchb = lds + wave * (WTM * 36);
    const int row0 = m0 + wm * WTM, col0 = n0 + wn * 32;
    int rv = p.M - row0, cv = p.N - col0;
    rv = rv < 0 ? 0 : (rv > WTM ? WTM : rv);
    cv = cv < 0 ? 0 : (cv > 32 ? 32 : cv);
    auto out_row = [&](int m) {
      int b, ho, wo;
      row_to_pixel(p, m, b, ho, wo);
      return (b * p.Ho + ho) * p.Wo + wo;
    };
    if (p.bns_y != nullptr) {
      const BnsEpilogue be{p.bns_y, p.bns_ldy, p.bns_mean, p.bns_invstd, p.bns_scale, p.bns_shift, p.bns_act, p.bns_db, p.bns_dg,
                           (long long)(tile_m * WARPS_M + wm) * p.N};
      store_tiles<TM, 1, false, true>(acc, patchb, p.y, p.ldy, row0, col0, rv, cv, p.bias, p.accumulate != 0, lane, out_row, &be);
    } else {
      store_tiles<TM, 1>(acc, patchb, p.y, p.ldy, row0, col0, rv, cv, p.bias, p.accumulate != 0, lane, out_row);
    }
  }
  if (p.stat != nullptr) {
    const int group = tile_m * WARPS_M + wm;
    const long long gsz = (long long)p.stat_rows * p.N;
    const int col = n0 + wn * 32 + col_l;
    const float k0 = __shfl(acc[0][0][0], lane & 31, 64);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
        if (row < p.M) {
          const float d = acc[i][0][r] - k0;
          s1 += d;
          s2 += d * d;
        }
      }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (lane < 32 && col < p.N) {
      const long long o = (long long)group * p.N + col;
      p.stat[o] = k0;
      p.stat[gsz + o] = s1;
      p.stat[2 * gsz + o] = s2;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// PERSISTENT form of the exact-fp32 LDS-DMA gather kernel for POINTWISE convs (round 5).  The 1x1 layers of a bottleneck network
// are short contractions on big maps: 8-16 K-steps per 128x128 tile.  Per-block stamps of gather_f32_dma_kernel on 256 -> 1024
// channels at 32x32 (profiles/EXPERIMENTS.md 5.1): prologue 4.0 us + K loop 24.9 us + epilogue 6.6 us -- the matrix pipe is busy
// 74 % of the launch, and what idles it is the chain kernel arguments -> row setup -> first DMA round trip at one end and the
// store drain at the other, once per TILE.  Here a block walks tiles blockIdx, blockIdx + grid, ...: ONE continuous stream of
// K-steps through the two-stage ring -- when a tile's last K-steps are being multiplied the DMAs of the next tile's first ones
// are already in flight -- and the epilogue works out of an 8-row patch of its own (store_tiles_rows8) beside the ring, so a
// block still takes 73 KB of LDS and two share a CU.  Pointwise only (rows are pixels: no index arithmetic per tile beyond two
// multiplications), channels a multiple of the K-step, no split-K; bias / accumulate / fused BatchNorm statistics / fused
// BatchNorm-backward sums as the tile-per-block kernel.
// (register budget: two 8-wave blocks, or three 4-wave blocks, per CU -- 4 / 3 waves per SIMD)
template <int BM, int BN, int WARPS_M, int WARPS_N, bool BNS>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N, (WARPS_M * WARPS_N == 8 ? 4 : 3)) void gather_f32_pw_kernel(
    const GatherConvParams p, int ntiles) {
  set_wave_prio(p.prio);
  constexpr int NW = WARPS_M * WARPS_N;
  static_assert(NW == 8 || NW == 4, "8 or 4 waves");
  constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N, TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1 && WTM % 32 == 0 && WTN % 32 == 0, "wave tile");
  constexpr int kStageDw = (BM + BN) * 32;
  constexpr int kPatch8 = 8 * (WTN + 4);
  __shared__ __attribute__((aligned(16))) float lds[2 * kStageDw + NW * kPatch8];
  unsigned* ldsw = reinterpret_cast<unsigned*>(lds);
  constexpr int kA = 0, kB = BM * 32;
  constexpr int GA = BM / 8 / NW, GB = BN / 8 / NW;
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "whole row groups per wave");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;
  const int gridN = (p.N + BN - 1) / BN;
  const int nblocks = (int)gridDim.x;
  const int ksteps = p.kt_total;

  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.w, p.w_bytes);
  const int lrow = lane >> 3, lslot = lane & 7;

  // ---- issue side: K-steps of this block's tiles, in order
  int i_vt = (int)blockIdx.x - nblocks;      // virtual tile whose K-steps are being issued (opened below)
  int i_kt = ksteps;                         // (== ksteps: the first issue opens a tile)
  bool i_done = false;
  uint32_t a_cur[GA], b_cur[GB];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  auto issue = [&](int st) {
    if (i_kt == ksteps) {
      i_vt += nblocks;
      if (i_vt < ntiles) {
        const int t = remap_tile(p.xcd_remap, i_vt, ntiles);
        const int tn = t % gridN, tm = t / gridN;
#pragma unroll
        for (int g = 0; g < GA; ++g) {
          const int row = 8 * (wave + NW * g) + lrow;
          const int m = tm * BM + row;
          a_cur[g] = m < p.M ? (uint32_t)m * (uint32_t)p.ldx * 4u + (uint32_t)((lslot ^ ((row >> 1) & 7)) * 16) : kOOB;
        }
#pragma unroll
        for (int g = 0; g < GB; ++g) {
          const int row = 8 * (wave + NW * g) + lrow;
          const int n = tn * BN + row;
          b_cur[g] = n < p.N ? (uint32_t)n * (uint32_t)p.K * 4u + (uint32_t)((lslot ^ ((row >> 1) & 7)) * 16) : kOOB;
        }
        i_kt = 0;
      } else {
        i_done = true;
      }
    }
    if (i_done) return;          // nothing left: the stage keeps its old bytes, which nobody multiplies
    unsigned* sb = ldsw + st * kStageDw;
#pragma unroll
    for (int g = 0; g < GA; ++g)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr)(sb + kA + 8 * (wave + NW * g) * 32), 16, (int)a_cur[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < GB; ++g)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr)(sb + kB + 8 * (wave + NW * g) * 32), 16, (int)b_cur[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < GA; ++g) a_cur[g] += (uint32_t)(BK * 4);      // (kOOB + anything a launch can add stays out of range)
#pragma unroll
    for (int g = 0; g < GB; ++g) b_cur[g] += (uint32_t)(BK * 4);
    ++i_kt;
  };

  // ---- compute side
  f32x16 acc[TM][TN];
  const int frag_row = lane & 31, frag_h = lane >> 5;
  auto swz32 = [](int row, int slot) -> int { return row * 32 + ((slot ^ ((row >> 1) & 7)) << 2); };
  f32x4 fa[2][2 * TM], fb[2][2 * TN];
  auto read_frags = [&](int set, int st, int half) {
    const float* sb = lds + st * kStageDw;
#pragma unroll
    for (int gg = 0; gg < 2; ++gg) {
      const int slot = 2 * (half * 2 + gg) + frag_h;
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[set][gg * TM + i] = *reinterpret_cast<const f32x4*>(&sb[kA + swz32(wm * WTM + i * 32 + frag_row, slot)]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[set][gg * TN + j] = *reinterpret_cast<const f32x4*>(&sb[kB + swz32(wn * WTN + j * 32 + frag_row, slot)]);
    }
  };
  auto mfmas = [&](int set) {
#pragma unroll
    for (int gg = 0; gg < 2; ++gg)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][gg * TM + i][e], fb[set][gg * TN + j][e], acc[i][j], 0, 0,
                                                             0);
  };

  float* patch8 = lds + 2 * kStageDw + wave * kPatch8;
  const int col_l = lane & 31, row_h = (lane >> 5) * 4;
  auto epilogue = [&](int tile_m, int tile_n) {
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int row0 = m0 + wm * WTM, col0 = n0 + wn * WTN;
    int rv = p.M - row0, cv = p.N - col0;
    rv = rv < 0 ? 0 : (rv > WTM ? WTM : rv);
    cv = cv < 0 ? 0 : (cv > WTN ? WTN : cv);
    auto out_row = [](int m) { return m; };      // pointwise: GEMM row m is pixel m
    if constexpr (BNS) {
      const BnsEpilogue be{p.bns_y, p.bns_ldy, p.bns_mean, p.bns_invstd, p.bns_scale, p.bns_shift, p.bns_act, p.bns_db, p.bns_dg,
                           (long long)(tile_m * WARPS_M + wm) * p.N};
      store_tiles_rows8<TM, TN, true>(acc, patch8, p.y, p.ldy, row0, col0, rv, cv, p.bias, p.accumulate != 0, lane, out_row, &be);
    } else {
      store_tiles_rows8<TM, TN>(acc, patch8, p.y, p.ldy, row0, col0, rv, cv, p.bias, p.accumulate != 0, lane, out_row);
    }
    if (p.stat != nullptr) {       // fused BatchNorm statistics, as gather_f32_dma_kernel
      const int group = tile_m * WARPS_M + wm;
      const long long gsz = (long long)p.stat_rows * p.N;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * WTN + j * 32 + col_l;
        const float k0 = __shfl(acc[0][j][0], lane & 31, 64);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
            if (row < p.M) {
              const float d = acc[i][j][r] - k0;
              s1 += d;
              s2 += d * d;
            }
          }
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (lane < 32 && col < p.N) {
          const long long o = (long long)group * p.N + col;
          p.stat[o] = k0;
          p.stat[gsz + o] = s1;
          p.stat[2 * gsz + o] = s2;
        }
      }
    }
  };

  issue(0);
  issue(1);
  // K-step 0 has landed (this wave's share): all but the youngest GA + GB DMAs -- or everything, when the block has one K-step
  if (ksteps == 1 && (int)blockIdx.x + nblocks >= ntiles) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (GA + GB == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (GA + GB == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (GA + GB == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (GA + GB == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_frags(0, 0, 0);
  int st = 0;
  for (int c_vt = (int)blockIdx.x; c_vt < ntiles; c_vt += nblocks) {
    const int t = remap_tile(p.xcd_remap, c_vt, ntiles);
    const int tile_n = t % gridN, tile_m = t / gridN;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < ksteps; ++it) {
      const int st1 = st ^ 1;
      read_frags(1, st, 1);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next K-step (of this tile or of the next) has landed
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave is done reading stage `st`
      __builtin_amdgcn_s_barrier();
      read_frags(0, st1, 0);      // (the next TILE's first half-step at a tile boundary; stale bytes after the last tile)
      __builtin_amdgcn_sched_barrier(0);
      issue(st);                  // stage `st` is free now
      mfmas(1);
      __builtin_amdgcn_sched_barrier(0);
      st = st1;
    }
    epilogue(tile_m, tile_n);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no DMA may land after the block has given its LDS back
}

// fp32 [M][ld] -> bf16 hi / lo planes [M][ldp] (hi = bf16_rne(x), lo = bf16_rne(x - hi)); columns [C, ldp) are zeroed
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, int ldx, long long M, int C,
                                                           uint16_t* __restrict__ hi, uint16_t* __restrict__ lo, int ldp) {
  PSEG_HELPER_PRIO();
  const ResidualSel rs;
  const int c8n = ldp / 8;
  const long long total = M * c8n;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / c8n;
    const int c = (int)(i - r * c8n) * 8;
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
    if (c < C) v0 = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
    if (c + 4 < C) v1 = *reinterpret_cast<const f32x4*>(x + r * ldx + c + 4);
    u32x2 l0[2], l1[2];
    split4n<2>(v0, l0, rs);
    split4n<2>(v1, l1, rs);
    *reinterpret_cast<u32x4*>(hi + r * ldp + c) = u32x4{l0[0][0], l0[0][1], l1[0][0], l1[0][1]};
    *reinterpret_cast<u32x4*>(lo + r * ldp + c) = u32x4{l0[1][0], l0[1][1], l1[1][0], l1[1][1]};
  }
}


template <int BM, int BN, int WARPS_M, int WARPS_N, bool SKIP>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
  static_assert(WARPS_M * WARPS_N == 4, "4 waves");
  constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1 && WTM % 32 == 0 && WTN % 32 == 0, "wave tile");
  constexpr int CPR_A = BM / 4, CPR_B = BN / 4;            // 16-byte chunks per pixel row
  constexpr int RPP_A = 256 / CPR_A, RPP_B = 256 / CPR_B;  // pixel rows covered per pass
  constexpr int AR = (BK + RPP_A - 1) / RPP_A, BR = (BK + RPP_B - 1) / RPP_B;

  constexpr int kStage = 2 * BK * (BM + BN), kPatch = 4 * WTM * (WTN + 4);
  __shared__ __attribute__((aligned(16))) float lds[kStage > kPatch ? kStage : kPatch];
  float* As = lds;                // [2][BK][BM]   (dy^T tile)
  float* Bs = lds + 2 * BK * BM;  // [2][BK][BN]   (gathered x tile)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;
  const int gridN = (p.K + BN - 1) / BN;
  int wg_tile, wg_split;
  wgrad_block((int)gridDim.x, wg_tile, wg_split);
  const int tile_n = wg_tile % gridN;
  const int tile_m = wg_tile / gridN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t dr = make_rsrc(p.dy, p.dy_bytes);

  const int ca = tid % CPR_A, pra = tid / CPR_A;
  const int cb = tid % CPR_B, prb = tid / CPR_B;
  const int a_col = m0 + ca * 4;
  const bool a_cok = a_col < p.Cout;
  const int b_col = n0 + cb * 4;
  const bool b_cok = b_col < p.K;
  int b_dh, b_dw, b_c;
  {
    const int kk = b_cok ? b_col : 0;
    const int tap = kk / p.Cin;
    b_c = kk - tap * p.Cin;
    const int r = tap / p.kw;
    const int s = tap - r * p.kw;
    b_dh = r * p.dil - p.pad;
    b_dw = s * p.dil - p.pad;
  }

  const int p_begin = wg_split * p.pix_per_split;
  int p_end = p_begin + p.pix_per_split;
  if (p_end > p.P) p_end = p.P;

  // ---- pixel walkers: slot i of this thread gathers pixel pt + prb + RPP_B*i of K-step pt.  (b, ho, wo) is decoded
  // by division only when the walk jumps (start, skipped steps); consecutive steps advance by BK pixels with carries.
  int w_ho[BR], w_wo[BR], w_img[BR];
  const int himg = p.Hi * p.Wi;
  int w_pt = -1;  // K-step the walkers currently point at
  const bool patch = p.patch_mode != 0;
  // patch mode: in-patch coordinates of this thread's rows are constants; a K-step adds a block-uniform origin
  int a_rel[AR], b_rel[BR], b_hh[BR], b_ww[BR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int row = pra + RPP_A * i;
    a_rel[i] = ((row / p.patch_w) * p.Wo + row % p.patch_w) * p.ldy + a_col;
  }
#pragma unroll
  for (int i = 0; i < BR; ++i) {
    const int row = prb + RPP_B * i;
    b_hh[i] = (row / p.patch_w) * p.stride + b_dh;
    b_ww[i] = (row % p.patch_w) * p.stride + b_dw;
    b_rel[i] = (b_hh[i] * p.Wi + b_ww[i]) * p.ldx + b_c;
  }
  auto seek = [&](int pt) {
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      const int pix = pt + prb + RPP_B * i;
      const int b = pix / p.HoWo;
      const int rem = pix - b * p.HoWo;
      w_ho[i] = rem / p.Wo;
      w_wo[i] = rem - w_ho[i] * p.Wo;
      w_img[i] = b * himg;
    }
    w_pt = pt;
  };

  // block-uniform: is K-step [pt, pt+BK) pure padding for this block's tap?
  const int t_dh = (n0 / p.Cin / p.kw) * p.dil - p.pad;
  const int t_dw = ((n0 / p.Cin) % p.kw) * p.dil - p.pad;
  auto step_dead = [&](int pt) -> bool { return wg_step_dead(p, pt, p_end, t_dh, t_dw); };
  auto next_valid = [&](int pt) -> int {
    if (SKIP)
      while (pt < p_end && step_dead(pt)) pt += BK;
    return pt;
  };

  f32x4 areg[AR], breg[BR];

  auto load_tile = [&](int pt) {
    if (patch) {
      // whole K-steps only (P, pix_per_split multiples of 32): no pixel-tail tests
      int pb, h0, w0;
      wg_patch_origin(p, pt, pb, h0, w0);
      const int a_base = ((pb * p.Ho + h0) * p.Wo + w0) * p.ldy;
      const int hs = h0 * p.stride, ws = w0 * p.stride;
      const int b_base = ((pb * p.Hi + hs) * p.Wi + ws) * p.ldx;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const bool ok = (pra + RPP_A * i < BK) && a_cok;
        areg[i] = buf_load4(dr, ok ? (uint32_t)((a_base + a_rel[i]) * 4) : kOOB);
      }
#pragma unroll
      for (int i = 0; i < BR; ++i) {
        const bool ok = (prb + RPP_B * i < BK) && b_cok && ((unsigned)(hs + b_hh[i]) < (unsigned)p.Hi) &&
                        ((unsigned)(ws + b_ww[i]) < (unsigned)p.Wi);
        breg[i] = buf_load4(xr, ok ? (uint32_t)((b_base + b_rel[i]) * 4) : kOOB);
      }
      return;
    }
    if (SKIP) {
      if (pt != w_pt) seek(pt);
    } else if (w_pt < 0) {
      seek(pt);
    }
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int row = pra + RPP_A * i;
      const int pix = pt + row;
      const bool ok = (row < BK) && a_cok && (pix < p_end);
      const uint32_t off = ok ? (uint32_t)((pix * p.ldy + a_col) * 4) : kOOB;
      areg[i] = buf_load4(dr, off);
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      const int row = prb + RPP_B * i;
      const int pix = pt + row;
      const int hi = w_ho[i] * p.stride + b_dh;
      const int wi = w_wo[i] * p.stride + b_dw;
      const bool ok = (row < BK) && b_cok && (pix < p_end) && ((unsigned)hi < (unsigned)p.Hi) &&
                      ((unsigned)wi < (unsigned)p.Wi);
      const uint32_t off = ok ? (uint32_t)(((w_img[i] + hi * p.Wi + wi) * p.ldx + b_c) * 4) : kOOB;
      breg[i] = buf_load4(xr, off);
      // advance this walker by BK pixels
      w_wo[i] += BK;
      while (w_wo[i] >= p.Wo) {
        w_wo[i] -= p.Wo;
        if (++w_ho[i] == p.Ho) {
          w_ho[i] = 0;
          w_img[i] += himg;
        }
      }
    }
    w_pt = pt + BK;
  };

  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int row = pra + RPP_A * i;
      if (row < BK) *reinterpret_cast<f32x4*>(&As[(buf * BK + row) * BM + ca * 4]) = areg[i];
    }
#pragma unroll
    for (int i = 0; i < BR; ++i) {
      const int row = prb + RPP_B * i;
      if (row < BK) *reinterpret_cast<f32x4*>(&Bs[(buf * BK + row) * BN + cb * 4]) = breg[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frag_col = lane & 31;
  const int frag_h = lane >> 5;

  // Same three-tile pipeline as the gather kernel: tile `cur` is multiplied from LDS buffer `buf`, `nxt` is complete in
  // buf^1, `stg` waits in the staging registers (its global loads were issued one whole K-step earlier).  Two fragment
  // sets alternate between the 16-pixel halves of a K-step: the ds_read_b32 words of the next half are fetched under the
  // current MFMA burst, the barrier has no data to wait for, and the staged tile is written + the tile after it requested
  // under the second burst.
  // Operands TM / TN at a time: MFMA tile i of the wave takes rows TM*t + i (t = lane & 31), so one ds_read_b32 / b64 /
  // b128 per k delivers the lane's operand for ALL TM row tiles (same for the TN column tiles).  A half-step is then 16
  // LDS reads for TM = TN = 2 instead of 32 ds_read_b32 -- the LGKM counter holds 15 outstanding operations, and the
  // second half of a 32-read burst used to stall the wave in front of its MFMAs.  (The output rows / columns are mapped
  // back in the epilogue: store_tiles<..., IL = true>.)
  typedef float fvm __attribute__((ext_vector_type(TM)));
  typedef float fvn __attribute__((ext_vector_type(TN)));
  float fa[2][8][TM], fb[2][8][TN];
  auto read_frags = [&](int set, int buf, int half) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int kk = half * 16 + 2 * s + frag_h;
      if constexpr (TM == 1) {
        fa[set][s][0] = As[(buf * BK + kk) * BM + wm * WTM + frag_col];
      } else {
        const fvm v = *reinterpret_cast<const fvm*>(&As[(buf * BK + kk) * BM + wm * WTM + TM * frag_col]);
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[set][s][i] = v[i];
      }
      if constexpr (TN == 1) {
        fb[set][s][0] = Bs[(buf * BK + kk) * BN + wn * WTN + frag_col];
      } else {
        const fvn v = *reinterpret_cast<const fvn*>(&Bs[(buf * BK + kk) * BN + wn * WTN + TN * frag_col]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[set][s][j] = v[j];
      }
    }
  };
  auto mfmas = [&](int set) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][s][i], fb[set][s][j], acc[i][j], 0, 0, 0);
  };

  {
    const int t0 = next_valid(p_begin);
    if (t0 < p_end) {
      load_tile(t0);
      store_tile(0);
      int cur = t0, nxt = next_valid(t0 + BK), stg = p_end;
      if (nxt < p_end) {
        load_tile(nxt);
        store_tile(1);
        stg = next_valid(nxt + BK);
        if (stg < p_end) load_tile(stg);
      }
      if (nxt > p_end) nxt = p_end;
      if (stg > p_end) stg = p_end;
      __syncthreads();
      read_frags(0, 0, 0);
      int buf = 0;
      while (cur < p_end) {
        int after = p_end;
        if (stg < p_end) {
          after = next_valid(stg + BK);
          if (after > p_end) after = p_end;
        }
        read_frags(1, buf, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        read_frags(0, buf ^ 1, 0);   // (garbage on the last step: never multiplied)
        __builtin_amdgcn_sched_barrier(0);
        if (stg < p_end) store_tile(buf);
        if (after < p_end) load_tile(after);
        mfmas(1);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
        nxt = stg;
        stg = after;
        buf ^= 1;
      }
      __syncthreads();   // the epilogue reuses the staging buffers as output patches
    }
  }

  float* out = p.dw + (long long)wg_split * p.slab_stride;
  {
    float* patch = lds + wave * (WTM * (WTN + 4));
    const int row0 = m0 + wm * WTM, col0 = n0 + wn * WTN;
    int rv = p.Cout - row0, cv = p.K - col0;
    rv = rv < 0 ? 0 : (rv > WTM ? WTM : rv);
    cv = cv < 0 ? 0 : (cv > WTN ? WTN : cv);
    store_tiles<TM, TN, true>(acc, patch, out, p.K, row0, col0, rv, cv, nullptr, p.accumulate != 0, lane,
                        [](int m) { return m; });
  }
}

// ------------------------------------------------------------------------------------------------
// Exact-fp32 weight gradient staged by LDS-DMA.  The LDS image of wgrad_kernel -- [32 pixels][BM] / [32 pixels][BN] floats
// -- is exactly how the operands lie in memory (a pixel's channels are contiguous), so `buffer_load_dwordx4 ... lds`
// fills it without any swizzle: one wave-instruction = 1 KiB = 1024 / (4*BM) whole pixel rows.  No staging registers, no
// ds_write; two-stage ring, 8 waves per block and ~100 VGPRs: two blocks (16 waves) share a CU where the register-staged
// kernel has 8 waves, which is what hides a block's prologue / slab epilogue behind the other's MFMAs.
// Pixel order of the contraction: patch mode only (host-checked: p.patch_mode, whole 32-pixel K-steps).
// Round 5: (a) the pieces of a K-step are dealt in 16-byte CHUNK order -- chunk 64 * piece + lane of the [32 px][BN] image -- so a
// tile's pixel row need not be a whole number of pieces: 32 x 288 on NINE waves takes the whole 3x3 filter of a 32-channel layer
// (HRNet's fine branch) as ONE column tile, dy fetched once per K-step instead of once per 128 columns; (b) skip_rows == 4: pixels
// in plain row-major order with each lane deriving its pixel by two divisions -- maps that do not tile into 32-pixel patches
// no longer fall back to the register-staged kernel.
template <int BM, int BN, int WARPS_M, int WARPS_N, bool SKIP>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N) void wgrad_f32_dma_kernel(const WgradParams p) {
  static_assert(WARPS_M * WARPS_N == 8 || WARPS_M * WARPS_N == 9, "8 or 9 waves");
  constexpr int NW = WARPS_M * WARPS_N;
  constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1 && WTM % 32 == 0 && WTN % 32 == 0, "wave tile");
  constexpr int kStageF = BK * (BM + BN);
  constexpr int kPatch = NW * WTM * (WTN + 4);
  __shared__ __attribute__((aligned(16))) float lds[2 * kStageF > kPatch ? 2 * kStageF : kPatch];
  constexpr int kA = 0, kB = BK * BM;                           // inside a stage
  // DMA pieces: RA / RB pixel rows per wave-instruction, IA / IB instructions per wave and K-step
  constexpr int RA = 256 / BM;                                  // dy rows per piece: 1024 B / (4*BM B per row)
  constexpr int CB = BN / 4;                                    // 16-byte chunks per x row
  // pieces per K-step: PA of dy, PB of x, dealt round-robin over the waves (piece q = wave + NW * g).  A 32-row tile has
  // PA = 4 < NW: waves 0..3 carry one dy piece each, the others none -- the per-wave DMA count is wave-uniform, not block-uniform
  constexpr int PA = BK / RA, PB = BK * CB / 64;
  constexpr int IA = (PA + NW - 1) / NW, IB = (PB + NW - 1) / NW;
  static_assert(PA >= 1 && PB >= 1 && BK % RA == 0 && (BK * CB) % 64 == 0 && (PA % NW == 0 || PA < NW) && PB % NW == 0, "pieces per wave");
  constexpr bool kAPartial = PA < NW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;
  const int gridN = (p.K + BN - 1) / BN;
  int wg_tile, wg_split;
  wgrad_block((int)gridDim.x, wg_tile, wg_split);
  int tile_n = wg_tile % gridN;
  const int tile_m = wg_tile / gridN;
  if (SKIP && p.lpt_per > 0) {
    // longest-first order of the column tiles (see WgradParams::lpt_per)
    const int tpt = p.Cin / BN, ntap = p.K / p.Cin, grp = ntap * p.lpt_per;
    const int h = tile_n / grp, rem = tile_n - h * grp;
    const int rank = rem / p.lpt_per, c = rem - rank * p.lpt_per;
    tile_n = p.tap_order[rank] * tpt + h * p.lpt_per + c;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t dr = make_rsrc(p.dy, p.dy_bytes);

  int p_begin = wg_split * p.pix_per_split;
  int p_end = p_begin + p.pix_per_split;
  if (p_end > p.P) p_end = p.P;
  // packed contraction (skip_rows == 3): the live rectangle of this block's tap, block-uniform
  const bool packed = SKIP && p.skip_rows == 3;
  int pk_r0 = 0, pk_c0 = 0, pk_total = 0, pk_dh = 0, pk_dw = 0;
  FastDiv pk_area_div, pk_lw_div;
  if (packed) {
    const int tap = n0 / p.Cin;
    pk_dh = (tap / p.kw) * p.dil - p.pad;
    pk_dw = (tap % p.kw) * p.dil - p.pad;
    pk_r0 = pk_dh < 0 ? -pk_dh : 0;
    pk_c0 = pk_dw < 0 ? -pk_dw : 0;
    int r1 = p.Hi - pk_dh, c1 = p.Wi - pk_dw;
    r1 = r1 > p.Ho ? p.Ho : r1;
    c1 = c1 > p.Wo ? p.Wo : c1;
    const int lh = r1 > pk_r0 ? r1 - pk_r0 : 0, lw = c1 > pk_c0 ? c1 - pk_c0 : 0;
    const int area = lh * lw;
    pk_total = (p.P / p.HoWo) * area;
    pk_area_div = FastDiv((uint32_t)(area > 0 ? area : 1));
    pk_lw_div = FastDiv((uint32_t)(lw > 0 ? lw : 1));
    const int steps = (pk_total + BK - 1) / BK, nsplit = (int)gridDim.z;
    const int per = (steps + nsplit - 1) / nsplit;
    p_begin = wg_split * per * BK;
    p_end = p_begin + per * BK;
    if (p_end > steps * BK) p_end = steps * BK;
    if (p_begin > p_end) p_begin = p_end;
  }

  // ---- per-lane constants of the DMA pieces (patch mode: K-step origin + these)
  int a_rel[IA];      // float offset inside dy relative to the K-step's first pixel, or -1 (column beyond Cout)
  int b_rel[IB], b_hh[IB], b_ww[IB];
#pragma unroll
  for (int g = 0; g < IA; ++g) {
    const int row = RA * (wave + NW * g) + lane / (BM / 4);       // pixel row of the K-step
    const int col = m0 + 4 * (lane % (BM / 4));
    a_rel[g] = col < p.Cout ? ((row / p.patch_w) * p.Wo + row % p.patch_w) * p.ldy + col : -1;
    if (packed) a_rel[g] = col < p.Cout ? col : -1;               // (packed: the column; the pixel comes per K-step)
  }
  const bool rowmajor = p.skip_rows == 4;      // plain pixel order, per-lane pixel derivation (no patches)
  int b_prow[IB];                              // pixel row (0..31) of the K-step this lane's chunk of piece g belongs to
#pragma unroll
  for (int g = 0; g < IB; ++g) {
    const int ci = 64 * (wave + NW * g) + lane;                   // chunk of the [32 px][BN] image
    const int row = ci / CB;
    const int col = n0 + 4 * (ci - row * CB);
    b_prow[g] = row;
    if (col < p.K) {
      const int tap = col / p.Cin;
      const int c = col - tap * p.Cin;
      const int r = tap / p.kw, sx = tap - r * p.kw;
      b_hh[g] = (row / p.patch_w) * p.stride + r * p.dil - p.pad;
      b_ww[g] = (row % p.patch_w) * p.stride + sx * p.dil - p.pad;
      b_rel[g] = (b_hh[g] * p.Wi + b_ww[g]) * p.ldx + c;
      if (packed) b_rel[g] = c;                                   // (packed: the channel inside the block's tap)
      if (rowmajor) {                                             // (row-major: the tap's offsets and the channel)
        b_hh[g] = r * p.dil - p.pad;
        b_ww[g] = sx * p.dil - p.pad;
        b_rel[g] = c;
      }
    } else {
      b_hh[g] = b_ww[g] = -(1 << 28);     // never in range
      b_rel[g] = (packed || rowmajor) ? -1 : 0;
    }
  }
  if (rowmajor) {
#pragma unroll
    for (int g = 0; g < IA; ++g) {
      const int col = m0 + 4 * (lane % (BM / 4));
      a_rel[g] = col < p.Cout ? col : -1;
    }
  }

  const int t_dh = (n0 / p.Cin / p.kw) * p.dil - p.pad;
  const int t_dw = ((n0 / p.Cin) % p.kw) * p.dil - p.pad;
  auto next_valid = [&](int pt) -> int {
    if (SKIP && !packed)
      while (pt < p_end && wg_step_dead(p, pt, p_end, t_dh, t_dw)) pt += BK;
    return pt;
  };

  typedef __attribute__((address_space(3))) void* lds_ptr;
  // packed pixel q of this block's tap -> float offsets of its dy pixel and of its source pixel (always inside the image)
  auto packed_pixel = [&](int q, int& a_pix, int& b_pix) {
    const uint32_t b = pk_area_div.div((uint32_t)q);
    const uint32_t rem = (uint32_t)q - b * pk_area_div.d;
    const uint32_t hh = pk_lw_div.div(rem);
    const uint32_t ww = rem - hh * pk_lw_div.d;
    const int ho = pk_r0 + (int)hh, wo = pk_c0 + (int)ww;
    a_pix = (((int)b * p.Ho + ho) * p.Wo + wo) * p.ldy;
    b_pix = (((int)b * p.Hi + ho + pk_dh) * p.Wi + wo + pk_dw) * p.ldx;
  };
  auto issue = [&](int pt, int st) {     // pt >= p_end: all-zero dummy pieces
    float* sb = lds + st * kStageF;
    if (packed) {
      const bool live_step = pt < p_end;
#pragma unroll
      for (int g = 0; g < IA; ++g) {
        if (kAPartial && wave >= PA) break;
        const int q = pt + RA * (wave + NW * g) + lane / (BM / 4);
        int a_pix, b_pix;
        packed_pixel(q < pk_total ? q : 0, a_pix, b_pix);
        const uint32_t off = (live_step && q < pk_total && a_rel[g] >= 0) ? (uint32_t)((a_pix + a_rel[g]) * 4) : kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(dr, (lds_ptr)(sb + kA + RA * (wave + NW * g) * BM), 16, (int)off, 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < IB; ++g) {
        const int q = pt + b_prow[g];
        int a_pix, b_pix;
        packed_pixel(q < pk_total ? q : 0, a_pix, b_pix);
        const uint32_t off = (live_step && q < pk_total && b_rel[g] >= 0) ? (uint32_t)((b_pix + b_rel[g]) * 4) : kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr)(sb + kB + 256 * (wave + NW * g)), 16, (int)off, 0, 0, 0);
      }
      return;
    }
    if (rowmajor) {
      // pixel pt + row of the [B, Ho, Wo] raster, by two divisions per piece; pixels at or past p_end read nothing
      auto pixel = [&](int q, int& b, int& ho, int& wo) {
        const uint32_t bb = p.rm_howo.div((uint32_t)q);
        const uint32_t rem = (uint32_t)q - bb * p.rm_howo.d;
        const uint32_t hh = p.rm_wo.div(rem);
        b = (int)bb;
        ho = (int)hh;
        wo = (int)(rem - hh * p.rm_wo.d);
      };
#pragma unroll
      for (int g = 0; g < IA; ++g) {
        if (kAPartial && wave >= PA) break;
        const int q = pt + RA * (wave + NW * g) + lane / (BM / 4);
        int b, ho, wo;
        pixel(q < p_end ? q : 0, b, ho, wo);
        const uint32_t off = (q < p_end && a_rel[g] >= 0) ? (uint32_t)((((b * p.Ho + ho) * p.Wo + wo) * p.ldy + a_rel[g]) * 4) : kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(dr, (lds_ptr)(sb + kA + RA * (wave + NW * g) * BM), 16, (int)off, 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < IB; ++g) {
        const int q = pt + b_prow[g];
        int b, ho, wo;
        pixel(q < p_end ? q : 0, b, ho, wo);
        const int hi = ho * p.stride + b_hh[g], wi = wo * p.stride + b_ww[g];
        const bool ok = q < p_end && b_rel[g] >= 0 && ((unsigned)hi < (unsigned)p.Hi) && ((unsigned)wi < (unsigned)p.Wi);
        const uint32_t off = ok ? (uint32_t)((((b * p.Hi + hi) * p.Wi + wi) * p.ldx + b_rel[g]) * 4) : kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr)(sb + kB + 256 * (wave + NW * g)), 16, (int)off, 0, 0, 0);
      }
      return;
    }
    int pb = 0, h0 = 0, w0 = 0;
    const bool live = pt < p_end;
    if (live) wg_patch_origin(p, pt, pb, h0, w0);
    const int a_base = ((pb * p.Ho + h0) * p.Wo + w0) * p.ldy;
    const int hs = h0 * p.stride, ws = w0 * p.stride;
    const int b_base = ((pb * p.Hi + hs) * p.Wi + ws) * p.ldx;
#pragma unroll
    for (int g = 0; g < IA; ++g) {
      if (kAPartial && wave >= PA) break;      // (wave-uniform)
      const uint32_t off = (live && a_rel[g] >= 0) ? (uint32_t)((a_base + a_rel[g]) * 4) : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(dr, (lds_ptr)(sb + kA + RA * (wave + NW * g) * BM), 16, (int)off, 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < IB; ++g) {
      const bool ok = live && ((unsigned)(hs + b_hh[g]) < (unsigned)p.Hi) && ((unsigned)(ws + b_ww[g]) < (unsigned)p.Wi);
      const uint32_t off = ok ? (uint32_t)((b_base + b_rel[g]) * 4) : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr)(sb + kB + 256 * (wave + NW * g)), 16, (int)off, 0, 0, 0);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frag_col = lane & 31;
  const int frag_h = lane >> 5;
  typedef float fvm __attribute__((ext_vector_type(TM)));
  typedef float fvn __attribute__((ext_vector_type(TN)));
  float fa[2][8][TM], fb[2][8][TN];
  auto read_frags = [&](int set, int st, int half) {
    const float* sb = lds + st * kStageF;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int kk = half * 16 + 2 * s + frag_h;
      if constexpr (TM == 1) {
        fa[set][s][0] = sb[kA + kk * BM + wm * WTM + frag_col];
      } else {
        const fvm v = *reinterpret_cast<const fvm*>(&sb[kA + kk * BM + wm * WTM + TM * frag_col]);
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[set][s][i] = v[i];
      }
      if constexpr (TN == 1) {
        fb[set][s][0] = sb[kB + kk * BN + wn * WTN + frag_col];
      } else {
        const fvn v = *reinterpret_cast<const fvn*>(&sb[kB + kk * BN + wn * WTN + TN * frag_col]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[set][s][j] = v[j];
      }
    }
  };
  auto mfmas = [&](int set) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][s][i], fb[set][s][j], acc[i][j], 0, 0, 0);
  };

  {
    int q0 = next_valid(p_begin);
    if (q0 < p_end) {
      int q1 = next_valid(q0 + BK);
      issue(q0, 0);
      issue(q1, 1);
      int q2 = q1 < p_end ? next_valid(q1 + BK) : p_end;
      // K-step q0 has landed (this wave's share), q1 stays in flight: all but the youngest (pieces of this wave per K-step)
      if constexpr (kAPartial) {
        static_assert(IA == 1 && IB == 4, "partial-A tile: 5 or 4 pieces per wave");      // (32x256 on 8 waves, 32x288 on 9)
        if (wave < PA) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else if constexpr (IA + IB == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if constexpr (IA + IB == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      read_frags(0, 0, 0);
      int st = 0;
      while (q0 < p_end) {
        read_frags(1, st, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the next K-step has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave is done reading stage `st`
        __builtin_amdgcn_s_barrier();
        read_frags(0, st ^ 1, 0);     // (zeros on the last step: never multiplied)
        __builtin_amdgcn_sched_barrier(0);
        issue(q2, st);                // stage `st` is free now
        mfmas(1);
        __builtin_amdgcn_sched_barrier(0);
        q0 = q1;
        q1 = q2;
        q2 = q2 < p_end ? next_valid(q2 + BK) : p_end;
        st ^= 1;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // dummy pieces must not land in the output patches
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }

  float* out = p.dw + (long long)wg_split * p.slab_stride;
  {
    float* patch = lds + wave * (WTM * (WTN + 4));
    const int row0 = m0 + wm * WTM, col0 = n0 + wn * WTN;
    int rv = p.Cout - row0, cv = p.K - col0;
    rv = rv < 0 ? 0 : (rv > WTM ? WTM : rv);
    cv = cv < 0 ? 0 : (cv > WTN ? WTN : cv);
    store_tiles<TM, TN, true>(acc, patch, out, p.K, row0, col0, rv, cv, nullptr, p.accumulate != 0, lane,
                              [](int m) { return m; });
  }
}

// ------------------------------------------------------------------------------------------------
// Split-bf16 weight gradient.  dW[Cout][K] = dY^T * A with the contraction over pixels; the bf16 MFMA wants 8 consecutive
// k (= pixels) per lane, but both operands arrive pixel-major (4 consecutive CHANNELS per 16-byte load).  Each loader
// thread therefore takes one 4-channel chunk of 8 consecutive pixels (8 loads), transposes them in registers into four
// 8-pixel runs, splits each run into NL bf16 limbs and writes them as 16-byte k-slots of the channel-major LDS image
// ([row = channel or K-column][32 pixels], same swizzle as the gather kernel).  Threads [0, BM) load dY^T, threads
// [256-BN, 256) gather A; the MFMA block, pixel-split slabs, row skipping and epilogue are those of the fp32 kernel.
// ------------------------------------------------------------------------------------------------
// HALO-STAGED weight gradient of a 3x3 (unit stride, dilation 1, padding 1) for NARROW filters, exact fp32 (round 5).
// wgrad_f32_dma_kernel<32, 288, 1, 9> fetches the x operand of a 32-pixel K-step as nine shifted copies -- [32 px][9 taps x 32 ch] =
// 36 KB + 4 KB of dy for 1024 matrix cycles of each of its nine waves: 40 bytes per clock and CU against an LDS-DMA path that
// sustains ~27 -- and a 32 / 64-row tile has no other way to be wide.  Here a K-step is a 2 x 16 STRIP of output pixels; x comes in
// ONCE per K-step as the 4 x 18 halo strip of a 32-channel chunk (9 KB), dy as [32 px][BM] (4 / 8 KB), and wave t (tap t) reads its
// B fragments from the halo strip at its tap's shift: dw[co][t][ci] += sum_p dy[p][co] x[p + t][ci].  Nine waves = nine taps, each
// with BM x 32 accumulators; a block = (row tile of BM filters, one 32-channel chunk of the input, one pixel split).  13 / 17 KB
// per K-step, 41 KB of LDS (the nine epilogue patches).  Same pixel splits / slabs / reduction as wgrad_f32_dma_kernel.
//   LDS reads are single dwords (pixel-major images, as wgrad_f32_dma_kernel): lanes 0-31 = 32 consecutive floats of one pixel
//   row, lanes 32-63 of the NEXT pixel (k = 2 s + (lane >> 5)): adjacent pixels are adjacent 128-byte rows -- the two halves of the
//   64 banks -- at every tap shift (18 is even), so no swizzle.
template <int BM>
__global__ __launch_bounds__(576) void wgrad_f32_halo_kernel(const WgradParams p) {
  constexpr int NW = 9, TM = BM / 32;
  static_assert(BM == 32 || BM == 64, "row tile");
  constexpr int kHS = 4 * 18;                                  // halo strip pixels
  constexpr int kA = 0, kB = BK * BM, kStageF = BK * BM + kHS * 32;
  constexpr int kPatch = NW * 32 * 36;
  __shared__ __attribute__((aligned(16))) float lds[2 * kStageF > kPatch ? 2 * kStageF : kPatch];
  // DMA pieces per K-step: dy rows of BM floats (BM * 4 bytes): 1024 / (4 BM) pixel rows per wave-instruction -> NA instructions
  constexpr int RA = 256 / BM, NA = BK / RA;                   // 8 rows, 4 instructions (BM = 32) / 4 rows, 8 instructions (64)
  static_assert(NA <= NW, "one dy piece per wave at most");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gridN = p.Cin >> 5;
  int wg_tile, wg_split;
  wgrad_block((int)gridDim.x, wg_tile, wg_split);
  const int tile_n = wg_tile % gridN, tile_m = wg_tile / gridN;
  const int m0 = tile_m * BM, c0 = tile_n * 32;

  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t dr = make_rsrc(p.dy, p.dy_bytes);
  const int p_begin = wg_split * p.pix_per_split;
  int p_end = p_begin + p.pix_per_split;
  if (p_end > p.P) p_end = p.P;

  // this lane's share of a K-step's DMAs, relative to the strip's top-left output pixel (h0, w0) of image b
  int a_rel = -1;                                              // dy: float offset, or -1 (no piece / column beyond Cout)
  if (wave < NA) {
    const int row = RA * wave + lane / (BM / 4);               // pixel of the strip: (row >> 4, row & 15)
    const int col = m0 + 4 * (lane % (BM / 4));
    if (col < p.Cout) a_rel = ((row >> 4) * p.Wo + (row & 15)) * p.ldy + col;
  }
  const int hidx = 8 * wave + (lane >> 3);                     // halo pixel (hidx / 18 - 1, hidx % 18 - 1) relative to (h0, w0)
  const int h_dh = hidx / 18 - 1, h_dw = hidx % 18 - 1;
  const int h_rel = (h_dh * p.Wi + h_dw) * p.ldx + c0 + 4 * (lane & 7);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  unsigned* ldsw = reinterpret_cast<unsigned*>(lds);
  auto issue = [&](int pt, int st) {
    unsigned* sb = ldsw + st * kStageF;
    const bool live = pt < p_end;
    int pb = 0, h0 = 0, w0 = 0;
    if (live) wg_patch_origin(p, pt, pb, h0, w0);
    if (wave < NA) {                                           // (wave-uniform)
      const uint32_t off = (live && a_rel >= 0) ? (uint32_t)((((pb * p.Ho + h0) * p.Wo + w0) * p.ldy + a_rel) * 4) : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(dr, (lds_ptr)(sb + kA + RA * wave * BM), 16, (int)off, 0, 0, 0);
    }
    const bool ok = live && ((unsigned)(h0 + h_dh) < (unsigned)p.Hi) && ((unsigned)(w0 + h_dw) < (unsigned)p.Wi);
    const uint32_t off = ok ? (uint32_t)((((pb * p.Hi + h0) * p.Wi + w0) * p.ldx + h_rel) * 4) : kOOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr)(sb + kB + 8 * wave * 32), 16, (int)off, 0, 0, 0);
  };

  f32x16 acc[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int frag_col = lane & 31, frag_h = lane >> 5;
  const int tr = wave / 3, ts = wave - 3 * tr;                 // this wave's tap: x pixel = output pixel + (tr - 1, ts - 1)
  float fa[2][8][TM], fb[2][8];
  auto read_frags = [&](int set, int st, int half) {
    const float* sb = lds + st * kStageF;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int kk = half * 16 + 2 * s + frag_h;               // pixel of the strip: (half, 2 s + frag_h)
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[set][s][i] = sb[kA + kk * BM + i * 32 + frag_col];
      fb[set][s] = sb[kB + ((half + tr) * 18 + (2 * s + frag_h) + ts) * 32 + frag_col];
    }
  };
  auto mfmas = [&](int set) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][s][i], fb[set][s], acc[i], 0, 0, 0);
  };

  if (p_begin < p_end) {
    int q0 = p_begin;
    issue(q0, 0);
    issue(q0 + BK, 1);
    // K-step q0 has landed (this wave's share) when only the second K-step's pieces of this wave are outstanding
    if (wave < NA) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(0, 0, 0);
    int st = 0;
    while (q0 < p_end) {
      read_frags(1, st, 1);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the next K-step has landed
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave is done reading stage `st`
      __builtin_amdgcn_s_barrier();
      read_frags(0, st ^ 1, 0);     // (zeros past the end: never multiplied)
      __builtin_amdgcn_sched_barrier(0);
      issue(q0 + 2 * BK, st);       // stage `st` is free now
      mfmas(1);
      __builtin_amdgcn_sched_barrier(0);
      q0 += BK;
      st ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // dummy pieces must not land in the output patches
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  float* out = p.dw + (long long)wg_split * p.slab_stride;
  float* patch = lds + wave * (32 * 36);
  const int col0 = wave * p.Cin + c0;                          // K index of (tap, channel): tap * Cin + channel
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row0 = m0 + i * 32;
    int rv = p.Cout - row0;
    rv = rv < 0 ? 0 : (rv > 32 ? 32 : rv);
    const f32x16 (&acc1)[1][1] = reinterpret_cast<const f32x16 (&)[1][1]>(acc[i]);
    store_tiles<1, 1>(acc1, patch, out, p.K, row0, col0, rv, 32, nullptr, p.accumulate != 0, lane, [](int m) { return m; });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();                           // the patch is rewritten by the next row half
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
}

template <int NL>
__device__ __forceinline__ void split8n(float (&v)[8], u32x4 (&limb)[NL], const ResidualSel& rs) {
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    unsigned pk[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      pk[q] = pack_bf16(v[2 * q], v[2 * q + 1]);
      if (l + 1 < NL) {
        v[2 * q] = resid_lo(pk[q], v[2 * q], rs);
        v[2 * q + 1] = resid_hi(pk[q], v[2 * q + 1], rs);
      }
    }
    limb[l] = u32x4{pk[0], pk[1], pk[2], pk[3]};
  }
}

template <int BM, int BN, int WARPS_M, int WARPS_N, bool SKIP, int NL>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N) void wgrad_limb_kernel(const WgradParams p) {
  static_assert(WARPS_M * WARPS_N == 4 || WARPS_M * WARPS_N == 8, "4 waves, or 8 for the 256-row tile");
  constexpr int NT = 64 * WARPS_M * WARPS_N;
  static_assert(BM + BN <= NT && BK == 32, "one loader slot per thread");
  constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int kStage = 2 * (BM + BN) * 16 * NL, kPatch = (NT / 64) * WTM * (WTN + 4);
  __shared__ __attribute__((aligned(16))) float lds[kStage > kPatch ? kStage : kPatch];
  unsigned* ldsw = reinterpret_cast<unsigned*>(lds);
  constexpr int kAsz = 2 * BM * 16, kBsz = 2 * BN * 16, kBbase = NL * kAsz;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;
  const int gridN = (p.K + BN - 1) / BN;
  int wg_tile, wg_split;
  wgrad_block((int)gridDim.x, wg_tile, wg_split);
  const int tile_n = wg_tile % gridN;
  const int tile_m = wg_tile / gridN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t dr = make_rsrc(p.dy, p.dy_bytes);
  const ResidualSel rsel;

  // loader roles
  const bool is_a = tid < BM;
  const bool is_b = tid >= NT - BN;
  const int sa = tid, sb = tid - (NT - BN);
  const int ca = sa % (BM / 4), ga = sa / (BM / 4);        // channel chunk, 8-pixel group (0..3)
  const int cb = (is_b ? sb : 0) % (BN / 4), gb = (is_b ? sb : 0) / (BN / 4);
  const int a_col = m0 + ca * 4;
  const bool a_cok = is_a && a_col < p.Cout;
  const int b_col = n0 + cb * 4;
  const bool b_cok = is_b && b_col < p.K;
  int b_dh, b_dw, b_c;
  {
    const int kk = b_cok ? b_col : 0;
    const int tap = kk / p.Cin;
    b_c = kk - tap * p.Cin;
    const int r = tap / p.kw;
    const int s = tap - r * p.kw;
    b_dh = r * p.dil - p.pad;
    b_dw = s * p.dil - p.pad;
  }

  const int p_begin = wg_split * p.pix_per_split;
  int p_end = p_begin + p.pix_per_split;
  if (p_end > p.P) p_end = p.P;

  const int t_dh = (n0 / p.Cin / p.kw) * p.dil - p.pad;
  const int t_dw = ((n0 / p.Cin) % p.kw) * p.dil - p.pad;
  const bool patch = p.patch_mode != 0;
  auto step_dead = [&](int pt) -> bool { return wg_step_dead(p, pt, p_end, t_dh, t_dw); };
  auto next_valid = [&](int pt) -> int {
    if (SKIP)
      while (pt < p_end && step_dead(pt)) pt += BK;
    return pt;
  };

  const int himg = p.Hi * p.Wi;
  f32x4 reg[8];

  // B loader: (image base, ho, wo) of this thread's first pixel, advanced incrementally (a K-step moves 32 pixels on);
  // a division only on the first step and on row-skipping jumps.
  int bs_pix = -(1 << 30), bs_img = 0, bs_ho = 0, bs_wo = 0;
  // patch mode (the map tiles into patch_h x patch_w = 32-pixel patches, patch_w % 8 == 0): this thread's 8-pixel run
  // sits at a constant place inside every patch; a K-step adds the block-uniform patch origin
  const int pg = is_a ? ga : gb;
  const int pm_ih = (pg * 8) / p.patch_w, pm_iw = (pg * 8) % p.patch_w;
  const int pm_a_rel = (pm_ih * p.Wo + pm_iw) * p.ldy + a_col;
  const int pm_hh = pm_ih * p.stride + b_dh, pm_ww = pm_iw * p.stride + b_dw;
  const int pm_b_rel = (pm_hh * p.Wi + pm_ww) * p.ldx + b_c;
  auto load_tile = [&](int pt) {
    if (patch) {
      int pb, h0, w0;
      wg_patch_origin(p, pt, pb, h0, w0);
      if (is_a) {
        const uint32_t off0 = (uint32_t)((((pb * p.Ho + h0) * p.Wo + w0) * p.ldy + pm_a_rel) * 4);
        const uint32_t dpx = (uint32_t)p.ldy * 4u;
#pragma unroll
        for (int j = 0; j < 8; ++j) reg[j] = buf_load4(dr, a_cok ? off0 + (uint32_t)j * dpx : kOOB);
      } else if (is_b) {
        const int hs = h0 * p.stride, ws = w0 * p.stride;
        const bool rok = b_cok && ((unsigned)(hs + pm_hh) < (unsigned)p.Hi);
        const int wi0 = ws + pm_ww;
        const uint32_t off0 = (uint32_t)((((pb * p.Hi + hs) * p.Wi + ws) * p.ldx + pm_b_rel) * 4);   // may wrap; used when ok
        const uint32_t dpx = (uint32_t)(p.stride * p.ldx) * 4u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool ok = rok && ((unsigned)(wi0 + j * p.stride) < (unsigned)p.Wi);
          reg[j] = buf_load4(xr, ok ? off0 + (uint32_t)j * dpx : kOOB);
        }
      }
      return;
    }
    if (is_a) {
      const int pix0 = pt + ga * 8;
      const int nval = a_cok ? p_end - pix0 : 0;
      const uint32_t off0 = (uint32_t)((pix0 * p.ldy + a_col) * 4);
      const uint32_t dpx = (uint32_t)p.ldy * 4u;
#pragma unroll
      for (int j = 0; j < 8; ++j) reg[j] = buf_load4(dr, j < nval ? off0 + (uint32_t)j * dpx : kOOB);
    } else if (is_b) {
      const int pix0 = pt + gb * 8;
      if (pix0 == bs_pix + BK) {
        bs_wo += BK;
        while (bs_wo >= p.Wo) {
          bs_wo -= p.Wo;
          if (++bs_ho == p.Ho) {
            bs_ho = 0;
            bs_img += himg;
          }
        }
      } else {
        const int b = pix0 / p.HoWo;
        const int rem = pix0 - b * p.HoWo;
        bs_ho = rem / p.Wo;
        bs_wo = rem - bs_ho * p.Wo;
        bs_img = b * himg;
      }
      bs_pix = pix0;
      const int nval = b_cok ? p_end - pix0 : 0;
      if (bs_wo + 8 <= p.Wo) {
        // the 8 pixels lie in one output row (always, when Wo % 8 == 0): one row test, offsets a constant stride apart
        const int hi = bs_ho * p.stride + b_dh;
        const int wi0 = bs_wo * p.stride + b_dw;
        const bool rok = (unsigned)hi < (unsigned)p.Hi;
        const uint32_t off0 = (uint32_t)(((bs_img + hi * p.Wi + wi0) * p.ldx + b_c) * 4);   // may wrap; used when ok
        const uint32_t dpx = (uint32_t)(p.stride * p.ldx) * 4u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool ok = rok && j < nval && ((unsigned)(wi0 + j * p.stride) < (unsigned)p.Wi);
          reg[j] = buf_load4(xr, ok ? off0 + (uint32_t)j * dpx : kOOB);
        }
      } else {
        int ho = bs_ho, wo = bs_wo, img = bs_img;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int hi = ho * p.stride + b_dh, wi = wo * p.stride + b_dw;
          const bool ok = j < nval && ((unsigned)hi < (unsigned)p.Hi) && ((unsigned)wi < (unsigned)p.Wi);
          const uint32_t off = ok ? (uint32_t)(((img + hi * p.Wi + wi) * p.ldx + b_c) * 4) : kOOB;
          reg[j] = buf_load4(xr, off);
          if (++wo == p.Wo) {
            wo = 0;
            if (++ho == p.Ho) {
              ho = 0;
              img += himg;
            }
          }
        }
      }
    }
  };

  auto store_tile = [&](int buf) {
    if (!(is_a || is_b)) return;
    const int rows = is_a ? BM : BN;
    const int base = is_a ? 0 : kBbase;
    const int isz = is_a ? kAsz : kBsz;
    const int c4 = is_a ? ca : cb, g = is_a ? ga : gb;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = reg[j][e];
      u32x4 limb[NL];
      split8n<NL>(v, limb, rsel);
      const int o = buf * rows * 16 + swz(c4 * 4 + e, g);
#pragma unroll
      for (int l = 0; l < NL; ++l) *reinterpret_cast<u32x4*>(&ldsw[base + l * isz + o]) = limb[l];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frag_row = lane & 31;

  // Same pipeline as the gather kernel: two fragment sets alternate between the 16-pixel halves of a K-step; tile `cur`
  // is multiplied from LDS buffer `buf`, `nxt` is complete in buf^1, `stg` waits in the staging registers (loaded one
  // whole K-step earlier) and is split + written into `buf` after the barrier, under the second MFMA burst.
  f32x4 fa[2][NL * TM], fb[2][NL * TN];
  auto read_frags = [&](int set, int buf, int half) {
    const int slot = half * 2 + (lane >> 5);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int o = buf * BM * 16 + swz(wm * WTM + i * 32 + frag_row, slot);
#pragma unroll
      for (int l = 0; l < NL; ++l) fa[set][l * TM + i] = *reinterpret_cast<const f32x4*>(&ldsw[l * kAsz + o]);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int o = buf * BN * 16 + swz(wn * WTN + j * 32 + frag_row, slot);
#pragma unroll
      for (int l = 0; l < NL; ++l) fb[set][l * TN + j] = *reinterpret_cast<const f32x4*>(&ldsw[kBbase + l * kBsz + o]);
    }
  };
  auto mfmas = [&](int set) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int ord = NL - 1; ord >= 0; --ord)
#pragma unroll
          for (int la = 0; la <= ord; ++la)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[set][la * TM + i]),
                                                                __builtin_bit_cast(bf16x8, fb[set][(ord - la) * TN + j]),
                                                                acc[i][j], 0, 0, 0);
  };

  {
    const int t0 = next_valid(p_begin);
    if (t0 < p_end) {
      load_tile(t0);
      store_tile(0);
      int cur = t0, nxt = next_valid(t0 + BK), stg = p_end;
      if (nxt < p_end) {
        load_tile(nxt);
        store_tile(1);
        stg = next_valid(nxt + BK);
        if (stg < p_end) load_tile(stg);
      }
      if (nxt > p_end) nxt = p_end;
      if (stg > p_end) stg = p_end;
      __syncthreads();
      read_frags(0, 0, 0);
      int buf = 0;
      while (cur < p_end) {
        int after = p_end;
        if (stg < p_end) {
          after = next_valid(stg + BK);
          if (after > p_end) after = p_end;
        }
        read_frags(1, buf, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        read_frags(0, buf ^ 1, 0);   // (garbage on the last step: never multiplied)
        __builtin_amdgcn_sched_barrier(0);
        if (stg < p_end) store_tile(buf);
        if (after < p_end) load_tile(after);
        mfmas(1);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
        nxt = stg;
        stg = after;
        buf ^= 1;
      }
      __syncthreads();   // the epilogue reuses the staging buffers as output patches
    }
  }

  float* out = p.dw + (long long)wg_split * p.slab_stride;
  {
    float* patch = lds + wave * (WTM * (WTN + 4));
    const int row0 = m0 + wm * WTM, col0 = n0 + wn * WTN;
    int rv = p.Cout - row0, cv = p.K - col0;
    rv = rv < 0 ? 0 : (rv > WTM ? WTM : rv);
    cv = cv < 0 ? 0 : (cv > WTN ? WTN : cv);
    store_tiles<TM, TN>(acc, patch, out, p.K, row0, col0, rv, cv, nullptr, p.accumulate != 0, lane,
                        [](int m) { return m; });
  }
}

// ------------------------------------------------------------------------------------------------
// Fixed-order reduction of split slabs: out[m*ld + n] = (acc ? out : 0) + bias[n] + sum_z slab[z][m][n].
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, long long slab_stride,
                                                          int nslab, float* __restrict__ out, int ld, long long M,
                                                          int N, const float* __restrict__ bias, int accumulate) {
  const long long total = M * N;
  if ((N & 3) == 0 && (ld & 3) == 0 && (slab_stride & 3) == 0 && (((uintptr_t)slabs | (uintptr_t)out) & 15) == 0) {
    // 16 bytes per lane, slabs in a fixed order; two elements in flight per lane
    const long long total4 = total >> 2;
    const int n4 = N >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
      const long long m = i / n4;
      const int n = (int)(i - m * n4) * 4;
      const float* sp = slabs + i * 4;
      f32x4 v = *reinterpret_cast<const f32x4*>(sp);
      // slabs added strictly in order, their loads sixteen / four at a time: the narrow layers of the encoder split into
      // up to 256 slabs of a few thousand elements -- a handful of blocks, each lane a chain of nslab round trips
      // (230 us per launch beside the data gradients; the weight-gradient stream ends the fp32 step 0.5 ms late)
      int z = 1;
      for (; z + 15 < nslab; z += 16) {
        f32x4 t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) t[k] = *reinterpret_cast<const f32x4*>(sp + (long long)(z + k) * slab_stride);
#pragma unroll
        for (int k = 0; k < 16; ++k) v += t[k];
      }
      for (; z + 3 < nslab; z += 4) {
        f32x4 t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = *reinterpret_cast<const f32x4*>(sp + (long long)(z + k) * slab_stride);
#pragma unroll
        for (int k = 0; k < 4; ++k) v += t[k];
      }
      for (; z < nslab; ++z) v += *reinterpret_cast<const f32x4*>(sp + (long long)z * slab_stride);
      if (bias != nullptr) v += *reinterpret_cast<const f32x4*>(bias + n);
      float* op = out + m * ld + n;
      if (accumulate) v += *reinterpret_cast<const f32x4*>(op);
      *reinterpret_cast<f32x4*>(op) = v;
    }
    return;
  }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long m = i / N;
    const int n = (int)(i - m * N);
    float v = 0.f;
    for (int z = 0; z < nslab; ++z) v += slabs[(long long)z * slab_stride + i];
    if (bias != nullptr) v += bias[n];
    const long long o = m * ld + n;
    if (accumulate) v += out[o];
    out[o] = v;
  }
}

// Every split weight gradient of a backward pass reduced in ONE launch (pseg_slab_reduce_batch): jobs[j] = {slabs, out,
// elements (a multiple of 4), slab count, index of the job's first block}; a block covers kSlabBlock consecutive
// elements of one job and finds it by bisection.  Fixed slab order, same arithmetic as slab_reduce_kernel.
constexpr int kSlabBlock = 4096;   // floats per block: 256 lanes x 4 x 16 bytes
__global__ __launch_bounds__(256) void slab_reduce_batch_kernel(const long long* __restrict__ jobs, int n, int accumulate) {
  const long long b = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid * 5 + 4] <= b) lo = mid;
    else hi = mid - 1;
  }
  const long long* job = jobs + lo * 5;
  const float* slabs = reinterpret_cast<const float*>(job[0]);
  float* out = reinterpret_cast<float*>(job[1]);
  const long long elems = job[2];
  const int nslab = (int)job[3];
  const long long base = (b - job[4]) * kSlabBlock;
#pragma unroll
  for (int u = 0; u < kSlabBlock / 1024; ++u) {
    const long long i = base + (long long)u * 1024 + threadIdx.x * 4;
    if (i < elems) {
      f32x4 v = *reinterpret_cast<const f32x4*>(slabs + i);
      int z = 1;      // in order, loads sixteen / four at a time (as in slab_reduce_kernel)
      for (; z + 15 < nslab; z += 16) {
        f32x4 t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) t[k] = *reinterpret_cast<const f32x4*>(slabs + (long long)(z + k) * elems + i);
#pragma unroll
        for (int k = 0; k < 16; ++k) v += t[k];
      }
      for (; z + 3 < nslab; z += 4) {
        f32x4 t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = *reinterpret_cast<const f32x4*>(slabs + (long long)(z + k) * elems + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) v += t[k];
      }
      for (; z < nslab; ++z) v += *reinterpret_cast<const f32x4*>(slabs + (long long)z * elems + i);
      if (accumulate) v += *reinterpret_cast<const f32x4*>(out + i);
      *reinterpret_cast<f32x4*>(out + i) = v;
    }
  }
}

int launch_slab_reduce(const float* slabs, long long slab_stride, int nslab, float* out, int ld, long long M, int N,
                       const float* bias, int accumulate, hipStream_t st) {
  const long long total = M * N;
  const int blocks = (int)(total / 256 + 1 < 4096 ? total / 256 + 1 : 4096);
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, st, slabs, slab_stride, nslab, out, ld, M, N, bias,
                     accumulate);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

// w[Cout][taps][Cin] -> wT[Cin][taps][Cout]   (32x32 LDS tile per tap)
__global__ __launch_bounds__(256) void filter_transpose_kernel(const float* __restrict__ w, float* __restrict__ wT,
                                                               int Cout, int taps, int Cin) {
  PSEG_HELPER_PRIO();
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < Cout && ci < Cin) ? w[((long long)co * taps + t) * Cin + ci] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < Cin && co < Cout) wT[((long long)ci * taps + t) * Cout + co] = tile[tx][r];
  }
}


template <typename P, typename F, int NTILES>
static int launch_tiles(const F (&fns)[2][NTILES], bool skip, TileCfg c, dim3 grid, const P& p, hipStream_t st) {
  int idx = -1;
  if (c.bm == 128 && c.bn == 128) idx = 0;
  else if (c.bm == 128 && c.bn == 64) idx = 1;
  else if (c.bm == 128 && c.bn == 32) idx = 2;
  else if (c.bm == 64 && c.bn == 128) idx = 3;
  else if (c.bm == 32 && c.bn == 128) idx = 4;
  else if (c.bm == 256 && c.bn == 128 && NTILES > 5) idx = 5;   // 8 waves
  if (idx < 0 || fns[skip ? 1 : 0][idx] == nullptr) {
    set_error("no kernel for tile %dx%d", c.bm, c.bn);
    return PSEG_ERR_ARG;
  }
  hipLaunchKernelGGL(fns[skip ? 1 : 0][idx], grid, dim3(c.bm == 256 ? 512 : 256), 0, st, p);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}



// debug: per-block phase timestamps of gather_f32_dma_kernel (pseg_debug_conv_trace; tools/conv_phases.py)
static unsigned long long* g_conv_trace = nullptr;

// pre-split bf16 limb planes of both operands of a gather GEMM (see gather_limb_dma_kernel)
struct LimbPlanes {
  const uint16_t* xh;
  const uint16_t* xl;
  const uint16_t* wh;
  const uint16_t* wl;
  int ldxp;
  long long xp_bytes, wp_bytes;
};

// one instantiation of the persistent pointwise kernel: grid = the blocks the device holds at once; 0 = launched, 1 = not worth it
// (fewer tiles than resident blocks: every block would own one tile) or not launchable
template <int BM, int BN, int WM, int WN>
static int launch_pw(int ntiles, hipStream_t st, const GatherConvParams& p) {
  static int resident = 0;
  if (resident == 0) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gather_f32_pw_kernel<BM, BN, WM, WN, false>, 64 * WM * WN, 0) != hipSuccess ||
        per_cu < 1) {
      resident = -1;
    } else {
      constexpr long long lds_bytes = (2LL * (BM + BN) * 32 + WM * WN * 8 * (BN / WN + 4)) * 4;
      const int by_lds = (int)((160 * 1024) / lds_bytes);
      if (per_cu > by_lds) per_cu = by_lds;
      static const int forced = env_int("PSEG_CONV_PW_BPC", 0);
      if (forced > 0 && forced < per_cu) per_cu = forced;
      resident = prop.multiProcessorCount * per_cu;
    }
  }
  int grid = resident;
  if (cfg().conv_pw_resident > 0) grid = cfg().conv_pw_resident;      // (tests: several tiles per block on small problems)
  if (grid <= 0 || ntiles <= grid) return 1;
  if (p.bns_y != nullptr)
    hipLaunchKernelGGL((gather_f32_pw_kernel<BM, BN, WM, WN, true>), dim3((unsigned)grid), dim3(64 * WM * WN), 0, st, p, ntiles);
  else
    hipLaunchKernelGGL((gather_f32_pw_kernel<BM, BN, WM, WN, false>), dim3((unsigned)grid), dim3(64 * WM * WN), 0, st, p, ntiles);
  return 0;
}

// ... and its GENERIC form: channels of the gathered tensor a multiple of 4 but not of the K-step (no tap skipping, natural or
// pointwise row order, no split-K)
static bool f32dma_generic(const FwdPlan& pl, int precision, int Cin, bool skip_taps, int row_perm) {
  static const int on = env_int("PSEG_CONV_F32DMA_GENERIC", 1);
  const bool tile_ok = (pl.tile.bm == 128 && (pl.tile.bn == 128 || pl.tile.bn == 64 || (pl.tile.bn == 32 && cfg().conv_dma32 != 0))) ||
                       (pl.tile.bm == 64 && pl.tile.bn == 128);
  return on != 0 && precision == 0 && Cin % BK != 0 && Cin % 4 == 0 && pl.splits == 1 && cfg().conv_f32dma != 0 && !skip_taps &&
         (row_perm == 0 || row_perm == 3) && tile_ok;
}

// BatchNorm-backward partial sums fused into a data gradient (GatherConvParams::bns_*)
struct BnsArgs {
  const float* y;
  int ldy;
  const float* mean;
  const float* invstd;
  const float* scale;
  const float* shift;
  int act;
  float* db;
  float* dg;
  int rows;
};

// can the LDS-DMA limb kernel run this gather problem?  (whole K-steps inside a tap, the 256x128 tile, no split-K)
static bool dma_plan_ok(const FwdPlan& pl, int Cin, int N) {
  return pl.tile.bm == kDmaBM && pl.tile.bn == kDmaBN && pl.splits == 1 && Cin % BK == 0 && N >= kDmaBN &&
         cfg().conv_nodma == 0;
}

static int waves_m(TileCfg t) { return t.bm == 256 ? 4 : (t.bn == 32 ? 4 : (t.bm == 32 ? 1 : 2)); }

// does the exact-fp32 LDS-DMA kernel run this gather problem?  (whole K-steps inside a tap, one of its four tiles, no split-K)
static bool f32dma_covers(const FwdPlan& pl, int precision, int Cin, int K, int taps, bool skip_taps) {
  const bool tile_ok = (pl.tile.bm == 128 && (pl.tile.bn == 128 || pl.tile.bn == 64 || (pl.tile.bn == 32 && cfg().conv_dma32 != 0))) ||
                       (pl.tile.bm == 64 && pl.tile.bn == 128);
  return precision == 0 && Cin % BK == 0 && pl.splits == 1 && cfg().conv_f32dma != 0 && K % BK == 0 && taps <= 32 &&
         !(cfg().conv_f32dma == 2 && skip_taps) && tile_ok;
}

// does the halo-staged kernel run this problem?  (see gather_f32_halo_kernel)
static bool halo_f32_covers(const FwdPlan& pl, const GatherConvParams& p, int Cin, int taps, int taps_w, int s_out, int s_in, int dstep,
                            int off0, int Hi, int Wi, int Ho, int Wo, bool skip) {
  const bool tile_ok = pl.tile.bm == 128 && ((pl.tile.bn == 32 && cfg().conv_dma32 != 0) || (pl.tile.bn == 64 && cfg().conv_halo >= 2));
  return cfg().conv_halo != 0 && tile_ok && pl.splits == 1 && !skip && taps == 9 && taps_w == 3 && s_out == 1 &&
         s_in == 1 && (dstep == 1 || dstep == -1) && off0 == -dstep && Cin % 32 == 0 && Hi == Ho && Wi == Wo && Ho % kHaloPH == 0 &&
         Wo % kHaloPW == 0 && p.row_perm == 0;
}

static int run_gather(const float* x, long long x_bytes, int ldx, const float* w, float* y, int ldy, const float* bias,
                      float* stat, int B, int Hi, int Wi, int Cin, int Ho, int Wo, int N, int taps_w,
                      int K, int s_out, int s_in, int dstep, int off0, int accumulate, int precision,
                      const unsigned* amax_a, const unsigned* amax_b, void* workspace, int64_t workspace_bytes,
                      hipStream_t st, const LimbPlanes* planes = nullptr, const BnsArgs* bns = nullptr) {
  const long long M = (long long)B * Ho * Wo;
  PSEG_REQUIRE(M > 0 && M < (1LL << 31) && N > 0 && K > 0, "conv: empty or oversized problem M=%lld N=%d K=%d", M, N, K);
  PSEG_REQUIRE(Cin % 4 == 0 && ldx % 4 == 0, "conv: Cin (%d) and ldx (%d) must be multiples of 4", Cin, ldx);
  PSEG_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0, "conv: x / w must be 16-byte aligned");
  const long long w_bytes = (long long)N * K * 4;
  PSEG_REQUIRE(x_bytes < kMaxBytes && w_bytes < kMaxBytes, "conv: tensor exceeds 2 GiB (x %lld, w %lld bytes)", x_bytes,
               w_bytes);
  PSEG_REQUIRE(((M - 1) * ldy + N) * 4 < (1LL << 40), "conv: output too large");
  DilGeom geom;
  const bool has_geom = dil_geom(geom, Ho, Wo, Hi, Wi, (K / Cin) / taps_w, taps_w, Cin, s_out, s_in, dstep, off0);
  // the 256x128 / 8-wave tile (one block per CU) of the limb kernels: measured 1 % SLOWER in the round-2 training step than
  // two 128-row blocks per CU (mixed policy 35.85 vs 35.45 ms) -- opt-in (PSEG_CONV_BIG=1); the pre-split DMA kernel is
  // built on it
  const bool allow_big = stat == nullptr && (precision == 1 || precision == 3) &&
                         (planes != nullptr || cfg().conv_big != 0 || cfg().conv_forcebig != 0);
  FwdPlan pl = plan_gather(M, N, K, allow_big, has_geom ? &geom : nullptr);
  if (planes != nullptr && !(precision == 1 && stat == nullptr && bias == nullptr && dma_plan_ok(pl, Cin, N))) {
    set_error("conv: the pre-split (LDS-DMA) limb kernel does not cover this problem (M=%lld N=%d K=%d Cin=%d)", M, N, K, Cin);
    return PSEG_ERR_ARG;
  }

  GatherConvParams p;
  p.x = x;
  p.w = w;
  p.bias = bias;
  p.stat = stat;
  p.stat_rows = pl.gridM * waves_m(pl.tile);
  p.x_bytes = (uint32_t)x_bytes;
  p.w_bytes = (uint32_t)w_bytes;
  p.ldx = ldx;
  p.Hi = Hi;
  p.Wi = Wi;
  p.Cin = Cin;
  p.Ho = Ho;
  p.Wo = Wo;
  p.HoWo = Ho * Wo;
  p.M = (int)M;
  p.N = N;
  p.K = K;
  p.kw = taps_w;
  p.s_out = s_out;
  p.s_in = s_in;
  p.dstep = dstep;
  p.off0 = off0;
  p.kt_total = pl.kt_total;
  p.kt_per_split = pl.kt_per_split;
  const int taps = K / Cin;
  const int adil = dstep < 0 ? -dstep : dstep;
  p.ntaps = taps;
  p.ktiles_per_tap = Cin / BK;
  p.skip_taps = (adil >= 4 && taps > 1 && taps <= 32 && Cin % BK == 0 && cfg().conv_noskip == 0) ? 1 : 0;
  // stride-2 data gradient (s_in == 2): parity-homogeneous tiles + tap skipping (needs whole tiles per class, no split-K)
  p.xcd_remap = cfg().conv_noxcd == 0 ? 1 : 0;
  p.prio = dstep < 0 ? cfg().dgrad_prio : 0;
  p.row_perm = 0;
  p.patch_w = p.patch_hw = p.patches_per_row = 1;
  if (K == Cin && s_out == 1 && s_in == 1 && off0 == 0 && Hi == Ho && Wi == Wo) {
    // 1x1, unit stride: GEMM row m is pixel m of the source.  Present the tensor as a 1 x M image so that the kernel's
    // per-row prologue is free of divisions (short-K layers -- K = 64 is two K-steps -- are prologue / epilogue bound)
    p.row_perm = 3;
    p.Hi = 1;
    p.Wi = (int)M;
    p.Ho = 1;
    p.Wo = (int)M;
    p.HoWo = (int)M;
  }
  if (pl.patch_w > 0 && p.skip_taps) {
    p.row_perm = 2;
    p.patch_w = pl.patch_w;
    p.patch_hw = pl.patch_h * pl.patch_w;
    p.patches_per_row = Wo / pl.patch_w;
  }
  if (pl.banded && p.skip_taps) {
    p.row_perm = 4;
    p.band = pl.band;
    p.xcd_remap = 2;     // the descending-cost order is the schedule (band_makespan)
  }
  if (s_in == 2 && Ho % 2 == 0 && Wo % 2 == 0 && ((Ho / 2) * (Wo / 2)) % pl.tile.bm == 0 && pl.splits == 1 &&
      taps <= 32 && Cin % BK == 0 && cfg().conv_noskip == 0) {
    p.row_perm = 1;
    p.skip_taps = 1;
  }
  const dim3 grid((unsigned)(pl.gridM * pl.gridN), 1, (unsigned)pl.splits);
  if (pl.splits == 1) {
    p.y = y;
    p.ldy = ldy;
    p.accumulate = accumulate;
    p.slab_stride = 0;
  } else {
    const long long need = (long long)pl.splits * M * N * 4;
    if (workspace == nullptr || workspace_bytes < need) {
      set_error("conv: split-K needs %lld workspace bytes, got %lld", need, (long long)workspace_bytes);
      return PSEG_ERR_WORKSPACE;
    }
    p.y = (float*)workspace;
    p.ldy = N;
    p.accumulate = 0;
    p.bias = nullptr;
    p.stat = nullptr;
    p.slab_stride = M * N;
  }
  p.trace = g_conv_trace;
  p.xh = p.xl = p.wh = p.wl = nullptr;
  p.xp_bytes = p.wp_bytes = 0;
  p.ldxp = 0;
  p.bns_y = nullptr;
  p.bns_ldy = 0;
  p.bns_mean = p.bns_invstd = p.bns_scale = p.bns_shift = nullptr;
  p.bns_act = 0;
  p.bns_db = p.bns_dg = nullptr;
  if (bns != nullptr) {
    // only the exact-fp32 LDS-DMA kernel carries the fused sums: the caller asked pseg_conv2d_dgrad_bnstat_rows first
    if (!(f32dma_covers(pl, precision, Cin, K, taps, p.skip_taps != 0) && planes == nullptr && accumulate == 0 &&
          bns->rows == pl.gridM * waves_m(pl.tile))) {
      set_error("conv: the fused BatchNorm-backward sums are not available for this data gradient (M=%lld N=%d K=%d, rows %d)", M,
                N, K, bns->rows);
      return PSEG_ERR_ARG;
    }
    p.bns_y = bns->y;
    p.bns_ldy = bns->ldy;
    p.bns_mean = bns->mean;
    p.bns_invstd = bns->invstd;
    p.bns_scale = bns->scale;
    p.bns_shift = bns->shift;
    p.bns_act = bns->act;
    p.bns_db = bns->db;
    p.bns_dg = bns->dg;
  }
  if (planes != nullptr) {
    PSEG_REQUIRE(planes->xp_bytes < kMaxBytes && planes->wp_bytes < kMaxBytes && planes->ldxp % 8 == 0 &&
                     ((uintptr_t)planes->xh & 15) == 0 && ((uintptr_t)planes->xl & 15) == 0 &&
                     ((uintptr_t)planes->wh & 15) == 0 && ((uintptr_t)planes->wl & 15) == 0,
                 "conv: limb planes must be 16-byte aligned, ld %% 8 == 0, < 2 GiB");
    p.xh = planes->xh;
    p.xl = planes->xl;
    p.wh = planes->wh;
    p.wl = planes->wl;
    p.xp_bytes = (uint32_t)planes->xp_bytes;
    p.wp_bytes = (uint32_t)planes->wp_bytes;
    p.ldxp = planes->ldxp;
    p.precision = 1;
    p.amax_a = p.amax_b = nullptr;
    if (p.skip_taps) hipLaunchKernelGGL(gather_limb_dma_kernel<true>, grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL(gather_limb_dma_kernel<false>, grid, dim3(512), 0, st, p);
    g_last_conv_kernel = PSEG_KERNEL_GATHER_LIMB_DMA;
    PSEG_LAUNCH_CHECK();
    return PSEG_OK;
  }
  if (f32dma_generic(pl, precision, Cin, p.skip_taps != 0, p.row_perm) && planes == nullptr && bns == nullptr) {
    // channel counts off the K-step grid (stem, classifier data gradient, MobileNetV2 widths): the same kernel with per-slot
    // (tap, channel) derivation; two-stage ring
    p.precision = 0;
    p.amax_a = p.amax_b = nullptr;
    p.gen_spt = FastDiv((uint32_t)(Cin / 4));
    p.gen_kw = FastDiv((uint32_t)taps_w);
    if (pl.tile.bm == 128 && pl.tile.bn == 128)
      hipLaunchKernelGGL((gather_f32_dma_kernel<128, 128, 2, 4, false, 2, true>), grid, dim3(512), 0, st, p);
    else if (pl.tile.bm == 128 && pl.tile.bn == 64)
      hipLaunchKernelGGL((gather_f32_dma_kernel<128, 64, 2, 2, false, 2, true>), grid, dim3(256), 0, st, p);
    else if (pl.tile.bm == 64 && pl.tile.bn == 128)
      hipLaunchKernelGGL((gather_f32_dma_kernel<64, 128, 2, 2, false, 2, true>), grid, dim3(256), 0, st, p);
    else
      hipLaunchKernelGGL((gather_f32_dma_kernel<128, 32, 4, 1, false, 2, true>), grid, dim3(256), 0, st, p);
    g_last_conv_kernel = PSEG_KERNEL_GATHER_RING_GENERIC;
    PSEG_LAUNCH_CHECK();
    return PSEG_OK;
  }
  if (f32dma_covers(pl, precision, Cin, K, taps, p.skip_taps != 0)) {   // (3 = two-stage ring for the tap-skipping problems as well)
    // Exact-fp32 problems whose K-steps never straddle a tap run on the LDS-DMA kernel (same tile, same statistics
    // layout).  PSEG_CONV_F32DMA: 3 (default) = two-stage ring for every such problem, 2 = only for those without tap
    // skipping (51.0 -> 49.4 ms; the tap-skipping ASPP / stride-2 problems add 49.0 -> 48.4) -- 64 / 48 KB of LDS
    // and ~100 VGPRs, so TWO blocks of 8 waves (128x128) or THREE of 4 (128x64) share a CU and one block's prologue /
    // epilogue hides behind the others' MFMAs: medium and short-K layers +5-10 % (128x128 maps, 64 channels: 98 -> 109 TF),
    // fp32 step 51.0 -> 49.4 ms; 1 = three-stage ring everywhere (one block per CU: equal to the register-staged kernel in
    // the step); 0 = register-staged kernel only.  The large layers sit at the sustained fp32-MFMA rate either way.
    p.precision = 0;
    p.amax_a = p.amax_b = nullptr;
    const bool sk = p.skip_taps != 0;
    // pointwise convs with short contractions and more tiles than the device holds blocks: the persistent kernel
    // (PSEG_CONV_PW=0: off; PSEG_CONV_PW_KT: longest contraction, in K-steps, that takes it)
    if (cfg().conv_pw != 0 && p.row_perm == 3 && !sk && pl.kt_total <= cfg().conv_pw_kt && cfg().conv_f32dma >= 2) {
      const int ntiles = pl.gridM * pl.gridN;
      int rc = 1;
      if (pl.tile.bm == 128 && pl.tile.bn == 128) rc = launch_pw<128, 128, 2, 4>(ntiles, st, p);
      else if (pl.tile.bm == 128 && pl.tile.bn == 64) rc = launch_pw<128, 64, 2, 2>(ntiles, st, p);
      else if (pl.tile.bm == 64 && pl.tile.bn == 128) rc = launch_pw<64, 128, 2, 2>(ntiles, st, p);
      if (rc == 0) {
        g_last_conv_kernel = PSEG_KERNEL_GATHER_POINTWISE;
        PSEG_LAUNCH_CHECK();
        return PSEG_OK;
      }
    }
    // narrow 3x3 (128x32 plan tile): the halo-staged kernel where the map is made of whole 8 x 16 patches
    if (halo_f32_covers(pl, p, Cin, taps, taps_w, s_out, s_in, dstep, off0, Hi, Wi, Ho, Wo, sk)) {
      p.row_perm = 2;
      p.patch_w = kHaloPW;
      p.patch_hw = kHaloPH * kHaloPW;
      p.patches_per_row = Wo / kHaloPW;
      if (pl.tile.bn == 32) hipLaunchKernelGGL((gather_f32_halo_kernel<4, 1>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((gather_f32_halo_kernel<2, 2>), grid, dim3(256), 0, st, p);
      g_last_conv_kernel = PSEG_KERNEL_GATHER_HALO;
      PSEG_LAUNCH_CHECK();
      return PSEG_OK;
    }
    bool launched = true;
    const bool two = cfg().conv_f32dma >= 2;   // two-stage ring: 64 / 48 KB of LDS, 2 / 3 blocks per CU
#define PSEG_DMA_LAUNCH(BM_, BN_, WM_, WN_, NTHR)                                                                    \
  do {                                                                                                               \
    if (two && sk) hipLaunchKernelGGL((gather_f32_dma_kernel<BM_, BN_, WM_, WN_, true, 2>), grid, dim3(NTHR), 0, st, p);   \
    else if (two) hipLaunchKernelGGL((gather_f32_dma_kernel<BM_, BN_, WM_, WN_, false, 2>), grid, dim3(NTHR), 0, st, p);   \
    else if (sk) hipLaunchKernelGGL((gather_f32_dma_kernel<BM_, BN_, WM_, WN_, true, 3>), grid, dim3(NTHR), 0, st, p);     \
    else hipLaunchKernelGGL((gather_f32_dma_kernel<BM_, BN_, WM_, WN_, false, 3>), grid, dim3(NTHR), 0, st, p);            \
  } while (0)
    if (pl.tile.bm == 128 && pl.tile.bn == 128) {
      PSEG_DMA_LAUNCH(128, 128, 2, 4, 512);
    } else if (pl.tile.bm == 128 && pl.tile.bn == 64) {
      PSEG_DMA_LAUNCH(128, 64, 2, 2, 256);
    } else if (pl.tile.bm == 64 && pl.tile.bn == 128) {
      PSEG_DMA_LAUNCH(64, 128, 2, 2, 256);
    } else if (pl.tile.bm == 128 && pl.tile.bn == 32 && cfg().conv_dma32 != 0) {
      PSEG_DMA_LAUNCH(128, 32, 4, 1, 256);     // narrow outputs (HRNet's 32-channel branch, the 21-class classifier)
#undef PSEG_DMA_LAUNCH
    } else {
      launched = false;
    }
    if (launched) {
      g_last_conv_kernel = PSEG_KERNEL_GATHER_RING;
      PSEG_LAUNCH_CHECK();
      return PSEG_OK;
    }
  }
  typedef void (*Kfn)(const GatherConvParams);
#define PSEG_GATHER_ROW(SK, PR, BIG)                                                                   \
  {gather_conv_kernel<128, 128, 2, 2, SK, PR>, gather_conv_kernel<128, 64, 2, 2, SK, PR>,              \
   gather_conv_kernel<128, 32, 4, 1, SK, PR>, gather_conv_kernel<64, 128, 2, 2, SK, PR>,               \
   gather_conv_kernel<32, 128, 1, 4, SK, PR>, BIG}
  // the 256x128 tile (8 waves, one block per CU) exists for the LDS-staging-bound two-limb variants
  static const Kfn fns32[2][6] = {PSEG_GATHER_ROW(false, 0, nullptr), PSEG_GATHER_ROW(true, 0, nullptr)};
  static const Kfn fnsb3[2][6] = {PSEG_GATHER_ROW(false, 1, (gather_conv_kernel<256, 128, 4, 2, false, 1>)),
                                  PSEG_GATHER_ROW(true, 1, (gather_conv_kernel<256, 128, 4, 2, true, 1>))};
  static const Kfn fnsb6[2][6] = {PSEG_GATHER_ROW(false, 2, nullptr), PSEG_GATHER_ROW(true, 2, nullptr)};
  static const Kfn fnsh3[2][6] = {PSEG_GATHER_ROW(false, 3, (gather_conv_kernel<256, 128, 4, 2, false, 3>)),
                                  PSEG_GATHER_ROW(true, 3, (gather_conv_kernel<256, 128, 4, 2, true, 3>))};
#undef PSEG_GATHER_ROW
  p.precision = precision;
  p.amax_a = amax_a;
  p.amax_b = amax_b;
  if (precision == 3 && (amax_a == nullptr || amax_b == nullptr)) {
    set_error("conv: PSEG_PREC_FP16X3 needs the amax of both operands");
    return PSEG_ERR_ARG;
  }
  g_last_conv_kernel = PSEG_KERNEL_GATHER_REGISTER;
  int rc = precision == 3   ? launch_tiles<GatherConvParams, Kfn, 6>(fnsh3, p.skip_taps != 0, pl.tile, grid, p, st)
           : precision == 2 ? launch_tiles<GatherConvParams, Kfn, 6>(fnsb6, p.skip_taps != 0, pl.tile, grid, p, st)
           : precision == 1 ? launch_tiles<GatherConvParams, Kfn, 6>(fnsb3, p.skip_taps != 0, pl.tile, grid, p, st)
                            : launch_tiles<GatherConvParams, Kfn, 6>(fns32, p.skip_taps != 0, pl.tile, grid, p, st);
  if (rc != PSEG_OK) return rc;
  if (pl.splits > 1) {
    const long long total = M * N;
    const int blocks = (int)(total / 256 + 1 < 4096 ? total / 256 + 1 : 4096);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, M * N, pl.splits, y,
                       ldy, M, N, bias, accumulate);
    PSEG_LAUNCH_CHECK();
  }
  return PSEG_OK;
}


}  // namespace pseg

using namespace pseg;

extern "C" {

int pseg_abi_version(void) { return PSEG_ABI_VERSION; }
int pseg_debug_conv_trace(void* buffer) {
  if (!kConvTrace && buffer != nullptr) {
    pseg::set_error("debug_conv_trace: the library was built without -DPSEG_CONV_TRACE=1 (PSEG_BUILD_TRACE=1 ... build --force)");
    return PSEG_ERR_ARG;
  }
  pseg::g_conv_trace = (unsigned long long*)buffer;
  return PSEG_OK;
}
int pseg_debug_last_conv_kernel(void) { return pseg::g_last_conv_kernel; }
int pseg_config_reload(void) {
  pseg::cfg_load();
  return PSEG_OK;
}
const char* pseg_last_error(void) { return pseg::last_error(); }

// the forward plan of a conv as far as the statistics layout depends on it (tile shape; K does not enter)
static FwdPlan plan_fwd_stats(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil) {
  const long long M = (long long)B * Ho * Wo;
  const int H = (Ho - 1) * stride - 2 * pad + dil * (kh - 1) + 1, W = (Wo - 1) * stride - 2 * pad + dil * (kw - 1) + 1;
  DilGeom geom;
  const bool has_geom = dil_geom(geom, Ho, Wo, H, W, kh, kw, Cin, stride, 1, dil, -pad);
  return plan_gather(M, Cout, kh * kw * Cin, false, has_geom ? &geom : nullptr);
}

int pseg_conv2d_stat_rows(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil) {
  FwdPlan pl = plan_fwd_stats(B, Ho, Wo, Cin, Cout, kh, kw, stride, pad, dil);
  return pl.gridM * waves_m(pl.tile);
}

int pseg_conv2d_stat_group(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil) {
  FwdPlan pl = plan_fwd_stats(B, Ho, Wo, Cin, Cout, kh, kw, stride, pad, dil);
  return pl.tile.bm / waves_m(pl.tile);
}

int64_t pseg_conv2d_fwd_workspace_bytes(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw) {
  const long long M = (long long)B * Ho * Wo;
  FwdPlan pl = plan_gather(M, Cout, kh * kw * Cin);
  return pl.splits > 1 ? (int64_t)pl.splits * M * Cout * 4 : 0;
}

int pseg_conv2d_fwd(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int B, int H, int W,
                    int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int dil, int accumulate,
                    int precision, const float* amax_x, const float* amax_w, float* stat, void* workspace,
                    int64_t workspace_bytes, void* stream) {
  PSEG_REQUIRE(x && w && y, "conv2d_fwd: null pointer");
  PSEG_REQUIRE(stride >= 1 && dil >= 1 && pad >= 0 && kh >= 1 && kw >= 1, "conv2d_fwd: bad geometry");
  PSEG_REQUIRE(Ho == (H + 2 * pad - dil * (kh - 1) - 1) / stride + 1 && Wo == (W + 2 * pad - dil * (kw - 1) - 1) / stride + 1,
               "conv2d_fwd: Ho/Wo (%d,%d) inconsistent with H/W (%d,%d) k=%dx%d s=%d p=%d d=%d", Ho, Wo, H, W, kh, kw,
               stride, pad, dil);
  const int K = kh * kw * Cin;
  FwdPlan pl = plan_fwd_stats(B, Ho, Wo, Cin, Cout, kh, kw, stride, pad, dil);
  if (stat != nullptr && pl.splits > 1) {
    set_error("conv2d_fwd: fused statistics are unavailable when the plan splits K; use pseg_col_stats");
    return PSEG_ERR_ARG;
  }
  PSEG_REQUIRE(precision >= 0 && precision <= 3, "conv2d_fwd: precision must be one of PSEG_PREC_*");
  return run_gather(x, nhwc_bytes(B, H, W, Cin, ldx), ldx, w, y, ldy, bias, stat, B, H, W, Cin, Ho, Wo,
                    Cout, kw, K, stride, 1, dil, -pad, accumulate, precision, (const unsigned*)amax_x,
                    (const unsigned*)amax_w, workspace, workspace_bytes, (hipStream_t)stream);
}

int pseg_conv2d_dgrad(const float* dy, int ldy, const float* wT, float* dx, int ldx, int B, int H, int W, int Cin, int Ho,
                      int Wo, int Cout, int kh, int kw, int stride, int pad, int dil, int accumulate, int precision,
                      const float* amax_dy, const float* amax_w, void* workspace, int64_t workspace_bytes,
                      void* stream) {
  PSEG_REQUIRE(dy && wT && dx, "conv2d_dgrad: null pointer");
  PSEG_REQUIRE(precision >= 0 && precision <= 3, "conv2d_dgrad: precision must be one of PSEG_PREC_*");
  PSEG_REQUIRE(stride >= 1 && dil >= 1 && pad >= 0, "conv2d_dgrad: bad geometry");
  // GEMM rows = input pixels (B,H,W); contraction over (r,s,co); gather source = dy [B,Ho,Wo,Cout]
  const int K = kh * kw * Cout;
  return run_gather(dy, nhwc_bytes(B, Ho, Wo, Cout, ldy), ldy, wT, dx, ldx, nullptr, nullptr, B, Ho, Wo, Cout, H,
                    W, Cin, kw, K, 1, stride, -dil, pad, accumulate, precision, (const unsigned*)amax_dy,
                    (const unsigned*)amax_w, workspace, workspace_bytes, (hipStream_t)stream);
}

// geometry of the dgrad gather problem as run_gather sees it -> does the LDS-DMA kernel (the one with the fused sums) take it?
static bool dgrad_bnstat_plan(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int dil,
                              FwdPlan& pl) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 || kh * kw > 32 || Cin % 4 != 0) return false;
  const long long M = (long long)B * H * W;
  if (M >= (1LL << 31)) return false;
  const int K = kh * kw * Cout;
  DilGeom geom;
  const bool has_geom = dil_geom(geom, H, W, Ho, Wo, kh, kw, Cout, 1, stride, -dil, pad);
  pl = plan_gather(M, Cin, K, false, has_geom ? &geom : nullptr);
  const int adil = dil;
  const bool skip = (adil >= 4 && kh * kw > 1 && Cout % BK == 0 && cfg().conv_noskip == 0) ||
                    (stride == 2 && H % 2 == 0 && W % 2 == 0 && ((H / 2) * (W / 2)) % pl.tile.bm == 0 && pl.splits == 1 &&
                     Cout % BK == 0 && cfg().conv_noskip == 0);
  return f32dma_covers(pl, 0, Cout, K, kh * kw, skip);
}

int pseg_conv2d_dgrad_bnstat_rows(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad,
                                  int dil) {
  FwdPlan pl;
  if (!dgrad_bnstat_plan(B, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, dil, pl)) return 0;
  return pl.gridM * waves_m(pl.tile);
}

int pseg_conv2d_dgrad_bnstat(const float* dy, int ldy, const float* wT, float* dx, int ldx, int B, int H, int W, int Cin, int Ho,
                             int Wo, int Cout, int kh, int kw, int stride, int pad, int dil, const float* y_prev, int ldy_prev,
                             const float* mean, const float* invstd, const float* scale, const float* shift, int act,
                             float* part_db, float* part_dg, int part_rows, void* stream) {
  PSEG_REQUIRE(dy && wT && dx && y_prev && mean && invstd && scale && shift && part_db && part_dg,
               "conv2d_dgrad_bnstat: null pointer");
  PSEG_REQUIRE(stride >= 1 && dil >= 1 && pad >= 0, "conv2d_dgrad_bnstat: bad geometry");
  PSEG_REQUIRE(act == PSEG_ACT_NONE || act == PSEG_ACT_RELU || act == PSEG_ACT_RELU6, "conv2d_dgrad_bnstat: unknown activation");
  PSEG_REQUIRE(ldy_prev % 4 == 0 && ldy_prev >= Cin && ((uintptr_t)y_prev & 15) == 0 && ((uintptr_t)part_db & 15) == 0 &&
                   ((uintptr_t)part_dg & 15) == 0 && ((uintptr_t)mean & 15) == 0 && ((uintptr_t)invstd & 15) == 0 &&
                   ((uintptr_t)scale & 15) == 0 && ((uintptr_t)shift & 15) == 0,
               "conv2d_dgrad_bnstat: y_prev / coefficient / partial pointers must be 16-byte aligned, ldy_prev %% 4 == 0");
  PSEG_REQUIRE(part_rows > 0 && part_rows == pseg_conv2d_dgrad_bnstat_rows(B, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, dil),
               "conv2d_dgrad_bnstat: part_rows (%d) is not pseg_conv2d_dgrad_bnstat_rows() of this problem (0 = not covered)",
               part_rows);
  const BnsArgs bns{y_prev, ldy_prev, mean, invstd, scale, shift, act, part_db, part_dg, part_rows};
  const int K = kh * kw * Cout;
  return run_gather(dy, nhwc_bytes(B, Ho, Wo, Cout, ldy), ldy, wT, dx, ldx, nullptr, nullptr, B, Ho, Wo, Cout, H, W, Cin, kw, K, 1,
                    stride, -dil, pad, 0, PSEG_PREC_FP32, nullptr, nullptr, nullptr, 0, (hipStream_t)stream, nullptr, &bns);
}

int pseg_split_planes(const float* x, int ldx, int64_t M, int C, uint16_t* hi, uint16_t* lo, int ldp, void* stream) {
  PSEG_REQUIRE(x && hi && lo && M > 0 && C > 0, "split_planes: bad argument");
  PSEG_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldp % 8 == 0 && ldp >= C, "split_planes: C %% 4, ldx %% 4, ldp %% 8, ldp >= C");
  PSEG_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)hi & 15) == 0 && ((uintptr_t)lo & 15) == 0, "split_planes: alignment");
  const long long total = (long long)M * (ldp / 8);
  const int blocks = (int)(total / 256 + 1 < 8192 ? total / 256 + 1 : 8192);
  hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, (long long)M, C, hi, lo,
                     ldp);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

// plan of the data gradient as a gather GEMM: rows = input pixels, N = Cin, gather source = dy (Cout channels)
static FwdPlan plan_dgrad(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int dil) {
  DilGeom geom;
  const bool has_geom = dil_geom(geom, H, W, Ho, Wo, kh, kw, Cout, 1, stride, -dil, pad);
  return plan_gather((long long)B * H * W, Cin, kh * kw * Cout, true, has_geom ? &geom : nullptr);
}

int pseg_conv2d_dgrad_planes_ok(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad,
                                int dil) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  return dma_plan_ok(plan_dgrad(B, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, dil), Cout, Cin) ? 1 : 0;
}

int pseg_conv2d_dgrad_planes(const uint16_t* dy_hi, const uint16_t* dy_lo, int ldp, const uint16_t* wT_hi,
                             const uint16_t* wT_lo, float* dx, int ldx, int B, int H, int W, int Cin, int Ho, int Wo,
                             int Cout, int kh, int kw, int stride, int pad, int dil, int accumulate, void* stream) {
  PSEG_REQUIRE(dy_hi && dy_lo && wT_hi && wT_lo && dx, "conv2d_dgrad_planes: null pointer");
  PSEG_REQUIRE(stride >= 1 && dil >= 1 && pad >= 0 && ldp >= Cout, "conv2d_dgrad_planes: bad geometry");
  const int K = kh * kw * Cout;
  LimbPlanes pln{dy_hi, dy_lo, wT_hi, wT_lo, ldp, (((long long)B * Ho * Wo - 1) * ldp + Cout) * 2, (long long)Cin * K * 2};
  // (x / w of run_gather are unused on this path; pass the planes for the alignment checks)
  return run_gather(reinterpret_cast<const float*>(dy_hi), 16, 4, reinterpret_cast<const float*>(wT_hi), dx, ldx, nullptr,
                    nullptr, B, Ho, Wo, Cout, H, W, Cin, kw, K, 1, stride, -dil, pad, accumulate, PSEG_PREC_BF16X3, nullptr,
                    nullptr, nullptr, 0, (hipStream_t)stream, &pln);
}

// All filters of a model in ONE launch: jobs[j] = {w, wT, Cout, taps, Cin, first 32x32 tile of job j} (6 x int64, device
// memory, tile offsets ascending); block b finds its job by bisection.
__global__ __launch_bounds__(256) void filter_transpose_batch_kernel(const long long* __restrict__ jobs, int n) {
  PSEG_HELPER_PRIO();
  __shared__ float tile[32][33];
  const long long b = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid * 6 + 5] <= b) lo = mid;
    else hi = mid - 1;
  }
  const long long* job = jobs + lo * 6;
  const float* w = reinterpret_cast<const float*>(job[0]);
  float* wT = reinterpret_cast<float*>(job[1]);
  const int Cout = (int)job[2], taps = (int)job[3], Cin = (int)job[4];
  const int tci = (Cin + 31) / 32, tco = (Cout + 31) / 32;
  int local = (int)(b - job[5]);
  const int t = local / (tci * tco);
  local -= t * tci * tco;
  if (t >= taps) return;
  const int co0 = (local / tci) * 32, ci0 = (local % tci) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < Cout && ci < Cin) ? w[((long long)co * taps + t) * Cin + ci] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < Cin && co < Cout) wT[((long long)ci * taps + t) * Cout + co] = tile[tx][r];
  }
}

int pseg_filter_transpose_batch(const int64_t* jobs, int n, int64_t total_tiles, void* stream) {
  PSEG_REQUIRE(jobs && n > 0 && total_tiles > 0 && total_tiles < (1LL << 31), "filter_transpose_batch: bad argument");
  hipLaunchKernelGGL(filter_transpose_batch_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(jobs), n);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

int pseg_filter_transpose(const float* w, float* wT, int Cout, int taps, int Cin, void* stream) {
  PSEG_REQUIRE(w && wT && Cout > 0 && taps > 0 && Cin > 0, "filter_transpose: bad argument");
  PSEG_REQUIRE(taps <= 65535, "filter_transpose: too many taps");
  dim3 grid((unsigned)cdiv(Cin, 32), (unsigned)cdiv(Cout, 32), (unsigned)taps);
  hipLaunchKernelGGL(filter_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, wT, Cout, taps, Cin);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

int64_t pseg_conv2d_wgrad_workspace_bytes(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw) {
  // the plan depends on the arithmetic (the limb kernels may take the 256-row tile): size for the larger of the two
  const WgradPlan a = plan_wgrad((long long)B * Ho * Wo, Cout, kh * kw * Cin, false, false);
  const WgradPlan a2 = plan_wgrad((long long)B * Ho * Wo, Cout, kh * kw * Cin, false, false, 0, false);
  const WgradPlan b = plan_wgrad((long long)B * Ho * Wo, Cout, kh * kw * Cin, true, true);
  const WgradPlan c = plan_wgrad((long long)B * Ho * Wo, Cout, kh * kw * Cin, false, true);
  int splits = a.splits > b.splits ? a.splits : b.splits;
  if (c.splits > splits) splits = c.splits;
  if (a2.splits > splits) splits = a2.splits;
  return splits > 1 ? (int64_t)splits * Cout * kh * kw * Cin * 4 : 0;
}

// defer != 0: a split plan leaves its slabs in the workspace (the caller reduces them later, pseg_slab_reduce_batch)
static int run_wgrad(const float* x, int ldx, const float* dy, int ldy, float* dw, int B, int H, int W, int Cin, int Ho,
                     int Wo, int Cout, int kh, int kw, int stride, int pad, int dil, int accumulate, int precision,
                     void* workspace, int64_t workspace_bytes, void* stream, int defer, int concurrent) {
  PSEG_REQUIRE(x && dy && dw, "conv2d_wgrad: null pointer");
  PSEG_REQUIRE(precision >= 0 && precision <= 2, "conv2d_wgrad: precision must be PSEG_PREC_FP32 / _BF16X3 / _BF16X6");
  PSEG_REQUIRE(Cin % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "conv2d_wgrad: Cin, ldx, ldy must be multiples of 4");
  PSEG_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0, "conv2d_wgrad: x / dy must be 16-byte aligned");
  const long long P = (long long)B * Ho * Wo;
  const int K = kh * kw * Cin;
  PSEG_REQUIRE(P > 0 && P < (1LL << 31), "conv2d_wgrad: bad pixel count");
  const long long xb = nhwc_bytes(B, H, W, Cin, ldx), db = nhwc_bytes(B, Ho, Wo, Cout, ldy);
  // dy chunks are read 4 channels at a time: the last chunk of a row may run up to 3 floats past Cout (inside ldy)
  PSEG_REQUIRE(xb < kMaxBytes && db < kMaxBytes, "conv2d_wgrad: tensor exceeds 2 GiB");
  PSEG_REQUIRE((Cout + 3) / 4 * 4 <= ldy, "conv2d_wgrad: ldy must cover Cout rounded up to 4");
  WgradPlan pl = plan_wgrad(P, Cout, K, precision == 1, precision != 0, 0, concurrent != 0);
  WgradParams p;
  p.x = x;
  p.dy = dy;
  p.x_bytes = (uint32_t)xb;
  p.dy_bytes = (uint32_t)(((P - 1) * ldy + (Cout + 3) / 4 * 4) * 4);
  p.ldx = ldx;
  p.ldy = ldy;
  p.Hi = H;
  p.Wi = W;
  p.Cin = Cin;
  p.Ho = Ho;
  p.Wo = Wo;
  p.HoWo = Ho * Wo;
  p.Cout = Cout;
  p.K = K;
  p.P = (int)P;
  p.kw = kw;
  p.stride = stride;
  p.pad = pad;
  p.dil = dil;
  p.pix_per_split = pl.pix_per_split;
  // a column tile must sit inside one tap for the skip test to be block-uniform
  const bool can_skip = (dil >= 4 && kh * kw > 1 && Cin % pl.tile.bn == 0 && cfg().conv_noskip == 0);
  p.skip_rows = can_skip ? 1 : 0;
  p.patch_mode = 0;
  p.patch_h = 1;
  p.patch_w = BK;
  if (P % BK == 0 && ((long long)Ho * Wo) % BK == 0 && cfg().conv_noskip == 0) {
    // pixel order of the contraction: 32-pixel K-steps as PH x PW patches of the output map (PW % 8 == 0).  Addresses of
    // a K-step are then a block-uniform origin plus thread constants, and a dilated tap is dead for the whole step when
    // its rows OR its columns are out of range: take the shape that leaves the fewest live (K-step, tap) pairs
    // (32x32 map: rate 12 -> 4x8 = 0.63 against 0.75 row-major, rate 18 -> 2x16 = 0.42 against 0.63).
    DilGeom g{Ho, Wo, H, W, kh, kw, dil, -pad};
    double best = 2.0;
    for (int pw = BK; pw >= 8; pw /= 2) {
      const int ph = BK / pw;
      if (pw > Wo || Wo % pw != 0 || Ho % ph != 0) continue;
      const double f = (can_skip && stride == 1) ? live_fraction(g, ph, pw) : 1.0;
      if (f < best - 1e-9) {
        best = f;
        p.patch_mode = 1;
        p.patch_h = ph;
        p.patch_w = pw;
      }
    }
    if (p.patch_mode) {
      p.ppr = FastDiv((uint32_t)(Wo / p.patch_w));
      p.ppi = FastDiv((uint32_t)((Ho / p.patch_h) * (Wo / p.patch_w)));
      if (can_skip) p.skip_rows = 2;
    }
  }
  p.lpt_per = 0;
  for (int t = 0; t < 9; ++t) p.tap_order[t] = t;
  // narrow 3x3 filters on few channels: the halo-staged kernel (wgrad_f32_halo_kernel) on the plan's pixel splits -- a block is
  // (row tile of 32 filters, one 32-channel chunk of the input, one split); K-steps are 2 x 16 strips of the output map.
  // Measured (profiles/EXPERIMENTS.md 5.13): 32 -> 32 on 128x128 53 -> 45 us, HRNet fp32 14.19 -> 14.07 ms; the classifier (384
  // channels: 583 -> 540 us alone) is SLOWER inside the two-stream DeepLabV3+ step (42.93 -> 43.03 ms) and 64-filter layers lose
  // outright (64 -> 64: 40 -> 72 us) -- hence at most 64 input channels and 32 filters; PSEG_WGRAD_HALO=2 lifts the channel cap.
  const bool halo = precision == 0 && cfg().wgrad_halo != 0 && cfg().wgrad_f32dma != 0 && kh == 3 && kw == 3 && stride == 1 &&
                    dil == 1 && pad == 1 && Cin % 32 == 0 && (Cin <= 64 || cfg().wgrad_halo >= 2) && Cout <= 32 && H == Ho && W == Wo && Ho % 2 == 0 && Wo % 16 == 0 &&
                    P % BK == 0 && pl.pix_per_split % BK == 0 && (Cout + 3) / 4 * 4 <= ldy;
  if (halo) {
    p.skip_rows = 0;
    p.patch_mode = 1;
    p.patch_h = 2;
    p.patch_w = 16;
    p.ppr = FastDiv((uint32_t)(Wo / 16));
    p.ppi = FastDiv((uint32_t)((Ho / 2) * (Wo / 16)));
    pl.tile.bm = 32;
    pl.tile.bn = 288;
    pl.gridM = cdiv(Cout, 32);
    pl.gridN = Cin / 32;
  }
  const bool dma_tile = (pl.tile.bm == 128 && (pl.tile.bn == 128 || pl.tile.bn == 64)) || (pl.tile.bm == 64 && pl.tile.bn == 128) ||
                        (pl.tile.bm == 32 && (pl.tile.bn == 256 || pl.tile.bn == 288));
  p.rm_howo = FastDiv((uint32_t)(Ho * Wo));
  p.rm_wo = FastDiv((uint32_t)Wo);
  if (precision == 0 && !p.patch_mode && cfg().wgrad_f32dma != 0 && dma_tile && P < (1LL << 31)) {
    // a map that does not tile into 32-pixel patches: the LDS-DMA kernel in plain row-major pixel order (per-lane pixel
    // derivation; no dead-step skipping), instead of the register-staged kernel
    p.skip_rows = 4;
  }
  static const int packed_on = env_int("PSEG_WGRAD_PACKED", 1);
  if (precision == 0 && can_skip && stride == 1 && p.patch_mode && cfg().wgrad_f32dma != 0 && dma_tile && kh * kw <= 9 &&
      (Cout + 3) / 4 * 4 <= ldy && packed_on != 0) {
    // dilated conv on the LDS-DMA kernel: packed live rectangles instead of 32-pixel patches (WgradParams::lpt_per)
    p.skip_rows = 3;
    // longest-first order of the taps (live area, descending; ties keep the tap order: deterministic)
    int area[9];
    const int ntap = kh * kw;
    for (int t = 0; t < ntap; ++t) {
      const int dh = (t / kw) * dil - pad, dwv = (t % kw) * dil - pad;
      int r0 = dh < 0 ? -dh : 0, c0 = dwv < 0 ? -dwv : 0;
      int r1 = H - dh < Ho ? H - dh : Ho, c1 = W - dwv < Wo ? W - dwv : Wo;
      area[t] = (r1 > r0 ? r1 - r0 : 0) * (c1 > c0 ? c1 - c0 : 0);
    }
    for (int i = 1; i < ntap; ++i)
      for (int j = i; j > 0 && area[p.tap_order[j]] > area[p.tap_order[j - 1]]; --j) {
        const int t = p.tap_order[j];
        p.tap_order[j] = p.tap_order[j - 1];
        p.tap_order[j - 1] = t;
      }
    // one group = what one XCD walks in order (wgrad_block: eight contiguous ranges of the (split, tile) pairs)
    const int tpt = Cin / pl.tile.bn;
    const long long total = (long long)pl.gridM * pl.gridN * pl.splits;
    int groups = 1;
    if (total % 8 == 0 && pl.gridN % (total / 8) == 0) groups = (int)(pl.gridN / (total / 8));
    if (groups < 1 || tpt % groups != 0) groups = 1;
    p.lpt_per = tpt / groups;
  }
  const long long wsz = (long long)Cout * K;
  if (pl.splits == 1) {
    p.dw = dw;
    p.accumulate = accumulate;
    p.slab_stride = 0;
  } else {
    const long long need = (long long)pl.splits * wsz * 4;
    if (workspace == nullptr || workspace_bytes < need) {
      set_error("conv2d_wgrad: needs %lld workspace bytes, got %lld", need, (long long)workspace_bytes);
      return PSEG_ERR_WORKSPACE;
    }
    p.dw = (float*)workspace;
    p.accumulate = 0;
    p.slab_stride = wsz;
  }
  const dim3 grid((unsigned)(pl.gridM * pl.gridN), 1, (unsigned)pl.splits);
  if (precision == 0 && (p.patch_mode || p.skip_rows == 4) && cfg().wgrad_f32dma != 0 && (Cout + 3) / 4 * 4 <= ldy) {
    // exact-fp32 weight gradient on the LDS-DMA kernel (8 waves, two blocks per CU); same tile, same split plan
    const bool sk = p.skip_rows != 0 && p.skip_rows != 4;
    bool launched = true;
    hipStream_t st = (hipStream_t)stream;
    g_last_conv_kernel = halo ? PSEG_KERNEL_WGRAD_HALO : PSEG_KERNEL_WGRAD_DMA;
    if (halo) {
      hipLaunchKernelGGL(wgrad_f32_halo_kernel<32>, grid, dim3(576), 0, st, p);
    } else if (pl.tile.bm == 128 && pl.tile.bn == 128) {
      if (sk) hipLaunchKernelGGL((wgrad_f32_dma_kernel<128, 128, 2, 4, true>), grid, dim3(512), 0, st, p);
      else hipLaunchKernelGGL((wgrad_f32_dma_kernel<128, 128, 2, 4, false>), grid, dim3(512), 0, st, p);
    } else if (pl.tile.bm == 128 && pl.tile.bn == 64) {
      if (sk) hipLaunchKernelGGL((wgrad_f32_dma_kernel<128, 64, 4, 2, true>), grid, dim3(512), 0, st, p);
      else hipLaunchKernelGGL((wgrad_f32_dma_kernel<128, 64, 4, 2, false>), grid, dim3(512), 0, st, p);
    } else if (pl.tile.bm == 64 && pl.tile.bn == 128) {
      if (sk) hipLaunchKernelGGL((wgrad_f32_dma_kernel<64, 128, 2, 4, true>), grid, dim3(512), 0, st, p);
      else hipLaunchKernelGGL((wgrad_f32_dma_kernel<64, 128, 2, 4, false>), grid, dim3(512), 0, st, p);
    } else if (pl.tile.bm == 32 && pl.tile.bn == 256) {
      // narrow outputs (the 21-class classifier, HRNet's 32-channel branch): 32 x 256 tile, the x operand by DMA
      if (sk) hipLaunchKernelGGL((wgrad_f32_dma_kernel<32, 256, 1, 8, true>), grid, dim3(512), 0, st, p);
      else hipLaunchKernelGGL((wgrad_f32_dma_kernel<32, 256, 1, 8, false>), grid, dim3(512), 0, st, p);
    } else if (pl.tile.bm == 32 && pl.tile.bn == 288) {
      // 3x3 on 32 channels: the whole filter as one column tile, nine waves
      hipLaunchKernelGGL((wgrad_f32_dma_kernel<32, 288, 1, 9, false>), grid, dim3(576), 0, st, p);
    } else {
      launched = false;
    }
    if (launched) {
      PSEG_LAUNCH_CHECK();
      if (pl.splits > 1 && !defer) {
        const int blocks = (int)(wsz / 256 + 1 < 4096 ? wsz / 256 + 1 : 4096);
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, wsz, pl.splits, dw,
                           K, (long long)Cout, K, (const float*)nullptr, accumulate);
        PSEG_LAUNCH_CHECK();
      }
      return PSEG_OK;
    }
  }
  g_last_conv_kernel = precision == 0 ? PSEG_KERNEL_WGRAD_REGISTER : PSEG_KERNEL_WGRAD_LIMB;
  if (precision == 0 && pl.tile.bm == 32 && pl.tile.bn == 288) {
    set_error("conv2d_wgrad: the 32 x 288 tile runs on the LDS-DMA kernel only (PSEG_WGRAD_F32DMA=0 with PSEG_WGRAD_NARROW288=1?)");
    return PSEG_ERR_ARG;
  }
  if (precision == 0 && pl.tile.bm == 32 && pl.tile.bn == 256) {
    // (the 32 x 256 tile off the DMA kernel -- map sizes that do not tile into 32-pixel patches: register-staged, two
    // accumulators per wave)
    if (p.skip_rows != 0) hipLaunchKernelGGL((wgrad_kernel<32, 256, 1, 4, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((wgrad_kernel<32, 256, 1, 4, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
    PSEG_LAUNCH_CHECK();
    if (pl.splits > 1 && !defer) {
      const int blocks = (int)(wsz / 256 + 1 < 4096 ? wsz / 256 + 1 : 4096);
      hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, wsz,
                         pl.splits, dw, K, (long long)Cout, K, (const float*)nullptr, accumulate);
      PSEG_LAUNCH_CHECK();
    }
    return PSEG_OK;
  }
  typedef void (*Kfn)(const WgradParams);
  static const Kfn fns[2][5] = {
      {wgrad_kernel<128, 128, 2, 2, false>, wgrad_kernel<128, 64, 2, 2, false>, wgrad_kernel<128, 32, 4, 1, false>,
       wgrad_kernel<64, 128, 2, 2, false>, wgrad_kernel<32, 128, 1, 4, false>},
      {wgrad_kernel<128, 128, 2, 2, true>, wgrad_kernel<128, 64, 2, 2, true>, wgrad_kernel<128, 32, 4, 1, true>,
       wgrad_kernel<64, 128, 2, 2, true>, wgrad_kernel<32, 128, 1, 4, true>}};
#define PSEG_WLIMB_ROW(SK, NLIMB)                                                                                 \
  wgrad_limb_kernel<128, 128, 2, 2, SK, NLIMB>, wgrad_limb_kernel<128, 64, 2, 2, SK, NLIMB>,                      \
      wgrad_limb_kernel<128, 32, 4, 1, SK, NLIMB>, wgrad_limb_kernel<64, 128, 2, 2, SK, NLIMB>,                   \
      wgrad_limb_kernel<32, 128, 1, 4, SK, NLIMB>
  static const Kfn fnsb3[2][6] = {{PSEG_WLIMB_ROW(false, 2), wgrad_limb_kernel<256, 128, 4, 2, false, 2>},
                                  {PSEG_WLIMB_ROW(true, 2), wgrad_limb_kernel<256, 128, 4, 2, true, 2>}};
  static const Kfn fnsb6[2][5] = {{PSEG_WLIMB_ROW(false, 3)}, {PSEG_WLIMB_ROW(true, 3)}};
#undef PSEG_WLIMB_ROW
  int rc = precision == 2   ? launch_tiles<WgradParams, Kfn, 5>(fnsb6, p.skip_rows != 0, pl.tile, grid, p, (hipStream_t)stream)
           : precision == 1 ? launch_tiles<WgradParams, Kfn, 6>(fnsb3, p.skip_rows != 0, pl.tile, grid, p, (hipStream_t)stream)
                            : launch_tiles<WgradParams, Kfn, 5>(fns, p.skip_rows != 0, pl.tile, grid, p, (hipStream_t)stream);
  if (rc != PSEG_OK) return rc;
  if (pl.splits > 1 && !defer) {
    const int blocks = (int)(wsz / 256 + 1 < 4096 ? wsz / 256 + 1 : 4096);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, wsz,
                       pl.splits, dw, K, (long long)Cout, K, (const float*)nullptr, accumulate);
    PSEG_LAUNCH_CHECK();
  }
  return PSEG_OK;
}

int pseg_conv2d_wgrad(const float* x, int ldx, const float* dy, int ldy, float* dw, int B, int H, int W, int Cin, int Ho,
                      int Wo, int Cout, int kh, int kw, int stride, int pad, int dil, int accumulate, int precision,
                      int concurrent, void* workspace, int64_t workspace_bytes, void* stream) {
  return run_wgrad(x, ldx, dy, ldy, dw, B, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, dil, accumulate, precision,
                   workspace, workspace_bytes, stream, 0, concurrent);
}

int pseg_conv2d_wgrad_splits(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw, int precision, int concurrent) {
  if (B <= 0 || Ho <= 0 || Wo <= 0 || Cin <= 0 || Cout <= 0 || precision < 0 || precision > 2) return 0;
  return plan_wgrad((long long)B * Ho * Wo, Cout, kh * kw * Cin, precision == 1, precision != 0, 0, concurrent != 0).splits;
}

int pseg_conv2d_wgrad_slabs(const float* x, int ldx, const float* dy, int ldy, float* slabs, int B, int H, int W, int Cin,
                            int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int dil, int precision,
                            int concurrent, int64_t slab_bytes, void* stream) {
  PSEG_REQUIRE(pseg_conv2d_wgrad_splits(B, Ho, Wo, Cin, Cout, kh, kw, precision, concurrent) > 1,
               "conv2d_wgrad_slabs: this plan does not split -- call pseg_conv2d_wgrad");
  // (dw is unused by a split plan; the slabs pointer stands in for the null check)
  return run_wgrad(x, ldx, dy, ldy, slabs, B, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, dil, 0, precision, slabs,
                   slab_bytes, stream, 1, concurrent);
}

int pseg_slab_reduce_batch(const int64_t* jobs, int n, int64_t total_blocks, int accumulate, void* stream) {
  PSEG_REQUIRE(jobs && n > 0 && total_blocks > 0 && total_blocks < (1LL << 31), "slab_reduce_batch: bad argument");
  hipLaunchKernelGGL(slab_reduce_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(jobs), n, accumulate);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

int pseg_slab_reduce_block(void) { return kSlabBlock; }

}  // extern "C"
