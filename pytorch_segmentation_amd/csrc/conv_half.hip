// Half-precision (`-mp`) convolution for gfx950: fp16 activations / gradients / filters in HBM, single-pass
// v_mfma_f32_32x32x16_f16 with fp32 accumulation.  This is the arithmetic the reference asks apex for with `-mp`
// (train.py:70,102-105,138; README.md:12: fp16 compute, fp32 master weights, loss scaling).
//
//  gather_h_kernel   forward and data gradient: C[M = pixels][N] = A[M][K] * Bw[N][K]^T, both operands K-contiguous fp16.
//     Same implicit-GEMM formulation, GEMM-row orders (patch / parity / liveness-class sorted), tap skipping and fused
//     BatchNorm statistics as gather_f32_dma_kernel (conv_mfma.hip); tiles go global -> LDS by `buffer_load_dwordx4 ... lds`
//     into a two-stage ring of [rows][64 halves = 128 B] images (16-byte k-slot XOR (row >> 1) & 7, applied on the source
//     side), one ds_read_b128 = the 8 consecutive k of one MFMA operand.  Output fp16 (or fp32 for the class logits, which
//     the loss reads), written in 16-byte pieces through a wave-private LDS patch.
//     GENERIC = true: channel counts that are not multiples of 64 (HRNet's 32-channel branch, MobileNetV2, the stem): a
//     K-step may straddle taps, every lane derives (tap, channel) of its own 16-byte slot.
//  wgrad_h_kernel    weight gradient: dW[Cout][K] = dY[P][Cout]^T * A[P][K], contraction over pixels.  Both operands lie
//     pixel-major in memory ([pixel][channels]) and are DMA'd as they lie into [64 px][cols] LDS images; the MFMA wants 8
//     consecutive k (= pixels) of one channel per lane, which `ds_read_b64_tr_b16` delivers from the pixel-major image (a
//     4-pixel x 16-channel block per 16 lanes, transposed on the way out).  16-byte chunks are XOR-swizzled by the pixel
//     row (on the source side of the DMA) so that the four rows of a block fall on different banks.  fp32 slabs per pixel
//     split, reduced in a fixed order by the same slab kernels as the fp32 path (bit-reproducible).
#include "conv_common.h"
#include "half_io.h"

#include <type_traits>

namespace pseg {

constexpr int BKH = 64;    // long K-step of the gather kernel in halves (128-byte LDS rows); the short one is 32
// (the weight-gradient kernel contracts over BKP = 64 or 32 pixels per K-step: two or one 32-pixel sub-steps)

// s_waitcnt vmcnt(N) with a compile-time N (the instruction takes an immediate)
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 24, "vmcnt immediate");
#define PSEG_VMCNT_CASE(V) else if constexpr (N == V) asm volatile("s_waitcnt vmcnt(" #V ")" ::: "memory")
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PSEG_VMCNT_CASE(1); PSEG_VMCNT_CASE(2); PSEG_VMCNT_CASE(3); PSEG_VMCNT_CASE(4); PSEG_VMCNT_CASE(5); PSEG_VMCNT_CASE(6);
  PSEG_VMCNT_CASE(7); PSEG_VMCNT_CASE(8); PSEG_VMCNT_CASE(9); PSEG_VMCNT_CASE(10); PSEG_VMCNT_CASE(11); PSEG_VMCNT_CASE(12);
  PSEG_VMCNT_CASE(13); PSEG_VMCNT_CASE(14); PSEG_VMCNT_CASE(15); PSEG_VMCNT_CASE(16); PSEG_VMCNT_CASE(17); PSEG_VMCNT_CASE(18);
  PSEG_VMCNT_CASE(19); PSEG_VMCNT_CASE(20); PSEG_VMCNT_CASE(21); PSEG_VMCNT_CASE(22); PSEG_VMCNT_CASE(23); PSEG_VMCNT_CASE(24);
#undef PSEG_VMCNT_CASE
}

// ds_read_b64_tr_b16 as inline assembly, with the byte offset as an immediate.  Why not the builtin: hipcc treats the
// intrinsic as a read of ANY LDS byte and puts `s_waitcnt vmcnt(0)` in front of it whenever an LDS-DMA (`buffer_load ... lds`, a
// pending LDS write on the VM counter) is in flight -- which drains the operand ring once per K-step in the weight gradient
// and once per tile in the persistent kernel's epilogue (round 4: found in the disassembly).  An asm statement is invisible to
// that pass; its result is NOT covered by the compiler's own waits either: the caller waits (`s_waitcnt lgkmcnt`) before use.
typedef short s16x4t __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ s16x4t tr_read_asm(uint32_t lds_byte_addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  s16x4t v;
#if defined(__HIP_DEVICE_COMPILE__) && defined(PSEG_TR_BUILTIN) && PSEG_TR_BUILTIN
  // (A/B build, `python -m pytorch_segmentation_amd.csrc.build --trbuiltin`: the compiler-visible read, with its vmcnt(0))
  typedef __attribute__((address_space(3))) s16x4t* lds_s16x4p;
  v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4p)(uintptr_t)(lds_byte_addr + OFF));
#elif defined(__HIP_DEVICE_COMPILE__)
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_byte_addr), "i"(OFF));
#else
  v = s16x4t{0, 0, 0, 0};
  (void)lds_byte_addr;
#endif
  return v;
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// keeps a value live without code (ablation builds: the compiler must not delete the work that produced it)
__device__ __forceinline__ void keep_alive(const f32x16& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" ::"v"(v));
#else
  (void)v;
#endif
}

// Lab build (-DPSEG_LAB=1: `python -m pytorch_segmentation_amd.csrc.build --lab` -> libpseg_amd_lab.so, PSEG_LIB_PATH selects it):
// variants that were measured and NOT picked stay reachable there and nowhere else -- PSEG_HCONV_TILE = 1..4 (128x128 on four
// waves, 256x128 on eight / sixteen, 256x256), PSEG_HCONV_ABLATE (stores / statistics / DMAs / fragment reads / MFMAs switched
// off: results are then wrong).  The product library holds neither the instantiations nor the branches.
#ifndef PSEG_LAB
#define PSEG_LAB 0
#endif
#if PSEG_LAB
#define PSEG_ABLATE(hp_) ((hp_).ablate)
#else
#define PSEG_ABLATE(hp_) 0
#endif

struct HGatherParams {
  GatherConvParams g;    // x / w / y are fp16 here (y fp32 when y_f32); element strides as in the fp32 kernels
  int y_f32;
  FastDiv cin_div, kw_div;   // GENERIC: k -> (tap, channel), tap -> (row, column)
  FastDiv howo_div, wo_div;  // gather_hp_kernel: GEMM row -> (image, row, column)
  int ntiles;                // gather_hp_kernel: tiles of the launch (a persistent block walks blockIdx, blockIdx + grid, ...)
  int ablate;                // diagnostics (PSEG_HCONV_ABLATE, profiles/scripts/exp_ablate.sh; results are then WRONG): 1 no stores, 2 no
                             // statistics, 4 no operand DMAs after the prologue, 8 no MFMAs, 16 no fragment reads
};

// One 32-row tile row of a wave's accumulators -> global memory through a wave-private [32][WTN + 4] fp32 patch, 8 columns
// per lane: 16-byte stores of fp16 (or two of fp32).  Called once per tile row, so the patch is a quarter / half of what the
// whole wave tile would need: LDS per block is set by the operand ring, not by the epilogue (more blocks per CU).
// BNS (a data gradient that feeds a BatchNorm backward; GatherConvParams::bns_y, fp16 here): the lanes that store eight channels
// of a row of dx read the producing layer's y at the same place and keep sum(g) / sum(g * x_hat) per channel, g = the value AS
// STORED (rounded to fp16: what bn_bwd_reduce_kernel would read back) under the activation mask of the forward pass's own
// expression.  The coefficients of the lane's eight channels are loaded once per tile (HBnsCoef), the sums are folded over
// the lanes of a channel octet after the last tile row (bns_fold8).
struct HBnsCoef {
  f32x4 mu[2], is[2], sc[2], sh[2];
};
struct HBnsSums {
  f32x4 s1[2], s2[2];
};

__device__ __forceinline__ void bns_load_coef(const GatherConvParams& p, int col, bool ok, HBnsCoef& c) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    c.mu[h] = ok ? *reinterpret_cast<const f32x4*>(p.bns_mean + col + 4 * h) : z;
    c.is[h] = ok ? *reinterpret_cast<const f32x4*>(p.bns_invstd + col + 4 * h) : z;
    c.sc[h] = ok ? *reinterpret_cast<const f32x4*>(p.bns_scale + col + 4 * h) : z;
    c.sh[h] = ok ? *reinterpret_cast<const f32x4*>(p.bns_shift + col + 4 * h) : z;
  }
}

// eight stored channels of one row (ov) against the layer's y at `yrow` (channels col ..)
__device__ __forceinline__ void bns_add8(const f16x8v& ov, const f16x8v& yv, int act, const HBnsCoef& c, HBnsSums& s) {
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = (float)yv[4 * h + e] - c.mu[h][e];
      const float pre = d * c.sc[h][e] + c.sh[h][e];       // the forward pass's own expression: the mask is the one it applied
      bool on = true;
      if (act == PSEG_ACT_RELU) on = pre > 0.f;
      else if (act == PSEG_ACT_RELU6) on = (pre > 0.f) && (pre < 6.f);
      const float g = on ? (float)ov[4 * h + e] : 0.f;
      s.s1[h][e] += g;
      s.s2[h][e] += g * (d * c.is[h][e]);
    }
}

template <int TN, bool BNS, typename RowMap>
__device__ __forceinline__ void store_row32(const f32x16 (&acc)[TN], float* patch, void* out, bool out_f32, long long ld, int row0,
                                            int col0, int rows_valid, int cols_valid, const float* bias, bool accumulate,
                                            int lane, RowMap&& out_row, const GatherConvParams& bns, const HBnsCoef& bc,
                                            HBnsSums& bs) {
  constexpr int WTN = TN * 32, LDW = WTN + 4;
  const int col_l = lane & 31;
  const int row_h = (lane >> 5) * 4;
  // BNS: the rows of y this lane will need are requested before the accumulators make their way through the patch
  constexpr int kC8 = WTN / 8, kRPI = 64 / kC8;
  f16x8v ypre[32 / kRPI];
  if constexpr (BNS) {
#pragma unroll
    for (int it = 0; it < 32 / kRPI; ++it) {
      const int row = it * kRPI + lane / kC8;
      const int colp = (lane % kC8) * 8;
      ypre[it] = f16x8v{0, 0, 0, 0, 0, 0, 0, 0};
      if (colp < cols_valid && row < rows_valid)
        ypre[it] = *reinterpret_cast<const f16x8v*>(reinterpret_cast<const half_t*>(bns.bns_y) +
                                                    (long long)out_row(row0 + row) * bns.bns_ldy + col0 + colp);
    }
  }
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + row_h) * LDW + j * 32 + col_l] = acc[j][r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  constexpr int C8 = WTN / 8;          // 8-column chunks per row
  constexpr int RPI = 64 / C8;         // rows per wave-instruction
  const int c8 = lane % C8, rr = lane / C8;
  const int col = c8 * 8;
  const bool cok = col < cols_valid;   // cols_valid is a multiple of 8
  f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
  if (bias != nullptr && cok) {
    b0 = *reinterpret_cast<const f32x4*>(bias + col0 + col);
    b1 = *reinterpret_cast<const f32x4*>(bias + col0 + col + 4);
  }
#pragma unroll
  for (int it = 0; it < 32 / RPI; ++it) {
    const int row = it * RPI + rr;
    if (cok && row < rows_valid) {
      f32x4 v0 = *reinterpret_cast<const f32x4*>(&patch[row * LDW + col]) + b0;
      f32x4 v1 = *reinterpret_cast<const f32x4*>(&patch[row * LDW + col + 4]) + b1;
      const long long orow = (long long)out_row(row0 + row);
      const long long o = orow * ld + col0 + col;
      if (out_f32) {
        float* gp = reinterpret_cast<float*>(out) + o;
        if (accumulate) {
          v0 += *reinterpret_cast<const f32x4*>(gp);
          v1 += *reinterpret_cast<const f32x4*>(gp + 4);
        }
        *reinterpret_cast<f32x4*>(gp) = v0;
        *reinterpret_cast<f32x4*>(gp + 4) = v1;
      } else {
        half_t* gp = reinterpret_cast<half_t*>(out) + o;
        if (accumulate) {
          const f16x8v old = *reinterpret_cast<const f16x8v*>(gp);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v0[e] += (float)old[e];
            v1[e] += (float)old[4 + e];
          }
        }
        f16x8v ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ov[e] = (half_t)v0[e];
          ov[4 + e] = (half_t)v1[e];
        }
        *reinterpret_cast<f16x8v*>(gp) = ov;
        if constexpr (BNS) bns_add8(ov, ypre[it], bns.bns_act, bc, bs);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();      // the patch is rewritten by the next tile row
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int TN, typename RowMap>
__device__ __forceinline__ void store_row32(const f32x16 (&acc)[TN], float* patch, void* out, bool out_f32, long long ld, int row0,
                                            int col0, int rows_valid, int cols_valid, const float* bias, bool accumulate,
                                            int lane, RowMap&& out_row) {
  GatherConvParams none;       // (never read: BNS == false)
  HBnsCoef bc;
  HBnsSums bs;
  store_row32<TN, false>(acc, patch, out, out_f32, ld, row0, col0, rows_valid, cols_valid, bias, accumulate, lane, out_row, none,
                         bc, bs);
}

// KB: halves per K-step = 64 (128-byte LDS rows, 8 rows per DMA wave-instruction) or 32 (64-byte rows, 16 rows per
// wave-instruction).  The short K-step halves the ring -- with the 32-row epilogue patch a 128x128 / 8-wave block needs 36 KB of
// LDS and four of them share a CU (32 waves) -- which is what the many short, latency-bound launches of a training step want
// (a layer's blocks spend most of their life waiting for their first tile and for their stores); the long K-step has half
// the barriers per MAC and serves the deep contractions.
// STAGES: depth of the operand ring.  A block keeps STAGES - 1 tiles in flight; with one fp16 MFMA pass per tile the matrix
// work of a K-step (128-512 cycles) is far shorter than a memory round trip (1500+ cycles under load), so the rate of a CU is
// bytes in flight / latency (Little's law): the ring depth x resident blocks, not the issue rate, is what sets it.
template <int BM, int BN, int WARPS_M, int WARPS_N, bool SKIP, bool GENERIC, int KB, int STAGES, bool BNS = false>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N) void gather_h_kernel(const HGatherParams hp) {
  const GatherConvParams& p = hp.g;
  set_wave_prio(p.prio);
  static_assert(!(SKIP && GENERIC), "tap skipping needs whole K-steps per tap");
  static_assert(KB == 64 || KB == 32, "K-step");
  constexpr int NW = WARPS_M * WARPS_N;
  static_assert(NW == 16 || NW == 8 || NW == 4, "16, 8 or 4 waves");
  constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N, TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1 && WTM % 32 == 0 && WTN % 32 == 0, "wave tile");
  constexpr int RDW = KB / 2;                     // dwords per LDS row (32 or 16)
  constexpr int NSLOT = KB / 8;                   // 16-byte k-slots per row (8 or 4)
  constexpr int RPG = 64 / NSLOT;                 // rows per DMA wave-instruction (8 or 16)
  constexpr int kStageDw = (BM + BN) * RDW;
  constexpr int kPatch = NW * 32 * (WTN + 4);
  static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
  constexpr int kLds = STAGES * kStageDw > kPatch ? STAGES * kStageDw : kPatch;
  static_assert(kLds * 4 <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) float lds[kLds];
  unsigned* ldsw = reinterpret_cast<unsigned*>(lds);
  constexpr int kA = 0, kB = BM * RDW;
  // DMA row groups (RPG rows each): wave w owns A groups w, w + NW, ... and B groups likewise
  constexpr int GA = BM / RPG / NW, GB = (BN / RPG + NW - 1) / NW, NG = GA + GB;
  static_assert((BM / RPG) % NW == 0 && GA >= 1, "whole A row groups per wave");
  constexpr bool kBPartial = (BN / RPG) % NW != 0;   // narrow tiles: fewer B groups than waves

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

  // image swizzle: k-slot s of row r lives at physical slot s ^ f(r); f(r) = (r >> 1) & 7 for 128-byte rows, (r >> 2) & 3
  // for 64-byte rows (the 16-lane groups of the fragment ds_read_b128 then cover all 64 banks once)
  auto fsw = [](int row) -> int { return KB == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); };
  const int lrow = lane / NSLOT, lslot = lane % NSLOT;
  // logical k-slot of this lane: the same for every row group (rows RPG * (wave + NW * g) + lrow: f(row) does not depend
  // on g while NW is even)
  const int lslot_log = lslot ^ fsw(RPG * wave + lrow);
  int a_bh[GA], a_bw[GA], a_img[GA];
  bool a_ok[GA];
#pragma unroll
  for (int g = 0; g < GA; ++g) {
    const int row = RPG * (wave + NW * g) + lrow;
    const int m = m0 + row;
    const bool ok = m < p.M;
    int b, ho, wo;
    row_to_pixel(p, ok ? m : 0, b, ho, wo);
    a_ok[g] = ok;
    a_bh[g] = ho * p.s_out + p.off0;
    a_bw[g] = wo * p.s_out + p.off0;
    a_img[g] = b * p.Hi * p.Wi;
  }
  uint32_t b_rowoff[GB];
  bool b_ok[GB];
  int b_grp[GB];
#pragma unroll
  for (int g = 0; g < GB; ++g) {
    // (narrow tile: a wave beyond the tile's B groups re-loads group (wave mod groups) -- the same bytes into the same
    // place as its owner, harmless -- so that every wave issues, and counts, the same number of DMAs per K-step)
    b_grp[g] = kBPartial ? (wave + NW * g) % (BN / RPG) : (wave + NW * g);
    const int row = RPG * b_grp[g] + lrow;
    b_ok[g] = (n0 + row) < p.N;
    b_rowoff[g] = b_ok[g] ? (uint32_t)(n0 + row) * (uint32_t)p.K * 2u + (uint32_t)(lslot_log * 16) : kOOB;
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
  int n_steps;
  int s_chunk = 0, s_lin = 0;
  unsigned s_tm = tapmask;
  if (!SKIP) {
    n_steps = kt_end;            // every K-step, in order (any number of taps)
  } else {
    tapmask &= (p.ntaps >= 32) ? 0xFFFFFFFFu : ((1u << p.ntaps) - 1u);      // (host: skipping only with <= 32 taps)
    s_tm = tapmask;
    n_steps = __builtin_popcount(tapmask) * p.ktiles_per_tap;
  }
  auto next_kt = [&]() -> int {     // next live K-step (tap major), kt_end when exhausted
    if (!SKIP) {
      const int kt = s_lin < kt_end ? s_lin : kt_end;
      ++s_lin;
      return kt;
    }
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
    if (kt < kt_end) {
      if (GENERIC) {
        // this lane's 16-byte slot: k = kt * KB + 8 * slot -> (tap, channel); slots beyond K read as zeros
        const int kb = kt * KB + lslot_log * 8;
        const bool kin = kb < p.K;
        const uint32_t tap = hp.cin_div.div((uint32_t)kb);
        const int c = kb - (int)tap * p.Cin;
        const uint32_t kr = hp.kw_div.div(tap);
        const int ks = (int)tap - (int)kr * p.kw;
#pragma unroll
        for (int g = 0; g < GA; ++g) {
          int hn, wn_;
          const bool ok = row_tap_ok(g, (int)kr * p.dstep, ks * p.dstep, hn, wn_) && kin;
          ao[g] = ok ? (uint32_t)((a_img[g] + hn * p.Wi + wn_) * p.ldx + c) * 2u : kOOB;
        }
#pragma unroll
        for (int g = 0; g < GB; ++g)
          bo[g] = (kin && b_ok[g]) ? b_rowoff[g] - (uint32_t)(lslot_log * 16) + (uint32_t)kb * 2u : kOOB;
      } else {
        const int tap = kt / p.ktiles_per_tap;
        if (tap != tap_cur) {
          tap_cur = tap;
          const int kr = tap / p.kw, ks = tap - kr * p.kw;
#pragma unroll
          for (int g = 0; g < GA; ++g) {
            int hn, wn_;
            const bool ok = row_tap_ok(g, kr * p.dstep, ks * p.dstep, hn, wn_);
            a_off[g] = ok ? (uint32_t)((a_img[g] + hn * p.Wi + wn_) * p.ldx) * 2u + (uint32_t)(lslot_log * 16) : kOOB;
          }
        }
        const uint32_t kc_b = (uint32_t)((kt - tap * p.ktiles_per_tap) * KB) * 2u;
#pragma unroll
        for (int g = 0; g < GA; ++g) ao[g] = a_off[g] + kc_b;     // kOOB + kc_b stays out of range
#pragma unroll
        for (int g = 0; g < GB; ++g) bo[g] = b_rowoff[g] + (uint32_t)kt * (uint32_t)(KB * 2);
      }
    } else {
#pragma unroll
      for (int g = 0; g < GA; ++g) ao[g] = kOOB;
#pragma unroll
      for (int g = 0; g < GB; ++g) bo[g] = kOOB;
    }
    unsigned* sb = ldsw + st * kStageDw;
#pragma unroll
    for (int g = 0; g < GA; ++g)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr)(sb + kA + RPG * (wave + NW * g) * RDW), 16, (int)ao[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < GB; ++g)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr)(sb + kB + RPG * b_grp[g] * RDW), 16, (int)bo[g], 0, 0, 0);
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
  auto swz = [&](int row, int slot) -> int { return row * RDW + ((slot ^ fsw(row)) << 2); };
  constexpr int HK = KB / 32;           // MFMAs (16-deep k-steps) per half of a K-step: 2 or 1
  f32x4 fa[2][HK * TM], fb[2][HK * TN];   // [set][gg * T + tile]: the 8 k of MFMA gg of a half-step
  auto read_frags = [&](int set, int st, int half) {
    const float* sb = lds + st * kStageDw;
#pragma unroll
    for (int gg = 0; gg < HK; ++gg) {
      const int slot = 2 * (half * HK + gg) + frag_h;
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[set][gg * TM + i] = *reinterpret_cast<const f32x4*>(&sb[kA + swz(wm * WTM + i * 32 + frag_row, slot)]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[set][gg * TN + j] = *reinterpret_cast<const f32x4*>(&sb[kB + swz(wn * WTN + j * 32 + frag_row, slot)]);
    }
  };
  auto mfmas = [&](int set) {
#pragma unroll
    for (int gg = 0; gg < HK; ++gg)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, fa[set][gg * TM + i]),
                                                             __builtin_bit_cast(f16x8v, fb[set][gg * TN + j]), acc[i][j],
                                                             0, 0, 0);
  };
  // counted waits: NG DMAs per tile and wave
  if constexpr (GENERIC) {
    if (n_steps > 0) {
#pragma unroll
      for (int s0 = 0; s0 < STAGES; ++s0) issue(next_kt(), s0);
      wait_vmcnt<(STAGES - 1) * NG>();   // tile 0 has landed (this wave's share); STAGES - 1 tiles stay in flight
      __builtin_amdgcn_s_barrier();      // ... and everybody's
      read_frags(0, 0, 0);
      int st = 0;
      for (int it = 0; it < n_steps; ++it) {
        const int st1 = st == STAGES - 1 ? 0 : st + 1;
        read_frags(1, st, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0);
        __builtin_amdgcn_sched_barrier(0);
        wait_vmcnt<(STAGES - 2) * NG>();                     // the next tile has landed; STAGES - 2 more stay in flight
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
  } else if (n_steps > 0) {
    // ---- the K loop with its bookkeeping taken out (round 4).  Counters of round 3's loop on the DeepLabV3+ shapes: 4.8 VALU +
    // 6 SALU instructions per MFMA on the deep contractions, 12 + 12 on the K = 512 pointwise layers (a division of the K-step
    // index by the steps per tap, the tap test and the address rebuild on every step) -- the waves were issuing bookkeeping, not
    // waiting for memory.  Now a K-step costs one add per DMA: a tap is OPENED once (per-row validity and base offsets: the only
    // place that multiplies), its channel chunks advance both operands by KB * 2 bytes, and the ring stage is a compile-time
    // constant of the unrolled step, so fragment and DMA addresses are loop-invariant registers + immediates.
    unsigned taps_left = SKIP ? tapmask : 0u;
    int tap_next = 0, chunks_left = 0;
    uint32_t a_cur[GA], b_cur[GB];
    const uint32_t tap_bytes = (uint32_t)p.Cin * 2u;
    auto open_tap = [&]() {
      int tap = -1;
      if (SKIP) {
        if (taps_left != 0u) {
          tap = __builtin_ctz(taps_left);
          taps_left &= taps_left - 1u;
        }
      } else if (tap_next < p.ntaps) {
        tap = tap_next++;
      }
      if (tap < 0) {                  // exhausted: the remaining (dummy) DMAs move nothing
#pragma unroll
        for (int g = 0; g < GA; ++g) a_cur[g] = kOOB;
#pragma unroll
        for (int g = 0; g < GB; ++g) b_cur[g] = kOOB;
        chunks_left = 0x7fffffff;
        return;
      }
      const int kr = (int)hp.kw_div.div((uint32_t)tap), ks = tap - kr * p.kw;
#pragma unroll
      for (int g = 0; g < GA; ++g) {
        int hn, wn_;
        const bool ok = row_tap_ok(g, kr * p.dstep, ks * p.dstep, hn, wn_);
        a_cur[g] = ok ? (uint32_t)((a_img[g] + hn * p.Wi + wn_) * p.ldx) * 2u + (uint32_t)(lslot_log * 16) : kOOB;
      }
#pragma unroll
      for (int g = 0; g < GB; ++g) b_cur[g] = b_rowoff[g] + (uint32_t)tap * tap_bytes;     // (kOOB rows stay out of range)
      chunks_left = p.ktiles_per_tap;
    };
    auto issue_c = [&](auto stc) {
      constexpr int ST = decltype(stc)::value;
      if (chunks_left == 0) open_tap();
      unsigned* sb = ldsw + ST * kStageDw;
      const uint32_t kill = (PSEG_ABLATE(hp) & 4) ? kOOB : 0u;      // (an out-of-range DMA moves no byte but is issued and counted)
#pragma unroll
      for (int g = 0; g < GA; ++g)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr)(sb + kA + RPG * (wave + NW * g) * RDW), 16, (int)(a_cur[g] | kill), 0, 0, 0);
#pragma unroll
      for (int g = 0; g < GB; ++g)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr)(sb + kB + RPG * b_grp[g] * RDW), 16, (int)(b_cur[g] | kill), 0, 0, 0);
#pragma unroll
      for (int g = 0; g < GA; ++g) a_cur[g] += (uint32_t)(KB * 2);
#pragma unroll
      for (int g = 0; g < GB; ++g) b_cur[g] += (uint32_t)(KB * 2);
      --chunks_left;
    };
    // loop-invariant fragment addresses (dwords inside a stage): [half * HK + gg] of the wave's first A / B tile row
    int fa_off[2 * HK], fb_off[2 * HK];
#pragma unroll
    for (int q = 0; q < 2 * HK; ++q) {
      fa_off[q] = kA + swz(wm * WTM + frag_row, 2 * q + frag_h);
      fb_off[q] = kB + swz(wn * WTN + frag_row, 2 * q + frag_h);
    }
    auto read_c = [&](auto stc, auto setc, auto halfc) {
      constexpr int ST = decltype(stc)::value, SET = decltype(setc)::value, HALF = decltype(halfc)::value;
      const float* sb = lds + ST * kStageDw;
#pragma unroll
      for (int gg = 0; gg < HK; ++gg) {
#pragma unroll
        for (int i = 0; i < TM; ++i)      // (tile rows 32 apart keep the swizzle: f(row + 32) = f(row))
          fa[SET][gg * TM + i] = *reinterpret_cast<const f32x4*>(&sb[fa_off[HALF * HK + gg] + i * 32 * RDW]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          fb[SET][gg * TN + j] = *reinterpret_cast<const f32x4*>(&sb[fb_off[HALF * HK + gg] + j * 32 * RDW]);
      }
    };
    typedef std::integral_constant<int, 0> c0;
    typedef std::integral_constant<int, 1> c1;
    issue_c(c0{});
    if constexpr (STAGES >= 2) issue_c(c1{});
    if constexpr (STAGES >= 3) issue_c(std::integral_constant<int, 2>{});
    if constexpr (STAGES >= 4) issue_c(std::integral_constant<int, 3>{});
    wait_vmcnt<(STAGES - 1) * NG>();   // tile 0 has landed (this wave's share); STAGES - 1 tiles stay in flight
    __builtin_amdgcn_s_barrier();      // ... and everybody's
    read_c(c0{}, c0{}, c0{});
#define PSEG_GH_STEP(S)                                                                                          \
  {                                                                                                              \
    typedef std::integral_constant<int, (S)> cs;                                                                 \
    typedef std::integral_constant<int, ((S) + 1) % STAGES> cs1;                                                 \
    if (!(PSEG_ABLATE(hp) & 16)) read_c(cs{}, c1{}, c1{});                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    if (!(PSEG_ABLATE(hp) & 8)) mfmas(0);                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    wait_vmcnt<(STAGES - 2) * NG>();                   /* the next tile has landed; STAGES - 2 more in flight */  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* this wave is done reading stage S */                   \
    __builtin_amdgcn_s_barrier();                                                                                \
    if (!(PSEG_ABLATE(hp) & 16)) read_c(cs1{}, c0{}, c0{});  /* (zeros on the last step: never multiplied) */          \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    issue_c(cs{});                                     /* stage S is free now */                                 \
    if (!(PSEG_ABLATE(hp) & 8)) mfmas(1);                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
  }
    for (int it = 0;;) {
      PSEG_GH_STEP(0)
      if (++it == n_steps) break;
      PSEG_GH_STEP(1)
      if (++it == n_steps) break;
      if constexpr (STAGES >= 3) {
        PSEG_GH_STEP(2)
        if (++it == n_steps) break;
      }
      if constexpr (STAGES >= 4) {
        PSEG_GH_STEP(3)
        if (++it == n_steps) break;
      }
    }
#undef PSEG_GH_STEP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // dummy DMAs must not land in the output patches
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue: bias / accumulate / row map, fused BatchNorm statistics
  const int col_l = lane & 31;
  const int row_h = (lane >> 5) * 4;
  if (PSEG_ABLATE(hp) & 1) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) keep_alive(acc[i][j]);
  } else {
    float* patch = lds + wave * (32 * (WTN + 4));
    const int col0 = n0 + wn * WTN;
    int cv = p.N - col0;
    cv = cv < 0 ? 0 : (cv > WTN ? WTN : cv);
    auto rowmap = [&](int m) {
      if (!p.row_perm) return m;
      int b, ho, wo;
      row_to_pixel(p, m, b, ho, wo);
      return (b * p.Ho + ho) * p.Wo + wo;
    };
    if constexpr (BNS) {
      // (the host only launches this instantiation for an fp16 result without bias / accumulation)
      constexpr int C8 = WTN / 8;
      const int c8 = lane % C8, rr = lane / C8;
      const bool cok = c8 * 8 < cv;
      HBnsCoef bc;
      HBnsSums bs;
      bns_load_coef(p, col0 + c8 * 8, cok, bc);
#pragma unroll
      for (int h = 0; h < 2; ++h) bs.s1[h] = bs.s2[h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row0 = m0 + wm * WTM + i * 32;
        int rv = p.M - row0;
        rv = rv < 0 ? 0 : (rv > 32 ? 32 : rv);
        store_row32<TN, true>(acc[i], patch, p.y, false, p.ldy, row0, col0, rv, cv, nullptr, false, lane, rowmap, p, bc, bs);
      }
#pragma unroll
      for (int o = C8; o < 64; o <<= 1)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            bs.s1[h][e] += __shfl_xor(bs.s1[h][e], o, 64);
            bs.s2[h][e] += __shfl_xor(bs.s2[h][e], o, 64);
          }
      if (rr == 0 && cok) {
        const long long o = (long long)(tile_m * WARPS_M + wm) * p.N + col0 + c8 * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          *reinterpret_cast<f32x4*>(p.bns_db + o + 4 * h) = bs.s1[h];
          *reinterpret_cast<f32x4*>(p.bns_dg + o + 4 * h) = bs.s2[h];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row0 = m0 + wm * WTM + i * 32;
        int rv = p.M - row0;
        rv = rv < 0 ? 0 : (rv > 32 ? 32 : rv);
        store_row32<TN>(acc[i], patch, p.y, hp.y_f32 != 0, p.ldy, row0, col0, rv, cv, p.bias, p.accumulate != 0, lane, rowmap);
      }
    }
  }
  if (p.stat != nullptr && !(PSEG_ABLATE(hp) & 2)) {
    // BatchNorm statistics of the tensor AS STORED: an fp16 result is rounded before it is summed, so that the layer
    // normalises exactly the values its backward pass and the next layer read (what a BatchNorm fed by an fp16 conv sees)
    const bool f32out = hp.y_f32 != 0;
    auto rnd = [&](float v) -> float { return f32out ? v : (float)(half_t)v; };
    const int group = tile_m * WARPS_M + wm;
    const long long gsz = (long long)p.stat_rows * p.N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * WTN + j * 32 + col_l;
      const float k0 = __shfl(rnd(acc[0][j][0]), lane & 31, 64);
      float s1 = 0.f, s2 = 0.f;
      if (m0 + BM <= p.M) {       // (block-uniform: every row of the tile is a pixel -- no per-element test)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float d = rnd(acc[i][j][r]) - k0;
            s1 += d;
            s2 += d * d;
          }
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
            if (row < p.M) {
              const float d = rnd(acc[i][j][r]) - k0;
              s1 += d;
              s2 += d * d;
            }
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

// ------------------------------------------------------------------------------------------------ persistent gather kernel
// Round 4.  What the ablation of gather_h_kernel on the DeepLabV3+ shapes showed (profiles/scripts/exp_ablate.sh, profiles/EXPERIMENTS.md):
// with stores, statistics, operand DMAs, fragment reads and MFMAs ALL switched off, half of the time of a launch is still
// there -- a block's life is a latency chain (kernel arguments, row -> pixel arithmetic, the first tile's round trip, eight
// waves meeting at a barrier, the LDS round trip of the epilogue, the stores' drain) and a CU holds only two or three blocks
// to hide one chain behind another; the short contractions (K <= 512: 2-16 K-steps) are nothing but chain.
// Here a block is PERSISTENT: it walks tiles blockIdx, blockIdx + grid, ... and the operand ring never drains -- the DMAs of
// the next tile's first K-steps are issued before the current tile's last MFMA, its row -> pixel arithmetic runs inside
// that issue slot, and the epilogue works out of a wave-private patch of its own while they land:
//   * one continuous stream of (tile, tap, channel chunk) K-steps; a tap is opened once, chunks advance by KB * 2 bytes;
//   * the ring stage is a compile-time constant of the unrolled step (a switch re-enters the unrolled sequence at the
//     phase the previous tile ended on);
//   * epilogue for fp16 results: the 32x32 accumulator tile goes to LDS TRANSPOSED, four 8-byte writes per lane
//     (ds_write_b64: a lane's four consecutive rows of one column), and comes back through ds_read_b64_tr_b16 as 8
//     consecutive channels per lane for 16-byte stores -- 8 LDS instructions per tile where the fp32 patch of
//     gather_h_kernel takes 20, and 2.3 KB of patch per wave instead of 4.6; no block barrier anywhere in it.
// Covers what most launches of a training step are: every tap live (no skipping), channels a multiple of the K-step, fp16
// result without bias / accumulation, rows in natural order.  Everything else stays on gather_h_kernel.
typedef short s16x4g __attribute__((ext_vector_type(4)));
typedef short s16x8g __attribute__((ext_vector_type(8)));

template <int BM, int BN, int WARPS_M, int WARPS_N, int KB, int STAGES, bool BNS = false>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N) void gather_hp_kernel(const HGatherParams hp) {
  const GatherConvParams& p = hp.g;
  set_wave_prio(p.prio);
  static_assert(KB == 64 || KB == 32, "K-step");
  constexpr int NW = WARPS_M * WARPS_N;
  static_assert(NW == 8 || NW == 4, "8 or 4 waves");
  constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N, TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1 && WTM % 32 == 0 && WTN % 32 == 0, "wave tile");
  constexpr int RDW = KB / 2, NSLOT = KB / 8, RPG = 64 / NSLOT;
  constexpr int kStageDw = (BM + BN) * RDW;
  static_assert(STAGES == 2 || STAGES == 3, "ring depth");
  constexpr int kPitch = 72;                         // bytes per patch row = one output column: 32 rows x 2 B + 8 (bank spread)
  constexpr int kPatchB = 32 * kPitch;
  constexpr int kRingB = STAGES * kStageDw * 4;
  constexpr int kLdsB = kRingB + NW * kPatchB;
  static_assert(kLdsB <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[kLdsB];
  float* lds = reinterpret_cast<float*>(lds_raw);
  unsigned* ldsw = reinterpret_cast<unsigned*>(lds_raw);
  constexpr int kA = 0, kB = BM * RDW;
  constexpr int GA = BM / RPG / NW, GB = (BN / RPG + NW - 1) / NW, NG = GA + GB;
  static_assert((BM / RPG) % NW == 0 && GA >= 1, "whole A row groups per wave");
  constexpr bool kBPartial = (BN / RPG) % NW != 0;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;
  const int gridN = (p.N + BN - 1) / BN;
  const int ntiles = hp.ntiles;
  const int nblocks = (int)gridDim.x;

  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.w, p.w_bytes);
  auto fsw = [](int row) -> int { return KB == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); };
  const int lrow = lane / NSLOT, lslot = lane % NSLOT;
  const int lslot_log = lslot ^ fsw(RPG * wave + lrow);
  int b_grp[GB];
#pragma unroll
  for (int g = 0; g < GB; ++g) b_grp[g] = kBPartial ? (wave + NW * g) % (BN / RPG) : (wave + NW * g);

  // ---- issue side: the stream of K-steps over this block's tiles
  int i_vt = (int)blockIdx.x;          // next virtual tile to open
  int tap_next = p.ntaps;              // (== ntaps: the first open_tap opens a tile)
  int chunks_left = 0;
  int a_bh[GA], a_bw[GA], a_img[GA];
  bool a_ok[GA];
  uint32_t b_rowoff[GB];
  uint32_t a_cur[GA], b_cur[GB];
  const uint32_t tap_bytes = (uint32_t)p.Cin * 2u;
  auto open_tile = [&]() -> bool {
    if (i_vt >= ntiles) return false;
    const int t = remap_tile(p.xcd_remap, i_vt, ntiles);
    i_vt += nblocks;
    const int tile_n = t % gridN, tile_m = t / gridN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
#pragma unroll
    for (int g = 0; g < GA; ++g) {
      const int m = m0 + RPG * (wave + NW * g) + lrow;
      const bool ok = m < p.M;
      // rows in natural order only (the host sends the permuted orders -- patches, parity classes, liveness classes -- to
      // gather_h_kernel): pixel m of the [B, Ho, Wo] row space, or of the 1 x M image of a pointwise conv (HoWo = Wo = M)
      const uint32_t mm = ok ? (uint32_t)m : 0u;
      const uint32_t b = hp.howo_div.div(mm);
      const uint32_t rem = mm - b * (uint32_t)p.HoWo;
      const uint32_t ho = hp.wo_div.div(rem);
      const uint32_t wo = rem - ho * (uint32_t)p.Wo;
      a_ok[g] = ok;
      a_bh[g] = (int)ho * p.s_out + p.off0;
      a_bw[g] = (int)wo * p.s_out + p.off0;
      a_img[g] = (int)b * p.Hi * p.Wi;
    }
#pragma unroll
    for (int g = 0; g < GB; ++g) {
      const int row = n0 + RPG * b_grp[g] + lrow;
      b_rowoff[g] = row < p.N ? (uint32_t)row * (uint32_t)p.K * 2u + (uint32_t)(lslot_log * 16) : kOOB;
    }
    tap_next = 0;
    return true;
  };
  auto open_tap = [&]() {
    if (tap_next >= p.ntaps && !open_tile()) {      // exhausted: the remaining (dummy) DMAs fetch nothing
#pragma unroll
      for (int g = 0; g < GA; ++g) a_cur[g] = kOOB;
#pragma unroll
      for (int g = 0; g < GB; ++g) b_cur[g] = kOOB;
      chunks_left = 0x7fffffff;
      return;
    }
    const int tap = tap_next++;
    const int kr = (int)hp.kw_div.div((uint32_t)tap), ks = tap - kr * p.kw;
    const int dh = kr * p.dstep, dw = ks * p.dstep;
#pragma unroll
    for (int g = 0; g < GA; ++g) {
      int hn = a_bh[g] + dh, wn_ = a_bw[g] + dw;
      bool ok = a_ok[g];
      if (p.s_in != 1) {
        ok = ok && (hn % p.s_in == 0) && (wn_ % p.s_in == 0);
        hn /= p.s_in;
        wn_ /= p.s_in;
      }
      ok = ok && ((unsigned)hn < (unsigned)p.Hi) && ((unsigned)wn_ < (unsigned)p.Wi);
      a_cur[g] = ok ? (uint32_t)((a_img[g] + hn * p.Wi + wn_) * p.ldx) * 2u + (uint32_t)(lslot_log * 16) : kOOB;
    }
#pragma unroll
    for (int g = 0; g < GB; ++g) b_cur[g] = b_rowoff[g] + (uint32_t)tap * tap_bytes;     // (kOOB rows stay out of range)
    chunks_left = p.ktiles_per_tap;
  };
  typedef __attribute__((address_space(3))) void* lds_ptr;
  auto issue_c = [&](auto stc) {
    constexpr int ST = decltype(stc)::value;
    if (chunks_left == 0) open_tap();
    unsigned* sb = ldsw + ST * kStageDw;
#pragma unroll
    for (int g = 0; g < GA; ++g)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr)(sb + kA + RPG * (wave + NW * g) * RDW), 16, (int)a_cur[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < GB; ++g)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr)(sb + kB + RPG * b_grp[g] * RDW), 16, (int)b_cur[g], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < GA; ++g) a_cur[g] += (uint32_t)(KB * 2);
#pragma unroll
    for (int g = 0; g < GB; ++g) b_cur[g] += (uint32_t)(KB * 2);
    --chunks_left;
  };

  // ---- compute side
  f32x16 acc[TM][TN];
  const int frag_row = lane & 31, frag_h = lane >> 5;
  auto swz = [&](int row, int slot) -> int { return row * RDW + ((slot ^ fsw(row)) << 2); };
  constexpr int HK = KB / 32;
  f32x4 fa[2][HK * TM], fb[2][HK * TN];
  int fa_off[2 * HK], fb_off[2 * HK];
#pragma unroll
  for (int q = 0; q < 2 * HK; ++q) {
    fa_off[q] = kA + swz(wm * WTM + frag_row, 2 * q + frag_h);
    fb_off[q] = kB + swz(wn * WTN + frag_row, 2 * q + frag_h);
  }
  auto read_c = [&](auto stc, auto setc, auto halfc) {
    constexpr int ST = decltype(stc)::value, SET = decltype(setc)::value, HALF = decltype(halfc)::value;
    const float* sb = lds + ST * kStageDw;
#pragma unroll
    for (int gg = 0; gg < HK; ++gg) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[SET][gg * TM + i] = *reinterpret_cast<const f32x4*>(&sb[fa_off[HALF * HK + gg] + i * 32 * RDW]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[SET][gg * TN + j] = *reinterpret_cast<const f32x4*>(&sb[fb_off[HALF * HK + gg] + j * 32 * RDW]);
    }
  };
  auto mfmas = [&](int set) {
#pragma unroll
    for (int gg = 0; gg < HK; ++gg)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, fa[set][gg * TM + i]),
                                                             __builtin_bit_cast(f16x8v, fb[set][gg * TN + j]), acc[i][j],
                                                             0, 0, 0);
  };

  // ---- epilogue of one tile: transposed fp16 patch, fused BatchNorm statistics of the values as stored
  unsigned char* patch = lds_raw + kRingB + wave * kPatchB;
  const int col_l = lane & 31, hh = lane >> 5;
  const int g16 = lane >> 4, i16 = lane & 15, tqq = i16 >> 2, tpp = i16 & 3;
  const uint32_t tr_addr = lds_addr_of(patch + (8 * g16 + tqq) * kPitch + 8 * tpp);   // block (patch rows 8 g16 + qq, columns 4 pp ..)
  auto epilogue = [&](int cm0, int cn0, int ctile_m) {
    half_t* out = reinterpret_cast<half_t*>(p.y);
    const bool full = cm0 + BM <= p.M;
    const long long gsz = (long long)p.stat_rows * p.N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col0 = cn0 + wn * WTN + j * 32;
      float k0 = 0.f, s1 = 0.f, s2 = 0.f;
      // BNS (see store_row32): this lane stores channels col0 + 8 g16 .. + 7 of rows row0 + i16 and row0 + 16 + i16
      const bool bns_cok = col0 + 8 * g16 < p.N;
      HBnsCoef bc;
      HBnsSums bs;
      f16x8v ypre[TM][2];
      if constexpr (BNS) {
        bns_load_coef(p, col0 + 8 * g16, bns_cok, bc);
#pragma unroll
        for (int h = 0; h < 2; ++h) bs.s1[h] = bs.s2[h] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the rows of y this lane will need, requested before the accumulators make their way through the patch
        const half_t* yb = reinterpret_cast<const half_t*>(p.bns_y) + col0 + 8 * g16;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int hb = 0; hb < 2; ++hb) {
            const int row = cm0 + wm * WTM + i * 32 + 16 * hb + i16;
            ypre[i][hb] = f16x8v{0, 0, 0, 0, 0, 0, 0, 0};
            if (row < p.M && bns_cok) ypre[i][hb] = *reinterpret_cast<const f16x8v*>(yb + (long long)row * p.bns_ldy);
          }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row0 = cm0 + wm * WTM + i * 32;
        // accumulator registers 4q .. 4q + 3 are rows 8q + 4h + 0 .. 3 of column (lane & 31)
        f16x4v hq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) hq[q][e] = (half_t)acc[i][j][4 * q + e];
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f16x4v*>(patch + col_l * kPitch + (8 * q + 4 * hh) * 2) = hq[q];
        if (p.stat != nullptr) {
          if (i == 0) k0 = __shfl((float)hq[0][0], lane & 31, 64);
          if (full) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float d = (float)hq[q][e] - k0;
                s1 += d;
                s2 += d * d;
              }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (row0 + 8 * q + 4 * hh + e < p.M) {
                  const float d = (float)hq[q][e] - k0;
                  s1 += d;
                  s2 += d * d;
                }
          }
        }
        // (the patch is private to this wave and LDS executes a wave's instructions in order: the writes are visible to the
        // reads below once they have been counted out -- no fence, which would also wait for the operand DMAs of the NEXT tile
        // that are in flight by design)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // ds_read_b64_tr_b16 per 16-lane group: the block of 4 patch rows (= output columns 8 g16 + [0, 4) or + [4, 8)) x 16
        // patch columns (= output rows r0 .. r0 + 15); lane 4 qq + pp supplies (patch row qq, columns 4 pp ..), lane i16 receives
        // output row r0 + i16, the four columns.  Two reads = 8 consecutive channels = one 16-byte store; the four lane groups
        // cover the four channel octets of a row: 64 contiguous bytes per output row and instruction.
        {
          const s16x4t lo0 = tr_read_asm<0>(tr_addr), hi0 = tr_read_asm<4 * kPitch>(tr_addr);
          const s16x4t lo1 = tr_read_asm<32>(tr_addr), hi1 = tr_read_asm<4 * kPitch + 32>(tr_addr);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          const int col = col0 + 8 * g16;
          const int row_a = row0 + i16, row_b = row0 + 16 + i16;
          const s16x8g va = s16x8g{lo0[0], lo0[1], lo0[2], lo0[3], hi0[0], hi0[1], hi0[2], hi0[3]};
          const s16x8g vb = s16x8g{lo1[0], lo1[1], lo1[2], lo1[3], hi1[0], hi1[1], hi1[2], hi1[3]};
          if (row_a < p.M && col < p.N) *reinterpret_cast<s16x8g*>(out + (long long)row_a * p.ldy + col) = va;
          if (row_b < p.M && col < p.N) *reinterpret_cast<s16x8g*>(out + (long long)row_b * p.ldy + col) = vb;
          if constexpr (BNS) {
            if (row_a < p.M && bns_cok) bns_add8(__builtin_bit_cast(f16x8v, va), ypre[i][0], p.bns_act, bc, bs);
            if (row_b < p.M && bns_cok) bns_add8(__builtin_bit_cast(f16x8v, vb), ypre[i][1], p.bns_act, bc, bs);
          }
        }
        __builtin_amdgcn_wave_barrier();      // (the patch is rewritten by the next tile: its reads above have been waited for)
      }
      if constexpr (BNS) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1)
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              bs.s1[h][e] += __shfl_xor(bs.s1[h][e], o, 64);
              bs.s2[h][e] += __shfl_xor(bs.s2[h][e], o, 64);
            }
        if (i16 == 0 && bns_cok) {
          const long long o = (long long)(ctile_m * WARPS_M + wm) * p.N + col0 + 8 * g16;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<f32x4*>(p.bns_db + o + 4 * h) = bs.s1[h];
            *reinterpret_cast<f32x4*>(p.bns_dg + o + 4 * h) = bs.s2[h];
          }
        }
      }
      if (p.stat != nullptr) {
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        const int col = col0 + col_l;
        if (lane < 32 && col < p.N) {
          const long long o = (long long)(ctile_m * WARPS_M + wm) * p.N + col;
          p.stat[o] = k0;
          p.stat[gsz + o] = s1;
          p.stat[2 * gsz + o] = s2;
        }
      }
    }
  };

  const int n_steps = p.ntaps * p.ktiles_per_tap;
  typedef std::integral_constant<int, 0> c0;
  typedef std::integral_constant<int, 1> c1;
  typedef std::integral_constant<int, 2> c2;
  issue_c(c0{});
  issue_c(c1{});
  if constexpr (STAGES >= 3) issue_c(c2{});
  wait_vmcnt<(STAGES - 1) * NG>();   // step 0 has landed (this wave's share); STAGES - 1 steps stay in flight
  __builtin_amdgcn_s_barrier();      // ... and everybody's
  read_c(c0{}, c0{}, c0{});
#define PSEG_GHP_STEP(S)                                                                                         \
  {                                                                                                              \
    typedef std::integral_constant<int, (S)> cs;                                                                 \
    typedef std::integral_constant<int, ((S) + 1) % STAGES> cs1;                                                 \
    read_c(cs{}, c1{}, c1{});                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    mfmas(0);                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    wait_vmcnt<(STAGES - 2) * NG>();                   /* the next step has landed; STAGES - 2 more in flight */  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* this wave is done reading stage S */                   \
    __builtin_amdgcn_s_barrier();                                                                                \
    read_c(cs1{}, c0{}, c0{});                         /* (the next TILE's first step at a tile boundary) */     \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    issue_c(cs{});                                     /* stage S is free now */                                 \
    mfmas(1);                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
  }
#define PSEG_GHP_CASE(S)                     \
  case (S):                                  \
    PSEG_GHP_STEP(S)                         \
    if (--left == 0) {                       \
      stage = ((S) + 1) % STAGES;            \
      running = false;                       \
      break;                                 \
    }
  int stage = 0;       // ring phase the next K-step computes from: carried over tile boundaries
  for (int c_vt = (int)blockIdx.x; c_vt < ntiles; c_vt += nblocks) {
    const int t = remap_tile(p.xcd_remap, c_vt, ntiles);
    const int ctile_n = t % gridN, ctile_m = t / gridN;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int left = n_steps;
    bool running = true;
    while (running) {
      if constexpr (STAGES == 2) {
        switch (stage) {
          PSEG_GHP_CASE(0)
          PSEG_GHP_CASE(1)
        }
      } else {
        switch (stage) {
          PSEG_GHP_CASE(0)
          PSEG_GHP_CASE(1)
          PSEG_GHP_CASE(2)
        }
      }
      if (running) stage = 0;
    }
    epilogue(ctile_m * BM, ctile_n * BN, ctile_m);
  }
#undef PSEG_GHP_CASE
#undef PSEG_GHP_STEP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no DMA may land after the block has given its LDS back
}

#if PSEG_LAB
// ------------------------------------------------------------------------------------------------
// HALO-STAGED 3x3 (round 5; VERDICT r4 item 3 i) -- LAB BUILD ONLY: measured slower than gather_h_kernel (layer-4 3x3 d = 2 forward
// 101-108 us against 83, layer 3 30 against 27; profiles/EXPERIMENTS.md 5.2 has the numbers and the reason: the halo buffers take the
// LDS of the second resident block, and eight waves per CU do not cover their own barrier / LDS latency).  gather_h_kernel fetches the A operand of a 3x3 conv once per TAP: nine
// [128 pixels][64 channels] tiles per channel chunk, eight of them the same pixels shifted by one or two positions -- and
// the operand stream into LDS (the L2 -> LDS DMA path, ~27 B/clk/CU) is exactly what bounds these kernels (round 4: MFMA busy
// 0.41-0.43 on layer 4; a 128x128 tile moves 32 KB per 512 matrix cycles).  Here an M tile is an 8 x 16 PATCH of output pixels
// and the A operand of a channel chunk is DMA'd ONCE, as the (8 + 2d) x (16 + 2d) halo patch of input pixels (d = dilation 1 or 2:
// 180 / 240 pixels x 128 B); the nine taps read their fragments from that image at shifted pixel positions.  Per chunk and tile:
// 23 / 31 KB of A instead of 144 KB, 9 x 16 KB of B as before -- 19.5 KB per K-step instead of 32.
//   LDS: two halo buffers (chunk c is multiplied while chunk c + 1 lands) + a three-stage ring of B tiles = 108 KB, one 8-wave
//   block per CU; K runs CHUNK-major (chunk, then tap) -- another summation order than gather_h_kernel, same products.
//   The halo image keeps the ring's layout per pixel (128-byte rows, 16-byte k-slot XOR (pixel >> 1) & 7 -- applied on the source
//   side of the DMA, keyed on the HALO pixel index), so a fragment read is the same ds_read_b128, at a per-tap row.
// Covers: 3x3, unit stride, dilation 1 or 2, channels % 64 == 0, maps that tile into 8 x 16 patches, fp16 result without bias /
// accumulation, forward (+ fused BatchNorm statistics) and data gradient (the same gather with the taps reversed).
constexpr int kHaloPH = 8, kHaloPW = 16, kHaloMaxPix = 256;      // (8 + 2d) x (16 + 2d) <= 240 pixels, buffers of 32 whole DMA pieces

template <int STAGES>
__global__ __launch_bounds__(512) void gather_hh_kernel(const HGatherParams hp) {
  const GatherConvParams& p = hp.g;
  set_wave_prio(p.prio);
  constexpr int BM = 128, BN = 128, WARPS_M = 2, WARPS_N = 4, NW = 8, KB = 64;
  static_assert(STAGES >= 3 && STAGES <= 6, "B ring depth");
  constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N, TM = WTM / 32, TN = WTN / 32;     // 64 x 32 per wave: TM = 2, TN = 1
  constexpr int RDW = 32;                                  // dwords per pixel / filter row of a chunk
  constexpr int kHaloDw = kHaloMaxPix * RDW;               // one halo buffer
  constexpr int kBStageDw = BN * RDW;
  constexpr int kBBase = 2 * kHaloDw;
  constexpr int kLds = 2 * kHaloDw + STAGES * kBStageDw;   // 112 KB with three B stages, 160 KB with six (the epilogue patches reuse it)
  static_assert(kLds * 4 <= 160 * 1024, "LDS");
  static_assert(NW * 32 * (WTN + 4) <= kLds, "epilogue patch");
  __shared__ __attribute__((aligned(16))) float lds[kLds];
  unsigned* ldsw = reinterpret_cast<unsigned*>(lds);
  constexpr int GB = BN / 8 / NW;                          // B DMAs per wave and K-step (2)
  constexpr int GAH = kHaloMaxPix / 8 / NW;                // halo DMAs per wave and chunk: always 4 (pieces past the halo fetch
                                                           // nothing and land in the buffer's unused tail) -- a fixed count for vmcnt

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
  // the tile's patch: image b, top-left output pixel (y0, x0)   (row_perm == 2 order: patches_per_row, patch_hw set by the host)
  const int ppi = p.HoWo / BM;                             // patches per image
  const int b_img = tile_m / ppi, pidx = tile_m - b_img * ppi;
  const int y0 = (pidx / p.patches_per_row) * kHaloPH, x0 = (pidx % p.patches_per_row) * kHaloPW;
  const int d = p.dstep < 0 ? -p.dstep : p.dstep;
  const int HW = kHaloPW + 2 * d, HPIX = (kHaloPH + 2 * d) * HW;

  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.w, p.w_bytes);
  const int lrow = lane >> 3, lslot = lane & 7;

  // ---- halo DMA pieces of this lane: halo pixel 8 * piece + lrow, physical slot lslot <- logical slot lslot ^ f(pixel)
  uint32_t h_off[GAH];
#pragma unroll
  for (int g = 0; g < GAH; ++g) {
    const int piece = wave + NW * g;
    const int hpix = 8 * piece + lrow;
    const int iy = y0 - d + hpix / HW, ix = x0 - d + hpix % HW;
    const bool ok = hpix < HPIX && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
    // swizzle keyed on the halo COLUMN (not the linear pixel index): a fragment read's 16-lane service group holds columns
    // x .. x + 3, x + 12 .. x + 15 of one patch row and x + 4 .. x + 11 of the next -- sixteen different columns mod 16 at every
    // tap, so the group covers all 64 banks once (keyed on the linear index, rows 18 / 20 pixels apart collided 2- / 4-way)
    h_off[g] = ok ? (uint32_t)(((b_img * p.Hi + iy) * p.Wi + ix) * p.ldx) * 2u + (uint32_t)((lslot ^ (((hpix % HW) >> 1) & 7)) * 16) : kOOB;
  }
  // ---- B rows of this lane
  uint32_t b_row[GB];
#pragma unroll
  for (int g = 0; g < GB; ++g) {
    const int row = 8 * (wave + NW * g) + lrow;
    b_row[g] = (n0 + row) < p.N ? (uint32_t)(n0 + row) * (uint32_t)p.K * 2u + (uint32_t)((lslot ^ ((row >> 1) & 7)) * 16) : kOOB;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int nchunks = p.Cin / KB;
  const int n_steps = nchunks * 9;
  const uint32_t tap_bytes = (uint32_t)p.Cin * 2u;

  int h_chunk = 0;                       // next chunk whose halo is to be issued
  auto issue_halo = [&]() {
    if (h_chunk >= nchunks) return;
    unsigned* hb = ldsw + (h_chunk & 1) * kHaloDw;
#pragma unroll
    for (int g = 0; g < GAH; ++g) {
      const int piece = wave + NW * g;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr)(hb + piece * 8 * RDW), 16, (int)h_off[g], 0, 0, 0);
      h_off[g] += (uint32_t)(KB * 2);    // (kOOB + anything a launch adds stays out of range)
    }
    ++h_chunk;
  };
  int h_age = 3;                         // K-steps since the last halo issue inside the loop (>= 3: its DMAs no longer count)
  int i_tap = 0, i_chunk = 0;            // the K-step the next B DMA belongs to
  uint32_t b_cur[GB];
#pragma unroll
  for (int g = 0; g < GB; ++g) b_cur[g] = b_row[g];
  auto issue_b = [&](auto stc) {
    constexpr int ST = decltype(stc)::value;
    unsigned* sb = ldsw + kBBase + ST * kBStageDw;
    const bool live = i_chunk < nchunks;
#pragma unroll
    for (int g = 0; g < GB; ++g)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr)(sb + 8 * (wave + NW * g) * RDW), 16, (int)(live ? b_cur[g] : kOOB), 0, 0, 0);
    if (!live) return;
    if (++i_tap == 9) {                  // next chunk: back to tap 0, 64 channels on
      i_tap = 0;
      ++i_chunk;
#pragma unroll
      for (int g = 0; g < GB; ++g) b_cur[g] = b_cur[g] - 8u * tap_bytes + (uint32_t)(KB * 2);
    } else {
#pragma unroll
      for (int g = 0; g < GB; ++g) b_cur[g] += tap_bytes;
    }
  };

  // ---- compute side
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int frag_row = lane & 31, frag_h = lane >> 5;
  // halo pixel of this lane's fragment row of tile row i at the CENTRE tap
  int hp0[TM], hx0[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int R = wm * WTM + i * 32 + frag_row;
    hx0[i] = (R & 15) + d;
    hp0[i] = ((R >> 4) + d) * HW + hx0[i];
  }
  int fb_off[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = wn * WTN + frag_row;
    fb_off[q] = row * RDW + (((2 * q + frag_h) ^ ((row >> 1) & 7)) << 2);
  }
  f32x4 fa[2][2 * TM], fb[2][2 * TN];
  int c_tap = 0, c_chunk = 0;            // the K-step being multiplied
  int fa_base[TM], fa_x[TM];             // halo pixel row (dwords) and its swizzle term, of the current tap
  auto open_tap = [&](int tap) {
    const int kr = tap / 3, ks = tap - 3 * kr;
    const int ox = p.off0 + ks * p.dstep;
    const int sh = (p.off0 + kr * p.dstep) * HW + ox;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      fa_base[i] = (hp0[i] + sh) * RDW;
      fa_x[i] = ((hx0[i] + ox) >> 1) & 7;
    }
  };
  auto read_a = [&](int set, int half, const float* hb) {
#pragma unroll
    for (int gg = 0; gg < 2; ++gg) {
      const int slot = 2 * (half * 2 + gg) + frag_h;
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[set][gg * TM + i] = *reinterpret_cast<const f32x4*>(&hb[fa_base[i] + ((slot ^ fa_x[i]) << 2)]);
    }
  };
  auto read_b = [&](auto stc, int set, int half) {
    constexpr int ST = decltype(stc)::value;
    const float* sb = lds + kBBase + ST * kBStageDw;
#pragma unroll
    for (int gg = 0; gg < 2; ++gg)
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[set][gg * TN + j] = *reinterpret_cast<const f32x4*>(&sb[fb_off[half * 2 + gg] + j * 32 * RDW]);
  };
  auto mfmas = [&](int set) {
#pragma unroll
    for (int gg = 0; gg < 2; ++gg)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, fa[set][gg * TM + i]),
                                                             __builtin_bit_cast(f16x8v, fb[set][gg * TN + j]), acc[i][j], 0, 0, 0);
  };

  typedef std::integral_constant<int, 0> c0;
  typedef std::integral_constant<int, 1> c1;
  typedef std::integral_constant<int, 2> c2;
  issue_halo();                          // chunk 0 (older than every B DMA below: a counted wait for a B tile covers it)
  issue_b(c0{});
  issue_halo();                          // chunk 1 lands while chunk 0 is multiplied
  issue_b(c1{});
  issue_b(c2{});
  if constexpr (STAGES >= 4) issue_b(std::integral_constant<int, 3>{});
  if constexpr (STAGES >= 5) issue_b(std::integral_constant<int, 4>{});
  if constexpr (STAGES >= 6) issue_b(std::integral_constant<int, 5>{});
  wait_vmcnt<(STAGES - 1) * GB>();       // halos + B tile 0 have landed (this wave's share)
  // In the loop a halo is issued BEHIND the B tile of its slot: [B(s + 3)] [halo x 4].  vmcnt counts in issue order, so the two
  // following waits -- for B(s + 2) and B(s + 3), both older than the halo -- may leave the halo's four DMAs in flight on top of
  // the youngest B tile; the third wait (for B(s + 4), younger) retires it: three K-steps to land instead of one.
  __builtin_amdgcn_s_barrier();
  open_tap(0);
  read_a(0, 0, lds);
  read_b(c0{}, 0, 0);
#define PSEG_HH_STEP(S)                                                                                                        \
  {                                                                                                                            \
    typedef std::integral_constant<int, (S)> cs;                                                                               \
    typedef std::integral_constant<int, ((S) + 1) % STAGES> cs1;                                                               \
    const float* hb_cur = lds + (c_chunk & 1) * kHaloDw;                                                                       \
    read_a(1, 1, hb_cur);                                                                                                      \
    read_b(cs{}, 1, 1);                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                         \
    mfmas(0);                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                         \
    if (h_age < 2) wait_vmcnt<(STAGES - 2) * GB + GAH>(); /* ... a halo issued one / two slots ago may stay in flight */       \
    else wait_vmcnt<(STAGES - 2) * GB>();              /* the next B tile (and any older halo) has landed */                   \
    ++h_age;                                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* this wave is done reading B stage S and this K-step's A rows */      \
    __builtin_amdgcn_s_barrier();                                                                                              \
    const bool new_chunk = c_tap == 8;                                                                                         \
    if (new_chunk) {                                                                                                           \
      c_tap = 0;                                                                                                               \
      ++c_chunk;                                                                                                               \
    } else {                                                                                                                   \
      ++c_tap;                                                                                                                 \
    }                                                                                                                          \
    open_tap(c_tap);                                                                                                           \
    read_a(0, 0, lds + (c_chunk & 1) * kHaloDw);       /* (stale bytes after the last step: never multiplied) */               \
    read_b(cs1{}, 0, 0);                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                         \
    issue_b(cs{});                                     /* B stage S is free now */                                             \
    if (new_chunk && h_chunk < nchunks) {              /* the buffer of the chunk just finished is free: chunk + 2 -> it */    \
      issue_halo();                                                                                                            \
      h_age = 0;                                                                                                               \
    }                                                                                                                          \
    mfmas(1);                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                         \
  }
  for (int it = 0;;) {
    PSEG_HH_STEP(0)
    if (++it == n_steps) break;
    PSEG_HH_STEP(1)
    if (++it == n_steps) break;
    PSEG_HH_STEP(2)
    if (++it == n_steps) break;
    if constexpr (STAGES >= 4) {
      PSEG_HH_STEP(3)
      if (++it == n_steps) break;
    }
    if constexpr (STAGES >= 5) {
      PSEG_HH_STEP(4)
      if (++it == n_steps) break;
    }
    if constexpr (STAGES >= 6) {
      PSEG_HH_STEP(5)
      if (++it == n_steps) break;
    }
  }
#undef PSEG_HH_STEP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // dummy DMAs must not land in the output patches
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- epilogue: as gather_h_kernel (fp16 result, patch-ordered rows, fused BatchNorm statistics of the values as stored)
  const int col_l = lane & 31;
  {
    float* patch = lds + wave * (32 * (WTN + 4));
    const int col0 = n0 + wn * WTN;
    int cv = p.N - col0;
    cv = cv < 0 ? 0 : (cv > WTN ? WTN : cv);
    auto rowmap = [&](int m) {
      int b, ho, wo;
      row_to_pixel(p, m, b, ho, wo);
      return (b * p.Ho + ho) * p.Wo + wo;
    };
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row0 = m0 + wm * WTM + i * 32;
      store_row32<TN>(acc[i], patch, p.y, false, p.ldy, row0, col0, 32, cv, nullptr, false, lane, rowmap);
    }
  }
  if (p.stat != nullptr) {
    auto rnd = [&](float v) -> float { return (float)(half_t)v; };
    const int group = tile_m * WARPS_M + wm;
    const long long gsz = (long long)p.stat_rows * p.N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * WTN + j * 32 + col_l;
      const float k0 = __shfl(rnd(acc[0][j][0]), lane & 31, 64);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float dd = rnd(acc[i][j][r]) - k0;
          s1 += dd;
          s2 += dd * dd;
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
// HALO-STAGED 3x3 with the filter as a RING OF TAPS, fp16 (round 5 -- the form of gather_f32_halo_kernel that paid in fp32, after the
// lab kernel above lost: what it lost on was residency -- two halo buffers + a ring of whole-chunk tap tiles, 112-160 KB, ONE block
// per CU).  An M tile is an 8 x 16 patch of output pixels; a 64-channel chunk of the A operand is DMA'd once, as the
// (8 + 2d) x (16 + 2d) halo patch (23 / 30 KB for dilation 1 / 2), SINGLE-buffered; the nine taps read their fragments from it; the
// chunk's filter slice goes through a three-stage ring of single taps ([BN columns][64 channels]: 16 / 8 KB).  78 KB (128 columns,
// eight waves: two blocks per CU -- what gather_h_kernel has) / 54 KB (64 columns, four waves: two to three).  Operand bytes per
// chunk and 128x128 tile: 30 + 144 KB instead of 288.  One barrier per tap, as gather_h_kernel has one per K-step.
// Layouts (halo rows keyed on the halo column, filter rows as the ring kernels') and fragment reads as gather_f32_halo_kernel --
// a 16-byte k-slot holds eight halves, v_mfma_f32_32x32x16_f16 takes one per operand.
// Covers: 3x3, unit stride, dilation 1 or 2 with matching padding (forward / data gradient), channels % 64 == 0, maps of whole 8 x 16
// patches, the 128x128 / 128x64 plan tiles with every tap live; bias / accumulate / fp32 or fp16 result / fused statistics as
// gather_h_kernel.  LAB BUILD ONLY (PSEG_HCONV_HALO2=1 there): measured SLOWER than gather_h_kernel -- layer-4 3x3 d = 2 91 / 90 us against
// 85 / 83, layer 3 35 / 32 against 29 / 26, layer 2 33 / 32 against 29 / 28, layer 1 48 / 42 against 41 / 42 (profiles/EXPERIMENTS.md
// 5.14): a 64-channel chunk is nine taps of eight fp16 MFMAs per wave -- 1.1 us -- and the single halo buffer drains the DMA queue
// at every chunk boundary; the fp32 kernel's chunk is four times as long.
constexpr int kH2PH = 8, kH2PW = 16, kH2MaxRows = (kH2PH + 4) * (kH2PW + 4);        // 240 halo pixels at dilation 2 (180 at 1)
template <int WARPS_N>
__global__ __launch_bounds__(128 * WARPS_N, WARPS_N == 4 ? 4 : 3) void gather_hr_kernel(const HGatherParams hp) {
  const GatherConvParams& p = hp.g;
  set_wave_prio(p.prio);
  constexpr int WARPS_M = 2, NW = WARPS_M * WARPS_N, BN = 32 * WARPS_N, TM = 2;
  constexpr int RDW = 32;                                   // dwords per LDS row: 64 halves
  constexpr int kA = 0, kB = kH2MaxRows * RDW, kRing = kB + 3 * BN * RDW;
  constexpr int kEpi = NW * 32 * 36;
  constexpr int kLds = kRing > kEpi ? kRing : kEpi;
  __shared__ __attribute__((aligned(16))) float lds[kLds];
  unsigned* ldsw = reinterpret_cast<unsigned*>(lds);
  constexpr int GA = kH2MaxRows / 8 / NW + ((kH2MaxRows / 8) % NW != 0);      // halo row groups per wave: 4 (eight waves) / 8 (four)
  constexpr int GB = BN / 8 / NW;                                              // row groups of one tap per wave: 2
  static_assert(GB == 2, "two filter DMAs per wave and tap");

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
  const int d = p.dstep < 0 ? -p.dstep : p.dstep;
  const int HC = kH2PW + 2 * d, HROWS = (kH2PH + 2 * d) * HC;                 // halo columns / pixels
  const int img = m0 / p.HoWo;
  const int patch = (m0 - img * p.HoWo) / (kH2PH * kH2PW);
  const int ph = patch / p.patches_per_row, pw = patch - ph * p.patches_per_row;
  const int h0 = ph * kH2PH - d, w0 = pw * kH2PW - d;                           // image position of halo pixel (0, 0)

  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.w, p.w_bytes);
  const int lrow = lane >> 3, lslot = lane & 7;
  uint32_t a_off[GA], b_off[GB];
#pragma unroll
  for (int g = 0; g < GA; ++g) {
    const int idx = 8 * (wave + NW * g) + lrow;
    const int hr = idx / HC, hc = idx - hr * HC;
    const int y = h0 + hr, x = w0 + hc;
    const bool ok = idx < HROWS && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi;
    a_off[g] = ok ? (uint32_t)(((img * p.Hi + y) * p.Wi + x) * p.ldx) * 2u + (uint32_t)((lslot ^ ((hc >> 1) & 7)) * 16) : kOOB;
  }
#pragma unroll
  for (int g = 0; g < GB; ++g) {
    const int n = 8 * (wave + NW * g) + lrow;
    b_off[g] = (n0 + n) < p.N ? (uint32_t)(n0 + n) * (uint32_t)p.K * 2u + (uint32_t)((lslot ^ ((n >> 1) & 7)) * 16) : kOOB;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr;
  auto issue_a = [&](int chunk) {
    const uint32_t kc = (uint32_t)chunk * 128u;
#pragma unroll
    for (int g = 0; g < GA; ++g)
      if (wave + NW * g < kH2MaxRows / 8)        // (wave-uniform)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr)(ldsw + kA + 8 * (wave + NW * g) * RDW), 16,
                                                 (int)(a_off[g] == kOOB ? kOOB : a_off[g] + kc), 0, 0, 0);
  };
  auto issue_b = [&](int chunk, int t) {
    const uint32_t kc = (uint32_t)chunk * 128u + (uint32_t)(t * p.Cin) * 2u;
#pragma unroll
    for (int g = 0; g < GB; ++g)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr)(ldsw + kB + ((t % 3) * BN + 8 * (wave + NW * g)) * RDW), 16,
                                               (int)(b_off[g] == kOOB ? kOOB : b_off[g] + kc), 0, 0, 0);
  };

  f32x16 acc[TM][1];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

  const int frag_row = lane & 31, frag_h = lane >> 5;
  const int pcol = frag_row & 15;
  const int hbase = (2 * wm * TM + (frag_row >> 4) + d) * HC + (pcol + d);        // strip i: + 2 i rows
  int a_col_dw[3], a_swz[3];
#pragma unroll
  for (int ts = 0; ts < 3; ++ts) {
    const int os = p.off0 + ts * p.dstep;            // -d, 0, +d (forward) or +d, 0, -d (data gradient)
    a_col_dw[ts] = os * RDW;
    a_swz[ts] = ((pcol + d + os) >> 1) & 7;
  }
  const int b_frag = (wn * 32 + frag_row) * RDW, b_swz = (frag_row >> 1) & 7;
  f32x4 fa[TM][4], fb[4];
  auto read_a = [&](int t) {
    const int tr = t / 3, ts = t - tr * 3;
    const int orow = p.off0 + tr * p.dstep;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int arow = kA + (hbase + (2 * i + orow) * HC) * RDW + a_col_dw[ts];
#pragma unroll
      for (int q = 0; q < 4; ++q) fa[i][q] = *reinterpret_cast<const f32x4*>(&lds[arow + (((2 * q + frag_h) ^ a_swz[ts]) << 2)]);
    }
  };
  auto read_b = [&](int t) {
    const int brow = kB + (t % 3) * BN * RDW + b_frag;
#pragma unroll
    for (int q = 0; q < 4; ++q) fb[q] = *reinterpret_cast<const f32x4*>(&lds[brow + (((2 * q + frag_h) ^ b_swz) << 2)]);
  };
  auto mfmas = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < TM; ++i)
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, fa[i][q]), __builtin_bit_cast(f16x8v, fb[q]),
                                                           acc[i][0], 0, 0, 0);
  };

  const int nchunks = p.Cin >> 6;
  for (int c = 0; c < nchunks; ++c) {
    issue_a(c);
    issue_b(c, 0);
    issue_b(c, 1);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      // in order: the halo image and taps <= t have landed when at most one younger tap (GB DMAs) is outstanding
      if (t < 8) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + 2 < 9) issue_b(c, t + 2);             // into the stage tap t - 1 was read from
      read_a(t);
      read_b(t);
      __builtin_amdgcn_sched_barrier(0);
      mfmas();
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // everybody is done reading: the next chunk (or the output patches) may land
  }

  // ---- epilogue: as gather_h_kernel (bias / accumulate / row map, fused BatchNorm statistics of the values as stored)
  const int col_l = lane & 31;
  const int row_h = (lane >> 5) * 4;
  {
    float* patchb = lds + wave * (32 * 36);
    const int col0 = n0 + wn * 32;
    int cv = p.N - col0;
    cv = cv < 0 ? 0 : (cv > 32 ? 32 : cv);
    auto rowmap = [&](int m) {
      int b, ho, wo;
      row_to_pixel(p, m, b, ho, wo);
      return (b * p.Ho + ho) * p.Wo + wo;
    };
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row0 = m0 + wm * 64 + i * 32;
      int rv = p.M - row0;
      rv = rv < 0 ? 0 : (rv > 32 ? 32 : rv);
      store_row32<1>(acc[i], patchb, p.y, hp.y_f32 != 0, p.ldy, row0, col0, rv, cv, p.bias, p.accumulate != 0, lane, rowmap);
    }
  }
  if (p.stat != nullptr) {
    const bool f32out = hp.y_f32 != 0;
    auto rnd = [&](float v) -> float { return f32out ? v : (float)(half_t)v; };
    const int group = tile_m * WARPS_M + wm;
    const long long gsz = (long long)p.stat_rows * p.N;
    const int col = n0 + wn * 32 + col_l;
    const float k0 = __shfl(rnd(acc[0][0][0]), lane & 31, 64);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
        if (row < p.M) {
          const float dd = rnd(acc[i][0][r]) - k0;
          s1 += dd;
          s2 += dd * dd;
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
#endif  // PSEG_LAB

// ------------------------------------------------------------------------------------------------ weight gradient
struct HWgradParams {
  WgradParams g;           // x / dy are fp16 here
  FastDiv howo_div, wo_div;
};

// chunk swizzle of the pixel-major LDS images: XOR of the 16-byte chunk index of pixel row `row`, by row length

template <int ROWBYTES>
__device__ __forceinline__ int wg_swz(int row) {
  if constexpr (ROWBYTES >= 256) return (row & 3) << 2;
  else if constexpr (ROWBYTES == 128) return ((row >> 1) & 1) << 2;
  else return 0;
}

template <int BM, int BN, int WARPS_M, int WARPS_N, bool SKIP, int STAGES, int BKP>
__global__ __launch_bounds__(256) void wgrad_h_kernel(const HWgradParams hp) {
  const WgradParams& p = hp.g;
  static_assert(WARPS_M * WARPS_N == 4, "4 waves");
  constexpr int NW = 4;
  constexpr int WTM = BM / WARPS_M, WTN = BN / WARPS_N;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(TM >= 1 && TN >= 1 && WTM % 32 == 0 && WTN % 32 == 0, "wave tile");
  constexpr int RBA = BM * 2, RBB = BN * 2;                     // bytes per pixel row of the images
  constexpr int kStageB = BKP * (RBA + RBB);                    // bytes per stage
  static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
  static_assert(BKP == 64 || BKP == 32, "K-step");
  constexpr int NSUB = BKP / 32;                                // 32-pixel sub-steps per K-step
  constexpr int kPatchB = NW * 32 * (WTN + 4) * 4;              // one 32-row tile row per wave at a time (store_row32)
  constexpr int kLdsB = STAGES * kStageB > kPatchB ? STAGES * kStageB : kPatchB;
  static_assert(kLdsB <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[kLdsB];
  constexpr int kA = 0, kB = BKP * RBA;                         // byte offsets inside a stage
  // DMA pieces (1 KiB = one wave-instruction): RA / RB pixel rows each, IA / IB pieces per wave and K-step
  constexpr int RA = 1024 / RBA, RB = 1024 / RBB;
  constexpr int PA = BKP / RA, PB = BKP / RB;                   // pieces per K-step
  constexpr int IA = (PA + NW - 1) / NW, IB = (PB + NW - 1) / NW;
  // (fewer pieces than waves -- a 32-column operand with the short K-step: the spare waves re-load piece (wave mod pieces),
  // the same bytes into the same place, so that every wave issues and counts the same number of DMAs)
  static_assert((PA % NW == 0 || PA < NW) && (PB % NW == 0 || PB < NW), "whole pieces per wave");
  static_assert(RA >= 4 && RB >= 4 && RA <= 32 && RB <= 32, "a piece covers whole 4-row groups inside one 32-pixel sub-step");
  constexpr int CA = RBA / 16, CB = RBB / 16;                   // 16-byte chunks per pixel row

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WARPS_N, wn = wave % WARPS_N;
  const int gridN = (p.K + BN - 1) / BN;
  int wg_tile, wg_split;
  wgrad_block((int)gridDim.x, wg_tile, wg_split);
  const int tile_n = wg_tile % gridN;
  const int tile_m = wg_tile / gridN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes);
  const __amdgpu_buffer_rsrc_t dr = make_rsrc(p.dy, p.dy_bytes);

  const int p_begin = wg_split * p.pix_per_split;
  int p_end = p_begin + p.pix_per_split;
  if (p_end > p.P) p_end = p.P;

  // ---- per-lane constants of the DMA pieces.  Piece q = wave + NW * g of the K-step covers pixel rows [R q, R q + R);
  // lane l fills (row l / C, physical chunk l % C) and therefore fetches the logical chunk (l % C) ^ swz(row).
  int a_row[IA], a_col[IA];        // pixel row inside the K-step (0..63), first channel of the chunk or -1
  int a_ih[IA], a_iw[IA];          // patch mode: position inside the 32-pixel patch
  int b_row[IB], b_c[IB], b_dh[IB], b_dw[IB], b_ih[IB], b_iw[IB];
  bool b_colok[IB];
  int a_piece[IA], b_piece[IB];
#pragma unroll
  for (int g = 0; g < IA; ++g) {
    a_piece[g] = (wave + NW * g) % PA;
    const int row = RA * a_piece[g] + lane / CA;
    const int col = m0 + 8 * ((lane % CA) ^ wg_swz<RBA>(row));
    a_row[g] = row;
    a_col[g] = col < p.Cout ? col : -1;
    const int rr = row & 31;
    a_ih[g] = rr / p.patch_w;
    a_iw[g] = rr - a_ih[g] * p.patch_w;
  }
#pragma unroll
  for (int g = 0; g < IB; ++g) {
    b_piece[g] = (wave + NW * g) % PB;
    const int row = RB * b_piece[g] + lane / CB;
    const int col = n0 + 8 * ((lane % CB) ^ wg_swz<RBB>(row));
    b_row[g] = row;
    b_colok[g] = col < p.K;
    const int tap = b_colok[g] ? col / p.Cin : 0;
    b_c[g] = col - tap * p.Cin;
    const int r = tap / p.kw, sx = tap - r * p.kw;
    b_dh[g] = r * p.dil - p.pad;
    b_dw[g] = sx * p.dil - p.pad;
    const int rr = row & 31;
    b_ih[g] = rr / p.patch_w;
    b_iw[g] = rr - b_ih[g] * p.patch_w;
  }

  // patch mode (the map tiles into 32-pixel patches: every DeepLabV3+ / UNet / HRNet layer but the smallest maps): the byte
  // offset of a piece is a block-uniform patch origin + a per-lane constant, so a K-step costs no address arithmetic for the dy
  // operand (the origin rides in the scalar offset of the DMA) and one add + the range test of its tap for the x operand
  // (round 4: the per-piece multiply chains were 9 VALU + 11 SALU instructions per MFMA)
  uint32_t a_loff[IA];
  int b_loff[IB], b_ch[IB], b_cw[IB];
#pragma unroll
  for (int g = 0; g < IA; ++g)
    a_loff[g] = a_col[g] >= 0 ? (uint32_t)((a_ih[g] * p.Wo + a_iw[g]) * p.ldy + a_col[g]) * 2u : kOOB;
#pragma unroll
  for (int g = 0; g < IB; ++g) {
    b_ch[g] = b_ih[g] * p.stride + b_dh[g];
    b_cw[g] = b_iw[g] * p.stride + b_dw[g];
    b_loff[g] = ((b_ch[g] * p.Wi + b_cw[g]) * p.ldx + b_c[g]) * 2;
  }

  const int t_dh = (n0 / p.Cin / p.kw) * p.dil - p.pad;
  const int t_dw = ((n0 / p.Cin) % p.kw) * p.dil - p.pad;
  // the stream of live 32-pixel sub-steps; a K-step takes two of them (the second may be none: zeros)
  auto next_valid = [&](int pt) -> int {
    if (SKIP)
      while (pt < p_end && wg_step_dead(p, pt, p_end, t_dh, t_dw)) pt += 32;
    return pt;
  };

  typedef __attribute__((address_space(3))) void* lds_ptr;
  auto issue = [&](int pt0, int pt1, int st) {     // sub-steps at pixels pt0 / pt1 (>= p_end: all-zero pieces)
    unsigned char* sb = lds_raw + st * kStageB;
    if (p.patch_mode) {
      uint32_t oa[NSUB], obx[NSUB];      // block-uniform byte origins of the sub-steps' patches in dy / x (kOOB: no such sub-step)
      int hs[NSUB], ws[NSUB];
#pragma unroll
      for (int s = 0; s < NSUB; ++s) {
        const int pt = s ? pt1 : pt0;
        int b = 0, h0 = 0, w0 = 0;
        const bool live = pt < p_end;
        if (live) wg_patch_origin(p, pt, b, h0, w0);
        hs[s] = h0 * p.stride;
        ws[s] = w0 * p.stride;
        oa[s] = live ? (uint32_t)(((b * p.Ho + h0) * p.Wo + w0) * p.ldy) * 2u : kOOB;
        obx[s] = live ? (uint32_t)(((b * p.Hi + hs[s]) * p.Wi + ws[s]) * p.ldx) * 2u : kOOB;
      }
#pragma unroll
      for (int g = 0; g < IA; ++g) {
        const int s = a_row[g] >> 5;
        // (a dead sub-step or a channel chunk beyond Cout: origin or lane constant is kOOB, the sum stays out of range)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(dr, (lds_ptr)(sb + kA + RA * a_piece[g] * RBA), 16, (int)(oa[s] + a_loff[g]), 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < IB; ++g) {
        const int s = b_row[g] >> 5;
        const int hi = hs[s] + b_ch[g], wi = ws[s] + b_cw[g];
        const bool ok = b_colok[g] && obx[s] != kOOB && ((unsigned)hi < (unsigned)p.Hi) && ((unsigned)wi < (unsigned)p.Wi);
        const uint32_t off = ok ? obx[s] + (uint32_t)b_loff[g] : kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr)(sb + kB + RB * b_piece[g] * RBB), 16, (int)off, 0, 0, 0);
      }
      return;
    }
    int ob[NSUB], oh[NSUB], ow[NSUB];
    bool live[NSUB];
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
      const int pt = s ? pt1 : pt0;
      live[s] = pt < p_end;
      ob[s] = oh[s] = ow[s] = 0;
      if (live[s] && p.patch_mode) wg_patch_origin(p, pt, ob[s], oh[s], ow[s]);
    }
#pragma unroll
    for (int g = 0; g < IA; ++g) {
      const int s = a_row[g] >> 5;
      const int pt = s ? pt1 : pt0;
      int b, ho, wo;
      bool ok = live[s] && a_col[g] >= 0;
      if (p.patch_mode) {
        b = ob[s];
        ho = oh[s] + a_ih[g];
        wo = ow[s] + a_iw[g];
      } else {
        const int pix = pt + (a_row[g] & 31);
        ok = ok && pix < p_end;
        const uint32_t bb = hp.howo_div.div((uint32_t)(ok ? pix : 0));
        const uint32_t rem = (uint32_t)(ok ? pix : 0) - bb * hp.howo_div.d;
        const uint32_t hh = hp.wo_div.div(rem);
        b = (int)bb;
        ho = (int)hh;
        wo = (int)(rem - hh * hp.wo_div.d);
      }
      const uint32_t off = ok ? (uint32_t)(((b * p.Ho + ho) * p.Wo + wo) * p.ldy + a_col[g]) * 2u : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(dr, (lds_ptr)(sb + kA + RA * a_piece[g] * RBA), 16, (int)off, 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < IB; ++g) {
      const int s = b_row[g] >> 5;
      const int pt = s ? pt1 : pt0;
      int b, ho, wo;
      bool ok = live[s] && b_colok[g];
      if (p.patch_mode) {
        b = ob[s];
        ho = oh[s] + b_ih[g];
        wo = ow[s] + b_iw[g];
      } else {
        const int pix = pt + (b_row[g] & 31);
        ok = ok && pix < p_end;
        const uint32_t bb = hp.howo_div.div((uint32_t)(ok ? pix : 0));
        const uint32_t rem = (uint32_t)(ok ? pix : 0) - bb * hp.howo_div.d;
        const uint32_t hh = hp.wo_div.div(rem);
        b = (int)bb;
        ho = (int)hh;
        wo = (int)(rem - hh * hp.wo_div.d);
      }
      const int hi = ho * p.stride + b_dh[g], wi = wo * p.stride + b_dw[g];
      ok = ok && ((unsigned)hi < (unsigned)p.Hi) && ((unsigned)wi < (unsigned)p.Wi);
      const uint32_t off = ok ? (uint32_t)(((b * p.Hi + hi) * p.Wi + wi) * p.ldx + b_c[g]) * 2u : kOOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr)(sb + kB + RB * b_piece[g] * RBB), 16, (int)off, 0, 0, 0);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- transposed fragment reads.  MFMA operand of lane l: row (channel) l & 31, k (pixels) 8 * (l >> 5) + 0..7.
  // ds_read_b64_tr_b16 per 16-lane group: a block of 4 pixel rows x 16 channels; lane 4q + pp supplies the address of
  // (row q, channels 4 pp .. 4 pp + 3), lane i receives channel i of the 4 rows.  Read r (0 / 1) of k16-step kk takes pixel
  // rows 16 kk + 8 (l >> 5) + 4 r + q, channels tile + 16 ((l >> 4) & 1) + ...: element e of read r is k = 4 r + e.
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1, th = lane >> 5;
  int offA[TM], offB[TN];          // byte offsets inside a stage for kk = 0, r = 0
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = 8 * th + tq;                                              // (+ 16 kk + 4 r: multiples of 4)
    const int chunk = (wm * WTM + 32 * i) / 8 + 2 * tg + (tp >> 1);
    offA[i] = kA + row * RBA + 16 * (chunk ^ wg_swz<RBA>(row)) + 8 * (tp & 1);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = 8 * th + tq;
    const int chunk = (wn * WTN + 32 * j) / 8 + 2 * tg + (tp >> 1);
    offB[j] = kB + row * RBB + 16 * (chunk ^ wg_swz<RBB>(row)) + 8 * (tp & 1);
  }
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  constexpr int HK = BKP / 32;           // 16-pixel MFMA steps per half of a K-step: 2 or 1
  // fragment halves as the transposing reads deliver them (inline assembly: see tr_read_asm -- the builtin made hipcc drain the
  // operand ring with `s_waitcnt vmcnt(0)` in front of every K-step's first read); [set][k2 * T + tile], k2 = 16-pixel step
  // inside the half.  The halves are joined into MFMA operands only AFTER the wait that covers them.
  s16x4t falo[2][HK * TM], fahi[2][HK * TM], fblo[2][HK * TN], fbhi[2][HK * TN];
  constexpr int kReadsPerCall = HK * (TM + TN) * 2;
  const uint32_t lds0 = lds_addr_of(lds_raw);
  auto read_frags = [&](auto setc, int st, auto halfc) {
    constexpr int SET = decltype(setc)::value, HALF = decltype(halfc)::value;
    const uint32_t sbase = lds0 + (uint32_t)(st * kStageB);
    static_for<HK>([&](auto k2c) {
      constexpr int k2 = decltype(k2c)::value, kk = HALF * HK + k2;
      static_for<TM>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const uint32_t a = sbase + (uint32_t)offA[i];
        falo[SET][k2 * TM + i] = tr_read_asm<(16 * kk) * RBA>(a);
        fahi[SET][k2 * TM + i] = tr_read_asm<(16 * kk + 4) * RBA>(a);
      });
      static_for<TN>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const uint32_t b = sbase + (uint32_t)offB[j];
        fblo[SET][k2 * TN + j] = tr_read_asm<(16 * kk) * RBB>(b);
        fbhi[SET][k2 * TN + j] = tr_read_asm<(16 * kk + 4) * RBB>(b);
      });
    });
  };
  auto mfmas = [&](auto setc) {
    constexpr int SET = decltype(setc)::value;
#pragma unroll
    for (int k2 = 0; k2 < HK; ++k2)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const s16x4t al = falo[SET][k2 * TM + i], ah = fahi[SET][k2 * TM + i];
          const s16x4t bl = fblo[SET][k2 * TN + j], bh = fbhi[SET][k2 * TN + j];
          const s16x8 av = s16x8{al[0], al[1], al[2], al[3], ah[0], ah[1], ah[2], ah[3]};
          const s16x8 bv = s16x8{bl[0], bl[1], bl[2], bl[3], bh[0], bh[1], bh[2], bh[3]};
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, av), __builtin_bit_cast(f16x8v, bv), acc[i][j],
                                                             0, 0, 0);
        }
  };
  // all but the youngest N LDS reads have returned (the counter has four bits)
  auto wait_older_reads = [&]() {
    constexpr int N = kReadsPerCall < 15 ? kReadsPerCall : 15;
    if constexpr (N == 15) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  typedef std::integral_constant<int, 0> c0;
  typedef std::integral_constant<int, 1> c1;
  {
    // the stream of live 32-pixel sub-steps, two per K-step; head[i] = first sub-step of the i-th tile of the ring (head[0] is
    // the one being multiplied, the others are in flight), p_end once the stream is exhausted (all-zero dummy pieces)
    constexpr int NG = IA + IB;
    int cursor = next_valid(p_begin);
    auto take = [&]() -> int {
      const int q = cursor;
      if (cursor < p_end) cursor = next_valid(cursor + 32);
      return q;
    };
    int head[STAGES];
    if (cursor < p_end) {
#pragma unroll
      for (int s0 = 0; s0 < STAGES; ++s0) {
        const int u = take(), v = NSUB == 2 ? take() : p_end;
        head[s0] = u;
        issue(u, v, s0);
      }
      wait_vmcnt<(STAGES - 1) * NG>();
      __builtin_amdgcn_s_barrier();
      read_frags(c0{}, 0, c0{});
      int st = 0;
      while (head[0] < p_end) {
        const int st1 = st == STAGES - 1 ? 0 : st + 1;
        read_frags(c1{}, st, c1{});
        wait_older_reads();             // set 0 (issued one phase ago) is in its registers; set 1 may still be on its way
        __builtin_amdgcn_sched_barrier(0);
        mfmas(c0{});
        __builtin_amdgcn_sched_barrier(0);
        wait_vmcnt<(STAGES - 2) * NG>();                    // the next K-step has landed; STAGES - 2 more stay in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // set 1 is in; this wave is done reading stage `st`
        __builtin_amdgcn_s_barrier();
        read_frags(c0{}, st1, c0{});    // (zeros on the last step: never multiplied)
        __builtin_amdgcn_sched_barrier(0);
        const int u = take(), v = NSUB == 2 ? take() : p_end;
        issue(u, v, st);              // stage `st` is free now
        mfmas(c1{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s0 = 0; s0 + 1 < STAGES; ++s0) head[s0] = head[s0 + 1];
        head[STAGES - 1] = u;
        st = st1;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // dummy pieces must not land in the output patches
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }

  float* out = p.dw + (long long)wg_split * p.slab_stride;
  {
    float* patch = reinterpret_cast<float*>(lds_raw) + wave * (32 * (WTN + 4));
    const int col0 = n0 + wn * WTN;
    int cv = p.K - col0;
    cv = cv < 0 ? 0 : (cv > WTN ? WTN : cv);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row0 = m0 + wm * WTM + i * 32;
      int rv = p.Cout - row0;
      rv = rv < 0 ? 0 : (rv > 32 ? 32 : rv);
      store_row32<TN>(acc[i], patch, out, true, p.K, row0, col0, rv, cv, nullptr, p.accumulate != 0, lane, [](int m) { return m; });
    }
  }
}

// ------------------------------------------------------------------------------------------------ filters
// Every dense conv filter of a model in ONE launch, from the fp32 master copy: the fp16 filter [Cout'][taps][Cin'] the
// forward convs read AND its transposed fp16 copy [Cin'][taps][Cout'] for the data gradients.  The fp16 copies may be padded
// further than the master ([Cout][taps][Cin] -> Cout' >= Cout, Cin' >= Cin: 8-channel granules; the padding is zero-filled).
// jobs[j] = {w fp32, w_h, wT_h, Cout, taps, Cin, Cout', Cin', first 32x32 tile of job j} (9 x int64, device memory, tile
// offsets ascending; tiles cover the PADDED extents); block b finds its job by bisection.
__global__ __launch_bounds__(256) void filter_prepare_h_kernel(const long long* __restrict__ jobs, int n) {
  PSEG_HELPER_PRIO();
  __shared__ float tile[32][33];
  const long long b = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid * 9 + 8] <= b) lo = mid;
    else hi = mid - 1;
  }
  const long long* job = jobs + lo * 9;
  const float* w = reinterpret_cast<const float*>(job[0]);
  half_t* wh = reinterpret_cast<half_t*>(job[1]);
  half_t* wT = reinterpret_cast<half_t*>(job[2]);
  const int Cout = (int)job[3], taps = (int)job[4], Cin = (int)job[5], CoutP = (int)job[6], CinP = (int)job[7];
  const int tci = (CinP + 31) / 32, tco = (CoutP + 31) / 32;
  int local = (int)(b - job[8]);
  const int t = local / (tci * tco);
  local -= t * tci * tco;
  if (t >= taps) return;
  const int co0 = (local / tci) * 32, ci0 = (local % tci) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = w[((long long)co * taps + t) * Cin + ci];
    if (wh != nullptr && co < CoutP && ci < CinP) wh[((long long)co * taps + t) * CinP + ci] = (half_t)v;
    tile[r][tx] = v;
  }
  __syncthreads();
  if (wT != nullptr)
    for (int r = ty; r < 32; r += 8) {
      const int ci = ci0 + r, co = co0 + tx;
      if (ci < CinP && co < CoutP) wT[((long long)ci * taps + t) * CoutP + co] = (half_t)tile[tx][r];
    }
}

// strided [M][C] conversion between fp32 and fp16 tensors with an optional multiplier: y = convert(x * scale), where
// scale = *dev_scale (device scalar: the dynamic loss scale) when non-null.  4 channels per lane.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void convert2d_kernel(const TI* __restrict__ x, int ldx, TO* __restrict__ y, int ldy,
                                                        uint32_t total, FastDiv c4div, const float* __restrict__ dev_scale) {
  PSEG_HELPER_PRIO();
  const float s = dev_scale != nullptr ? *dev_scale : 1.f;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t r = c4div.div(i);
    const uint32_t c = (i - r * c4div.d) * 4;
    stv4(y + (long long)r * ldy + c, ldv4(x + (long long)r * ldx + c) * s);
  }
}

// ------------------------------------------------------------------------------------------------ host side
static long long nhwc_bytes_h(int B, int H, int W, int C, int ld) { return (((long long)B * H * W - 1) * ld + C) * 2; }

// wave rows of a tile (= statistics groups per row tile)
static int waves_m_h(TileCfg t) { return (t.bn == 32 || (t.bm == 256 && t.bn == 128)) ? 4 : 2; }

// tiles the fp16 gather kernel is instantiated for
static TileCfg half_tile(TileCfg t) {
  if (t.bm == 256) t.bm = 128;
  if (t.bm == 32) t.bm = 64;
  if (t.bm == 64) t.bn = 128;
  if (t.bm == 128 && !(t.bn == 128 || t.bn == 64 || t.bn == 32)) t.bn = 128;
  return t;
}

// the plan of a gather problem on the fp16 kernels: tile and row order from the shared planner (the schedules of the
// dilated convs carry over), never split-K
// K-step (halves per LDS row) of a gather problem.  PSEG_HCONV_KB = 32 / 64 forces one.
static int hconv_kb(int Cin, int K) {
  static const int forced = env_int("PSEG_HCONV_KB", 0);
  if (forced == 32 || forced == 64) return forced;
  if (Cin % 64 != 0) return 32;          // 32-channel layers stay on the hoisted (tap-uniform) addressing with the short step
  // measured (tools/bench_conv_half.py over PSEG_HCONV_KB x PSEG_HCONV_STAGES): the short step with a three-deep ring wins
  // 5-20 % on the bandwidth-bound layers with K <= 576 (64 -> 256 channels on 128x128 maps: 40 vs 52 us against an HBM floor of
  // 31; 3x3 on 64 channels 42 vs 45) and loses 10-25 % on the deep contractions (layer-2..4 3x3, ASPP)
  return K <= 576 ? 32 : 64;
}

static FwdPlan plan_gather_h(long long M, int N, int K, int Cin, const DilGeom* geom) {
  const int kb = hconv_kb(Cin, K);
  // dilated convs: the tap-skipping schedules (patch / class-sorted rows, 64-row candidate tiles) are only searched when the
  // plain plan lands on a 4-wave tile; a 128x128 / 8-wave problem runs dense and row-major (see run_gather_h)
  FwdPlan pl = plan_gather(M, N, K, false, nullptr);
  if (geom != nullptr && Cin % kb == 0 && !(pl.tile.bm == 128 && pl.tile.bn == 128)) pl = plan_gather(M, N, K, false, geom);
  // one 128x128 block per CU beats two 128x64 blocks on the fp16 kernels when the problem has no dead taps to skip (round 4:
  // M = 16384 x N = 256, K = 2304 -- layer 3's 3x3 convs -- 32 -> 28 us, 26 on the persistent kernel): the kernels are bound by
  // the bytes a CU pulls through its LDS-DMA path, and the wider tile pulls a third fewer of them per MAC
  if (geom == nullptr && pl.tile.bm == 128 && pl.tile.bn == 64 && N >= 128 && K >= 1024 &&
      (long long)cdiv(M, 128) * cdiv(N, 128) >= 256 && cfg().conv_bm == 0) {
    pl.tile.bn = 128;
    pl.gridN = cdiv(N, 128);
  }
  const TileCfg t = half_tile(pl.tile);
  if (t.bm != pl.tile.bm || t.bn != pl.tile.bn || pl.splits > 1) {
    // (a substituted tile keeps the plain row order: patch / class schedules were costed for the planner's own tile)
    if (t.bm != pl.tile.bm) {
      pl.patch_h = pl.patch_w = 0;
      pl.banded = false;
    }
    pl.tile = t;
    pl.gridM = cdiv(M, t.bm);
    pl.gridN = cdiv(N, t.bn);
  }
  pl.splits = 1;
#if PSEG_LAB
  // experiments (tools/bench_conv_half.py): PSEG_HCONV_TILE = 1: 128x128 on four waves (64x64 wave tiles), 2: 256x128 on eight
  // waves (64x64 wave tiles), 3: 256x256 on eight waves (128x64 wave tiles)
  const int forced_tile = cfg().hconv_tile;
  if (forced_tile >= 1 && forced_tile <= 4 && Cin % kb == 0 && !pl.banded && pl.patch_w == 0) {
    pl.tile = forced_tile == 1 ? TileCfg{128, 128} : (forced_tile == 3 ? TileCfg{256, 256} : TileCfg{256, 128});
    pl.hwaves = forced_tile == 1 ? 4 : (forced_tile == 4 ? 16 : 8);
    pl.gridM = cdiv(M, pl.tile.bm);
    pl.gridN = cdiv(N, pl.tile.bn);
  }
#endif
  pl.kt_total = cdiv(K, kb);
  pl.kt_per_split = pl.kt_total;
  return pl;
}

// Instantiations that also exist with the fused BatchNorm-backward sums (BNS = true): the tiles / K-steps / ring depths the
// Bottleneck and BasicBlock data gradients of the reference's models are planned onto (every tap live, channels in whole
// K-steps).  Everything else answers pseg_conv2d_dgrad_bnstat_rows_h with 0 and the layer keeps its reduction pass.
template <int BM, int BN, int WM, int WN, int KB, int ST>
constexpr bool kHasBnsH = ((BM == 128 && BN == 128 && WM == 2 && WN == 4) || (BM == 128 && BN == 64 && WM == 2 && WN == 2) ||
                           (BM == 128 && BN == 32 && WM == 4 && WN == 1 && KB == 32)) &&
                          ((KB == 32 && ST == 3) || (KB == 64 && (ST == 2 || ST == 3)));
template <int BM, int BN, int WM, int WN, int KB, int ST>
constexpr bool kHasBnsHP = ((BM == 128 && BN == 128 && WM == 2 && WN == 4) || (BM == 128 && BN == 64 && WM == 2 && WN == 2) ||
                            (BM == 128 && BN == 32 && WM == 4 && WN == 1 && KB == 32)) &&
                           ((KB == 32 && ST == 3) || (KB == 64 && ST == 2));

// one instantiation of the gather kernel, if it exists (LDS budget; the 256x256 tile only with the short K-step: its
// accumulators take 128 registers).  bns: 0 plain; 1 launch the BNS instantiation; 2 only say whether it exists.
template <int BM, int BN, int WM, int WN, int KB, int ST>
static bool launch_gather_h(int variant, dim3 grid, hipStream_t st, const HGatherParams& hp, int bns = 0) {
  constexpr int NW = WM * WN;
  constexpr long long ring = (long long)ST * (BM + BN) * KB * 2, patch = (long long)NW * 32 * (BN / WN + 4) * 4;
  constexpr bool fits = (ring > patch ? ring : patch) <= 160 * 1024 && !(BM == 256 && BN == 256 && KB == 64);
  if constexpr (fits) {
    const dim3 block(64 * NW);
    if (bns != 0) {
      if constexpr (kHasBnsH<BM, BN, WM, WN, KB, ST>) {
        if (variant != 0) return false;
        if (bns == 1) hipLaunchKernelGGL((gather_h_kernel<BM, BN, WM, WN, false, false, KB, ST, true>), grid, block, 0, st, hp);
        return true;
      } else {
        return false;
      }
    }
    if (variant == 2) hipLaunchKernelGGL((gather_h_kernel<BM, BN, WM, WN, false, true, KB, ST>), grid, block, 0, st, hp);
    else if (variant == 1) hipLaunchKernelGGL((gather_h_kernel<BM, BN, WM, WN, true, false, KB, ST>), grid, block, 0, st, hp);
    else hipLaunchKernelGGL((gather_h_kernel<BM, BN, WM, WN, false, false, KB, ST>), grid, block, 0, st, hp);
    return true;
  }
  return false;
}

// one instantiation of the persistent kernel, if it exists; grid = min(tiles, CUs x resident blocks per CU)
template <int BM, int BN, int WM, int WN, int KB, int ST, bool BNS>
static bool launch_gather_hp_one(int ntiles, hipStream_t st, const HGatherParams& hp, bool dry) {
  constexpr int NW = WM * WN;
  constexpr long long lds_bytes = (long long)ST * (BM + BN) * KB * 2 + (long long)NW * 32 * 72;
  static int resident = 0;      // blocks of this instantiation the device holds at once
  if (resident == 0) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gather_hp_kernel<BM, BN, WM, WN, KB, ST, BNS>, 64 * NW, 0) != hipSuccess ||
        per_cu < 1)
      return false;
    const int by_lds = (int)((160 * 1024) / lds_bytes);
    if (per_cu > by_lds) per_cu = by_lds;
    static const int forced_bpc = env_int("PSEG_HCONV_PBPC", 0);
    if (forced_bpc > 0 && forced_bpc < per_cu) per_cu = forced_bpc;
    resident = prop.multiProcessorCount * per_cu;
  }
  if (dry) return true;
  const int blocks = ntiles < resident ? ntiles : resident;
  hipLaunchKernelGGL((gather_hp_kernel<BM, BN, WM, WN, KB, ST, BNS>), dim3((unsigned)blocks), dim3(64 * NW), 0, st, hp);
  return true;
}

template <int BM, int BN, int WM, int WN, int KB, int ST>
static bool launch_gather_hp(int ntiles, hipStream_t st, const HGatherParams& hp, int bns = 0) {
  constexpr int NW = WM * WN;
  constexpr long long lds_bytes = (long long)ST * (BM + BN) * KB * 2 + (long long)NW * 32 * 72;
  if constexpr (lds_bytes <= 160 * 1024 && !(BM == 256 && BN == 256)) {
    if (bns != 0) {
      if constexpr (kHasBnsHP<BM, BN, WM, WN, KB, ST>) return launch_gather_hp_one<BM, BN, WM, WN, KB, ST, true>(ntiles, st, hp, bns == 2);
      else return false;
    }
    return launch_gather_hp_one<BM, BN, WM, WN, KB, ST, false>(ntiles, st, hp, false);
  }
  return false;
}

// fused BatchNorm-backward partial sums of a data gradient (pseg_conv2d_dgrad_bnstat_h): the producing layer's y (fp16), its
// fp32 coefficient rows, where the [rows][N] partial sums go
struct HBnsArgs {
  const void* y;
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

// plan of a gather problem as run_gather_h makes it (shared with the queries)
static FwdPlan plan_run_h(int Hi, int Wi, int Cin, int Ho, int Wo, long long M, int N, int taps_w, int K, int s_out, int s_in,
                          int dstep, int off0) {
  const int kb = hconv_kb(Cin, K);
  const bool generic = Cin % kb != 0;
  DilGeom geom;
  const bool has_geom = !generic && dil_geom(geom, Ho, Wo, Hi, Wi, (K / Cin) / taps_w, taps_w, Cin, s_out, s_in, dstep, off0);
  return plan_gather_h(M, N, K, Cin, has_geom ? &geom : nullptr);
}

static int run_gather_h(const void* x, long long x_bytes, int ldx, const void* w, void* y, int ldy, int y_f32,
                        const float* bias, float* stat, int B, int Hi, int Wi, int Cin, int Ho, int Wo, int N, int taps_w,
                        int K, int s_out, int s_in, int dstep, int off0, int accumulate, hipStream_t st,
                        const HBnsArgs* bns = nullptr, bool bns_query = false) {
  const long long M = (long long)B * Ho * Wo;
  PSEG_REQUIRE(M > 0 && M < (1LL << 31) && N > 0 && K > 0, "conv_h: empty or oversized problem M=%lld N=%d K=%d", M, N, K);
  PSEG_REQUIRE(Cin % 8 == 0 && ldx % 8 == 0 && N % 8 == 0, "conv_h: Cin (%d), ldx (%d) and the output channels (%d) must be multiples of 8",
               Cin, ldx, N);
  PSEG_REQUIRE(ldy % (y_f32 ? 4 : 8) == 0 && ldy >= N, "conv_h: ldy (%d) must cover N and be a multiple of %d", ldy, y_f32 ? 4 : 8);
  PSEG_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)y & 15) == 0, "conv_h: x / w / y must be 16-byte aligned");
  const long long w_bytes = (long long)N * K * 2;
  PSEG_REQUIRE(x_bytes < kMaxBytes && w_bytes < kMaxBytes, "conv_h: tensor exceeds 2 GiB (x %lld, w %lld bytes)", x_bytes, w_bytes);
  PSEG_REQUIRE(((M - 1) * ldy + N) * 4 < (1LL << 40), "conv_h: output too large");
  const int kb = hconv_kb(Cin, K);
  const bool generic = Cin % kb != 0;
  FwdPlan pl = plan_run_h(Hi, Wi, Cin, Ho, Wo, M, N, taps_w, K, s_out, s_in, dstep, off0);

  HGatherParams hp;
  GatherConvParams& p = hp.g;
  p.x = reinterpret_cast<const float*>(x);
  p.w = reinterpret_cast<const float*>(w);
  p.y = reinterpret_cast<float*>(y);
  p.ldy = ldy;
  p.bias = bias;
  p.stat = stat;
  p.stat_rows = pl.gridM * waves_m_h(pl.tile);
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
  p.accumulate = accumulate;
  p.kt_total = pl.kt_total;
  p.kt_per_split = pl.kt_per_split;
  p.slab_stride = 0;
  const int taps = K / Cin;
  const int adil = dstep < 0 ? -dstep : dstep;
  p.ntaps = taps;
  p.ktiles_per_tap = generic ? 1 : Cin / kb;
  // dilated convs: K-steps of taps that are zero padding for the whole M tile are skipped -- on the 4-wave tiles only.  On the
  // 128x128 / 8-wave tile (the ASPP data gradients: N = 2048) the tap-skipping instantiation is SLOWER than the dense one
  // even where it skips half the K-steps (rate 18: 269 us patch-ordered / 290 class-sorted against 168 dense; rate 6: 272
  // against 206): one fp16 MFMA pass per tile leaves the kernel bound by its operand stream, a padding tap's DMA is an
  // out-of-range no-op that costs nothing, and the skip bookkeeping does (tools/bench_conv_half.py with PSEG_CONV_NOSKIP=1).
  p.skip_taps = (!generic && adil >= 4 && taps > 1 && taps <= 32 && cfg().conv_noskip == 0 &&
                 !(pl.tile.bm == 128 && pl.tile.bn == 128)) ? 1 : 0;
  p.xcd_remap = cfg().conv_noxcd == 0 ? 1 : 0;
  p.prio = dstep < 0 ? cfg().dgrad_prio : 0;
  p.row_perm = 0;
  p.patch_w = p.patch_hw = p.patches_per_row = 1;
  if (K == Cin && s_out == 1 && s_in == 1 && off0 == 0 && Hi == Ho && Wi == Wo) {
    p.row_perm = 3;     // 1x1, unit stride: the tensor is one long row of M pixels (no index arithmetic per row)
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
    p.xcd_remap = 2;
  }
  if (!generic && s_in == 2 && Ho % 2 == 0 && Wo % 2 == 0 && ((Ho / 2) * (Wo / 2)) % pl.tile.bm == 0 && taps <= 32 &&
      cfg().conv_noskip == 0) {
    p.row_perm = 1;     // stride-2 data gradient: parity-homogeneous tiles, 3/4 of the taps skipped
    p.skip_taps = 1;
  }
  p.trace = nullptr;
  p.bns_y = nullptr;
  p.bns_ldy = 0;
  p.bns_mean = p.bns_invstd = p.bns_scale = p.bns_shift = nullptr;
  p.bns_act = 0;
  p.bns_db = p.bns_dg = nullptr;
  if (bns != nullptr) {
    PSEG_REQUIRE(!y_f32 && bias == nullptr && !accumulate && stat == nullptr,
                 "conv_h: the fused BatchNorm-backward sums need an fp16 result without bias / accumulation / statistics");
    PSEG_REQUIRE(bns->rows == p.stat_rows, "conv_h: %d partial rows handed in, the plan has %d", bns->rows, p.stat_rows);
    p.bns_y = reinterpret_cast<const float*>(bns->y);
    p.bns_ldy = bns->ldy;
    p.bns_mean = bns->mean;
    p.bns_invstd = bns->invstd;
    p.bns_scale = bns->scale;
    p.bns_shift = bns->shift;
    p.bns_act = bns->act;
    p.bns_db = bns->db;
    p.bns_dg = bns->dg;
  }
  p.precision = 0;
  p.amax_a = p.amax_b = nullptr;
  p.xh = p.xl = p.wh = p.wl = nullptr;
  p.xp_bytes = p.wp_bytes = 0;
  p.ldxp = 0;
  hp.y_f32 = y_f32;
#if PSEG_LAB
  static const int ablate = env_int("PSEG_HCONV_ABLATE", 0);
  hp.ablate = ablate;
#else
  hp.ablate = 0;
#endif
  hp.cin_div = FastDiv((uint32_t)Cin);
  hp.kw_div = FastDiv((uint32_t)taps_w);
  hp.howo_div = FastDiv((uint32_t)p.HoWo);      // (after the pointwise rewrite above: HoWo = Wo = M there)
  hp.wo_div = FastDiv((uint32_t)p.Wo);
#if PSEG_LAB
  // halo-staged 3x3 (gather_hh_kernel): unit stride, dilation 1 / 2, channels in whole 64-chunks, maps of 8 x 16 patches, fp16
  // result without bias / accumulation, and a plan whose statistics layout is the kernel's (128-row tiles, two wave rows).
  // PSEG_HCONV_HALO=0: off.
  static const int halo_on = env_int("PSEG_HCONV_HALO", 0);      // (lab build, opt-in)
  if (halo_on != 0 && bns == nullptr && !generic && taps == 9 && taps_w == 3 && s_out == 1 && s_in == 1 && (adil == 1 || adil == 2) &&
      (off0 == -adil || off0 == adil) && Cin % 64 == 0 && Ho % kHaloPH == 0 && Wo % kHaloPW == 0 && Hi == Ho && Wi == Wo &&
      !y_f32 && bias == nullptr && !accumulate && pl.tile.bm == 128 && (pl.tile.bn == 128 || pl.tile.bn == 64) && N >= 64 &&
      p.row_perm == 0 && M % 128 == 0) {
    p.row_perm = 2;
    p.skip_taps = 0;
    p.patch_w = kHaloPW;
    p.patch_hw = kHaloPH * kHaloPW;
    p.patches_per_row = Wo / kHaloPW;
    const dim3 hgrid((unsigned)((M / 128) * cdiv(N, 128)), 1, 1);
    static const int hstages = env_int("PSEG_HCONV_HALO_STAGES", 6);
    if (hstages <= 3) hipLaunchKernelGGL(gather_hh_kernel<3>, hgrid, dim3(512), 0, st, hp);
    else if (hstages == 4) hipLaunchKernelGGL(gather_hh_kernel<4>, hgrid, dim3(512), 0, st, hp);
    else if (hstages == 5) hipLaunchKernelGGL(gather_hh_kernel<5>, hgrid, dim3(512), 0, st, hp);
    else hipLaunchKernelGGL(gather_hh_kernel<6>, hgrid, dim3(512), 0, st, hp);
    PSEG_LAUNCH_CHECK();
    return PSEG_OK;
  }
#endif
#if PSEG_LAB
  // halo-staged 3x3 with the filter as a ring of taps (gather_hr_kernel): unit stride, dilation 1 / 2, channels in whole 64-chunks,
  // maps of 8 x 16 patches, 128x128 / 128x64 plan tiles, every tap live.  PSEG_HCONV_HALO2=0: off.
  static const int halo2_on = env_int("PSEG_HCONV_HALO2", 0);
  if (halo2_on != 0 && bns == nullptr && !generic && taps == 9 && taps_w == 3 && s_out == 1 && s_in == 1 && (adil == 1 || adil == 2) &&
      off0 == -dstep && Cin % 64 == 0 && Ho % kH2PH == 0 && Wo % kH2PW == 0 && Hi == Ho && Wi == Wo && pl.tile.bm == 128 &&
      (pl.tile.bn == 128 || pl.tile.bn == 64) && p.row_perm == 0 && p.skip_taps == 0) {
    p.row_perm = 2;
    p.patch_w = kH2PW;
    p.patch_hw = kH2PH * kH2PW;
    p.patches_per_row = Wo / kH2PW;
    hp.howo_div = FastDiv((uint32_t)p.HoWo);
    hp.wo_div = FastDiv((uint32_t)p.Wo);
    const dim3 hgrid((unsigned)(pl.gridM * pl.gridN), 1, 1);
    if (pl.tile.bn == 128) hipLaunchKernelGGL(gather_hr_kernel<4>, hgrid, dim3(512), 0, st, hp);
    else hipLaunchKernelGGL(gather_hr_kernel<2>, hgrid, dim3(256), 0, st, hp);
    PSEG_LAUNCH_CHECK();
    return PSEG_OK;
  }
#endif
  const dim3 grid((unsigned)(pl.gridM * pl.gridN), 1, 1);
  const bool sk = p.skip_taps != 0;
  // ring depth (PSEG_HCONV_STAGES forces 2 / 3 / 4); never deeper than the K loop is long
  // Measured: the short K-step always wants three stages (its blocks are small: four to eight stay resident anyway); the long
  // one only on narrow tiles with a deep contraction, where a deeper ring costs no resident block the grid needs -- the ASPP
  // convs (128x64 tiles, K = 18432: 237 -> 197 / 208 -> 151 us with three stages) and the classifier (128x32, K = 3456: 238 ->
  // 158 us with four); 128x128 tiles lose their second resident block to a third stage (layer-4 3x3: 89 -> 98 us).
  static const int forced_stages = env_int("PSEG_HCONV_STAGES", 0);
  int stages = 2;
  if (kb == 32) stages = 3;
  else if (pl.tile.bn == 32 && K >= 2048) stages = 4;
  else if (pl.tile.bn == 64 && pl.tile.bm == 128 && K >= 2048) stages = 3;
  if (forced_stages >= 2 && forced_stages <= 4) stages = forced_stages;
  if (pl.kt_total < stages) stages = pl.kt_total < 2 ? 2 : pl.kt_total;
  bool launched = true;
  const int variant = generic ? 2 : (sk ? 1 : 0);
  const int bmode = bns == nullptr ? 0 : (bns_query ? 2 : 1);
  // the persistent kernel (gather_hp_kernel) takes the launches it covers; PSEG_HCONV_PERSIST=0 keeps everything on
  // gather_h_kernel (A/B runs)
  const int persist = cfg().hconv_persist;
  hp.ntiles = pl.gridM * pl.gridN;
  // ... where it pays: SHORT contractions (measured, tools/bench_conv_half.py with PSEG_HCONV_PERSIST=0/1: K <= 1280 -- 2 to 20
  // K-steps per tile -- 5-20 % faster, e.g. 64 -> 256 channels on 128x128 maps 41 -> 35 us = 4.8 TB/s of operand + result
  // traffic; the deep contractions lose 15-25 %: ring + patch leave one resident block per CU where gather_h_kernel holds two,
  // and with 32+ K-steps per tile there is no chain left to hide).  PSEG_HCONV_PERSIST=2 forces it everywhere it is valid.
  const int persist_max_kt = cfg().hconv_persist_kt;
  if (persist != 0 && variant == 0 && !y_f32 && bias == nullptr && !accumulate && (p.row_perm == 0 || p.row_perm == 3) &&
      pl.kt_total >= 1 && (pl.kt_total <= persist_max_kt || persist == 2 || (hp.ntiles <= 256 && pl.kt_total <= 48))) {
    const int pst = stages >= 3 ? 3 : 2;
    bool ok = false;
#define PSEG_HP_LAUNCH(BM_, BN_, WM_, WN_)                                                                    \
  do {                                                                                                         \
    if (kb == 32 && pst == 2) ok = launch_gather_hp<BM_, BN_, WM_, WN_, 32, 2>(hp.ntiles, st, hp, bmode);      \
    else if (kb == 32) ok = launch_gather_hp<BM_, BN_, WM_, WN_, 32, 3>(hp.ntiles, st, hp, bmode);             \
    else if (pst == 2) ok = launch_gather_hp<BM_, BN_, WM_, WN_, 64, 2>(hp.ntiles, st, hp, bmode);             \
    else ok = launch_gather_hp<BM_, BN_, WM_, WN_, 64, 3>(hp.ntiles, st, hp, bmode);                           \
  } while (0)
#if PSEG_LAB
    if (pl.tile.bm == 128 && pl.tile.bn == 128 && pl.hwaves == 4) PSEG_HP_LAUNCH(128, 128, 2, 2);
    else if (pl.tile.bm == 256 && pl.tile.bn == 128 && pl.hwaves == 16) ok = false;
    else if (pl.tile.bm == 256 && pl.tile.bn == 128) PSEG_HP_LAUNCH(256, 128, 4, 2);
    else
#endif
    if (pl.tile.bm == 128 && pl.tile.bn == 128) PSEG_HP_LAUNCH(128, 128, 2, 4);
    else if (pl.tile.bm == 128 && pl.tile.bn == 64) PSEG_HP_LAUNCH(128, 64, 2, 2);
    else if (pl.tile.bm == 64 && pl.tile.bn == 128) PSEG_HP_LAUNCH(64, 128, 2, 2);
    else if (pl.tile.bm == 128 && pl.tile.bn == 32) PSEG_HP_LAUNCH(128, 32, 4, 1);
#undef PSEG_HP_LAUNCH
    if (ok) {
      if (bns_query) return PSEG_OK;
      g_last_conv_kernel = PSEG_KERNEL_GATHER_H_PERSISTENT;
      PSEG_LAUNCH_CHECK();
      return PSEG_OK;
    }
  }
#define PSEG_H_LAUNCH(BM_, BN_, WM_, WN_)                                                                       \
  do {                                                                                                          \
    if (kb == 32 && stages == 2) launched = launch_gather_h<BM_, BN_, WM_, WN_, 32, 2>(variant, grid, st, hp, bmode);  \
    else if (kb == 32 && stages == 3) launched = launch_gather_h<BM_, BN_, WM_, WN_, 32, 3>(variant, grid, st, hp, bmode); \
    else if (kb == 32) launched = launch_gather_h<BM_, BN_, WM_, WN_, 32, 4>(variant, grid, st, hp, bmode);            \
    else if (stages == 2) launched = launch_gather_h<BM_, BN_, WM_, WN_, 64, 2>(variant, grid, st, hp, bmode);         \
    else if (stages == 3) launched = launch_gather_h<BM_, BN_, WM_, WN_, 64, 3>(variant, grid, st, hp, bmode);         \
    else launched = launch_gather_h<BM_, BN_, WM_, WN_, 64, 4>(variant, grid, st, hp, bmode);                          \
  } while (0)
#if PSEG_LAB
  if (pl.tile.bm == 128 && pl.tile.bn == 128 && pl.hwaves == 4) PSEG_H_LAUNCH(128, 128, 2, 2);
  else if (pl.tile.bm == 256 && pl.tile.bn == 128 && pl.hwaves == 16) PSEG_H_LAUNCH(256, 128, 4, 4);
  else if (pl.tile.bm == 256 && pl.tile.bn == 128) PSEG_H_LAUNCH(256, 128, 4, 2);
  else if (pl.tile.bm == 256 && pl.tile.bn == 256) PSEG_H_LAUNCH(256, 256, 2, 4);
  else
#endif
  if (pl.tile.bm == 128 && pl.tile.bn == 128) PSEG_H_LAUNCH(128, 128, 2, 4);
  else if (pl.tile.bm == 128 && pl.tile.bn == 64) PSEG_H_LAUNCH(128, 64, 2, 2);
  else if (pl.tile.bm == 64 && pl.tile.bn == 128) PSEG_H_LAUNCH(64, 128, 2, 2);
  else if (pl.tile.bm == 128 && pl.tile.bn == 32) PSEG_H_LAUNCH(128, 32, 4, 1);
  else launched = false;
#undef PSEG_H_LAUNCH
  if (!launched) {
    if (bns_query) return PSEG_ERR_ARG;
    set_error("conv_h: no kernel for tile %dx%d (K-step %d, %d stages)%s", pl.tile.bm, pl.tile.bn, kb, stages,
              bns != nullptr ? " with the fused BatchNorm-backward sums" : "");
    return PSEG_ERR_ARG;
  }
  if (bns_query) return PSEG_OK;
  g_last_conv_kernel = PSEG_KERNEL_GATHER_H;
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

// tiles the fp16 weight-gradient kernel is instantiated for (rows = Cout, columns = K)
static TileCfg half_wtile(TileCfg t) {
  if (t.bm == 256) t.bm = 128;
  return t;
}

static WgradPlan plan_wgrad_h(long long P, int Cout, int K) {
  // pixel splits for ONE resident block per CU (round 4): every split writes and re-reads a [Cout][K] fp32 slab, and under the
  // half policy that traffic was 40 % of the weight-gradient bytes (2.1 GB written + 2.1 GB read per DeepLabV3+ step for 157 MB
  // of gradients).  Half as many splits halve it; the step got 1.2 % faster with it (15.15 -> 14.97 ms: the blocks share the CUs
  // with the data gradients of the other stream anyway).  PSEG_WGRAD_BPC overrides.
  WgradPlan pl = plan_wgrad(P, Cout, K, false, true, 1);
  pl.tile = half_wtile(pl.tile);
  return pl;
}

static int run_wgrad_h(const void* x, int ldx, const void* dy, int ldy, float* dw, int B, int H, int W, int Cin, int Ho,
                       int Wo, int Cout, int kh, int kw, int stride, int pad, int dil, int accumulate, void* workspace,
                       int64_t workspace_bytes, void* stream, int defer) {
  PSEG_REQUIRE(x && dy && dw, "conv2d_wgrad_h: null pointer");
  PSEG_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "conv2d_wgrad_h: Cin, Cout, ldx, ldy must be multiples of 8");
  PSEG_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)dw & 15) == 0,
               "conv2d_wgrad_h: x / dy / dw must be 16-byte aligned");
  const long long P = (long long)B * Ho * Wo;
  const int K = kh * kw * Cin;
  PSEG_REQUIRE(P > 0 && P < (1LL << 31), "conv2d_wgrad_h: bad pixel count");
  const long long xb = nhwc_bytes_h(B, H, W, Cin, ldx), db = nhwc_bytes_h(B, Ho, Wo, Cout, ldy);
  PSEG_REQUIRE(xb < kMaxBytes && db < kMaxBytes, "conv2d_wgrad_h: tensor exceeds 2 GiB");
  WgradPlan pl = plan_wgrad_h(P, Cout, K);
  HWgradParams hp;
  WgradParams& p = hp.g;
  p.x = reinterpret_cast<const float*>(x);
  p.dy = reinterpret_cast<const float*>(dy);
  p.x_bytes = (uint32_t)xb;
  p.dy_bytes = (uint32_t)db;
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
  const bool can_skip = (dil >= 4 && kh * kw > 1 && Cin % pl.tile.bn == 0 && cfg().conv_noskip == 0);
  p.skip_rows = can_skip ? 1 : 0;
  p.patch_mode = 0;
  p.patch_h = 1;
  p.patch_w = 32;
  if (P % 32 == 0 && ((long long)Ho * Wo) % 32 == 0 && cfg().conv_noskip == 0) {
    DilGeom g{Ho, Wo, H, W, kh, kw, dil, -pad};
    double best = 2.0;
    for (int pw = 32; pw >= 8; pw /= 2) {
      const int ph = 32 / pw;
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
  hp.howo_div = FastDiv((uint32_t)(Ho * Wo));
  hp.wo_div = FastDiv((uint32_t)Wo);
  const long long wsz = (long long)Cout * K;
  if (pl.splits == 1) {
    p.dw = dw;
    p.accumulate = accumulate;
    p.slab_stride = 0;
  } else {
    const long long need = (long long)pl.splits * wsz * 4;
    if (workspace == nullptr || workspace_bytes < need) {
      set_error("conv2d_wgrad_h: needs %lld workspace bytes, got %lld", need, (long long)workspace_bytes);
      return PSEG_ERR_WORKSPACE;
    }
    p.dw = (float*)workspace;
    p.accumulate = 0;
    p.slab_stride = wsz;
  }
  const dim3 grid((unsigned)(pl.gridM * pl.gridN), 1, (unsigned)pl.splits);
  const bool sk = p.skip_rows != 0;
  hipStream_t st = (hipStream_t)stream;
  // ring depth: as deep as keeps the blocks the grid wants resident (measured: tools/bench_conv_half.py, PSEG_HWGRAD_STAGES)
  static const int forced_wst = env_int("PSEG_HWGRAD_STAGES", 0);
  int wst = 2;
  if (forced_wst >= 2 && forced_wst <= 4) wst = forced_wst;
  // pixels per K-step: 32 halves the ring (four blocks per CU instead of two) and pays on the deep 3x3 layers -- ASPP 361 -> 298
  // / 294 -> 238 / 250 -> 216 us, layer-4 3x3 140 -> 111 -- while the short launches (bounded by their slab traffic and
  // their prologue / epilogue) do not care and the narrow classifier loses 10 % (tools/bench_conv_half.py, PSEG_HWGRAD_BKP)
  static const int forced_wkp = env_int("PSEG_HWGRAD_BKP", 0);
  int wkp = (kh * kw > 1 && Cout >= 128 && K >= 4096) ? 32 : 64;
  if (forced_wkp == 32 || forced_wkp == 64) wkp = forced_wkp;
#define PSEG_HW_LAUNCH_S(BM_, BN_, WM_, WN_, ST_, KP_)                                                            \
  do {                                                                                                            \
    if (sk) hipLaunchKernelGGL((wgrad_h_kernel<BM_, BN_, WM_, WN_, true, ST_, KP_>), grid, dim3(256), 0, st, hp); \
    else hipLaunchKernelGGL((wgrad_h_kernel<BM_, BN_, WM_, WN_, false, ST_, KP_>), grid, dim3(256), 0, st, hp);   \
  } while (0)
#define PSEG_HW_LAUNCH(BM_, BN_, WM_, WN_)                                       \
  do {                                                                           \
    if (wkp == 64 && wst == 2) PSEG_HW_LAUNCH_S(BM_, BN_, WM_, WN_, 2, 64);       \
    else if (wkp == 64) PSEG_HW_LAUNCH_S(BM_, BN_, WM_, WN_, 3, 64);              \
    else if (wst == 2) PSEG_HW_LAUNCH_S(BM_, BN_, WM_, WN_, 2, 32);               \
    else if (wst == 3) PSEG_HW_LAUNCH_S(BM_, BN_, WM_, WN_, 3, 32);               \
    else PSEG_HW_LAUNCH_S(BM_, BN_, WM_, WN_, 4, 32);                             \
  } while (0)
  if (pl.tile.bm == 128 && pl.tile.bn == 128) PSEG_HW_LAUNCH(128, 128, 2, 2);
  else if (pl.tile.bm == 128 && pl.tile.bn == 64) PSEG_HW_LAUNCH(128, 64, 2, 2);
  else if (pl.tile.bm == 128 && pl.tile.bn == 32) PSEG_HW_LAUNCH(128, 32, 4, 1);
  else if (pl.tile.bm == 64 && pl.tile.bn == 128) PSEG_HW_LAUNCH(64, 128, 2, 2);
  else if (pl.tile.bm == 32 && pl.tile.bn == 128) PSEG_HW_LAUNCH(32, 128, 1, 4);
  else {
    set_error("conv2d_wgrad_h: no kernel for tile %dx%d", pl.tile.bm, pl.tile.bn);
    return PSEG_ERR_ARG;
  }
#undef PSEG_HW_LAUNCH
#undef PSEG_HW_LAUNCH_S
  PSEG_LAUNCH_CHECK();
  if (pl.splits > 1 && !defer)
    return launch_slab_reduce((const float*)workspace, wsz, pl.splits, dw, K, (long long)Cout, K, nullptr, accumulate, st);
  return PSEG_OK;
}

}  // namespace pseg

using namespace pseg;

extern "C" {

static FwdPlan plan_fwd_stats_h(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil) {
  const long long M = (long long)B * Ho * Wo;
  const int H = (Ho - 1) * stride - 2 * pad + dil * (kh - 1) + 1, W = (Wo - 1) * stride - 2 * pad + dil * (kw - 1) + 1;
  DilGeom geom;
  const bool has_geom = Cin % hconv_kb(Cin, kh * kw * Cin) == 0 && dil_geom(geom, Ho, Wo, H, W, kh, kw, Cin, stride, 1, dil, -pad);
  return plan_gather_h(M, Cout, kh * kw * Cin, Cin, has_geom ? &geom : nullptr);
}

int pseg_conv2d_stat_rows_h(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil) {
  FwdPlan pl = plan_fwd_stats_h(B, Ho, Wo, Cin, Cout, kh, kw, stride, pad, dil);
  return pl.gridM * waves_m_h(pl.tile);
}

int pseg_conv2d_stat_group_h(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil) {
  FwdPlan pl = plan_fwd_stats_h(B, Ho, Wo, Cin, Cout, kh, kw, stride, pad, dil);
  return pl.tile.bm / waves_m_h(pl.tile);
}

int pseg_conv2d_fwd_h(const pseg_half_t* x, int ldx, const pseg_half_t* w, const float* bias, void* y, int ldy, int y_is_f32,
                      int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int dil,
                      int accumulate, float* stat, void* stream) {
  PSEG_REQUIRE(x && w && y, "conv2d_fwd_h: null pointer");
  PSEG_REQUIRE(stride >= 1 && dil >= 1 && pad >= 0 && kh >= 1 && kw >= 1, "conv2d_fwd_h: bad geometry");
  PSEG_REQUIRE(Ho == (H + 2 * pad - dil * (kh - 1) - 1) / stride + 1 && Wo == (W + 2 * pad - dil * (kw - 1) - 1) / stride + 1,
               "conv2d_fwd_h: Ho/Wo (%d,%d) inconsistent with H/W (%d,%d) k=%dx%d s=%d p=%d d=%d", Ho, Wo, H, W, kh, kw,
               stride, pad, dil);
  return run_gather_h(x, nhwc_bytes_h(B, H, W, Cin, ldx), ldx, w, y, ldy, y_is_f32, bias, stat, B, H, W, Cin, Ho, Wo, Cout,
                      kw, kh * kw * Cin, stride, 1, dil, -pad, accumulate, (hipStream_t)stream);
}

int pseg_conv2d_dgrad_h(const pseg_half_t* dy, int ldy, const pseg_half_t* wT, pseg_half_t* dx, int ldx, int B, int H, int W,
                        int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int dil, int accumulate,
                        void* stream) {
  PSEG_REQUIRE(dy && wT && dx, "conv2d_dgrad_h: null pointer");
  PSEG_REQUIRE(stride >= 1 && dil >= 1 && pad >= 0, "conv2d_dgrad_h: bad geometry");
  // GEMM rows = input pixels (B,H,W); contraction over (r,s,co); gather source = dy [B,Ho,Wo,Cout]
  return run_gather_h(dy, nhwc_bytes_h(B, Ho, Wo, Cout, ldy), ldy, wT, dx, ldx, 0, nullptr, nullptr, B, Ho, Wo, Cout, H, W,
                      Cin, kw, kh * kw * Cout, 1, stride, -dil, pad, accumulate, (hipStream_t)stream);
}

int pseg_conv2d_dgrad_bnstat_rows_h(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad,
                                    int dil) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 || Cin % 8 != 0 || Cout % 8 != 0 || stride < 1 ||
      dil < 1 || pad < 0)
    return 0;
  const long long M = (long long)B * H * W;
  if (M >= (1LL << 31) || nhwc_bytes_h(B, Ho, Wo, Cout, Cout) >= kMaxBytes || (long long)Cin * kh * kw * Cout * 2 >= kMaxBytes) return 0;
  const FwdPlan pl = plan_run_h(Ho, Wo, Cout, H, W, M, Cin, kw, kh * kw * Cout, 1, stride, -dil, pad);
  const int rows = pl.gridM * waves_m_h(pl.tile);
  // walk the launch decisions of this problem without launching: is the kernel it lands on instantiated with the sums?
  static float dummy[4] __attribute__((aligned(16)));
  const HBnsArgs probe{dummy, Cin, dummy, dummy, dummy, dummy, 0, dummy, dummy, rows};
  const int rc = run_gather_h(dummy, nhwc_bytes_h(B, Ho, Wo, Cout, Cout), Cout, dummy, dummy, Cin, 0, nullptr, nullptr, B, Ho, Wo, Cout,
                              H, W, Cin, kw, kh * kw * Cout, 1, stride, -dil, pad, 0, nullptr, &probe, true);
  return rc == PSEG_OK ? rows : 0;
}

int pseg_conv2d_dgrad_bnstat_h(const pseg_half_t* dy, int ldy, const pseg_half_t* wT, pseg_half_t* dx, int ldx, int B, int H, int W,
                               int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int dil,
                               const pseg_half_t* y_prev, int ldy_prev, const float* mean, const float* invstd, const float* scale,
                               const float* shift, int act, float* part_db, float* part_dg, int part_rows, void* stream) {
  PSEG_REQUIRE(dy && wT && dx && y_prev && mean && invstd && scale && shift && part_db && part_dg,
               "conv2d_dgrad_bnstat_h: null pointer");
  PSEG_REQUIRE(stride >= 1 && dil >= 1 && pad >= 0, "conv2d_dgrad_bnstat_h: bad geometry");
  PSEG_REQUIRE(act == PSEG_ACT_NONE || act == PSEG_ACT_RELU || act == PSEG_ACT_RELU6, "conv2d_dgrad_bnstat_h: unknown activation");
  PSEG_REQUIRE(ldy_prev % 8 == 0 && ldy_prev >= Cin && ((uintptr_t)y_prev & 15) == 0 && ((uintptr_t)part_db & 15) == 0 &&
                   ((uintptr_t)part_dg & 15) == 0 && ((uintptr_t)mean & 15) == 0 && ((uintptr_t)invstd & 15) == 0 &&
                   ((uintptr_t)scale & 15) == 0 && ((uintptr_t)shift & 15) == 0,
               "conv2d_dgrad_bnstat_h: y_prev / coefficient / partial pointers must be 16-byte aligned, ldy_prev %% 8 == 0");
  PSEG_REQUIRE(part_rows > 0 && part_rows == pseg_conv2d_dgrad_bnstat_rows_h(B, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, dil),
               "conv2d_dgrad_bnstat_h: part_rows (%d) is not pseg_conv2d_dgrad_bnstat_rows_h() of this problem", part_rows);
  const HBnsArgs bns{y_prev, ldy_prev, mean, invstd, scale, shift, act, part_db, part_dg, part_rows};
  return run_gather_h(dy, nhwc_bytes_h(B, Ho, Wo, Cout, ldy), ldy, wT, dx, ldx, 0, nullptr, nullptr, B, Ho, Wo, Cout, H, W, Cin, kw,
                      kh * kw * Cout, 1, stride, -dil, pad, 0, (hipStream_t)stream, &bns);
}

int64_t pseg_conv2d_wgrad_workspace_bytes_h(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw) {
  const WgradPlan a = plan_wgrad_h((long long)B * Ho * Wo, Cout, kh * kw * Cin);
  return a.splits > 1 ? (int64_t)a.splits * Cout * kh * kw * Cin * 4 : 0;
}

int pseg_conv2d_wgrad_h(const pseg_half_t* x, int ldx, const pseg_half_t* dy, int ldy, float* dw, int B, int H, int W,
                        int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int dil, int accumulate,
                        void* workspace, int64_t workspace_bytes, void* stream) {
  return run_wgrad_h(x, ldx, dy, ldy, dw, B, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, dil, accumulate, workspace,
                     workspace_bytes, stream, 0);
}

int pseg_conv2d_wgrad_splits_h(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw) {
  if (B <= 0 || Ho <= 0 || Wo <= 0 || Cin <= 0 || Cout <= 0) return 0;
  return plan_wgrad_h((long long)B * Ho * Wo, Cout, kh * kw * Cin).splits;
}

int pseg_conv2d_wgrad_slabs_h(const pseg_half_t* x, int ldx, const pseg_half_t* dy, int ldy, float* slabs, int B, int H,
                              int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int dil,
                              int64_t slab_bytes, void* stream) {
  PSEG_REQUIRE(pseg_conv2d_wgrad_splits_h(B, Ho, Wo, Cin, Cout, kh, kw) > 1,
               "conv2d_wgrad_slabs_h: this plan does not split -- call pseg_conv2d_wgrad_h");
  return run_wgrad_h(x, ldx, dy, ldy, slabs, B, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, pad, dil, 0, slabs, slab_bytes,
                     stream, 1);
}

int pseg_filter_prepare_h(const int64_t* jobs, int n, int64_t total_tiles, void* stream) {
  PSEG_REQUIRE(jobs && n > 0 && total_tiles > 0 && total_tiles < (1LL << 31), "filter_prepare_h: bad argument");
  hipLaunchKernelGGL(filter_prepare_h_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(jobs), n);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

int pseg_convert2d(const void* x, int x_is_half, int ldx, void* y, int y_is_half, int ldy, int64_t M, int C,
                   const float* dev_scale, void* stream) {
  PSEG_REQUIRE(x && y && M > 0 && C > 0 && C % 4 == 0, "convert2d: need M > 0, C %% 4 == 0");
  PSEG_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C, "convert2d: ld %% 4 == 0, ld >= C");
  PSEG_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 7) == 0, "convert2d: alignment");
  PSEG_REQUIRE((long long)M * (C / 4) < (1LL << 31), "convert2d: tensor too large");
  const uint32_t total = (uint32_t)(M * (C / 4));
  long long blocks = ((long long)total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const FastDiv c4((uint32_t)(C / 4));
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)blocks), block(256);
  if (x_is_half && y_is_half)
    hipLaunchKernelGGL((convert2d_kernel<half_t, half_t>), grid, block, 0, st, (const half_t*)x, ldx, (half_t*)y, ldy, total, c4, dev_scale);
  else if (x_is_half)
    hipLaunchKernelGGL((convert2d_kernel<half_t, float>), grid, block, 0, st, (const half_t*)x, ldx, (float*)y, ldy, total, c4, dev_scale);
  else if (y_is_half)
    hipLaunchKernelGGL((convert2d_kernel<float, half_t>), grid, block, 0, st, (const float*)x, ldx, (half_t*)y, ldy, total, c4, dev_scale);
  else
    hipLaunchKernelGGL((convert2d_kernel<float, float>), grid, block, 0, st, (const float*)x, ldx, (float*)y, ldy, total, c4, dev_scale);
  PSEG_LAUNCH_CHECK();
  return PSEG_OK;
}

}  // extern "C"
