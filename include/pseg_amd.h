/* pseg_amd.h -- C ABI of libpseg_amd.so: the MI355X (gfx950) hot path of
 * WoodsGao/pytorch_segmentation, hand-written HIP.
 *
 * The reference has no native layer: its hot path is stock torch ops dispatched from
 * Python (conv2d / batch_norm / relu / interpolate / cross_entropy / max).  Each entry
 * point below names the reference call site (file:line under the reference tree) whose
 * arithmetic it replaces.  A reference maintainer binds these with ctypes (see
 * INTEGRATION.md); pytorch_segmentation_amd/_lib.py is exactly that binding.
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer owned by the caller (PyTorch's caching allocator
 *    in practice); the library never allocates or frees device memory.  Scratch is passed
 *    in as (workspace, workspace_bytes); pseg_*_workspace_bytes() gives the size.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All work is
 *    enqueued on it; nothing here synchronises the host, so calls are graph-capturable.
 *  - Activations are fp32 NHWC: element (b,h,w,c) of a tensor with pixel stride `ld`
 *    (in floats, ld >= C, ld % 4 == 0, base 16-byte aligned) lives at
 *    ((b*H + h)*W + w)*ld + c.  A channel slice of a wider concat buffer is simply a
 *    base pointer + ld of the wide buffer: concatenation costs no copy.
 *  - Conv weights are [Cout][kh][kw][Cin] (= a torch OIHW tensor in channels_last memory
 *    format), Cin % 4 == 0.  Logits / targets for the loss are NCHW fp32 / NHW int64, as
 *    the reference's nn.CrossEntropyLoss sees them.
 *  - Every tensor must be smaller than 2 GiB (32-bit buffer-descriptor addressing).
 *  - Return value: 0 on success, negative on error (PSEG_ERR_*); pseg_last_error() gives
 *    the message for the calling thread.  Nothing aborts the process.
 *  - Thread safety: functions are re-entrant; they keep no global state besides the
 *    thread-local error string (autograd calls backward from its own worker thread).
 */
#ifndef PSEG_AMD_H
#define PSEG_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSEG_OK 0
#define PSEG_ERR_ARG (-1)
#define PSEG_ERR_HIP (-2)
#define PSEG_ERR_WORKSPACE (-3)

/* conv arithmetic (all accumulate in fp32):
 *  FP32    exact fp32 products on v_mfma_f32_32x32x2_f32;
 *  BF16X3  each fp32 operand split into two bf16 limbs, a*b = ah*bh + ah*bl + al*bh on v_mfma_f32_32x32x16_bf16
 *          (~17 significant bits per product);
 *  BF16X6  three bf16 limbs (exactly the 24 mantissa bits) and the six partial products down to 2^-16:
 *          error ~2^-23 per product, i.e. fp32-equivalent results at 6/16 of the fp32-MFMA instruction time. */
#define PSEG_PREC_FP32 0
#define PSEG_PREC_BF16X3 1
#define PSEG_PREC_BF16X6 2
/*  FP16X3  two fp16 limbs (22-bit operands) of the operand scaled by an exact power of two taken from its per-tensor
 *          max|x| (`amax_*`: device pointer to one float, maintained with pseg_amax / the producer kernels);
 *          ~2^-22 per product at the BF16X3 cost.  fwd / dgrad only. */
#define PSEG_PREC_FP16X3 3

#define PSEG_ACT_NONE 0
#define PSEG_ACT_RELU 1
#define PSEG_ACT_RELU6 2

/* bumped whenever an existing prototype changes incompatibly; pseg_abi_version() returns the value the library was built with */
#define PSEG_ABI_VERSION 11
int pseg_abi_version(void);
const char* pseg_last_error(void);
/* The PSEG_CONV_* / PSEG_WGRAD_* planning overrides are read from the environment once, at the first launch;
 * call this after changing them at run time (tests do). */
int pseg_config_reload(void);

/* ------------------------------------------------------------------ convolution
 * Replaces torch conv2d as used by ConvNormAct / nn.Conv2d on the hot path:
 * models/aspp.py:12,27-30  models/deeplabv3plus.py:20,22  models/unet.py:19-23
 * (padding = (k-1)/2*dilation for ConvNormAct, 1 for the 3x3 cls_conv).
 * Implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate).
 *
 * pseg_conv2d_fwd:  y[b,ho,wo,:] = sum_{r,s,ci} x[b,ho*stride-pad+r*dil, wo*stride-pad+s*dil, ci] * w[:,r,s,ci] (+bias)
 *   stat (nullable): column statistics of y per row group, float [3][rows][Cout] with
 *   rows = pseg_conv2d_stat_rows(...), each group covering pseg_conv2d_stat_group(...) consecutive
 *   GEMM rows (= pixels in the kernel's own order: row-major, or patch-major for dilated convs): [0] = pivot K (the group's first sample), [1] = sum(y-K), [2] = sum((y-K)^2).
 *   Shifted sums keep the variance exact to fp32 even when |mean| >> std; consumed by
 *   pseg_bn_finalize (fuses the BatchNorm batch-statistics pass into the conv epilogue).
 *   accumulate != 0: y += result.
 */
int pseg_conv2d_fwd(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                    int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw,
                    int stride, int pad, int dil, int accumulate, int precision, const float* amax_x,
                    const float* amax_w, float* stat, void* workspace, int64_t workspace_bytes, void* stream);
int pseg_conv2d_stat_rows(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil);
int pseg_conv2d_stat_group(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil);
int64_t pseg_conv2d_fwd_workspace_bytes(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw);

/* dgrad: dx[b,h,w,ci] = sum_{r,s,co} dy[b,(h+pad-r*dil)/stride,(w+pad-s*dil)/stride,co] * w[co,r,s,ci]
 * wT is the transposed filter [Cin][kh][kw][Cout] made by pseg_filter_transpose.
 * accumulate != 0: dx += result (merges the two gradient paths of a residual block). */
int pseg_conv2d_dgrad(const float* dy, int ldy, const float* wT, float* dx, int ldx,
                      int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw,
                      int stride, int pad, int dil, int accumulate, int precision, const float* amax_dy,
                      const float* amax_w, void* workspace, int64_t workspace_bytes, void* stream);
/* The same data gradient (exact fp32, no accumulate) when dx is the gradient dz of the BatchNorm + activation layer that produced
 * the conv's input -- conv -> BN -> ReLU -> THIS conv, the interior of every ConvNormAct chain and ResNet bottleneck (reference
 * models/aspp.py:27-30, models/unet.py:19-23 via pytorch_modules' ConvNormAct; one consumer, no residual).  While the tile is
 * stored, the epilogue reads that layer's saved conv output y_prev at the same pixels, recomputes the activation mask from it
 * exactly as pseg_bn_act_fwd applied it ((y - mean) * scale + shift against 0 / 6), and writes the per-row-group partial sums
 * part_db[g][c] = sum dz * act', part_dg[g][c] = sum dz * act' * (y - mean) * invstd -- what pseg_bn_act_bwd_reduce would re-read
 * dz and y for: one activation-sized read less per layer, one launch fewer on the backward chain.  part_db / part_dg are
 * [part_rows][Cin] floats, fed to pseg_bn_bwd_finalize as they are.  pseg_conv2d_dgrad_bnstat_rows(...) -> part_rows, or 0 when
 * this problem does not run on the kernel that carries the sums (channel counts not multiples of 32, split-K plans): the caller
 * then uses pseg_conv2d_dgrad + pseg_bn_act_bwd_reduce. */
int pseg_conv2d_dgrad_bnstat_rows(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad,
                                  int dil);
int pseg_conv2d_dgrad_bnstat(const float* dy, int ldy, const float* wT, float* dx, int ldx, int B, int H, int W, int Cin, int Ho,
                             int Wo, int Cout, int kh, int kw, int stride, int pad, int dil, const float* y_prev, int ldy_prev,
                             const float* mean, const float* invstd, const float* scale, const float* shift, int act,
                             float* part_db, float* part_dg, int part_rows, void* stream);
int pseg_filter_transpose(const float* w, float* wT, int Cout, int taps, int Cin, void* stream);
/* BF16X3 data gradient on PRE-SPLIT operands (the fast form of the split-bf16 arithmetic: the consumer neither splits nor
 * stages through registers -- its tiles go global -> LDS by DMA).  pseg_split_planes writes the two bf16 limb planes of an
 * fp32 [M][C] tensor (hi = bf16(x), lo = bf16(x - hi); [M][ldp] uint16 each, ldp % 8 == 0, columns >= C zeroed); results
 * are bit-identical to pseg_conv2d_dgrad(..., PSEG_PREC_BF16X3) up to the accumulation grouping.
 * pseg_conv2d_dgrad_planes_ok(...) != 0 says whether the shape is covered (dy channels % 32 == 0, Cin >= 128, enough
 * 256x128 tiles to fill the chip); otherwise call pseg_conv2d_dgrad. */
int pseg_split_planes(const float* x, int ldx, int64_t M, int C, uint16_t* hi, uint16_t* lo, int ldp, void* stream);
int pseg_conv2d_dgrad_planes_ok(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride,
                                int pad, int dil);
int pseg_conv2d_dgrad_planes(const uint16_t* dy_hi, const uint16_t* dy_lo, int ldp, const uint16_t* wT_hi,
                             const uint16_t* wT_lo, float* dx, int ldx, int B, int H, int W, int Cin, int Ho, int Wo,
                             int Cout, int kh, int kw, int stride, int pad, int dil, int accumulate, void* stream);
/* every filter of a model in one launch.  jobs: device array of n records of six int64
 * {w (device address), wT (device address), Cout, taps, Cin, index of the record's first 32x32 tile}, tile indices
 * ascending from 0; a record covers taps * ceil(Cout/32) * ceil(Cin/32) tiles; total_tiles = their sum. */
int pseg_filter_transpose_batch(const int64_t* jobs, int n, int64_t total_tiles, void* stream);

/* wgrad: dw[co,r,s,ci] = sum_{b,ho,wo} dy[b,ho,wo,co] * x[b,ho*stride-pad+r*dil, wo*stride-pad+s*dil, ci]
 * Split over pixels into workspace slabs that are reduced in a fixed order (bit-reproducible).
 * accumulate != 0: dw += result (gradient accumulation over micro-batches, train.py --accumulate). */
int pseg_conv2d_wgrad(const float* x, int ldx, const float* dy, int ldy, float* dw,
                      int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw,
                      int stride, int pad, int dil, int accumulate, int precision, int concurrent,
                      void* workspace, int64_t workspace_bytes, void* stream);
/* concurrent != 0 (here and in pseg_conv2d_wgrad_splits / _slabs; the three must agree): the launch runs beside another
 * stream's kernels -- the training step puts its weight gradients on a second stream next to the data gradients -- and the
 * exact-fp32 plan then splits the pixels for ONE resident block per CU (half the slabs: 2.2 GB less written and re-read per
 * DeepLabV3+ step); 0: the launch has the chip to itself and gets two.  The workspace size covers both. */
int64_t pseg_conv2d_wgrad_workspace_bytes(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw);
/* The same weight gradient with the slab reduction DEFERRED: a training step runs ~60 (DeepLabV3+) to ~300 (HRNet) split
 * weight gradients, and their reductions are launch-bound one by one (23 us each for 37 MB on average).
 * pseg_conv2d_wgrad_splits: slab count of the plan (1: not split -- call pseg_conv2d_wgrad).  pseg_conv2d_wgrad_slabs
 * leaves the `splits` partial gradients in `slabs` ([splits][Cout][kh][kw][Cin] floats, slab_bytes >= that; the caller
 * keeps one such region per layer alive until the reduction).  pseg_slab_reduce_batch then reduces every layer of the
 * backward pass in ONE launch, slabs in a fixed order (bit-reproducible): jobs = device array of n records of five int64
 * {slabs (device address), dw (device address), elements per slab (% 4 == 0), slab count, index of the record's first
 * block}, block indices ascending from 0, a record covers ceil(elements / pseg_slab_reduce_block()) blocks, total_blocks
 * = their sum.  accumulate != 0: dw += sum (train.py --accumulate). */
int pseg_conv2d_wgrad_splits(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw, int precision, int concurrent);
int pseg_conv2d_wgrad_slabs(const float* x, int ldx, const float* dy, int ldy, float* slabs, int B, int H, int W, int Cin,
                            int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int dil, int precision,
                            int concurrent, int64_t slab_bytes, void* stream);
int pseg_slab_reduce_batch(const int64_t* jobs, int n, int64_t total_blocks, int accumulate, void* stream);
int pseg_slab_reduce_block(void);

/* ------------------------------------------------------------------ half-precision (`-mp`) convolution
 * The reference's mixed-precision mode (train.py:70 `-mp`, :102-105, :138; README.md:12 -- apex: fp16 compute, fp32 master
 * weights, loss scaling).  Activations, their gradients and the filters are IEEE binary16 in memory (pseg_half_t = the bit
 * pattern), element strides as in the fp32 entry points, C % 8 == 0, ld % 8 == 0, 16-byte aligned bases; products are exact
 * (fp16 x fp16 on v_mfma_f32_32x32x16_f16) and every sum is fp32.
 * pseg_conv2d_fwd_h: y fp16, or fp32 when y_is_f32 (the class logits, which the loss reads in fp32); stat as pseg_conv2d_fwd
 *   with the row grouping of pseg_conv2d_stat_rows_h / _group_h (taken from the fp32 accumulators).  bias is fp32.
 * pseg_conv2d_dgrad_h: wT = the transposed fp16 filter [Cin][kh][kw][Cout].
 * pseg_conv2d_wgrad_h: dw fp32 [Cout][kh][kw][Cin] (the master gradient arena); split / slab protocol as pseg_conv2d_wgrad.
 * pseg_filter_prepare_h: every dense filter of a model in one launch, from the fp32 master weights: the fp16 copy AND the
 *   transposed fp16 copy, both optionally padded further than the master (8-channel granules; padding zero-filled).  jobs:
 *   device array of n records of nine int64 {w fp32 [Cout][taps][Cin], w_h [CoutP][taps][CinP], wT_h [CinP][taps][CoutP]
 *   (device addresses; either output may be 0), Cout, taps, Cin, CoutP, CinP, index of the record's first 32x32 tile}; a
 *   record covers taps * ceil(CoutP/32) * ceil(CinP/32) tiles, tile indices ascending from 0.
 * pseg_convert2d: y[M][C] = convert(x[M][C] * scale) between fp32 / fp16 tensors (x_is_half / y_is_half), scale = *dev_scale
 *   when non-NULL (a device scalar: the dynamic loss scale multiplies the loss gradient on its way into fp16). */
typedef uint16_t pseg_half_t;
int pseg_conv2d_fwd_h(const pseg_half_t* x, int ldx, const pseg_half_t* w, const float* bias, void* y, int ldy, int y_is_f32,
                      int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int dil,
                      int accumulate, float* stat, void* stream);
int pseg_conv2d_stat_rows_h(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil);
int pseg_conv2d_stat_group_h(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil);
int pseg_conv2d_dgrad_h(const pseg_half_t* dy, int ldy, const pseg_half_t* wT, pseg_half_t* dx, int ldx, int B, int H, int W,
                        int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int dil, int accumulate,
                        void* stream);
/* pseg_conv2d_dgrad_bnstat_h: pseg_conv2d_dgrad_bnstat for the fp16 tensors (dx overwritten; y_prev fp16 with ldy_prev % 8 == 0;
 * the sums are taken over dx AS STORED, i.e. rounded to fp16, like pseg_bn_act_bwd_reduce_h reading it back).  Every fp16 gather
 * kernel carries the epilogue: pseg_conv2d_dgrad_bnstat_rows_h is 0 only for an invalid problem. */
int pseg_conv2d_dgrad_bnstat_rows_h(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad,
                                    int dil);
int pseg_conv2d_dgrad_bnstat_h(const pseg_half_t* dy, int ldy, const pseg_half_t* wT, pseg_half_t* dx, int ldx, int B, int H, int W,
                               int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int dil,
                               const pseg_half_t* y_prev, int ldy_prev, const float* mean, const float* invstd, const float* scale,
                               const float* shift, int act, float* part_db, float* part_dg, int part_rows, void* stream);
int pseg_conv2d_wgrad_h(const pseg_half_t* x, int ldx, const pseg_half_t* dy, int ldy, float* dw, int B, int H, int W,
                        int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int dil, int accumulate,
                        void* workspace, int64_t workspace_bytes, void* stream);
int64_t pseg_conv2d_wgrad_workspace_bytes_h(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw);
int pseg_conv2d_wgrad_splits_h(int B, int Ho, int Wo, int Cin, int Cout, int kh, int kw);
int pseg_conv2d_wgrad_slabs_h(const pseg_half_t* x, int ldx, const pseg_half_t* dy, int ldy, float* slabs, int B, int H,
                              int W, int Cin, int Ho, int Wo, int Cout, int kh, int kw, int stride, int pad, int dil,
                              int64_t slab_bytes, void* stream);
int pseg_filter_prepare_h(const int64_t* jobs, int n, int64_t total_tiles, void* stream);
int pseg_convert2d(const void* x, int x_is_half, int ldx, void* y, int y_is_half, int ldy, int64_t M, int C,
                   const float* dev_scale, void* stream);

/* fp16-storage variants of the bandwidth-bound passes: the same kernels instantiated on 2-byte activations (arguments and
 * semantics of the entry point without the suffix; statistics, per-channel vectors, filters of the depthwise convs and
 * every sum stay fp32; the amax / limb-plane arguments of the fp32 forms do not exist here).  Under `-mp` these halve the
 * bytes of the passes that are 40 % of a step's HBM traffic. */
int pseg_col_stats_h(const pseg_half_t* y, int ldy, int64_t M, int C, float* stat, void* stream);
int pseg_bn_act_fwd_h(const pseg_half_t* y, int ldy, const float* mean, const float* scale, const float* shift,
                      const pseg_half_t* residual, int ldr, int act, pseg_half_t* z, int ldz, int64_t M, int C,
                      uint32_t* mask_out, void* stream);
int pseg_bn_fwd_fused_h(const float* stat, int rows, int group, int64_t count, int C, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                        float* scale, float* shift, const pseg_half_t* y, int ldy, const pseg_half_t* residual, int ldr,
                        int act, pseg_half_t* z, int ldz, int64_t M, void* stream);
int pseg_bn_bwd_fused_h(const float* part_db, const float* part_dg, int rows, int64_t count, int C, float* dgamma,
                        float* dbeta, int accumulate, int frozen, const pseg_half_t* dz, int lddz, const pseg_half_t* z,
                        int ldz, const pseg_half_t* y, int ldy, const float* mean, const float* invstd, const float* scale,
                        const float* shift, int act, pseg_half_t* dy, int lddy, pseg_half_t* dres, int lddres,
                        int res_accumulate, int64_t M, void* stream);
int pseg_bn_act_bwd_reduce_h(const pseg_half_t* dz, int lddz, const pseg_half_t* z, int ldz, const pseg_half_t* y, int ldy,
                             const float* mean, const float* invstd, const float* scale, const float* shift, int act,
                             int64_t M, int C, float* part_db, float* part_dg, const uint32_t* mask, void* stream);
int pseg_bn_act_bwd_apply_h(const pseg_half_t* dz, int lddz, const pseg_half_t* z, int ldz, const pseg_half_t* y, int ldy,
                            const float* mean, const float* invstd, const float* scale, const float* shift, const float* c1,
                            const float* c2, int act, pseg_half_t* dy, int lddy, pseg_half_t* dres, int lddres,
                            int res_accumulate, int64_t M, int C, const uint32_t* mask, void* stream);
int pseg_act_bwd_h(const pseg_half_t* dz, int lddz, const pseg_half_t* z, int ldz, const float* scale, int act,
                   pseg_half_t* dy, int lddy, pseg_half_t* dres, int lddres, int res_accumulate, int64_t M, int C,
                   void* stream);
int pseg_col_sum_h(const pseg_half_t* dy, int ldy, int64_t M, int C, float* out, int accumulate, void* workspace,
                   int64_t workspace_bytes, void* stream);
int pseg_copy2d_h(const pseg_half_t* x, int ldx, pseg_half_t* y, int ldy, int64_t M, int C, int accumulate, void* stream);
int pseg_pool_sum_h(const pseg_half_t* x, int ldx, int B, int HW, int C, float scale, pseg_half_t* out, int ldo,
                    void* stream);
int pseg_broadcast_h(const pseg_half_t* x, int ldx, int B, int HW, int C, float scale, pseg_half_t* y, int ldy,
                     int accumulate, void* stream);
/* NHWC -> NHWC only (the NCHW forms serve the fp32 logits) */
int pseg_bilinear_fwd_h(const pseg_half_t* x, int ldx, int B, int Hi, int Wi, int C, pseg_half_t* y, int ldy, int Ho, int Wo,
                        int align_corners, void* stream);
int pseg_bilinear_bwd_h(const pseg_half_t* dy, int ldy, int B, int Hi, int Wi, int C, pseg_half_t* dx, int ldx, int Ho,
                        int Wo, int align_corners, int accumulate, void* stream);
int pseg_maxpool_fwd_h(const pseg_half_t* x, int ldx, int B, int H, int W, int C, pseg_half_t* y, int ldy, uint8_t* argmax,
                       int Ho, int Wo, int k, int stride, int pad, void* stream);
int pseg_maxpool_bwd_h(const pseg_half_t* dy, int ldy, const uint8_t* argmax, int B, int H, int W, int C, pseg_half_t* dx,
                       int ldx, int Ho, int Wo, int k, int stride, int pad, int accumulate, void* stream);
int pseg_dwconv_fwd_h(const pseg_half_t* x, int ldx, const float* w, pseg_half_t* y, int ldy, int B, int H, int W, int C,
                      int Ho, int Wo, int k, int stride, int pad, void* stream);
int pseg_dwconv_dgrad_h(const pseg_half_t* dy, int ldy, const float* w, pseg_half_t* dx, int ldx, int B, int H, int W, int C,
                        int Ho, int Wo, int k, int stride, int pad, void* stream);
int pseg_dwconv_wgrad_h(const pseg_half_t* x, int ldx, const pseg_half_t* dy, int ldy, float* dw, int B, int H, int W, int C,
                        int Ho, int Wo, int k, int stride, int pad, int accumulate, void* workspace, int64_t workspace_bytes,
                        void* stream);

/* ---- dynamic loss scaling + master-weight optimiser of the `-mp` path (apex O1/O2 semantics: train.py:102-105).
 * `state` = 8 floats on the device: [0] loss scale S, [1] 1/S, [2] clean steps since S last changed, [3] found-inf flag of
 * the step in flight, [4] optimiser steps applied, [5] steps skipped.  Nothing here synchronises the host.
 *   pseg_mp_state_init: S = init_scale, everything else 0.
 *   pseg_mp_check: raises state[3] when any of the n gradient floats is inf / nan (call after the gradient all-reduce, so
 *     every rank decides alike).
 *   pseg_sgd_step_mp / pseg_adam_step_mp: the steps of pseg_sgd_step / pseg_adam_step with grad_scale * (1/S) and -- when
 *     state[3] is raised -- no update at all; first-step / bias-correction counts come from state[4].
 *   pseg_mp_update: after a flagged step S *= backoff_factor (>= min_scale), after growth_interval clean ones S *= growth_factor
 *     (<= max_scale); clears the flag and counts the step as applied or skipped. */
int pseg_mp_state_init(float* state, float init_scale, void* stream);
int pseg_mp_check(const float* grad, int64_t n, float* state, void* stream);
int pseg_mp_update(float* state, float growth_factor, float backoff_factor, int growth_interval, float min_scale,
                   float max_scale, void* stream);
int pseg_sgd_step_mp(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum,
                     float weight_decay, int nesterov, float grad_scale, const float* state, void* stream);
int pseg_adam_step_mp(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                      float beta2, float eps, float weight_decay, int decoupled, float grad_scale, const float* state,
                      void* stream);

/* depthwise 3x3 (MobileNetV2 encoder of models/unet.py:16-17); w is [kh][kw][C]. */
int pseg_dwconv_fwd(const float* x, int ldx, const float* w, float* y, int ldy, int B, int H, int W, int C,
                    int Ho, int Wo, int k, int stride, int pad, void* stream);
int pseg_dwconv_dgrad(const float* dy, int ldy, const float* w, float* dx, int ldx, int B, int H, int W, int C,
                      int Ho, int Wo, int k, int stride, int pad, void* stream);
int pseg_dwconv_wgrad(const float* x, int ldx, const float* dy, int ldy, float* dw, int B, int H, int W, int C,
                      int Ho, int Wo, int k, int stride, int pad, int accumulate,
                      void* workspace, int64_t workspace_bytes, void* stream);
int64_t pseg_dwconv_wgrad_workspace_bytes(int B, int Ho, int Wo, int C, int k);

/* ------------------------------------------------------------------ BatchNorm (+activation, +residual)
 * Replaces nn.BatchNorm2d + activation inside ConvNormAct (training: batch statistics,
 * biased variance for normalisation, unbiased for running_var, momentum 0.1, eps 1e-5).
 *
 * pseg_col_stats: the same [3][rows][C] shifted statistics for any y[M][C] (when the producer did not
 *   emit them); rows = pseg_col_stats_rows(M, C), group size = pseg_col_stats_group(M, C) (chosen so the
 *   reduction grid fills the chip).  The backward partials of pseg_bn_act_bwd_reduce use the same row count.
 * pseg_bn_finalize: group statistics -> (Chan's parallel merge, in double) mean, invstd,
 *   scale=gamma*invstd, shift=beta; updates running_mean/var in place when non-NULL.
 * pseg_bn_eval_coeffs: the same four vectors from the running statistics (model.eval(), test.py:17).
 * pseg_bn_act_fwd: z = act((y - mean)*scale + shift (+ residual)); the mean is subtracted BEFORE scaling so
 *   channels with |mean| >> std keep full fp32 precision.  mean = scale = shift = NULL: plain act / residual add.
 */
int pseg_col_stats_rows(int64_t M, int C);
int pseg_col_stats_group(int64_t M, int C);
int pseg_col_stats(const float* y, int ldy, int64_t M, int C, float* stat, void* stream);
int pseg_bn_finalize(const float* stat, int rows, int group, int64_t count, int C,
                     const float* gamma, const float* beta, float* running_mean, float* running_var,
                     float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                     void* workspace, int64_t workspace_bytes, void* stream);
int64_t pseg_bn_finalize_workspace_bytes(int rows, int C);
int pseg_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, int C, float* mean, float* invstd, float* scale,
                        float* shift, void* stream);
int pseg_bn_act_fwd(const float* y, int ldy, const float* mean, const float* scale, const float* shift,
                    const float* residual, int ldr, int act, float* z, int ldz, int64_t M, int C, float* amax_z,
                    uint32_t* mask_out, void* stream);
/* amax_z (nullable): amax_z[0] = max(amax_z[0], max|z|), see pseg_amax.
 * mask_out (nullable; needs C % 32 == 0 and an activation): the activation bitmask of z, [M][C/32] words, bit c % 32 of
 * word c / 32 set when act'(z) = 1.  For a layer WITH a residual the backward passes cannot recompute the mask from y;
 * given this bitmask (argument `mask` of pseg_bn_act_bwd_reduce / _apply) they read 1 bit per element instead of z. */
/* Small tensors (launch-bound configurations): finalize folded into the apply pass, one launch instead of two, with
 * bit-identical coefficients.  pseg_bn_small_path(rows, M, C) != 0 says when the library recommends them
 * (rows = number of statistic / partial row groups). */
int pseg_bn_small_path(int rows, int64_t M, int C);
int pseg_bn_fwd_fused(const float* stat, int rows, int group, int64_t count, int C, const float* gamma,
                      const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                      float* mean, float* invstd, float* scale, float* shift, const float* y, int ldy,
                      const float* residual, int ldr, int act, float* z, int ldz, int64_t M, float* amax_z,
                      void* stream);
int pseg_bn_bwd_fused(const float* part_db, const float* part_dg, int rows, int64_t count, int C, float* dgamma,
                      float* dbeta, int accumulate, int frozen, const float* dz, int lddz, const float* z, int ldz,
                      const float* y, int ldy, const float* mean, const float* invstd, const float* scale,
                      const float* shift, int act, float* dy, int lddy, float* dres, int lddres,
                      int res_accumulate, int64_t M, void* stream);
/* backward, two passes over (dz, y[, z]).  z may be NULL when there is no residual: the activation argument is then
 * recomputed from y as (y - mean)*scale + shift, bit-identically to the forward pass (one tensor read fewer per pass).
 *  reduce: dyh = dz * act'(z); partials of sum(dyh), sum(dyh * xhat)         [rows][C] each
 *  finalize: dgamma, dbeta (+= when accumulate), c1 = dbeta/M, c2 = dgamma/M; frozen != 0 (eval-mode BatchNorm:
 *            mean / invstd are the running statistics, constants of the graph): c1 = c2 = 0, so apply gives
 *            dy = scale * dyh while dgamma = sum(dyh * xhat), dbeta = sum(dyh) are still produced
 *  apply: dy = scale * (dyh - c1 - xhat*c2); dres (nullable) = dyh (+= when res_accumulate) */
int pseg_bn_act_bwd_reduce(const float* dz, int lddz, const float* z, int ldz, const float* y, int ldy,
                           const float* mean, const float* invstd, const float* scale, const float* shift, int act,
                           int64_t M, int C, float* part_db, float* part_dg, const uint32_t* mask, void* stream);
int pseg_bn_bwd_finalize(const float* part_db, const float* part_dg, int rows, int64_t count, int C,
                         float* dgamma, float* dbeta, int accumulate, int frozen, float* c1, float* c2,
                         void* stream);
int pseg_bn_act_bwd_apply(const float* dz, int lddz, const float* z, int ldz, const float* y, int ldy,
                          const float* mean, const float* invstd, const float* scale, const float* shift,
                          const float* c1, const float* c2, int act, float* dy, int lddy, float* dres, int lddres,
                          int res_accumulate, int64_t M, int C, const uint32_t* mask, uint16_t* dy_hi, uint16_t* dy_lo,
                          int ldp, void* stream);
/* dy_hi / dy_lo (nullable, together; ldp == C, C % 8 == 0): the pass also writes dy as bf16 limb planes (hi = bf16(dy),
 * lo = bf16(dy - hi); exactly what pseg_split_planes(dy) would produce) for pseg_conv2d_dgrad_planes -- the consumer's
 * tiles then go global -> LDS by DMA with no split arithmetic. */
/* eval-mode / frozen-statistics backward and plain activation backward:
 *   dy = scale * dz * act'(z)   (scale NULL -> 1) ; dres as above */
int pseg_act_bwd(const float* dz, int lddz, const float* z, int ldz, const float* scale, int act,
                 float* dy, int lddy, float* dres, int lddres, int res_accumulate, int64_t M, int C,
                 void* stream);
/* column sums of dy[M][C] (bias gradient of the cls_conv, models/deeplabv3plus.py:22) */
int pseg_col_sum(const float* dy, int ldy, int64_t M, int C, float* out, int accumulate,
                 void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ pooling / resize / layout
 * pseg_pool_sum: out[b,c] = scale * sum_p x[b,p,c]   (AdaptiveAvgPool2d(1), models/aspp.py:11,
 *   and the backward of the pooled branch's broadcast)
 * pseg_broadcast: y[b,p,c] = x[b,c] * scale            (bilinear from 1x1, models/aspp.py:16-19; GAP backward)
 * pseg_bilinear_*: F.interpolate(mode='bilinear') NHWC -> NHWC slice
 *   (models/deeplabv3plus.py:34-37,40-43; models/unet.py:30-55; utils/utils.py:18-20)
 *   out_nchw != 0: the output (fwd) / incoming gradient (bwd) is a contiguous NCHW tensor, which is
 *   what the loss and argmax consume.
 */
int pseg_pool_sum(const float* x, int ldx, int B, int HW, int C, float scale, float* out, int ldo, void* stream);
int pseg_broadcast(const float* x, int ldx, int B, int HW, int C, float scale, float* y, int ldy,
                   int accumulate, void* stream);
int pseg_bilinear_fwd(const float* x, int ldx, int B, int Hi, int Wi, int C, float* y, int ldy, int Ho, int Wo,
                      int align_corners, int out_nchw, void* stream);
/* dy_nchw = 1 with scratch of pseg_bilinear_bwd_workspace_bytes(): separable two-pass form (the weights factor into
 * a column pass and a row pass); without scratch (workspace NULL) a single gather pass. */
int64_t pseg_bilinear_bwd_workspace_bytes(int B, int Hi, int Wi, int C, int Ho, int Wo, int dy_nchw);
int pseg_bilinear_bwd(const float* dy, int ldy, int B, int Hi, int Wi, int C, float* dx, int ldx, int Ho, int Wo,
                      int align_corners, int dy_nchw, int accumulate, void* workspace, int64_t workspace_bytes,
                      void* stream);
/* pseg_bn_act_maxpool_fwd: max-pooling of z = act((x - mean) * scale + shift) without z ever existing (the ResNet stem in training:
 * conv -> BatchNorm -> ReLU -> MaxPool(3, 2, 1), torchvision's resnet / the reference's pytorch_modules backbone; mean / scale /
 * shift = rows 0, 2, 3 of pseg_bn_finalize's coefficients).  Same values, same argmax as pseg_bn_act_fwd + pseg_maxpool_fwd. */
int pseg_bn_act_maxpool_fwd(const float* x, int ldx, const float* mean, const float* scale, const float* shift, int act, int B,
                            int H, int W, int C, float* y, int ldy, uint8_t* argmax, int Ho, int Wo, int k, int stride, int pad,
                            void* stream);
int pseg_bn_act_maxpool_fwd_h(const pseg_half_t* x, int ldx, const float* mean, const float* scale, const float* shift, int act,
                              int B, int H, int W, int C, pseg_half_t* y, int ldy, uint8_t* argmax, int Ho, int Wo, int k,
                              int stride, int pad, void* stream);
int pseg_maxpool_fwd(const float* x, int ldx, int B, int H, int W, int C, float* y, int ldy, uint8_t* argmax,
                     int Ho, int Wo, int k, int stride, int pad, void* stream);
int pseg_maxpool_bwd(const float* dy, int ldy, const uint8_t* argmax, int B, int H, int W, int C, float* dx,
                     int ldx, int Ho, int Wo, int k, int stride, int pad, int accumulate, void* stream);
/* NCHW [B][C][HW] -> NHWC [B][HW][ld] (channels >= C zero-filled up to Cpad) and back. */
int pseg_nchw_to_nhwc(const float* x, float* y, int ldy, int B, int C, int HW, int Cpad, void* stream);
int pseg_nhwc_to_nchw(const float* x, int ldx, float* y, int B, int C, int HW, void* stream);
/* fp32 NCHW image (C <= 8) -> fp16 NHWC with 8 channels per pixel (zero-filled): the half-precision policy's network input */
int pseg_nchw_to_nhwc_h(const float* x, pseg_half_t* y, int ldy, int B, int C, int HW, void* stream);
/* strided 2-D copy / add of an [M][C] block (concat-slice plumbing, residual adds) */
int pseg_copy2d(const float* x, int ldx, float* y, int ldy, int64_t M, int C, int accumulate, void* stream);

/* ------------------------------------------------------------------ loss / masks / metrics
 * pseg_ce_fwd_bwd: nn.CrossEntropyLoss() defaults (utils/utils.py:12,21): mean over non-ignored pixels,
 *   ignore_index -100.  ONE pass over the logits writes dlogits = (softmax - onehot)/n_valid and the
 *   loss partials.  loss_out[3]: [0] = mean loss, [1] = n_valid, [2] = n_out_of_range (as floats).  A pixel is valid
 *   iff target != ignore_index and 0 <= target < C -- the same predicate for the divisor, the sum and the gradient.
 *   Targets outside [0, C) other than ignore_index (torch raises IndexError on them) contribute nothing and are
 *   counted in loss_out[2] so the caller can raise without a second pass.  dlogits may be NULL (eval).
 * pseg_scale_inplace: dlogits *= *gscale (device scalar; returns immediately on the device when it is 1).
 * pseg_argmax: outputs.max(1)[1] (test.py:31), first index on ties.
 * pseg_confusion: per-class tp / fn / fp counts (test.py:34-46) accumulated into int64 counters[3][C].
 */
int64_t pseg_ce_workspace_bytes(int64_t npix);
int pseg_ce_fwd_bwd(const float* logits, const int64_t* target, int B, int C, int64_t HW, int64_t ignore_index,
                    float* dlogits, float* loss_out, void* workspace, int64_t workspace_bytes, void* stream);
/* The same loss on bilinearly up-sampled logits WITHOUT the full-resolution tensor (models/deeplabv3plus.py:40-43 +
 * utils/utils.py:18-21): logits_lr is the fp32 NHWC [B,h,w] x ld tensor of class logits (C <= 24 classes in the first
 * channels, ld % 4 == 0), target the [B,H,W] labels; the loss is CE(interpolate(logits, (H, W), 'bilinear', align_corners),
 * target) and dlogits_lr ([B,h,w] x ldd, nullable) receives its gradient with respect to the LOW-resolution logits
 * (channels [C, ldd) are written as 0).  Deterministic (no floating-point atomics).  pseg_ce_upsampled_ok(...) != 0 says
 * whether the scale factors are covered (about x4 or less per axis); otherwise use pseg_bilinear_fwd + pseg_ce_fwd_bwd +
 * pseg_bilinear_bwd. */
int pseg_ce_upsampled_ok(int h, int w, int C, int H, int W, int align_corners);
int64_t pseg_ce_upsampled_workspace_bytes(int B, int h, int w);
int pseg_ce_upsampled_fwd_bwd(const float* logits_lr, int ld, int B, int h, int w, int C, const int64_t* target, int H,
                              int W, int align_corners, int64_t ignore_index, float* dlogits_lr, int ldd,
                              float* loss_out, void* workspace, int64_t workspace_bytes, void* stream);
int pseg_scale_inplace(float* x, int64_t n, const float* gscale, void* stream);
int pseg_argmax(const float* logits, int B, int C, int64_t HW, int64_t* mask, void* stream);
int pseg_confusion(const int64_t* pred, const int64_t* target, int64_t n, int C, int64_t* counters, void* stream);

/* ------------------------------------------------------------------ optimiser (flat parameter arena)
 * One launch over the whole arena; grad_scale folds the 1/world_size of the data-parallel mean
 * (README.md:42-44, train.py:112-117) and the 1/accumulate of gradient accumulation (train.py:65).
 * decay_mask (nullable, uint8 per 4-element group) selects where weight decay applies. */
int pseg_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum,
                  float weight_decay, int nesterov, float grad_scale, int first_step, void* stream);
int pseg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int decoupled, float grad_scale,
                   int step, void* stream);
int pseg_fill(float* x, int64_t n, float value, void* stream);
/* amax_inout[0] = max(amax_inout[0], max |x[m*ld + c]|), m < M, c < C (atomic on the float's bit pattern; zero it first) */
int pseg_amax(const float* x, int64_t ld, int64_t M, int C, float* amax_inout, void* stream);
/* max|.| of many contiguous float arrays in one launch (every conv filter of a model, once per step).  jobs: device array
 * of n records of three int64 {device address, element count, index of the record's first 256-thread block}, block
 * indices ascending from 0; total_blocks = their sum.  out[j] = max|x_j| (the launch zeroes out[] first). */
int pseg_amax_batch(const int64_t* jobs, int n, int64_t total_blocks, float* out, void* stream);

/* debug: while `buffer` is non-NULL every block of the exact-fp32 LDS-DMA gather kernel writes five uint64 words to
 * buffer[5 * blockIdx.x ...] -- wall_clock64 at entry / first tile landed / last MFMA issued / stores drained, and HW_ID
 * (tools/conv_phases.py turns them into a phase table).  Debug builds only (-DPSEG_CONV_TRACE=1; the default build returns
 * an error for a non-NULL buffer): one global pointer, no stream ordering. */
int pseg_debug_conv_trace(void* buffer);
/* debug / tests: which kernel did the calling thread's LAST convolution entry point (pseg_conv2d_fwd / _dgrad* / _wgrad* and their
 * _h forms) enqueue?  A parity test of a specialised kernel can so assert that the kernel it means to test is the one that ran. */
#define PSEG_KERNEL_GATHER_REGISTER 1        /* gather_conv_kernel (register-staged; every precision, split-K) */
#define PSEG_KERNEL_GATHER_LIMB_DMA 2        /* gather_limb_dma_kernel (pre-split bf16 planes) */
#define PSEG_KERNEL_GATHER_RING 3            /* gather_f32_dma_kernel */
#define PSEG_KERNEL_GATHER_RING_GENERIC 4    /* gather_f32_dma_kernel, channel counts off the K-step grid */
#define PSEG_KERNEL_GATHER_POINTWISE 5       /* gather_f32_pw_kernel (persistent) */
#define PSEG_KERNEL_GATHER_HALO 6            /* gather_f32_halo_kernel */
#define PSEG_KERNEL_WGRAD_REGISTER 11        /* wgrad_kernel */
#define PSEG_KERNEL_WGRAD_LIMB 12            /* wgrad_limb_kernel */
#define PSEG_KERNEL_WGRAD_DMA 13             /* wgrad_f32_dma_kernel */
#define PSEG_KERNEL_WGRAD_HALO 14            /* wgrad_f32_halo_kernel */
#define PSEG_KERNEL_GATHER_H 21              /* gather_h_kernel (fp16) */
#define PSEG_KERNEL_GATHER_H_PERSISTENT 22   /* gather_hp_kernel (fp16) */
int pseg_debug_last_conv_kernel(void);

/* ---- lane executor: a captured hipGraph replayed as plain launches on a few streams (csrc/lanes.hip) --------------------
 * The reference drives its step from Python through torch/apex (train.py:63-72 via pytorch_modules' Trainer); for the
 * launch-bound configurations (BASELINE configs[1], configs[4]) the host, not the GPU, sets the step time.  A step that
 * was captured once (hipStreamBeginCapture -- torch.cuda.graph on the Python side) is replayed here without Python and
 * without hipGraphExec: pseg_lanes_build walks the graph (kernel / memset / empty nodes and their edges; a graph with memcpy or other
 * nodes is refused with an error -- their parameters cannot be read back reliably -- and the caller runs such a step EAGERLY:
 * hipGraphLaunch of a forked graph is not a fallback, DESIGN.md section 5 "Fault records"),
 * assigns the nodes to at most max_lanes stream-ordered lanes and turns the edges between lanes into events;
 * pseg_lanes_launch enqueues the whole step -- lane 0 on `stream`, the other lanes on streams the executor owns, all of
 * them after what `stream` holds so far, and `stream` continues after all of them.  The hipGraph_t (argument blocks,
 * private memory pool) must outlive the executor.  Same launches, dependency-respecting order: results are bit-identical
 * to eager execution.  pseg_lanes_info reports graph nodes, launching nodes, lanes used and cross-lane events. */
int pseg_lanes_build(void* hip_graph, int max_lanes, int64_t* handle);
int pseg_lanes_info(int64_t handle, int* nodes, int* launches, int* lanes, int* events);
int pseg_lanes_launch(int64_t handle, void* stream);
/* The lanes of every executor of a device run on one pool of streams that lives as long as the process (a stream per
 * executor would land on whatever hardware queue the runtime's round-robin has reached: GPU_MAX_HW_QUEUES = 4, two busy
 * lanes on one queue serialise).  pseg_lanes_reserve(lanes) creates -- and touches -- the streams for `lanes` lanes on the
 * current device NOW (the Trainer calls it before it captures its first step -- from its constructor for a model that lives on
 * the replay, so that the pool exists before the process's other streams); pseg_lanes_build reserves what is missing. */
int pseg_lanes_reserve(int lanes);
/* Drains the executor's device (hipDeviceSynchronize: the executor's events are bound to kernels of the last replay on the
 * pool streams AND on the caller's stream), then releases events and host state.  Must not be called while a stream capture
 * is open; on failure nothing is released and the handle stays valid. */
int pseg_lanes_destroy(int64_t handle);

/* Markers: where the REPLAYED step meets work the executor does not own -- the data-parallel gradient exchange
 * (train.py:33-35,112-117: DistributedDataParallel overlaps its bucket all-reduces with backward).  pseg_mark enqueues a
 * one-word memset (zero) of `word` on `stream`; inside a capture that is a memset node which depends on everything the
 * stream holds so far.  pseg_lanes_bind_markers(handle, base, count): every such node of the walked graph whose word lies
 * in base[0..count) becomes marker (word index); the replay records an event right after it on its lane.  After
 * pseg_lanes_launch, pseg_lanes_wait_marker(handle, id, stream) makes `stream` wait for all events of marker `id` (one
 * per stream the marker was set on) of that launch -- the caller then enqueues the bucket's all-reduce on `stream`, and it
 * overlaps the rest of the replayed backward.  (In a graph replayed any other way the markers are harmless memsets.) */
int pseg_mark(void* word, void* stream);
int pseg_lanes_bind_markers(int64_t handle, const void* base, int count, int* bound);
int pseg_lanes_wait_marker(int64_t handle, int id, void* stream);

/* ---- gradient exchange over RCCL (csrc/comm.hip) -----------------------------------------------------------------------
 * allreduce_bucket(flat_grad, stream): what DistributedDataParallel does for the reference inside its external Trainer
 * (README.md:42-44, train.py:33-35,112-117) -- the gradients of one bucket summed over the ranks, in place, on `stream`
 * (the caller's side stream: it overlaps the rest of backward).  A bucket is a contiguous range of the fp32 gradient arena.
 * RCCL is bound at run time (PSEG_RCCL_PATH if set, else the process's own librccl.so, else the loader path):
 * pseg_comm_available() tells whether that worked.  One rank calls pseg_comm_unique_id (128 bytes) and hands the id to the
 * others by whatever channel the host has; every rank then calls pseg_comm_init(id, nranks, rank) with its HIP device
 * current (collective: returns when all ranks have joined).  The mean's 1/nranks is the optimiser's grad_scale. */
int pseg_comm_available(void);
int pseg_comm_unique_id(void* id128);
int pseg_comm_init(const void* id128, int nranks, int rank, int64_t* comm);
int pseg_comm_destroy(int64_t comm);
int pseg_allreduce_bucket(int64_t comm, float* flat_grad, int64_t count, void* stream);
/* RCCL's version code (ncclGetVersion: major * 10000 + minor * 100 + patch), for run records. */
int pseg_comm_version(int* version);
/* The same sum as a reduce-scatter + all-gather pair, in place (on the fully connected 8-GPU xGMI
 * node a direct exchange uses all seven links of a GPU, a ring is bound by one): rank r first receives the sum of slice
 * [r * count_per_rank, (r + 1) * count_per_rank) of the bucket, then the reduced slices are gathered.  The bucket holds
 * nranks * count_per_rank elements (the caller all-reduces a remainder of fewer than nranks elements separately).
 * Replaces the same DistributedDataParallel exchange as pseg_allreduce_bucket (README.md:42-44, train.py:112-117). */
int pseg_reduce_scatter_bucket(int64_t comm, float* flat_grad, int64_t count_per_rank, int rank, void* stream);
int pseg_all_gather_bucket(int64_t comm, float* flat_grad, int64_t count_per_rank, int rank, void* stream);

/* ---- fault diagnostics (csrc/diag.hip) -----------------------------------------------------------------------------------
 * The reference's loop (train.py:71-72) dies with a Python traceback when something below it faults; a fault inside the HIP
 * runtime leaves no native frame in it.  pseg_fault_backtrace_enable() installs handlers for SIGSEGV / SIGBUS / SIGFPE /
 * SIGILL / SIGABRT that write the faulting thread's native frames to stderr (backtrace_symbols_fd) and then chain to the
 * handler installed before (Python's faulthandler) or re-raise with the default action.  PSEG_SEGV_BACKTRACE=1 in the
 * environment installs them when the library is loaded.  pseg_fault_backtrace_enabled() -> 0 / 1. */
int pseg_fault_backtrace_enable(void);
int pseg_fault_backtrace_enabled(void);

#ifdef __cplusplus
}
#endif
#endif /* PSEG_AMD_H */
