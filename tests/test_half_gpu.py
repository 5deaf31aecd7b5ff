"""Half-precision (`-mp`) path, op level: the fp16 kernels against stock torch CPU ops ON HALF-ROUNDED OPERANDS.

A product of two fp16 values is exact in fp32 and the kernels accumulate in fp32, so a result written as fp32 (class
logits, weight gradients, statistics) must agree with the fp64 CPU result of the same fp16-rounded operands to 1e-5 of
the tensor's peak -- no loose tolerance -- and a result written as fp16 must be that value rounded once: within half an
fp16 ulp of it, element by element (plus 1e-5 of the peak for the accumulation order).
Reference: the reference's `-mp` (apex fp16 compute, train.py:70,102-105,138; README.md:12)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import fill

pytestmark = pytest.mark.gpu

TOL32 = 1e-5


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    from pytorch_segmentation_amd import ops as _ops
    return _ops


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def assert_half_rounded(got, ref64, what=''):
    """got (fp16 values, any float dtype) == ref64 rounded once to fp16, up to the fp32 accumulation-order noise."""
    got = got.detach().double().cpu()
    ref64 = ref64.detach().double().cpu()
    peak = ref64.abs().max().item()
    # half an ulp of fp16 is 2^-11 relative for normal numbers (2^-25 absolute below 2^-14)
    band = ref64.abs() * 2.0 ** -11 * 1.0001 + 2.0 ** -25 + 2e-5 * peak * 2.0 ** -5 + 1e-5 * peak * 0 + 3e-6 * peak
    bad = (got - ref64).abs() > band
    assert not bad.any(), '%s: %d of %d elements off by more than one fp16 rounding (worst %.3e of peak)' % (
        what, int(bad.sum()), bad.numel(), ((got - ref64).abs().max() / (peak + 1e-30)).item())


def r8(n):
    return (n + 7) // 8 * 8


def to_act_h(ops, x, Cpad):
    return ops.Act.from_nchw(x.cuda(), Cpad, dtype=torch.float16)


def krsc(w, cout_pad, cin_pad):
    Cout, Cin, kh, kw = w.shape
    raw = torch.zeros(cout_pad, kh, kw, cin_pad)
    raw[:Cout, :, :, :Cin] = w.permute(0, 2, 3, 1)
    return raw.contiguous()


def h(x):
    """round to fp16 and back (the values the kernels see)"""
    return x.half().float()


CONV_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, dil
    (2, 64, 16, 16, 128, 1, 1, 0, 1),
    (2, 32, 20, 20, 64, 3, 1, 1, 1),          # Cin % 64 != 0: per-lane (tap, channel) addressing
    (2, 64, 24, 24, 32, 3, 1, 6, 6),
    (2, 64, 24, 24, 32, 3, 1, 12, 12),
    (2, 64, 24, 24, 32, 3, 1, 18, 18),
    (2, 32, 33, 35, 64, 3, 2, 1, 1),
    (2, 64, 16, 16, 128, 1, 2, 0, 1),
    (2, 3, 64, 64, 64, 7, 2, 3, 1),           # the stem (3 -> 8 padded input channels)
    (2, 384, 32, 32, 21, 3, 1, 1, 1),         # classifier: bias, fp32 logits
    (2, 2048, 8, 8, 256, 3, 1, 6, 6),
    (2, 1280, 4, 4, 256, 3, 1, 1, 1),
    (4, 256, 1, 1, 64, 1, 1, 0, 1),           # the pooled ASPP branch: M = B rows
    (2, 88, 16, 16, 2, 3, 1, 1, 1),
    (1, 160, 9, 9, 64, 3, 1, 1, 1),
    (2, 512, 8, 8, 512, 3, 1, 2, 2),
    (1, 16, 130, 130, 16, 3, 1, 1, 1),
    (4, 256, 32, 32, 128, 1, 1, 0, 1),
    (4, 320, 4, 4, 1280, 1, 1, 0, 1),
    (4, 256, 8, 8, 1024, 1, 1, 0, 1),
    (2, 128, 48, 48, 128, 3, 1, 1, 1),
    (4, 16, 128, 128, 32, 3, 1, 1, 1),
    (2, 128, 32, 32, 64, 3, 1, 18, 18),       # ASPP geometry: whole (tile, tap) pairs are padding -> skipped K-steps
    (2, 128, 32, 32, 64, 3, 1, 12, 12),
    (3, 256, 16, 16, 128, 3, 1, 6, 6),
    (2, 64, 32, 32, 64, 3, 2, 1, 1),          # stride-2 dgrad with parity-class row order
    (2, 64, 32, 32, 128, 1, 2, 0, 1),
    (1, 32, 64, 48, 96, 3, 2, 1, 1),
    (2, 24, 40, 40, 144, 1, 1, 0, 1),         # MobileNetV2 expansions
    (2, 144, 40, 40, 24, 1, 1, 0, 1),
    (8, 32, 64, 64, 32, 3, 1, 1, 1),          # HRNet fine branch
    (8, 64, 32, 32, 64, 3, 1, 1, 1),
    (8, 32, 64, 64, 64, 3, 2, 1, 1),          # HRNet fuse layer (stride 2)
]


def _setup(ops, case, with_bias):
    B, Cin, H, W, Cout, k, stride, pad, dil = case
    key = 'hconv/' + '_'.join(map(str, case))
    x = h(fill.uniform(key + '/x', (B, Cin, H, W)))
    w = h(fill.uniform(key + '/w', (Cout, Cin, k, k), (6.0 / (Cin * k * k)) ** 0.5))
    b = fill.uniform(key + '/b', (Cout,), 0.5) if with_bias else None
    cin_p, cout_p = r8(Cin), r8(Cout)
    xa = to_act_h(ops, x, cin_p)
    w_h = krsc(w, cout_p, cin_p).half().cuda()
    b_raw = None
    if with_bias:
        b_raw = torch.zeros(cout_p)
        b_raw[:Cout] = b
        b_raw = b_raw.cuda()
    Ho, Wo = ops.conv_out_size(H, k, stride, pad, dil), ops.conv_out_size(W, k, stride, pad, dil)
    return x, w, b, xa, w_h, b_raw, cin_p, cout_p, Ho, Wo


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv2d_fwd_half(ops, case):
    B, Cin, H, W, Cout, k, stride, pad, dil = case
    with_bias = Cout in (21, 2)
    x, w, b, xa, w_h, b_raw, cin_p, cout_p, Ho, Wo = _setup(ops, case, with_bias)
    ref = F.conv2d(x.double(), w.double(), b.double() if with_bias else None, stride, pad, dil)
    # fp32 output (what the classifier writes): exact products, fp32 sums
    y32 = ops.Act.empty(B, Ho, Wo, cout_p, 'cuda')
    stats = ops.conv2d_fwd(xa, w_h, b_raw, y32, k, k, stride, pad, dil, want_stats=not with_bias)
    assert rel(y32.to_nchw(Cout), ref) < TOL32
    if cout_p > Cout:
        assert y32.view4()[..., Cout:].abs().max().item() == 0.0
    if stats is not None:
        co = ops.bn_finalize(stats, y32.M, None, None, None, None, 0.0, 1e-5)
        mu, var = ref.mean((0, 2, 3)), ref.var((0, 2, 3), unbiased=False)
        assert rel(co[0][:Cout], mu) < 1e-4 * max(1.0, (var.sqrt().max() / (mu.abs().max() + 1e-30)).item())
        assert rel(co[1][:Cout], 1.0 / (var + 1e-5).sqrt()) < 1e-4
    # fp16 output: the same value rounded once
    y16 = ops.Act.empty(B, Ho, Wo, cout_p, 'cuda', dtype=torch.float16)
    ops.conv2d_fwd(xa, w_h, b_raw, y16, k, k, stride, pad, dil)
    assert_half_rounded(y16.to_nchw(Cout), ref, 'fwd')
    # accumulate: y = half(float(y) + conv)
    before = y16.to_nchw(Cout).double().cpu()
    ops.conv2d_fwd(xa, w_h, b_raw, y16, k, k, stride, pad, dil, accumulate=True)
    assert_half_rounded(y16.to_nchw(Cout), before + ref, 'fwd accumulate')


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv2d_dgrad_wgrad_half(ops, case):
    B, Cin, H, W, Cout, k, stride, pad, dil = case
    x, w, b, xa, w_h, b_raw, cin_p, cout_p, Ho, Wo = _setup(ops, case, False)
    key = 'hconvg/' + '_'.join(map(str, case))
    gy = h(fill.uniform(key, (B, Cout, Ho, Wo)))
    xr = x.double().requires_grad_()
    wr = w.double().requires_grad_()
    F.conv2d(xr, wr, None, stride, pad, dil).backward(gy.double())
    gya = to_act_h(ops, gy, cout_p)
    # transposed fp16 filter (what pseg_filter_prepare_h writes; here by torch)
    wT_h = w_h.view(cout_p, k * k, cin_p).permute(2, 1, 0).contiguous()
    dxa = ops.Act.empty(B, H, W, cin_p, 'cuda', dtype=torch.float16)
    ops.conv2d_dgrad(gya, wT_h, dxa, k, k, stride, pad, dil)
    assert_half_rounded(dxa.to_nchw(Cin), xr.grad, 'dgrad')
    before = dxa.to_nchw(Cin).double().cpu()
    ops.conv2d_dgrad(gya, wT_h, dxa, k, k, stride, pad, dil, accumulate=True)
    assert_half_rounded(dxa.to_nchw(Cin), before + xr.grad, 'dgrad accumulate')
    # weight gradient: fp32 output
    dw = torch.empty(cout_p, k, k, cin_p, device='cuda')
    ops.conv2d_wgrad(xa, gya, dw, k, k, stride, pad, dil)
    got = dw[:Cout, :, :, :Cin].permute(0, 3, 1, 2).cpu()
    assert rel(got, wr.grad) < TOL32
    if cout_p > Cout:
        assert dw[Cout:].abs().max().item() == 0.0
    if cin_p > Cin:
        assert dw[..., Cin:].abs().max().item() == 0.0
    ops.conv2d_wgrad(xa, gya, dw, k, k, stride, pad, dil, accumulate=True)
    got2 = dw[:Cout, :, :, :Cin].permute(0, 3, 1, 2).cpu()
    assert rel(got2, 2 * wr.grad) < TOL32
    dw2 = torch.empty_like(dw)
    ops.conv2d_wgrad(xa, gya, dw2, k, k, stride, pad, dil)
    dw3 = torch.empty_like(dw)
    ops.conv2d_wgrad(xa, gya, dw3, k, k, stride, pad, dil)
    assert torch.equal(dw2, dw3)        # bit-reproducible (fixed-order slabs)


# BASELINE.json configs[2] shapes (DeepLabV3+ R50, 512x512, batch 16) under the half policy: the launches `bench.py
# --precision half` / `train.py -mp` actually time, through the DEFAULT plan (no PSEG_* overrides) -- the 8-wave tiles with
# the XCD remap on 2048-block grids, the tap-skipping 128x64 forward tiles, the 32-pixel K-step weight gradient with its
# pixel splits + slab reduction.  The fp16 twin of tests/test_ops_gpu.py::test_conv2d_c3_shapes; the reference is stock
# torch conv2d + autograd in fp64 on the same half-rounded operands, the bounds are this file's own (1e-5 of the peak for
# results written in fp32, one fp16 rounding otherwise).
C3_CASES_H = [
    # name, B, Cin, H, W, Cout, k, pad, dil, bias
    ('aspp_d6', 16, 2048, 32, 32, 256, 3, 6, 6, False),      # reference models/aspp.py:28-29, rates models/deeplabv3plus.py:21
    ('aspp_d12', 16, 2048, 32, 32, 256, 3, 12, 12, False),
    ('aspp_d18', 16, 2048, 32, 32, 256, 3, 18, 18, False),
    ('aspp_1x1', 16, 2048, 32, 32, 256, 1, 0, 1, False),     # models/aspp.py:27
    ('aspp_project', 16, 1280, 32, 32, 256, 1, 0, 1, False), # models/aspp.py:30
    ('project', 16, 256, 128, 128, 128, 1, 0, 1, False),     # models/deeplabv3plus.py:20
    ('cls_conv', 16, 384, 128, 128, 21, 3, 1, 1, True),      # models/deeplabv3plus.py:22: bias, fp32 logits
    ('layer4_3x3', 16, 512, 32, 32, 512, 3, 2, 2, False),    # ResNet-50 OS16 layer 4 (dilation 2): the 128x128 / 8-wave tile
    ('layer1_1x1', 16, 64, 128, 128, 256, 1, 0, 1, False),   # bandwidth-bound expansion: short K-step, three-deep ring
]


@pytest.mark.parametrize('case', C3_CASES_H, ids=[c[0] for c in C3_CASES_H])
def test_conv2d_c3_shapes_half(ops, case):
    name, B, Cin, H, W, Cout, k, pad, dil, with_bias = case
    key = 'c3h/' + name
    x = h(fill.uniform(key + '/x', (B, Cin, H, W)).abs_())                    # post-ReLU activations
    w = h(fill.uniform(key + '/w', (Cout, Cin, k, k), (6.0 / (Cin * k * k)) ** 0.5))
    b = fill.uniform(key + '/b', (Cout,), 0.5) if with_bias else None
    gy = h(fill.uniform(key + '/gy', (B, Cout, H, W)))
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    ref = F.conv2d(xr, wr, b.double() if with_bias else None, 1, pad, dil)
    ref.backward(gy.double())
    ref = ref.detach()
    cin_p, cout_p = r8(Cin), r8(Cout)
    xa, gya = to_act_h(ops, x, cin_p), to_act_h(ops, gy, cout_p)
    w_h = krsc(w, cout_p, cin_p).half().cuda()
    wT_h = w_h.view(cout_p, k * k, cin_p).permute(2, 1, 0).contiguous()
    b_raw = None
    if with_bias:
        b_raw = torch.zeros(cout_p)
        b_raw[:Cout] = b
        b_raw = b_raw.cuda()
    # forward, fp32 result + fused BatchNorm statistics
    y32 = ops.Act.empty(B, H, W, cout_p, 'cuda')
    stats = ops.conv2d_fwd(xa, w_h, b_raw, y32, k, k, 1, pad, dil, want_stats=not with_bias)
    e_f = rel(y32.to_nchw(Cout), ref)
    assert e_f < TOL32, (name, 'fwd f32', e_f)
    if stats is not None:
        co = ops.bn_finalize(stats, y32.M, None, None, None, None, 0.0, 1e-5)
        var = ref.var((0, 2, 3), unbiased=False)
        assert rel(co[1][:Cout], 1.0 / (var + 1e-5).sqrt()) < 1e-4, (name, 'stats')
    # forward, fp16 result (what every layer but the classifier writes) + statistics of the values as stored
    y16 = ops.Act.empty(B, H, W, cout_p, 'cuda', dtype=torch.float16)
    stats16 = ops.conv2d_fwd(xa, w_h, b_raw, y16, k, k, 1, pad, dil, want_stats=not with_bias)
    assert_half_rounded(y16.to_nchw(Cout), ref, name + ' fwd f16')
    if stats16 is not None:
        co = ops.bn_finalize(stats16, y16.M, None, None, None, None, 0.0, 1e-5)
        stored = y16.to_nchw(Cout).double().cpu()
        assert rel(co[0][:Cout], stored.mean((0, 2, 3))) < 1e-4 * max(1.0, (stored.std((0, 2, 3)).max() / stored.mean((0, 2, 3)).abs().max()).item())
        assert rel(co[1][:Cout], 1.0 / (stored.var((0, 2, 3), unbiased=False) + 1e-5).sqrt()) < 1e-4, (name, 'stats of stored values')
    # data gradient (fp16 result)
    dxa = ops.Act.empty(B, H, W, cin_p, 'cuda', dtype=torch.float16)
    ops.conv2d_dgrad(gya, wT_h, dxa, k, k, 1, pad, dil)
    assert_half_rounded(dxa.to_nchw(Cin), xr.grad, name + ' dgrad')
    # weight gradient (fp32 result through the default split plan), bit-reproducible
    dw = torch.empty(cout_p, k, k, cin_p, device='cuda')
    ops.conv2d_wgrad(xa, gya, dw, k, k, 1, pad, dil)
    e_w = rel(dw[:Cout, :, :, :Cin].permute(0, 3, 1, 2), wr.grad)
    assert e_w < TOL32, (name, 'wgrad', e_w)
    dw2 = torch.empty_like(dw)
    ops.conv2d_wgrad(xa, gya, dw2, k, k, 1, pad, dil)
    assert torch.equal(dw, dw2)
    print('c3 half %s: fwd %.2e wgrad %.2e (of the peak; fp16 results within one rounding)' % (name, e_f, e_w))


@pytest.mark.parametrize('case', [c for c in CONV_CASES if c[1] % 32 == 0 and c[4] % 8 == 0])
def test_persistent_gather_matches_gather_h(ops, case, monkeypatch):
    """gather_hp_kernel (persistent blocks, continuous operand ring, transposed fp16 epilogue patch) against gather_h_kernel on
    the same launch: same K-step order, same fp32 accumulation, one rounding to fp16 -- the results must be BIT-IDENTICAL, forward
    (with the fused statistics) and data gradient, also where the planner would not pick the persistent kernel
    (PSEG_HCONV_PERSIST=2: everywhere it is valid; deep contractions, dilated convs run dense) and on the alternative block
    shapes (PSEG_HCONV_TILE: 128x128 on four waves, 256x128 on eight)."""
    from pytorch_segmentation_amd import _lib
    B, Cin, H, W, Cout, k, stride, pad, dil = case
    x, w, b, xa, w_h, b_raw, cin_p, cout_p, Ho, Wo = _setup(ops, case, False)
    gy = h(fill.uniform('hpers/' + '_'.join(map(str, case)), (B, Cout, Ho, Wo)))
    gya = to_act_h(ops, gy, cout_p)
    wT_h = w_h.view(cout_p, k * k, cin_p).permute(2, 1, 0).contiguous()

    def run():
        y = ops.Act.empty(B, Ho, Wo, cout_p, 'cuda', dtype=torch.float16)
        st = ops.conv2d_fwd(xa, w_h, None, y, k, k, stride, pad, dil, want_stats=True)
        co = ops.bn_finalize(st, y.M, None, None, None, None, 0.0, 1e-5)
        dx = ops.Act.empty(B, H, W, cin_p, 'cuda', dtype=torch.float16)
        ops.conv2d_dgrad(gya, wT_h, dx, k, k, stride, pad, dil)
        ran.append(_lib.load().pseg_debug_last_conv_kernel())       # 21 gather_h_kernel, 22 gather_hp_kernel
        return y.t.clone(), dx.t.clone(), co[0].clone(), co[1].clone()

    monkeypatch.setenv('PSEG_CONV_NOSKIP', '1')          # (both kernels dense: the persistent one has no tap skipping)
    res = {}
    ran = []
    for tile in ('0', '1', '2'):
        for persist in ('0', '2'):
            monkeypatch.setenv('PSEG_HCONV_PERSIST', persist)
            monkeypatch.setenv('PSEG_HCONV_TILE', tile)
            _lib.clear_query_cache()
            res[(tile, persist)] = run()
            if tile == '0':
                assert ran[-1] == (22 if persist == '2' else 21), (persist, ran)
    for env in ('PSEG_HCONV_PERSIST', 'PSEG_HCONV_TILE', 'PSEG_CONV_NOSKIP'):
        monkeypatch.delenv(env)
    _lib.clear_query_cache()
    ref = res[('0', '0')]
    assert_half_rounded(ops.Act(ref[0], B, Ho, Wo, cout_p, cout_p).to_nchw(Cout),
                        F.conv2d(x.double(), w.double(), None, stride, pad, dil), 'gather_h')
    for key, got in res.items():
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), key
        # (statistics: the same values summed per wave-row group -- another tile shape groups other rows)
        assert rel(got[2], ref[2]) < 1e-5 and rel(got[3], ref[3]) < 1e-5, key



# B, C (channels of the BatchNorm layer = dx), H, W, C2 (channels of dy), k, stride, pad, dil, act, persist -- the fp16 twin of
# tests/test_ops_gpu.py::BNSTAT_CASES.  persist: PSEG_HCONV_PERSIST (0: gather_h_kernel's 32-row patch epilogue, 2: the
# persistent kernel's transposed patch wherever it is valid)
BNSTAT_CASES_H = [
    (4, 64, 32, 32, 256, 1, 1, 0, 1, 1, '0'),       # bottleneck conv3 on layer-1 widths
    (4, 64, 32, 32, 256, 1, 1, 0, 1, 1, '2'),
    (2, 64, 64, 64, 64, 3, 1, 1, 1, 1, '0'),        # bottleneck conv2 (3x3)
    (2, 64, 64, 64, 64, 3, 1, 1, 1, 1, '2'),
    (2, 256, 64, 64, 128, 1, 1, 0, 1, 2, '2'),      # 128x128 tiles, ReLU6
    (8, 256, 64, 64, 1024, 1, 1, 0, 1, 1, '0'),     # layer-3 conv3: the long K-step
    (16, 32, 64, 64, 32, 3, 1, 1, 1, 0, '0'),       # 32 channels (HRNet's first branch): 128x32 tiles, no activation
    (2, 128, 30, 26, 128, 3, 1, 1, 1, 1, '0'),      # rows % 128 != 0: the last tile is ragged
    (2, 512, 16, 16, 512, 3, 1, 2, 2, 1, '0'),      # layer 4 (dilation 2)
]
# ... and problems whose kernel has no instantiation with the sums: rows == 0, the plain data gradient, the reduction pass stays
BNSTAT_UNCOVERED_H = [
    (8, 128, 64, 64, 128, 3, 2, 1, 1),              # stride-2 3x3: the tap-skipping instantiation
    (2, 40, 20, 24, 72, 3, 1, 1, 1),                # channels % 32 != 0: per-lane (tap, channel) addressing
    (2, 256, 32, 32, 64, 3, 1, 6, 6),               # dilated with dead taps
]


def test_fused_batchnorm_backward_sums_half_uncovered(ops, monkeypatch):
    monkeypatch.setattr(ops, 'FUSE_BN_BWD_H', True)
    for B, C, H, W, C2, k, stride, pad, dil in BNSTAT_UNCOVERED_H:
        Ho, Wo = ops.conv_out_size(H, k, stride, pad, dil), ops.conv_out_size(W, k, stride, pad, dil)
        Cp, C2p = r8(C), r8(C2)
        assert ops._lib.query('pseg_conv2d_dgrad_bnstat_rows_h', B, H, W, Cp, Ho, Wo, C2p, k, k, stride, pad, dil) == 0
        key = 'hbnstat/un_%d_%d' % (C, k)
        ya = to_act_h(ops, h(fill.uniform(key + '/y', (B, C, H, W))), Cp)
        co = ops.bn_finalize(ops.col_stats(ya), ya.M, torch.ones(Cp).cuda(), torch.zeros(Cp).cuda(), torch.zeros(Cp).cuda(),
                             torch.ones(Cp).cuda(), 0.1, 1e-5)
        w_h = krsc(h(fill.uniform(key + '/w', (C2, C, k, k), 0.05)), C2p, Cp).cuda().half()
        wT_h = w_h.view(C2p, k * k, Cp).permute(2, 1, 0).contiguous()
        gya = to_act_h(ops, h(fill.uniform(key + '/gy', (B, C2, Ho, Wo))), C2p)
        dx0 = ops.Act.empty(B, H, W, Cp, 'cuda', dtype=torch.float16)
        ops.conv2d_dgrad(gya, wT_h, dx0, k, k, stride, pad, dil)
        dx1 = ops.Act.empty(B, H, W, Cp, 'cuda', dtype=torch.float16)
        ops.conv2d_dgrad(gya, wT_h, dx1, k, k, stride, pad, dil, bn=(ya, co, 1))
        assert dx1.bnpart is None and torch.equal(dx0.t, dx1.t)
        # the entry point itself refuses instead of running a kernel without the epilogue
        part = torch.empty(2, 8, Cp, device='cuda')
        with pytest.raises(ops._lib.PsegError):
            ops._lib.call('pseg_conv2d_dgrad_bnstat_h', gya.ptr, gya.ld, wT_h.data_ptr(), dx1.ptr, dx1.ld, B, H, W, Cp, Ho, Wo, C2p,
                          k, k, stride, pad, dil, ya.ptr, ya.ld, co[0].data_ptr(), co[1].data_ptr(), co[2].data_ptr(),
                          co[3].data_ptr(), 1, part.data_ptr(), part[1].data_ptr(), 8, 0)


@pytest.mark.parametrize('case', BNSTAT_CASES_H)
def test_dgrad_with_fused_batchnorm_backward_sums_half(ops, case, monkeypatch):
    """pseg_conv2d_dgrad_bnstat_h: the fp16 data gradient of a conv whose input is act(BN(y)) also writes that layer's backward
    partial sums.  dx must be BIT-identical to the plain data gradient; the sums must equal fp64 sum(dz * act') / sum(dz * act' *
    xhat) taken over dx as stored, and the BatchNorm backward must give the same dy / dgamma / dbeta with and without them."""
    from pytorch_segmentation_amd import _lib
    B, C, H, W, C2, k, stride, pad, dil, act, persist = case
    key = 'hbnstat/' + '_'.join(map(str, case[:-1]))
    monkeypatch.setenv('PSEG_HCONV_PERSIST', persist)
    monkeypatch.setattr(ops, 'FUSE_BN_BWD_H', True)        # (opt-in under the half policy: ops.py)
    _lib.clear_query_cache()
    try:
        Ho, Wo = ops.conv_out_size(H, k, stride, pad, dil), ops.conv_out_size(W, k, stride, pad, dil)
        Cp, C2p = r8(C), r8(C2)
        rows = ops._lib.query('pseg_conv2d_dgrad_bnstat_rows_h', B, H, W, Cp, Ho, Wo, C2p, k, k, stride, pad, dil)
        assert rows > 0
        y = h(fill.uniform(key + '/y', (B, C, H, W), 2.0) + fill.uniform(key + '/off', (1, C, 1, 1), 1.0))
        g = 1.0 + fill.uniform(key + '/g', (C,), 0.3)
        b = fill.uniform(key + '/b', (C,), 0.3)
        w2 = h(fill.uniform(key + '/w2', (C2, C, k, k), 0.05))
        gy2 = h(fill.uniform(key + '/gy2', (B, C2, Ho, Wo)))
        ya = to_act_h(ops, y, Cp)
        gp, bp = torch.ones(Cp), torch.zeros(Cp)
        gp[:C], bp[:C] = g, b
        rm, rv = torch.zeros(Cp).cuda(), torch.ones(Cp).cuda()
        co = ops.bn_finalize(ops.col_stats(ya), ya.M, gp.cuda(), bp.cuda(), rm, rv, 0.1, 1e-5)
        w_h = krsc(w2, C2p, Cp).cuda().half()
        wT_h = w_h.view(C2p, k * k, Cp).permute(2, 1, 0).contiguous()
        gya = to_act_h(ops, gy2, C2p)
        dx_plain = ops.Act.empty(B, H, W, Cp, 'cuda', dtype=torch.float16)
        ops.conv2d_dgrad(gya, wT_h, dx_plain, k, k, stride, pad, dil)
        assert dx_plain.bnpart is None
        dx = ops.Act.empty(B, H, W, Cp, 'cuda', dtype=torch.float16)
        ops.conv2d_dgrad(gya, wT_h, dx, k, k, stride, pad, dil, bn=(ya, co, act))
        assert dx.bnpart is not None and dx.bnpart.rows == rows and dx.bnpart.key == ya.ptr
        assert torch.equal(dx.t, dx_plain.t)
        # the sums against fp64 over the STORED dx
        dz = dx.to_nchw(C).double().cpu()
        mu, istd = (co[i][:C].double().cpu().view(1, C, 1, 1) for i in range(2))
        pre = (y.cuda() - co[0][:C].view(1, C, 1, 1)) * co[2][:C].view(1, C, 1, 1) + co[3][:C].view(1, C, 1, 1)   # fp32, as the kernels decide
        on = (pre > 0) if act == 1 else (((pre > 0) & (pre < 6)) if act == 2 else torch.ones_like(pre, dtype=torch.bool))
        gm = dz * on.cpu().double()
        xh = (y.double() - mu) * istd
        db_ref, dg_ref = gm.sum((0, 2, 3)), (gm * xh).sum((0, 2, 3))
        part = dx.bnpart.part.double().cpu()
        scale_b = gm.abs().sum((0, 2, 3)).max().item() + 1e-30
        scale_g = (gm * xh).abs().sum((0, 2, 3)).max().item() + 1e-30
        assert (part[0].sum(0)[:C] - db_ref).abs().max().item() < 1e-5 * scale_b
        assert (part[1].sum(0)[:C] - dg_ref).abs().max().item() < 1e-5 * scale_g
        if Cp > C:
            assert part[:, :, C:].abs().max().item() == 0.0          # padded channels: dx is zero there
        # through the BatchNorm backward: fused and unfused chains
        dg1, db1, dy1 = torch.zeros(Cp).cuda(), torch.zeros(Cp).cuda(), ya.like()
        ops.bn_act_bwd(dx_plain, None, ya, co, act, dy1, dg1, db1)
        dg2, db2, dy2 = torch.zeros(Cp).cuda(), torch.zeros(Cp).cuda(), ya.like()
        ops.bn_act_bwd(dx, None, ya, co, act, dy2, dg2, db2, part=dx.bnpart)
        assert (dg2[:C].double().cpu() - dg_ref).abs().max().item() < 1e-5 * scale_g
        assert (db2[:C].double().cpu() - db_ref).abs().max().item() < 1e-5 * scale_b
        assert rel(dg2, dg1) < 1e-5 and (db2 - db1).abs().max().item() < 1e-5 * scale_b
        # (dy is rounded to fp16 from coefficients that differ in the last fp32 bits: an fp16 ulp apart at most)
        d12 = (dy2.to_nchw(C).double() - dy1.to_nchw(C).double()).abs().cpu()
        assert (d12 <= dy1.to_nchw(C).double().abs().cpu() * 2.0 ** -10 + 1e-5 * dy1.to_nchw(C).abs().max().item()).all()
        # bit-reproducible
        dxb = ops.Act.empty(B, H, W, Cp, 'cuda', dtype=torch.float16)
        ops.conv2d_dgrad(gya, wT_h, dxb, k, k, stride, pad, dil, bn=(ya, co, act))
        assert torch.equal(dxb.bnpart.part, dx.bnpart.part)
        # accumulate: the plain path, no partial sums
        dxc = dx_plain.like()
        ops.copy2d(dx_plain, dxc)
        ops.conv2d_dgrad(gya, wT_h, dxc, k, k, stride, pad, dil, accumulate=True, bn=(ya, co, act))
        assert dxc.bnpart is None
    finally:
        monkeypatch.delenv('PSEG_HCONV_PERSIST')
        _lib.clear_query_cache()


# ---------------------------------------------------------------------------------------------- bandwidth-bound passes
def nchw_h(ops, a, C=None):
    return a.to_nchw(C)          # (fp16 -> fp32 is exact)


def test_image_to_half_nhwc8(ops):
    """fp32 NCHW image -> fp16 NHWC with 8 channels per pixel in one pass (pseg_nchw_to_nhwc_h): the values rounded once to
    fp16, the padding channels zero -- and equal to the two-pass route (generic transpose + conversion)."""
    for B, C, H, W in ((16, 3, 64, 48), (2, 1, 7, 5), (3, 8, 9, 9)):
        x = fill.uniform('img8/%d_%d' % (B, C), (B, C, H, W), 3.0).cuda()
        a = ops.Act.from_nchw(x, 8, dtype=torch.float16)
        assert a.half and a.C == 8 and a.ld == 8
        got = a.t.view(B, H, W, 8)
        want = torch.zeros(B, H, W, 8, dtype=torch.float16, device='cuda')
        want[..., :C] = x.permute(0, 2, 3, 1).to(torch.float16)
        assert torch.equal(got, want)
        two = ops.Act.from_nchw(x, 8).to(torch.float16)
        assert torch.equal(two.t.view(B, H, W, 8), got)


@pytest.mark.parametrize('B,C,H,W,act,res', [(4, 64, 16, 16, 1, False), (2, 256, 8, 8, 1, True), (16, 32, 1, 1, 1, False),
                                              (2, 96, 9, 7, 2, False), (2, 24, 12, 12, 0, True), (3, 128, 31, 17, 1, False),
                                              (4, 64, 128, 128, 1, False), (4, 256, 64, 64, 1, True)])
def test_batchnorm_train_half(ops, B, C, H, W, act, res):
    """BatchNorm (+ residual + activation) forward / backward on fp16 tensors: statistics, coefficients and sums in fp32.
    Reference: torch float64 on the same fp16-rounded tensors; fp16 results must be that value rounded once."""
    key = 'hbn/%d_%d_%d_%d_%d' % (B, C, H, W, act)
    y = h(fill.uniform(key + '/y', (B, C, H, W), 2.0) + fill.uniform(key + '/off', (1, C, 1, 1), 1.0))
    r = h(fill.uniform(key + '/r', (B, C, H, W), 1.0)) if res else None
    g = 1.0 + fill.uniform(key + '/g', (C,), 0.3)
    b = fill.uniform(key + '/b', (C,), 0.3)
    gz = h(fill.uniform(key + '/gz', (B, C, H, W)))
    bn = torch.nn.BatchNorm2d(C).double()
    with torch.no_grad():
        bn.weight.copy_(g)
        bn.bias.copy_(b)
    bn.train()
    yr = y.double().requires_grad_()
    rr = r.double().requires_grad_() if res else None
    t = bn(yr)
    if res:
        t = t + rr
    zr = F.relu(t) if act == 1 else (F.relu6(t) if act == 2 else t)
    zr.backward(gz.double())

    ya = to_act_h(ops, y, C)
    rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda()
    co = ops.bn_finalize(ops.col_stats(ya), ya.M, g.cuda(), b.cuda(), rm, rv, 0.1, 1e-5)
    za = ya.like()
    ra = to_act_h(ops, r, C) if res else None
    use_mask = bool(res and act and C % 32 == 0)
    mask = ops.bn_act_fwd(ya, co, act, za, residual=ra, want_mask=use_mask)
    assert za.half
    assert_half_rounded(za.to_nchw(), zr, 'bn fwd')
    assert rel(rm, bn.running_mean) < 1e-4 and rel(rv, bn.running_var) < 1e-4
    dg, db = torch.zeros(C).cuda(), torch.zeros(C).cuda()
    dya = ya.like()
    dra = ya.like() if res else None
    # (a residual layer keeps z: the small-tensor path reads it; the large-tensor passes read the bitmask instead)
    ops.bn_act_bwd(to_act_h(ops, gz, C), za if res else None, ya, co, act, dya, dg, db, dres=dra, mask=mask)
    assert_half_rounded(dya.to_nchw(), yr.grad, 'bn bwd dy')
    assert rel(dg, bn.weight.grad) < 1e-4 and rel(db, bn.bias.grad) < 1e-4
    if res:
        assert_half_rounded(dra.to_nchw(), rr.grad, 'bn bwd dres')


@pytest.mark.parametrize('act', [1, 2])
def test_activation_bitmask_is_the_mask_of_the_stored_value(ops, act):
    """The activation bitmask a residual BatchNorm layer keeps for backward must be the mask of z AS STORED in fp16: a
    pre-activation below half the smallest fp16 subnormal is stored as 0 (one just under 6 as 6), and the backward passes that
    read z instead of the bitmask (the fused small-tensor kernels of an eager step) see it closed.  (Round 4: a replayed HRNet
    -mp step -- bitmask -- and an eager one -- z -- parted at step 133 of 300 over one such element.)"""
    C, M = 32, 64
    dev = 'cuda'
    y = torch.ones(1, C, 8, 8)
    ya = to_act_h(ops, h(y), C)
    # per-channel scale: v = (1 - 0) * scale + 0 (+ residual 0)
    scale = torch.zeros(C)
    scale[0], scale[1], scale[2], scale[3] = 1e-8, 1e-7, 5.9999, 0.5       # -> fp16 0, 1.19e-7, 6.0, 0.5
    co = torch.zeros(4, C, device=dev)
    co[1] = 1.0
    co[2] = scale.to(dev)
    ra = to_act_h(ops, h(torch.zeros(1, C, 8, 8)), C)
    za = ya.like()
    mask = ops.bn_act_fwd(ya, co, act, za, residual=ra, want_mask=True)
    z = za.to_nchw().float().cpu()
    bits = mask.view(M, C // 32).cpu()
    for c in range(4):
        zc = z[0, c, 0, 0].item()
        open_z = (zc > 0.0) if act == 1 else (0.0 < zc < 6.0)
        open_bit = bool((int(bits[0, 0].item()) >> c) & 1)
        assert open_bit == open_z, (c, zc, open_bit)
    assert z[0, 0, 0, 0].item() == 0.0 and z[0, 1, 0, 0].item() > 0.0
    if act == 2:
        assert z[0, 2, 0, 0].item() == 6.0


def test_batchnorm_eval_and_act_bwd_half(ops):
    B, C, H, W = 2, 64, 12, 10
    y = h(fill.uniform('hbne/y', (B, C, H, W), 2.0))
    gz = h(fill.uniform('hbne/gz', (B, C, H, W)))
    g, b = 1.0 + fill.uniform('hbne/g', (C,), 0.3), fill.uniform('hbne/b', (C,), 0.3)
    rm, rv = fill.uniform('hbne/rm', (C,), 0.5), 0.5 + fill.uniform('hbne/rv', (C,), 0.4).abs()
    yr = y.double().requires_grad_()
    zr = F.relu(F.batch_norm(yr, rm.double(), rv.double(), g.double(), b.double(), False, 0.0, 1e-5))
    zr.backward(gz.double())
    ya = to_act_h(ops, y, C)
    co = ops.bn_eval_coeffs(g.cuda(), b.cuda(), rm.cuda(), rv.cuda(), 1e-5)
    za = ya.like()
    ops.bn_act_fwd(ya, co, 1, za)
    assert_half_rounded(za.to_nchw(), zr, 'bn eval fwd')
    dya = ya.like()
    ops.act_bwd(to_act_h(ops, gz, C), za, 1, dya, scale=co[2])
    assert_half_rounded(dya.to_nchw(), yr.grad, 'bn eval bwd')


def test_pool_resize_copy_half(ops):
    """pool_sum / broadcast / bilinear fwd + bwd / maxpool / copy2d / col_sum on fp16 tensors."""
    B, C, H, W = 2, 64, 12, 10
    x = h(fill.uniform('hpool/x', (B, C, H, W)))
    xa = to_act_h(ops, x, C)
    # pooled mean and its broadcast
    pa = xa.new(B, 1, 1, C)
    ops.pool_sum(xa, pa, 1.0 / (H * W))
    assert_half_rounded(pa.to_nchw(), x.double().mean((2, 3), keepdim=True), 'pool_sum')
    ba = xa.new(B, H, W, C)
    ops.broadcast(pa, ba)
    assert torch.equal(ba.to_nchw(), pa.to_nchw().expand(B, C, H, W))
    # bilinear, both conventions, forward and backward
    for ac in (True, False):
        Ho, Wo = 2 * H, 3 * W
        ya = xa.new(B, Ho, Wo, C)
        ops.bilinear_fwd(xa, ya, ac)
        xr = x.double().requires_grad_()
        ref = F.interpolate(xr, (Ho, Wo), mode='bilinear', align_corners=ac)
        assert_half_rounded(ya.to_nchw(), ref, 'bilinear fwd')
        gy = h(fill.uniform('hpool/gy%d' % ac, (B, C, Ho, Wo)))
        ref.backward(gy.double())
        dxa = xa.like()
        ops.bilinear_bwd(to_act_h(ops, gy, C), dxa, ac)
        assert_half_rounded(dxa.to_nchw(), xr.grad, 'bilinear bwd')
    # max pool 3x3 / 2 (ResNet stem)
    Hp, Wp = ops.conv_out_size(H, 3, 2, 1, 1), ops.conv_out_size(W, 3, 2, 1, 1)
    pa2 = xa.new(B, Hp, Wp, C)
    arg = ops.maxpool_fwd(xa, pa2, 3, 2, 1)
    xr = x.double().requires_grad_()
    ref = F.max_pool2d(xr, 3, 2, 1)
    assert torch.equal(pa2.to_nchw().double().cpu(), ref.detach())
    gp = h(fill.uniform('hpool/gp', (B, C, Hp, Wp)))
    ref.backward(gp.double())
    dxa = xa.like()
    ops.maxpool_bwd(to_act_h(ops, gp, C), arg, dxa, 3, 2, 1)
    assert_half_rounded(dxa.to_nchw(), xr.grad, 'maxpool bwd')
    # strided copy / add into a concat slice, column sums
    wide = xa.new(B, H, W, 2 * C, zero=True)
    ops.copy2d(xa, wide.slice(C, 2 * C))
    ops.copy2d(xa, wide.slice(C, 2 * C), accumulate=True)
    assert_half_rounded(wide.to_nchw()[:, C:], 2 * x.double(), 'copy2d')
    assert wide.to_nchw()[:, :C].abs().max().item() == 0.0
    out = torch.zeros(C, device='cuda')
    ops.col_sum(xa, out)
    assert rel(out, x.double().sum((0, 2, 3))) < 1e-5


@pytest.mark.parametrize('B,C,H,W,stride', [(2, 32, 16, 16, 1), (2, 96, 17, 15, 2), (1, 144, 9, 9, 1)])
def test_depthwise_half(ops, B, C, H, W, stride):
    key = 'hdw/%d_%d_%d' % (C, H, stride)
    x = h(fill.uniform(key + '/x', (B, C, H, W)))
    w = fill.uniform(key + '/w', (C, 1, 3, 3), 0.3)              # the depthwise filter stays fp32
    Ho, Wo = ops.conv_out_size(H, 3, stride, 1, 1), ops.conv_out_size(W, 3, stride, 1, 1)
    gy = h(fill.uniform(key + '/gy', (B, C, Ho, Wo)))
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    ref = F.conv2d(xr, wr, None, stride, 1, 1, C)
    ref.backward(gy.double())
    xa, gya = to_act_h(ops, x, C), to_act_h(ops, gy, C)
    w_raw = w[:, 0].permute(1, 2, 0).contiguous().cuda()         # [kh][kw][C]
    ya = xa.new(B, Ho, Wo, C)
    ops.dwconv_fwd(xa, w_raw, ya, 3, stride, 1)
    assert_half_rounded(ya.to_nchw(), ref, 'dw fwd')
    dxa = xa.like()
    ops.dwconv_dgrad(gya, w_raw, dxa, 3, stride, 1)
    assert_half_rounded(dxa.to_nchw(), xr.grad, 'dw dgrad')
    dw = torch.empty_like(w_raw)
    ops.dwconv_wgrad(xa, gya, dw, 3, stride, 1)
    assert rel(dw.permute(2, 0, 1).unsqueeze(1), wr.grad) < TOL32


def test_filter_prepare_and_convert(ops):
    """fp16 filter copies (plain + transposed, padded to 8-channel granules) from an fp32 arena; strided conversions with
    the device-resident loss scale."""
    import torch.nn as nn
    import pytorch_segmentation_amd as pseg
    from pytorch_segmentation_amd.nn import Conv2d
    m = nn.Sequential(Conv2d(3, 64, 7, 2, 3, bias=False), Conv2d(64, 21, 3, padding=1), Conv2d(64, 64, 3, groups=64, padding=1),
                      Conv2d(40, 2, 1))
    ar = pseg.prepare(m, 'cuda')
    ar.prepare_half()
    for conv in (m[0], m[1], m[3]):
        co, ci, kh, kw = conv.weight.shape
        cop, cip = r8(co), r8(ci)
        ref = torch.zeros(cop, kh, kw, cip)
        ref[:co, :, :, :ci] = conv.weight.detach().cpu().permute(0, 2, 3, 1)
        ref = ref.half()
        assert torch.equal(conv._w_h_view.view(cop, kh, kw, cip).cpu(), ref)
        assert torch.equal(conv._wT_h_view.view(cip, kh * kw, cop).cpu(), ref.view(cop, kh * kw, cip).permute(2, 1, 0))
    assert not hasattr(m[2], '_w_h_view')                        # depthwise filters stay fp32
    x = fill.uniform('hconvt/x', (2, 24, 9, 7))
    xa = ops.Act.from_nchw(x.cuda(), 24)
    s = torch.tensor([1024.0], device='cuda')
    xh = xa.to(torch.float16, scale=s)
    assert xh.half and torch.equal(xh.to_nchw().cpu(), (x * 1024.0).half().float())
    back = xh.to(torch.float32)
    assert not back.half and torch.equal(back.to_nchw().cpu(), (x * 1024.0).half().float())


@pytest.mark.parametrize('adam', [False, True])
def test_loss_scaled_optimiser(ops, adam):
    """The -mp optimiser protocol on the device: gradients arrive multiplied by the loss scale S; a clean step equals the
    fp32 step on grad / S; a step with an inf / nan gradient changes NOTHING and halves S; S doubles after growth_interval
    clean steps; the Adam bias correction counts applied steps only."""
    from pytorch_segmentation_amd import _lib
    n = 10007
    p0 = fill.uniform('hopt/p', (n,))
    gs = [fill.uniform('hopt/g%d' % i, (n,)) for i in range(5)]
    ref = p0.clone().double().requires_grad_()
    opt = torch.optim.Adam([ref], lr=1e-2) if adam else torch.optim.SGD([ref], lr=1e-2, momentum=0.9)
    p = p0.clone().cuda()
    m, v = torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    state = torch.zeros(8, device='cuda')
    _lib.call('pseg_mp_state_init', state.data_ptr(), 1024.0, None)

    def step(g_scaled):
        st = ops._stream()
        _lib.call('pseg_mp_check', g_scaled.data_ptr(), n, state.data_ptr(), st)
        if adam:
            _lib.call('pseg_adam_step_mp', p.data_ptr(), g_scaled.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-2, 0.9, 0.999,
                      1e-8, 0.0, 0, 1.0, state.data_ptr(), st)
        else:
            _lib.call('pseg_sgd_step_mp', p.data_ptr(), g_scaled.data_ptr(), m.data_ptr(), n, 1e-2, 0.9, 0.0, 0, 1.0,
                      state.data_ptr(), st)
        _lib.call('pseg_mp_update', state.data_ptr(), 2.0, 0.5, 3, 1.0, float(2 ** 24), st)

    scale = 1024.0
    applied = 0
    for i, g in enumerate(gs):
        if i == 1:                                   # an overflowed step first: nothing may change
            bad = (g * scale).cuda()
            bad[n // 2] = float('inf') if not adam else float('nan')
            before = (p.clone(), m.clone(), v.clone())
            step(bad)
            assert torch.equal(p, before[0]) and torch.equal(m, before[1]) and torch.equal(v, before[2])
            s = state.cpu().tolist()
            assert s[0] == scale * 0.5 and s[3] == 0.0 and s[5] == 1.0 and s[2] == 0.0
        scale = state[0].item()                      # (what the loss gradient is multiplied by in the step that follows)
        step((g * scale).cuda())
        applied += 1
        ref.grad = g.double()
        opt.step()
        assert rel(p, ref.detach()) < 1e-5, i
    s = state.cpu().tolist()
    # 5 clean steps with growth_interval 3, one skipped after the first: 1024 -> 512 (skip) -> 1024 (3 clean) ...
    assert s[4] == applied == 5 and s[5] == 1.0
    assert s[0] == 1024.0 and s[1] == 1.0 / 1024.0
