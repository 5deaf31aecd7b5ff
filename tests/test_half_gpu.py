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
