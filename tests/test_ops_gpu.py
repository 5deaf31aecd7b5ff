"""Op-level parity of every C-ABI entry point against stock torch CPU fp32 ops (the arithmetic the reference
dispatches).  Tolerance: 1e-3 relative to the tensor's max magnitude is the contract (BASELINE.json north_star);
fp32 MFMA + fp32 accumulate lands around 1e-6..1e-5, so the asserts use TOL = 1e-4 to catch indexing bugs that a
loose bound would hide.  Integer outputs (argmax, counts) are bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import fill

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    from pytorch_segmentation_amd import ops as _ops
    return _ops


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def to_act(ops, x, Cpad=None):
    return ops.Act.from_nchw(x.cuda(), Cpad)


def krsc(w, cout_pad=None, cin_pad=None):
    """OIHW cpu weight -> raw [Cout_p][kh][kw][Cin_p] cuda tensor."""
    Cout, Cin, kh, kw = w.shape
    cout_pad = cout_pad or Cout
    cin_pad = cin_pad or Cin
    raw = torch.zeros(cout_pad, kh, kw, cin_pad)
    raw[:Cout, :, :, :Cin] = w.permute(0, 2, 3, 1)
    return raw.contiguous().cuda()


@pytest.mark.parametrize('B,C,H,W,Cpad', [(2, 3, 16, 20, 4), (2, 21, 8, 8, 24), (1, 64, 7, 9, 64), (3, 40, 5, 5, 40)])
def test_layout_roundtrip(ops, B, C, H, W, Cpad):
    x = fill.uniform('layout/%d' % C, (B, C, H, W))
    a = to_act(ops, x, Cpad)
    v = a.view4().cpu()
    assert torch.equal(v[..., :C], x.permute(0, 2, 3, 1))
    assert torch.equal(v[..., C:], torch.zeros(B, H, W, Cpad - C))
    assert torch.equal(a.to_nchw(C).cpu(), x)


CONV_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, dil
    (2, 64, 16, 16, 128, 1, 1, 0, 1),
    (2, 32, 20, 20, 64, 3, 1, 1, 1),
    (2, 64, 24, 24, 32, 3, 1, 6, 6),
    (2, 64, 24, 24, 32, 3, 1, 12, 12),
    (2, 64, 24, 24, 32, 3, 1, 18, 18),
    (2, 32, 33, 35, 64, 3, 2, 1, 1),
    (2, 64, 16, 16, 128, 1, 2, 0, 1),
    (2, 3, 64, 64, 64, 7, 2, 3, 1),
    (2, 384, 32, 32, 21, 3, 1, 1, 1),
    (2, 2048, 8, 8, 256, 3, 1, 6, 6),
    (2, 1280, 4, 4, 256, 3, 1, 1, 1),
    (4, 256, 1, 1, 64, 1, 1, 0, 1),
    (2, 88, 16, 16, 2, 3, 1, 1, 1),
    (1, 160, 9, 9, 64, 3, 1, 1, 1),
    (2, 512, 8, 8, 512, 3, 1, 2, 2),
    (1, 16, 130, 130, 16, 3, 1, 1, 1),
    (4, 256, 32, 32, 128, 1, 1, 0, 1),
    (4, 320, 4, 4, 1280, 1, 1, 0, 1),
    (4, 256, 8, 8, 1024, 1, 1, 0, 1),
    (2, 128, 48, 48, 128, 3, 1, 1, 1),
    (4, 16, 128, 128, 32, 3, 1, 1, 1),
    (2, 128, 32, 32, 64, 3, 1, 18, 18),   # ASPP geometry: whole (tile, tap) pairs are padding -> skipped K-steps
    (2, 128, 32, 32, 64, 3, 1, 12, 12),
    (3, 256, 16, 16, 128, 3, 1, 6, 6),
    (2, 64, 32, 32, 64, 3, 2, 1, 1),      # stride-2 dgrad with parity-class row order
    (2, 64, 32, 32, 128, 1, 2, 0, 1),
    (1, 32, 64, 48, 96, 3, 2, 1, 1),
    (2, 32, 64, 64, 32, 3, 1, 1, 1),      # HRNet's 32-channel branch: the 32 x 256 weight-gradient tile on the LDS-DMA kernel
    (2, 40, 9, 11, 24, 3, 1, 1, 1),       # ... and off it (a map that does not tile into 32-pixel patches): wgrad_kernel<32, 256>
    (2, 64, 32, 32, 32, 3, 2, 1, 1),      # stride-2 fusion conv of HRNet (32-row tile, strided gather)
    (2, 32, 9, 11, 32, 3, 1, 1, 1),       # K = 288 as ONE 32 x 288 tile on nine waves, row-major pixels (99 pixels per image)
    (8, 32, 32, 32, 32, 3, 1, 1, 1),      # ... and in patch order, with pixel splits
]


def _conv_setup(ops, case, with_bias):
    B, Cin, H, W, Cout, k, stride, pad, dil = case
    key = 'conv/' + '_'.join(map(str, case))
    x = fill.uniform(key + '/x', (B, Cin, H, W))
    w = fill.uniform(key + '/w', (Cout, Cin, k, k), (6.0 / (Cin * k * k)) ** 0.5)
    b = fill.uniform(key + '/b', (Cout,), 0.5) if with_bias else None
    cin_p, cout_p = (Cin + 3) // 4 * 4, (Cout + 3) // 4 * 4
    xa = to_act(ops, x, cin_p)
    w_raw = krsc(w, cout_p, cin_p)
    b_raw = None
    if with_bias:
        b_raw = torch.zeros(cout_p)
        b_raw[:Cout] = b
        b_raw = b_raw.cuda()
    Ho, Wo = ops.conv_out_size(H, k, stride, pad, dil), ops.conv_out_size(W, k, stride, pad, dil)
    return x, w, b, xa, w_raw, b_raw, cin_p, cout_p, Ho, Wo


# split-bf16 (three-pass) conv arithmetic: ~17 significant bits per product -> 1e-5-ish; asserted at 2e-4, five times
# inside the 1e-3 contract
PREC_TOL = {'fp32': TOL, 'bf16x3': 2e-4, 'bf16x6': TOL, 'fp16x3': TOL}


@pytest.mark.parametrize('prec', ['fp32', 'bf16x3', 'bf16x6', 'fp16x3'])
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv2d_fwd(ops, case, prec):
    TOL = PREC_TOL[prec]
    P = ops._PREC_NAMES[prec]
    B, Cin, H, W, Cout, k, stride, pad, dil = case
    with_bias = Cout in (21, 2)
    x, w, b, xa, w_raw, b_raw, cin_p, cout_p, Ho, Wo = _conv_setup(ops, case, with_bias)
    ref = F.conv2d(x, w, b, stride, pad, dil)
    ya = ops.Act.empty(B, Ho, Wo, cout_p, 'cuda')
    am = dict(amax_x=ops.amax_of(xa), amax_w=ops.amax_of(w_raw)) if prec == 'fp16x3' else {}
    stats = ops.conv2d_fwd(xa, w_raw, b_raw, ya, k, k, stride, pad, dil, want_stats=not with_bias, precision=P, **am)
    got = ya.to_nchw(Cout)
    assert rel(got, ref) < TOL
    if cout_p > Cout:
        assert ya.view4()[..., Cout:].abs().max().item() == 0.0
    if stats is not None:
        # fused epilogue statistics -> mean / invstd must match the two-pass fp64 values
        co = ops.bn_finalize(stats, ya.M, None, None, None, None, 0.0, 1e-5)
        r64 = ref.double()
        mu, var = r64.mean((0, 2, 3)), r64.var((0, 2, 3), unbiased=False)
        assert rel(co[0][:Cout], mu) < TOL * max(1.0, (var.sqrt().max() / (mu.abs().max() + 1e-30)).item())
        assert rel(co[1][:Cout], 1.0 / (var + 1e-5).sqrt()) < TOL
    # accumulate: y += conv
    ops.conv2d_fwd(xa, w_raw, b_raw, ya, k, k, stride, pad, dil, accumulate=True, precision=P, **am)
    assert rel(ya.to_nchw(Cout), 2 * ref) < TOL


@pytest.mark.parametrize('prec', ['fp32', 'bf16x3', 'bf16x6', 'fp16x3'])
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv2d_dgrad_wgrad(ops, case, prec):
    TOL = PREC_TOL[prec]
    P = ops._PREC_NAMES[prec]
    B, Cin, H, W, Cout, k, stride, pad, dil = case
    x, w, b, xa, w_raw, b_raw, cin_p, cout_p, Ho, Wo = _conv_setup(ops, case, False)
    key = 'convg/' + '_'.join(map(str, case))
    gy = fill.uniform(key, (B, Cout, Ho, Wo))
    xr = x.clone().requires_grad_()
    wr = w.clone().requires_grad_()
    F.conv2d(xr, wr, None, stride, pad, dil).backward(gy)
    gya = to_act(ops, gy, cout_p)
    # dgrad
    wT = ops.filter_transpose(w_raw, cout_p, k * k, cin_p)
    ref_wT = w_raw.view(cout_p, k * k, cin_p).permute(2, 1, 0).contiguous()
    assert torch.equal(wT.view(cin_p, k * k, cout_p), ref_wT)
    dxa = ops.Act.empty(B, H, W, cin_p, 'cuda')
    am = dict(amax_dy=ops.amax_of(gya), amax_w=ops.amax_of(wT)) if prec == 'fp16x3' else {}
    ops.conv2d_dgrad(gya, wT, dxa, k, k, stride, pad, dil, precision=P, **am)
    assert rel(dxa.to_nchw(Cin), xr.grad) < TOL
    ops.conv2d_dgrad(gya, wT, dxa, k, k, stride, pad, dil, accumulate=True, precision=P, **am)
    if prec == 'fp16x3':
        P = ops.PREC_BF16X3  # the weight gradient has no fp16-limb variant
        TOL = PREC_TOL['bf16x3']
    assert rel(dxa.to_nchw(Cin), 2 * xr.grad) < TOL
    # wgrad
    dw = torch.empty_like(w_raw)
    ops.conv2d_wgrad(xa, gya, dw, k, k, stride, pad, dil, precision=P)
    got = dw.view(cout_p, k, k, cin_p)[:Cout, :, :, :Cin].permute(0, 3, 1, 2).cpu()
    assert rel(got, wr.grad) < TOL
    ops.conv2d_wgrad(xa, gya, dw, k, k, stride, pad, dil, accumulate=True, precision=P)
    got2 = dw.view(cout_p, k, k, cin_p)[:Cout, :, :, :Cin].permute(0, 3, 1, 2).cpu()
    assert rel(got2, 2 * wr.grad) < TOL
    # bit-reproducible
    dw2 = torch.empty_like(w_raw)
    ops.conv2d_wgrad(xa, gya, dw2, k, k, stride, pad, dil, precision=P)
    dw3 = torch.empty_like(w_raw)
    ops.conv2d_wgrad(xa, gya, dw3, k, k, stride, pad, dil, precision=P)
    assert torch.equal(dw2, dw3)


# BASELINE.json configs[2] shapes (DeepLabV3+ R50, 512x512, batch 16): the launches bench.py actually times.  One CPU
# reference (stock torch fp32 conv forward + backward, seconds on the box's cores) per shape, then forward, data gradient
# and weight gradient under every arithmetic variant the policies use, through the default plan (no PSEG_* overrides):
# the tap-skipping 128x64 forward tiles, the 256x128 / 8-wave limb data-gradient tile and the 16384-pixel weight-gradient
# split meet the oracle here at full size.
C3_CASES = [
    # name, B, Cin, H, W, Cout, k, pad, dil, bias
    ('aspp_d6', 16, 2048, 32, 32, 256, 3, 6, 6, False),      # reference models/aspp.py:28-29, rates models/deeplabv3plus.py:21
    ('aspp_d12', 16, 2048, 32, 32, 256, 3, 12, 12, False),
    ('aspp_d18', 16, 2048, 32, 32, 256, 3, 18, 18, False),
    ('aspp_1x1', 16, 2048, 32, 32, 256, 1, 0, 1, False),     # models/aspp.py:27
    ('aspp_project', 16, 1280, 32, 32, 256, 1, 0, 1, False), # models/aspp.py:30
    ('project', 16, 256, 128, 128, 128, 1, 0, 1, False),     # models/deeplabv3plus.py:20
    ('cls_conv', 16, 384, 128, 128, 21, 3, 1, 1, True),      # models/deeplabv3plus.py:22
]


@pytest.mark.parametrize('case', C3_CASES, ids=[c[0] for c in C3_CASES])
def test_conv2d_c3_shapes(ops, case):
    name, B, Cin, H, W, Cout, k, pad, dil, with_bias = case
    key = 'c3/' + name
    x = fill.uniform(key + '/x', (B, Cin, H, W)).abs_()                       # post-ReLU activations
    w = fill.uniform(key + '/w', (Cout, Cin, k, k), (6.0 / (Cin * k * k)) ** 0.5)
    b = fill.uniform(key + '/b', (Cout,), 0.5) if with_bias else None
    gy = fill.uniform(key + '/gy', (B, Cout, H, W))
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    ref = F.conv2d(xr, wr, b, 1, pad, dil)
    ref.backward(gy)
    ref = ref.detach()
    cout_p = (Cout + 3) // 4 * 4
    xa, gya = to_act(ops, x, Cin), to_act(ops, gy, cout_p)
    w_raw = krsc(w, cout_p, Cin)
    b_raw = None
    if with_bias:
        b_raw = torch.zeros(cout_p)
        b_raw[:Cout] = b
        b_raw = b_raw.cuda()
    wT = ops.filter_transpose(w_raw, cout_p, k * k, Cin)
    errs = {}
    for prec in ('fp32', 'bf16x3', 'fp16x3'):
        tol, P = PREC_TOL[prec], ops._PREC_NAMES[prec]
        ya = ops.Act.empty(B, H, W, cout_p, 'cuda')
        am = dict(amax_x=ops.amax_of(xa), amax_w=ops.amax_of(w_raw)) if prec == 'fp16x3' else {}
        stats = ops.conv2d_fwd(xa, w_raw, b_raw, ya, k, k, 1, pad, dil, want_stats=not with_bias, precision=P, **am)
        e_f = rel(ya.to_nchw(Cout), ref)
        if stats is not None:    # the fused BatchNorm statistics of the same launch
            co = ops.bn_finalize(stats, ya.M, None, None, None, None, 0.0, 1e-5)
            r64 = ref.double()
            var = r64.var((0, 2, 3), unbiased=False)
            assert rel(co[1][:Cout], 1.0 / (var + 1e-5).sqrt()) < tol, (name, prec)
        dxa = ops.Act.empty(B, H, W, Cin, 'cuda')
        am = dict(amax_dy=ops.amax_of(gya), amax_w=ops.amax_of(wT)) if prec == 'fp16x3' else {}
        ops.conv2d_dgrad(gya, wT, dxa, k, k, 1, pad, dil, precision=P, **am)
        e_d = rel(dxa.to_nchw(Cin), xr.grad)
        e_w = None
        if prec != 'fp16x3':     # the weight gradient has no fp16-limb variant (the `limb` policy runs it on bf16 limbs)
            dw = torch.empty_like(w_raw)
            ops.conv2d_wgrad(xa, gya, dw, k, k, 1, pad, dil, precision=P)
            e_w = rel(dw.view(cout_p, k, k, Cin)[:Cout].permute(0, 3, 1, 2), wr.grad)
        errs[prec] = (e_f, e_d, e_w)
        assert e_f < tol and e_d < tol and (e_w is None or e_w < tol), (name, prec, errs)
    print('c3 %s max-norm rel err (fwd, dgrad, wgrad): %s' % (name, errs))


DMA_CASES = [
    # B, Cin, H, W, Cout, k, stride, pad, dil -- shapes the pre-split LDS-DMA limb kernel covers (dy channels % 32 == 0,
    # Cin >= 128, >= 256 tiles of 256x128): plain, dilated with patch-ordered rows, stride 2 with parity classes, ragged N
    (16, 512, 32, 32, 512, 3, 1, 2, 2),
    (16, 2048, 32, 32, 256, 3, 1, 18, 18),
    (16, 512, 32, 32, 2048, 1, 1, 0, 1),
    (16, 256, 64, 64, 256, 3, 2, 1, 1),
    (8, 192, 64, 64, 96, 3, 1, 1, 1),
]


@pytest.mark.parametrize('case', DMA_CASES)
def test_conv2d_dgrad_presplit_dma(ops, case):
    """pseg_split_planes + pseg_conv2d_dgrad_planes (BF16X3 arithmetic on pre-split bf16 limb planes, tiles staged by
    LDS-DMA): same products as the in-kernel-split limb kernel -- bit-identical results -- and the op tolerance against
    the exact-fp32 kernel; accumulate flag; limb planes round-trip (hi + lo reproduces x to ~2^-17)."""
    B, Cin, H, W, Cout, k, stride, pad, dil = case
    Ho, Wo = ops.conv_out_size(H, k, stride, pad, dil), ops.conv_out_size(W, k, stride, pad, dil)
    key = 'dma/' + '_'.join(map(str, case))
    dy = ops.Act(fill.uniform(key + '/dy', (B * Ho * Wo * Cout,)).cuda(), B, Ho, Wo, Cout, Cout)
    w = (fill.uniform(key + '/w', (Cout * k * k * Cin,)) * 0.05).cuda()
    wT = ops.filter_transpose(w, Cout, k * k, Cin)
    dx = [ops.Act.empty(B, H, W, Cin, 'cuda') for _ in range(3)]
    assert ops.dgrad_planes_ok(dy, dx[0], k, k, stride, pad, dil)
    ops.conv2d_dgrad(dy, wT, dx[0], k, k, stride, pad, dil, precision=ops.PREC_FP32)
    ops.conv2d_dgrad(dy, wT, dx[1], k, k, stride, pad, dil, precision=ops.PREC_BF16X3)
    dp, wp = ops.split_planes(dy), ops.split_planes(wT.view(Cin, k * k * Cout))
    hi = dp.hi.view(torch.bfloat16).float().view(dy.M, dp.ldp)[:, :Cout]
    lo = dp.lo.view(torch.bfloat16).float().view(dy.M, dp.ldp)[:, :Cout]
    xs = dy.t.view(dy.M, Cout)
    assert torch.equal(hi, xs.to(torch.bfloat16).float()) and ((hi + lo) - xs).abs().max().item() <= 2.0 ** -16 * xs.abs().max().item()
    ops.conv2d_dgrad_planes(dp, dy, wp, dx[2], k, k, stride, pad, dil)
    assert torch.equal(dx[2].t, dx[1].t)
    assert rel(dx[2].t, dx[0].t) < PREC_TOL['bf16x3']
    ops.conv2d_dgrad_planes(dp, dy, wp, dx[2], k, k, stride, pad, dil, accumulate=True)
    assert rel(dx[2].t, 2 * dx[0].t) < PREC_TOL['bf16x3']
    # shapes the kernel does not cover are refused, never mis-computed
    small = ops.Act.empty(2, 8, 8, 64, 'cuda')
    dys = ops.Act.empty(2, 8, 8, 32, 'cuda', zero=True)
    assert not ops.dgrad_planes_ok(dys, small, 3, 3, 1, 1, 1)
    from pytorch_segmentation_amd._lib import PsegError
    with pytest.raises(PsegError):
        ops.conv2d_dgrad_planes(ops.split_planes(dys), dys, ops.split_planes(torch.zeros(64, 9 * 32, device='cuda')), small, 3, 3, 1, 1, 1)


@pytest.mark.parametrize('case', [c for c in CONV_CASES if c[1] % 32 == 0])
def test_conv2d_f32_dma_variant(ops, case, monkeypatch, fresh_plans):
    """The exact-fp32 gather kernel staged by LDS-DMA (swizzled 128-byte rows; two- and three-stage ring) must
    give the same forward (+ fused BatchNorm statistics) and data gradient as the register-staged kernel: same products,
    same k order inside a K-step, so equal to rounding of the accumulation order (1e-6) -- on every geometry of the list
    whose K-steps stay inside a tap."""
    from pytorch_segmentation_amd import _lib
    B, Cin, H, W, Cout, k, stride, pad, dil = case
    x, w, b, xa, w_raw, b_raw, cin_p, cout_p, Ho, Wo = _conv_setup(ops, case, Cout in (21, 2))
    gy = fill.uniform('convg/' + '_'.join(map(str, case)), (B, Cout, Ho, Wo))
    gya = to_act(ops, gy, cout_p)
    wT = ops.filter_transpose(w_raw, cout_p, k * k, cin_p)

    def run():
        ya = ops.Act.empty(B, Ho, Wo, cout_p, 'cuda')
        st = ops.conv2d_fwd(xa, w_raw, b_raw, ya, k, k, stride, pad, dil, want_stats=b_raw is None, precision=ops.PREC_FP32)
        dxa = ops.Act.empty(B, H, W, cin_p, 'cuda')
        ops.conv2d_dgrad(gya, wT, dxa, k, k, stride, pad, dil, precision=ops.PREC_FP32)
        co = ops.bn_finalize(st, ya.M, None, None, None, None, 0.0, 1e-5) if st is not None else None
        return ya.to_nchw(Cout), dxa.to_nchw(Cin), co

    monkeypatch.setenv('PSEG_CONV_F32DMA', '0')      # register-staged kernel
    _lib.clear_query_cache()
    y0, d0, c0 = run()
    for mode in ('1', '3'):                          # three-stage ring / two-stage ring (3 = the default)
        monkeypatch.setenv('PSEG_CONV_F32DMA', mode)
        _lib.clear_query_cache()
        y1, d1, c1 = run()
        assert rel(y1, F.conv2d(x, w, b, stride, pad, dil)) < TOL
        assert rel(y1, y0) < 2e-6 and rel(d1, d0) < 2e-6, mode
        if c0 is not None:
            assert rel(c1[0], c0[0]) < 1e-5 and rel(c1[1], c0[1]) < 1e-5
    monkeypatch.delenv('PSEG_CONV_F32DMA')
    _lib.clear_query_cache()



# narrow 3x3 convs for the halo-staged kernel: (B, Cin, H, W, Cout, bias)
HALO_CASES = [(4, 32, 32, 32, 32, False), (2, 384, 16, 32, 21, True), (3, 64, 24, 16, 24, False)]


@pytest.mark.parametrize('bn', [32, 64])
@pytest.mark.parametrize('case', HALO_CASES)
def test_conv2d_halo_staged_narrow_3x3(ops, case, bn, monkeypatch, fresh_plans):
    """gather_f32_halo_kernel: an 8 x 16 patch of output pixels per block, the A operand of a 32-channel chunk DMA'd once as the
    10 x 18 halo patch, the nine taps read from it, the filter through a ring of single taps.  Forced onto the 128x32 (four waves
    down the rows) or 128x64 (2 x 2 waves; PSEG_CONV_HALO=2) plan tile and compared with the ring kernel on the same plan: forward (+ bias, + fused BatchNorm statistics), data gradient plain / accumulating / with the
    fused BatchNorm-backward sums.  The K order differs (chunk-major instead of tap-major): equal to rounding, and both against
    the fp64 CPU conv."""
    from pytorch_segmentation_amd import _lib
    B, Cin, H, W, Cout, bias = case
    Cp = (Cout + 3) // 4 * 4
    monkeypatch.setenv('PSEG_CONV_BM', '128')
    monkeypatch.setenv('PSEG_CONV_BN', str(bn))
    monkeypatch.setenv('PSEG_CONV_SPLITK', '1')        # (small maps: the planner would split K to fill the device -- another kernel)
    key = 'halo/' + '_'.join(map(str, case))
    x = fill.uniform(key + '/x', (B, Cin, H, W))
    w = fill.uniform(key + '/w', (Cout, Cin, 3, 3), (2.0 / (9 * Cin)) ** 0.5)
    b = fill.uniform(key + '/b', (Cout,)) if bias else None
    gy = fill.uniform(key + '/gy', (B, Cout, H, W))
    yprev = fill.uniform(key + '/yp', (B, Cin, H, W), 2.0)
    xa, ypa = to_act(ops, x), to_act(ops, yprev)
    gya = ops.Act.from_nchw(gy.cuda(), Cp)
    w_raw = torch.zeros(Cp, 3, 3, Cin)
    w_raw[:Cout] = w.permute(0, 2, 3, 1)
    w_raw = w_raw.cuda().contiguous()
    b_raw = None
    if bias:
        b_raw = torch.zeros(Cp)
        b_raw[:Cout] = b
        b_raw = b_raw.cuda()
    wT = ops.filter_transpose(w_raw, Cp, 9, Cin)
    co = ops.bn_finalize(ops.col_stats(ypa), ypa.M, torch.ones(Cin).cuda(), torch.zeros(Cin).cuda(), None, None, 0.0, 1e-5)
    base = fill.uniform(key + '/base', (B, Cin, H, W))

    def run():
        ya = ops.Act.empty(B, H, W, Cp, 'cuda')
        st = ops.conv2d_fwd(xa, w_raw, b_raw, ya, 3, 3, 1, 1, 1, want_stats=True, precision=ops.PREC_FP32)
        ran.append(_lib.load().pseg_debug_last_conv_kernel())
        cof = ops.bn_finalize(st, ya.M, None, None, None, None, 0.0, 1e-5)
        d_plain = ops.Act.empty(B, H, W, Cin, 'cuda')
        ops.conv2d_dgrad(gya, wT, d_plain, 3, 3, 1, 1, 1, precision=ops.PREC_FP32)
        ran.append(_lib.load().pseg_debug_last_conv_kernel())
        d_acc = to_act(ops, base)
        ops.conv2d_dgrad(gya, wT, d_acc, 3, 3, 1, 1, 1, accumulate=True, precision=ops.PREC_FP32)
        d_bn = ops.Act.empty(B, H, W, Cin, 'cuda')
        ops.conv2d_dgrad(gya, wT, d_bn, 3, 3, 1, 1, 1, precision=ops.PREC_FP32, bn=(ypa, co, 1))
        # (the data gradient gathers dy: its LDS-DMA forms -- ring, halo, the fused sums -- need dy channels % 32 == 0)
        assert (d_bn.bnpart is not None) == (Cp % 32 == 0)
        return ya, cof.clone(), d_plain, d_acc, d_bn, d_bn.bnpart.part.sum(1) if d_bn.bnpart is not None else None

    ran = []
    monkeypatch.setenv('PSEG_CONV_HALO', '0')
    _lib.clear_query_cache()
    ref = run()
    monkeypatch.setenv('PSEG_CONV_HALO', '2')
    _lib.clear_query_cache()
    got = run()
    # which kernels ran (pseg_debug_last_conv_kernel): 3 = the ring kernel, 4 = its form for channel counts off the K-step grid,
    # 6 = the halo-staged kernel -- forward always; the data gradient gathers dy, whose padded channel count must be % 32
    assert ran[0] == 3 and ran[2] == 6, ran
    assert ran[1] == (3 if Cp % 32 == 0 else 4) and ran[3] == (6 if Cp % 32 == 0 else 4), ran
    for env in ('PSEG_CONV_HALO', 'PSEG_CONV_BM', 'PSEG_CONV_BN', 'PSEG_CONV_SPLITK'):
        monkeypatch.delenv(env)
    _lib.clear_query_cache()
    y64 = F.conv2d(x.double(), w.double(), b.double() if bias else None, 1, 1, 1)
    xr = x.double().requires_grad_()
    F.conv2d(xr, w.double(), None, 1, 1, 1).backward(gy.double())
    for res in (ref, got):
        assert rel(res[0].to_nchw(Cout), y64) < 1e-5
        assert rel(res[2].to_nchw(), xr.grad) < 1e-5
        assert rel(res[3].to_nchw(), xr.grad + base.double()) < 1e-5
    assert rel(got[1], ref[1]) < 1e-5                                 # statistics -> mean / invstd
    assert torch.equal(got[4].t, got[2].t)                            # the sums ride on the same data gradient
    if got[5] is not None:
        assert rel(got[5], ref[5]) < 1e-5
    if Cp > Cout:
        assert got[0].t.view(-1, Cp)[:, Cout:].abs().max().item() == 0.0



@pytest.mark.parametrize('case', HALO_CASES + [(8, 32, 64, 64, 32, False), (2, 96, 20, 48, 8, False)])
def test_conv2d_wgrad_halo_staged(ops, case, monkeypatch, fresh_plans):
    """wgrad_f32_halo_kernel: a K-step is a 2 x 16 strip of output pixels, x comes in once as the 4 x 18 halo strip of a 32-channel
    chunk, wave t multiplies tap t.  Against the LDS-DMA weight-gradient kernel (PSEG_WGRAD_HALO=0) and the fp64 CPU gradient:
    single and multi-chunk inputs (PSEG_WGRAD_HALO=2 lifts the 64-channel cap), filters padded to 24 / 8 rows, pixel splits +
    slab reduction, accumulate, bit-reproducibility."""
    from pytorch_segmentation_amd import _lib
    B, Cin, H, W, Cout, _ = case
    Cp = (Cout + 3) // 4 * 4
    key = 'whalo/' + '_'.join(map(str, case))
    x = fill.uniform(key + '/x', (B, Cin, H, W))
    gy = fill.uniform(key + '/gy', (B, Cout, H, W))
    xa = to_act(ops, x)
    gya = ops.Act.from_nchw(gy.cuda(), Cp)
    xr = x.double()
    wr = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, wr, None, 1, 1, 1).backward(gy.double())
    ref64 = wr.grad.permute(0, 2, 3, 1)          # [Cout][kh][kw][Cin]

    def run():
        dw = torch.empty(Cp, 3, 3, Cin, device='cuda')
        ops.conv2d_wgrad(xa, gya, dw, 3, 3, 1, 1, 1, precision=ops.PREC_FP32)
        ran.append(_lib.load().pseg_debug_last_conv_kernel())
        dw2 = dw.clone()
        ops.conv2d_wgrad(xa, gya, dw2, 3, 3, 1, 1, 1, accumulate=True, precision=ops.PREC_FP32)
        dwc = torch.empty_like(dw)
        ops.conv2d_wgrad(xa, gya, dwc, 3, 3, 1, 1, 1, precision=ops.PREC_FP32, concurrent=True)
        return dw, dw2, dwc

    ran = []
    monkeypatch.setenv('PSEG_WGRAD_HALO', '0')
    _lib.clear_query_cache()
    ref = run()
    monkeypatch.setenv('PSEG_WGRAD_HALO', '2')
    _lib.clear_query_cache()
    got = run()
    again = run()
    assert ran[0] in (11, 13) and ran[1] == 14 and ran[2] == 14, ran      # 13 / 11: the LDS-DMA / register-staged kernel, 14: halo
    monkeypatch.delenv('PSEG_WGRAD_HALO')
    _lib.clear_query_cache()
    for res in (ref, got):
        assert rel(res[0][:Cout].cpu(), ref64) < 1e-5
        assert rel(res[1][:Cout].cpu(), 2 * ref64) < 1e-5
        assert rel(res[2][:Cout].cpu(), ref64) < 1e-5
        if Cp > Cout:
            assert res[0][Cout:].abs().max().item() == 0.0
    assert torch.equal(got[0], again[0]) and torch.equal(got[2], again[2])       # fixed-order slabs: bit-reproducible
    assert rel(got[0], ref[0]) < 1e-5



@pytest.mark.parametrize('half', [False, True])
@pytest.mark.parametrize('act', [1, 2, 0])
def test_bn_act_maxpool_fused(ops, half, act):
    """pseg_bn_act_maxpool_fwd(_h): max-pooling of act(BN(x)) without the activated map -- the same pooled values AND the same
    argmax bytes as bn_act_fwd followed by maxpool_fwd (ragged map, windows that hang over every border)."""
    B, C, H, W = 3, 64, 37, 29
    x = fill.uniform('bnpool/x%d' % act, (B, C, H, W), 2.0) + fill.uniform('bnpool/off', (1, C, 1, 1), 1.0)
    g = 1.0 + fill.uniform('bnpool/g', (C,), 0.3)
    b = fill.uniform('bnpool/b', (C,), 0.3)
    xa = ops.Act.from_nchw(x.cuda(), C, dtype=torch.float16) if half else to_act(ops, x)
    co = ops.bn_finalize(ops.col_stats(xa), xa.M, g.cuda(), b.cuda(), None, None, 0.0, 1e-5)
    z = xa.like()
    ops.bn_act_fwd(xa, co, act, z)
    Hp, Wp = ops.conv_out_size(H, 3, 2, 1, 1), ops.conv_out_size(W, 3, 2, 1, 1)
    p0 = xa.new(B, Hp, Wp, C)
    a0 = ops.maxpool_fwd(z, p0, 3, 2, 1)
    p1 = xa.new(B, Hp, Wp, C)
    a1 = ops.bn_act_maxpool_fwd(xa, co, act, p1, 3, 2, 1)
    assert torch.equal(p0.t, p1.t)
    assert torch.equal(a0, a1)
    assert ops.bn_act_maxpool_fwd(xa, co, act, p1, 3, 2, 1, want_argmax=False) is None and torch.equal(p0.t, p1.t)


# pointwise convs for the persistent kernel: (B, Cin, H, W, Cout).  Tiles: 128x128 (Cout 256), 128x64 (Cout 64), ragged M (30x30
# maps: the last row tile is partial) and ragged N (Cout 96: a 64-column tile half empty); K from 1 to 8 K-steps.
PW_CASES = [(4, 64, 32, 32, 256), (2, 256, 30, 30, 64), (2, 32, 64, 64, 128), (3, 128, 20, 20, 96), (2, 256, 32, 32, 256)]


@pytest.mark.parametrize('tile', [None, (128, 128), (64, 128)])
@pytest.mark.parametrize('grid', [3, 7])
@pytest.mark.parametrize('case', PW_CASES)
def test_conv2d_persistent_pointwise_kernel(ops, case, grid, tile, monkeypatch, fresh_plans):
    """gather_f32_pw_kernel: a block walks several tiles with ONE continuous stream of K-steps through the operand ring (the
    next tile's first DMAs are in flight while a tile is stored).  Forced onto small problems with a tiny grid
    (PSEG_CONV_PW_RESIDENT: 3 / 7 blocks for 8-64 tiles, so blocks own different numbers of tiles and ring phases carry over
    tile boundaries at both parities) and compared with the tile-per-block kernel (PSEG_CONV_PW=0): forward + fused BatchNorm
    statistics, data gradient plain / accumulating / with the fused BatchNorm-backward sums -- the same products in the same
    order per output element, so BIT-identical tensors; the sums (other lane order) to rounding.  tile: the planner's own choice
    (128x64 on these sizes, or 128x32, which the persistent kernel does not take) or a forced 128x128 / 64x128 tile, so that all
    three instantiations run."""
    from pytorch_segmentation_amd import _lib
    B, Cin, H, W, Cout = case
    if tile is not None:
        monkeypatch.setenv('PSEG_CONV_BM', str(tile[0]))
        monkeypatch.setenv('PSEG_CONV_BN', str(tile[1]))
    key = 'pw/' + '_'.join(map(str, case))
    x = fill.uniform(key + '/x', (B, Cin, H, W))
    w = fill.uniform(key + '/w', (Cout, Cin, 1, 1), (6.0 / Cin) ** 0.5)
    gy = fill.uniform(key + '/gy', (B, Cout, H, W))
    yprev = fill.uniform(key + '/yp', (B, Cin, H, W), 2.0)
    xa, gya, ypa = to_act(ops, x), to_act(ops, gy), to_act(ops, yprev)
    w_raw = krsc(w)
    wT = ops.filter_transpose(w_raw, Cout, 1, Cin)
    co = ops.bn_finalize(ops.col_stats(ypa), ypa.M, torch.ones(Cin).cuda(), torch.zeros(Cin).cuda(), None, None, 0.0, 1e-5)
    base = fill.uniform(key + '/base', (B, Cin, H, W))

    def run():
        ya = ops.Act.empty(B, H, W, Cout, 'cuda')
        st = ops.conv2d_fwd(xa, w_raw, None, ya, 1, 1, 1, 0, 1, want_stats=True, precision=ops.PREC_FP32)
        ran.append(_lib.load().pseg_debug_last_conv_kernel())
        cof = ops.bn_finalize(st, ya.M, None, None, None, None, 0.0, 1e-5)
        d_plain = ops.Act.empty(B, H, W, Cin, 'cuda')
        ops.conv2d_dgrad(gya, wT, d_plain, 1, 1, 1, 0, 1, precision=ops.PREC_FP32)
        d_acc = to_act(ops, base)
        ops.conv2d_dgrad(gya, wT, d_acc, 1, 1, 1, 0, 1, accumulate=True, precision=ops.PREC_FP32)
        d_bn = ops.Act.empty(B, H, W, Cin, 'cuda')
        ops.conv2d_dgrad(gya, wT, d_bn, 1, 1, 1, 0, 1, precision=ops.PREC_FP32, bn=(ypa, co, 1))
        part = d_bn.bnpart.part.sum(1) if d_bn.bnpart is not None else None
        return ya.t.clone(), cof.clone(), d_plain.t.clone(), d_acc.t.clone(), d_bn.t.clone(), part

    ran = []
    monkeypatch.setenv('PSEG_CONV_PW', '0')
    _lib.clear_query_cache()
    ref = run()
    monkeypatch.setenv('PSEG_CONV_PW', '1')
    monkeypatch.setenv('PSEG_CONV_PW_RESIDENT', str(grid))
    _lib.clear_query_cache()
    got = run()
    monkeypatch.delenv('PSEG_CONV_PW')
    monkeypatch.delenv('PSEG_CONV_PW_RESIDENT')
    if tile is not None:
        monkeypatch.delenv('PSEG_CONV_BM')
        monkeypatch.delenv('PSEG_CONV_BN')
    _lib.clear_query_cache()
    if tile is not None:       # (a forced tile is one the persistent kernel takes: 5 = it ran, 3 = the tile-per-block ring kernel)
        assert ran == [3, 5], ran
    assert rel(ops.Act(got[0], B, H, W, Cout, Cout).to_nchw(), F.conv2d(x, w)) < TOL
    assert torch.equal(got[0], ref[0]) and torch.equal(got[2], ref[2]) and torch.equal(got[3], ref[3]) and torch.equal(got[4], ref[4])
    assert rel(got[1][0], ref[1][0]) < 1e-5 and rel(got[1][1], ref[1][1]) < 1e-5
    assert (got[5] is None) == (ref[5] is None)
    if got[5] is not None:
        assert rel(got[5], ref[5]) < 1e-5


BIG_TILE_CASES = [c for c in CONV_CASES if c[1] >= 16] + [
    (2, 1024, 64, 64, 32, 1, 1, 0, 1),     # 32 x 8 = 256 tiles of 256x128: the planner picks the big tile by itself
    (4, 512, 64, 64, 16, 3, 2, 1, 1),      # stride-2 data gradient (parity-class rows) on the big tile
]


@pytest.fixture
def fresh_plans():
    """Planning overrides (PSEG_CONV_*) changed at run time: drop the memoised size queries before and after."""
    from pytorch_segmentation_amd import _lib
    _lib.clear_query_cache()
    yield
    _lib.clear_query_cache()


@pytest.mark.parametrize('prec', ['bf16x3', 'fp16x3'])
@pytest.mark.parametrize('case', BIG_TILE_CASES)
def test_conv2d_dgrad_big_tile(ops, case, prec, monkeypatch, fresh_plans):
    """The 256x128 tile / 8-wave variant of the limb gather kernel (what the data gradients of the wide layers run on):
    forced onto every geometry -- ragged M and N edges, tap skipping, parity-class row order, accumulate -- and compared
    both with the CPU reference and with the 128-row kernels."""
    TOL = PREC_TOL[prec]
    P = ops._PREC_NAMES[prec]
    B, Cin, H, W, Cout, k, stride, pad, dil = case
    x, w, b, xa, w_raw, b_raw, cin_p, cout_p, Ho, Wo = _conv_setup(ops, case, False)
    gy = fill.uniform('convg/' + '_'.join(map(str, case)), (B, Cout, Ho, Wo))
    dx_ref = torch.nn.grad.conv2d_input(x.shape, w, gy, stride, pad, dil)
    gya = to_act(ops, gy, cout_p)
    wT = ops.filter_transpose(w_raw, cout_p, k * k, cin_p)
    am = dict(amax_dy=ops.amax_of(gya), amax_w=ops.amax_of(wT)) if prec == 'fp16x3' else {}
    from pytorch_segmentation_amd import _lib
    small = ops.Act.empty(B, H, W, cin_p, 'cuda')
    monkeypatch.setenv('PSEG_CONV_NOBIG', '1')
    _lib.clear_query_cache()           # planning overrides change at run time in this test
    ops.conv2d_dgrad(gya, wT, small, k, k, stride, pad, dil, precision=P, **am)
    monkeypatch.delenv('PSEG_CONV_NOBIG')
    monkeypatch.setenv('PSEG_CONV_FORCEBIG', '1')
    _lib.clear_query_cache()
    big = ops.Act.empty(B, H, W, cin_p, 'cuda')
    ops.conv2d_dgrad(gya, wT, big, k, k, stride, pad, dil, precision=P, **am)
    assert rel(big.to_nchw(Cin), dx_ref) < TOL
    assert rel(big.to_nchw(Cin), small.to_nchw(Cin)) < 1e-5      # same products, only the split/accumulation grouping differs
    ops.conv2d_dgrad(gya, wT, big, k, k, stride, pad, dil, accumulate=True, precision=P, **am)
    assert rel(big.to_nchw(Cin), 2 * dx_ref) < TOL
    if prec != 'bf16x3':
        return
    # weight gradient on the 256(Cout) x 128 tile (forced) against the reference and the 128-row kernel
    dw_ref = torch.nn.grad.conv2d_weight(x, w.shape, gy, stride, pad, dil)

    def wgrad(acc=False, out=None):
        dw = torch.empty_like(w_raw) if out is None else out
        ops.conv2d_wgrad(xa, gya, dw, k, k, stride, pad, dil, accumulate=acc, precision=P)
        return dw

    dw_big = wgrad()
    got = dw_big.view(cout_p, k, k, cin_p)[:Cout, :, :, :Cin].permute(0, 3, 1, 2).cpu()
    assert rel(got, dw_ref) < TOL
    assert torch.equal(dw_big, wgrad())                           # bit-reproducible
    wgrad(True, dw_big)
    assert rel(dw_big.view(cout_p, k, k, cin_p)[:Cout, :, :, :Cin].permute(0, 3, 1, 2).cpu(), 2 * dw_ref) < TOL
    monkeypatch.delenv('PSEG_CONV_FORCEBIG')
    monkeypatch.setenv('PSEG_CONV_NOBIG', '1')
    _lib.clear_query_cache()
    dw_small = wgrad()
    monkeypatch.delenv('PSEG_CONV_NOBIG')
    monkeypatch.setenv('PSEG_CONV_FORCEBIG', '1')
    _lib.clear_query_cache()
    assert rel(wgrad(), dw_small) < 1e-5
    monkeypatch.delenv('PSEG_CONV_FORCEBIG')
    _lib.clear_query_cache()


def test_conv_into_concat_slice(ops):
    """Branches write straight into channel slices of one wide buffer (the reference's torch.cat)."""
    x = fill.uniform('cat/x', (2, 32, 12, 12))
    w1 = fill.uniform('cat/w1', (16, 32, 1, 1), 0.3)
    w2 = fill.uniform('cat/w2', (24, 32, 3, 3), 0.1)
    xa = to_act(ops, x)
    wide = ops.Act.empty(2, 12, 12, 40, 'cuda', zero=True)
    ops.conv2d_fwd(xa, krsc(w1), None, wide.slice(0, 16), 1, 1, 1, 0, 1)
    ops.conv2d_fwd(xa, krsc(w2), None, wide.slice(16, 40), 3, 3, 1, 1, 1)
    ref = torch.cat([F.conv2d(x, w1), F.conv2d(x, w2, padding=1)], 1)
    assert rel(wide.to_nchw(), ref) < TOL


@pytest.mark.parametrize('B,C,H,W,act,res', [(4, 64, 16, 16, 1, False), (2, 256, 8, 8, 1, True), (16, 32, 1, 1, 1, False),
                                              (2, 96, 9, 7, 2, False), (2, 24, 12, 12, 0, True), (3, 128, 31, 17, 1, False),
                                              (4, 64, 128, 128, 1, False)])
def test_batchnorm_train(ops, B, C, H, W, act, res):
    key = 'bn/%d_%d_%d_%d_%d' % (B, C, H, W, act)
    y = fill.uniform(key + '/y', (B, C, H, W), 2.0) + fill.uniform(key + '/off', (1, C, 1, 1), 1.0)
    r = fill.uniform(key + '/r', (B, C, H, W), 1.0) if res else None
    g = 1.0 + fill.uniform(key + '/g', (C,), 0.3)
    b = fill.uniform(key + '/b', (C,), 0.3)
    gz = fill.uniform(key + '/gz', (B, C, H, W))
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(g)
        bn.bias.copy_(b)
    bn.train()
    yr = y.clone().requires_grad_()
    rr = r.clone().requires_grad_() if res else None
    t = bn(yr)
    if res:
        t = t + rr
    zr = F.relu(t) if act == 1 else (F.relu6(t) if act == 2 else t)
    zr.backward(gz)

    ya = to_act(ops, y)
    rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda()
    co = ops.bn_finalize(ops.col_stats(ya), ya.M, g.cuda(), b.cuda(), rm, rv, 0.1, 1e-5)
    za = ya.like()
    ra = to_act(ops, r) if res else None
    ops.bn_act_fwd(ya, co, act, za, residual=ra)
    assert rel(za.to_nchw(), zr) < TOL
    assert rel(rm, bn.running_mean) < TOL and rel(rv, bn.running_var) < TOL
    dg, db = torch.zeros(C).cuda(), torch.zeros(C).cuda()
    dya = ya.like()
    dra = ya.like() if res else None
    ops.bn_act_bwd(to_act(ops, gz), za if res else None, ya, co, act, dya, dg, db, dres=dra)  # z=None: mask from y
    assert rel(dya.to_nchw(), yr.grad) < 5 * TOL
    assert rel(dg, bn.weight.grad) < 5 * TOL and rel(db, bn.bias.grad) < 5 * TOL
    if res:
        assert rel(dra.to_nchw(), rr.grad) < TOL
    if act and C % 32 == 0:
        # activation bitmask instead of z in backward (what the residual layers of the models use): the mask must be
        # exactly act'(z), and the backward passes fed with it bit-identical to the z-fed ones
        zb = ya.like()
        mask = ops.bn_act_fwd(ya, co, act, zb, residual=ra, want_mask=True)
        assert mask is not None and mask.numel() == ya.M * C // 32 and torch.equal(zb.t, za.t)
        zz = za.t.view(ya.M, C)
        on = (zz > 0) if act == 1 else ((zz > 0) & (zz < 6))
        bits = (mask.view(ya.M, C // 32, 1).to(torch.int64) >> torch.arange(32, device='cuda')) & 1
        assert torch.equal(bits.view(ya.M, C).bool(), on)
        dg2, db2, dyb = torch.zeros(C).cuda(), torch.zeros(C).cuda(), ya.like()
        drb = ya.like() if res else None
        dg1, db1, dy1 = torch.zeros(C).cuda(), torch.zeros(C).cuda(), ya.like()
        dr1 = ya.like() if res else None
        ops.bn_act_bwd(to_act(ops, gz), za, ya, co, act, dy1, dg1, db1, dres=dr1)
        ops.bn_act_bwd(to_act(ops, gz), za, ya, co, act, dyb, dg2, db2, dres=drb, mask=mask)   # (z: small-tensor path)
        assert torch.equal(dyb.t, dy1.t) and torch.equal(dg2, dg1) and torch.equal(db2, db1)
        if res:
            assert torch.equal(drb.t, dr1.t)
    if C % 8 == 0:
        # dy also as bf16 limb planes (feeds the pre-split data gradient): same dy, and exactly split_planes(dy);
        # the small-tensor path does not write planes (dy.planes stays None)
        dg3, db3, dyp = torch.zeros(C).cuda(), torch.zeros(C).cuda(), ya.like()
        ops.bn_act_bwd(to_act(ops, gz), za, ya, co, act, dyp, dg3, db3, want_planes=True)
        assert torch.equal(dyp.t, dya.t)
        if dyp.planes is not None:
            want = ops.split_planes(dyp)
            assert torch.equal(dyp.planes.hi, want.hi) and torch.equal(dyp.planes.lo, want.lo)
        else:
            assert ops.bn_small_path(ops._lib.query('pseg_col_stats_rows', ya.M, C), ya.M, C)
    # eval mode
    bn.eval()
    with torch.no_grad():
        ze = bn(y)
    co_e = ops.bn_eval_coeffs(g.cuda(), b.cuda(), bn.running_mean.cuda(), bn.running_var.cuda(), 1e-5)
    ops.bn_act_fwd(ya, co_e, 0, za)
    assert rel(za.to_nchw(), ze) < TOL



# conv -> BatchNorm -> act -> conv: the second conv's data gradient IS the first layer's dz.  (B, C_mid, H, W, C_out2, k, stride,
# pad, dil, act, slice): C_mid = channels of the BatchNorm layer = Cin of the second conv.  Tiles the fused form runs on: 128x64
# (C_mid 64), 128x128 (C_mid 256 on a big map), 64x128 / 128x32 via the planner's choices for the small problems; a stride-2 and a
# dilated second conv (parity-class / patch-ordered rows: the sums must follow the row permutation); y of the layer as a channel
# slice of a wider buffer.
BNSTAT_CASES = [
    (4, 64, 32, 32, 256, 1, 1, 0, 1, 1, False),      # bottleneck conv3 on layer-1 widths: K = 256, N = 64
    (2, 64, 64, 64, 64, 3, 1, 1, 1, 1, False),       # bottleneck conv2 (3x3)
    (2, 256, 64, 64, 64, 1, 1, 0, 1, 1, False),      # N = 256: 128x128 tiles
    (8, 128, 64, 64, 128, 3, 2, 1, 1, 1, True),      # stride-2 3x3 (ResNet layer 2 / 3 first block), y as a slice
    (8, 256, 32, 32, 64, 3, 1, 6, 6, 2, False),      # dilated 3x3 (tap skipping, patch / class-ordered rows), ReLU6
    (16, 32, 64, 64, 32, 3, 1, 1, 1, 0, False),      # 128x32 tiles, no activation (HRNet's activate=None layers)
]


@pytest.mark.parametrize('case', BNSTAT_CASES)
def test_dgrad_with_fused_batchnorm_backward_sums(ops, case):
    """pseg_conv2d_dgrad_bnstat: the data gradient of a conv whose input is act(BN(y)) also writes that layer's backward partial
    sums.  dx must be BIT-identical to the plain data gradient; the sums -- through pseg_bn_bwd_finalize, as the models use them
    -- must equal fp64 dbeta = sum(dz * act'), dgamma = sum(dz * act' * xhat) and give the same dy as the unfused reduce +
    finalize + apply chain (whose own parity with torch autograd is test_batchnorm_train's subject)."""
    B, C, H, W, C2, k, stride, pad, dil, act, sliced = case
    key = 'bnstat/' + '_'.join(map(str, case))
    Ho, Wo = ops.conv_out_size(H, k, stride, pad, dil), ops.conv_out_size(W, k, stride, pad, dil)
    rows = ops._lib.query('pseg_conv2d_dgrad_bnstat_rows', B, H, W, C, Ho, Wo, C2, k, k, stride, pad, dil)
    assert rows > 0, 'the fused form must cover this shape'
    y = fill.uniform(key + '/y', (B, C, H, W), 2.0) + fill.uniform(key + '/off', (1, C, 1, 1), 1.0)
    g = 1.0 + fill.uniform(key + '/g', (C,), 0.3)
    b = fill.uniform(key + '/b', (C,), 0.3)
    w2 = fill.uniform(key + '/w2', (C2, C, k, k), 0.05)
    gy2 = fill.uniform(key + '/gy2', (B, C2, Ho, Wo))
    if sliced:
        wide = ops.Act.empty(B, H, W, C + 32, 'cuda', zero=True)
        ya = wide.slice(32, 32 + C)
        ops.copy2d(to_act(ops, y), ya)
    else:
        ya = to_act(ops, y)
    rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda()
    co = ops.bn_finalize(ops.col_stats(ya), ya.M, g.cuda(), b.cuda(), rm, rv, 0.1, 1e-5)
    w_raw = krsc(w2)
    wT = ops.filter_transpose(w_raw, C2, k * k, C)
    gya = to_act(ops, gy2)
    dx_plain = ops.Act.empty(B, H, W, C, 'cuda')
    ops.conv2d_dgrad(gya, wT, dx_plain, k, k, stride, pad, dil)
    assert dx_plain.bnpart is None
    dx = ops.Act.empty(B, H, W, C, 'cuda')
    ops.conv2d_dgrad(gya, wT, dx, k, k, stride, pad, dil, bn=(ya, co, act))
    assert dx.bnpart is not None and dx.bnpart.rows == rows and dx.bnpart.key == ya.ptr
    assert torch.equal(dx.t, dx_plain.t)
    # the sums against fp64
    dz = dx.to_nchw().double().cpu()
    mu, istd, sc, sh = (co[i].double().cpu().view(1, C, 1, 1) for i in range(4))
    yd = y.double()
    pre = (y.cuda() - co[0].view(1, C, 1, 1)) * co[2].view(1, C, 1, 1) + co[3].view(1, C, 1, 1)      # fp32, as the kernels decide
    on = (pre > 0) if act == 1 else (((pre > 0) & (pre < 6)) if act == 2 else torch.ones_like(pre, dtype=torch.bool))
    gm = dz * on.cpu().double()
    db_ref = gm.sum((0, 2, 3))
    dg_ref = (gm * (yd - mu) * istd).sum((0, 2, 3))
    part = dx.bnpart.part.double().cpu()
    scale_b = gm.abs().sum((0, 2, 3)).max().item() + 1e-30
    scale_g = (gm * (yd - mu) * istd).abs().sum((0, 2, 3)).max().item() + 1e-30
    assert (part[0].sum(0) - db_ref).abs().max().item() < 1e-5 * scale_b
    assert (part[1].sum(0) - dg_ref).abs().max().item() < 1e-5 * scale_g
    # ... and through the BatchNorm backward: fused and unfused chains give the same dy / dgamma / dbeta
    dg1, db1, dy1 = torch.zeros(C).cuda(), torch.zeros(C).cuda(), ya.like()
    ops.bn_act_bwd(dx_plain, None, ya, co, act, dy1, dg1, db1)
    dg2, db2, dy2 = torch.zeros(C).cuda(), torch.zeros(C).cuda(), ya.like()
    ops.bn_act_bwd(dx, None, ya, co, act, dy2, dg2, db2, part=dx.bnpart)
    assert (dg2.double().cpu() - dg_ref).abs().max().item() < 1e-5 * scale_g
    assert (db2.double().cpu() - db_ref).abs().max().item() < 1e-5 * scale_b
    assert rel(dg2, dg1) < 1e-5 and (db2 - db1).abs().max().item() < 1e-5 * scale_b
    assert rel(dy2.to_nchw(), dy1.to_nchw()) < 1e-5
    # a stale / foreign partial is never used: wrong key -> the reduction pass runs
    dg3, db3, dy3 = torch.zeros(C).cuda(), torch.zeros(C).cuda(), ya.like()
    bogus = ops.BnPart(torch.full_like(dx.bnpart.part, 7.0), rows, ya.ptr + 16)
    ops.bn_act_bwd(dx_plain, None, ya, co, act, dy3, dg3, db3, part=bogus)
    assert torch.equal(dy3.t, dy1.t) and torch.equal(dg3, dg1)
    # bit-reproducible
    dxb = ops.Act.empty(B, H, W, C, 'cuda')
    ops.conv2d_dgrad(gya, wT, dxb, k, k, stride, pad, dil, bn=(ya, co, act))
    assert torch.equal(dxb.bnpart.part, dx.bnpart.part)


def test_fused_batchnorm_backward_sums_are_refused_where_unavailable(ops):
    """channel counts off the LDS-DMA kernel's grid (K-steps straddling taps), accumulate, other precisions: rows == 0 / plain path,
    and the C entry point refuses a mismatched part_rows instead of writing past the buffer."""
    q = ops._lib.query
    assert q('pseg_conv2d_dgrad_bnstat_rows', 2, 16, 16, 24, 16, 16, 24, 3, 3, 1, 1, 1) == 0      # dy channels % 32 != 0
    assert q('pseg_conv2d_dgrad_bnstat_rows', 2, 16, 16, 64, 16, 16, 64, 3, 3, 1, 1, 1) > 0
    B, C, H, W, C2 = 2, 64, 16, 16, 64
    ya = to_act(ops, fill.uniform('bnstat/refuse/y', (B, C, H, W)))
    co = ops.bn_finalize(ops.col_stats(ya), ya.M, torch.ones(C).cuda(), torch.zeros(C).cuda(), None, None, 0.0, 1e-5)
    wT = ops.filter_transpose(krsc(fill.uniform('bnstat/refuse/w', (C2, C, 3, 3), 0.05)), C2, 9, C)
    gya = to_act(ops, fill.uniform('bnstat/refuse/g', (B, C2, H, W)))
    dx = ops.Act.empty(B, H, W, C, 'cuda', zero=True)
    ops.conv2d_dgrad(gya, wT, dx, 3, 3, 1, 1, 1, accumulate=True, bn=(ya, co, 1))       # accumulate: plain path
    assert dx.bnpart is None
    ops.conv2d_dgrad(gya, wT, dx, 3, 3, 1, 1, 1, precision=ops.PREC_BF16X3, bn=(ya, co, 1))
    assert dx.bnpart is None
    part = torch.zeros(2, 3, C, device='cuda')
    with pytest.raises(ops._lib.PsegError, match='part_rows'):
        ops._lib.call('pseg_conv2d_dgrad_bnstat', gya.ptr, gya.ld, wT.data_ptr(), dx.ptr, dx.ld, B, H, W, C, H, W, C2, 3, 3, 1, 1, 1,
                      ya.ptr, ya.ld, co[0].data_ptr(), co[1].data_ptr(), co[2].data_ptr(), co[3].data_ptr(), 1,
                      part[0].data_ptr(), part[1].data_ptr(), 3, 0)


@pytest.mark.parametrize('B,C,H,W,act,res', [(4, 64, 16, 16, 1, False), (2, 256, 8, 8, 1, True), (2, 96, 9, 7, 2, False),
                                              (2, 24, 12, 12, 0, True)])
def test_batchnorm_eval_backward(ops, B, C, H, W, act, res):
    """Frozen-statistics BatchNorm (module.eval()) still back-propagates: dy = scale * dz * act', dgamma = sum(dz*act'*xhat),
    dbeta = sum(dz*act') with xhat from the RUNNING statistics -- what autograd computes for F.batch_norm(training=False)."""
    key = 'bne/%d_%d_%d_%d_%d' % (B, C, H, W, act)
    y = fill.uniform(key + '/y', (B, C, H, W), 2.0) + fill.uniform(key + '/off', (1, C, 1, 1), 1.0)
    r = fill.uniform(key + '/r', (B, C, H, W), 1.0) if res else None
    gz = fill.uniform(key + '/gz', (B, C, H, W))
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(1.0 + fill.uniform(key + '/g', (C,), 0.3))
        bn.bias.copy_(fill.uniform(key + '/b', (C,), 0.3))
        bn.running_mean.copy_(fill.uniform(key + '/rm', (C,), 0.8))
        bn.running_var.copy_(0.5 + fill.uniform(key + '/rv', (C,), 0.4).abs())
    bn.eval()
    yr = y.clone().requires_grad_()
    rr = r.clone().requires_grad_() if res else None
    t = bn(yr)
    if res:
        t = t + rr
    zr = F.relu(t) if act == 1 else (F.relu6(t) if act == 2 else t)
    zr.backward(gz)
    ya = to_act(ops, y)
    co = ops.bn_eval_coeffs(bn.weight.detach().cuda(), bn.bias.detach().cuda(), bn.running_mean.cuda(),
                            bn.running_var.cuda(), 1e-5)
    za = ya.like()
    ra = to_act(ops, r) if res else None
    ops.bn_act_fwd(ya, co, act, za, residual=ra)
    assert rel(za.to_nchw(), zr) < TOL
    dg, db = torch.zeros(C).cuda(), torch.zeros(C).cuda()
    dya = ya.like()
    dra = ya.like() if res else None
    ops.bn_act_bwd(to_act(ops, gz), za, ya, co, act, dya, dg, db, dres=dra, frozen=True)
    assert rel(dya.to_nchw(), yr.grad) < TOL
    assert rel(dg, bn.weight.grad) < 5 * TOL and rel(db, bn.bias.grad) < 5 * TOL
    if res:
        assert rel(dra.to_nchw(), rr.grad) < TOL


def test_batchnorm_statistics_are_cancellation_safe(ops):
    """|mean| >> std: E[y^2] - mean^2 in fp32 would lose the variance entirely; the shifted statistics must not."""
    B, C, H, W = 4, 32, 24, 24
    y = 100.0 + 1e-2 * fill.uniform('bnstab', (B, C, H, W))
    y[:, 5] = 7.0                      # a dead (constant) channel: variance exactly 0
    ya = to_act(ops, y)
    co = ops.bn_finalize(ops.col_stats(ya), ya.M, None, None, None, None, 0.0, 1e-5)
    y64 = y.double()
    var = y64.var((0, 2, 3), unbiased=False)
    assert rel(co[0], y64.mean((0, 2, 3))) < 1e-6
    assert rel(co[1], 1.0 / (var + 1e-5).sqrt()) < 1e-3
    # same through the conv epilogue: a 1x1 conv whose output has a large per-channel offset
    x = torch.cat([torch.ones(B, 4, H, W), 1e-3 * fill.uniform('bnstab/x', (B, 28, H, W))], 1)
    w = fill.uniform('bnstab/w', (64, 32, 1, 1), 1.0)
    w[:, :4] = 25.0
    ref = F.conv2d(x.double(), w.double())
    ya = ops.Act.empty(B, H, W, 64, 'cuda')
    st = ops.conv2d_fwd(to_act(ops, x), krsc(w), None, ya, 1, 1, 1, 0, 1, want_stats=True)
    co = ops.bn_finalize(st, ya.M, None, None, None, None, 0.0, 1e-5)
    assert rel(co[0], ref.mean((0, 2, 3))) < 1e-6
    assert rel(co[1], 1.0 / (ref.var((0, 2, 3), unbiased=False) + 1e-5).sqrt()) < 1e-3


def test_col_sum_and_copy(ops):
    x = fill.uniform('colsum', (3, 24, 17, 13))
    xa = to_act(ops, x)
    out = torch.zeros(24).cuda()
    ops.col_sum(xa, out)
    assert rel(out, x.sum((0, 2, 3))) < TOL
    ya = xa.like()
    ops.copy2d(xa, ya)
    ops.copy2d(xa, ya, accumulate=True)
    assert rel(ya.to_nchw(), 2 * x) < 1e-6


def test_pool_and_broadcast(ops):
    x = fill.uniform('gap', (3, 128, 10, 12))
    xa = to_act(ops, x)
    out = ops.Act.empty(3, 1, 1, 128, 'cuda')
    ops.pool_sum(xa, out, 1.0 / 120)
    assert rel(out.to_nchw(), F.adaptive_avg_pool2d(x, 1)) < TOL
    ya = xa.like()
    ops.broadcast(out, ya)
    ref = F.interpolate(F.adaptive_avg_pool2d(x, 1), size=(10, 12), mode='bilinear', align_corners=False)
    assert rel(ya.to_nchw(), ref) < TOL


RESIZE_CASES = [(2, 16, 8, 8, 32, 32, True), (2, 24, 16, 16, 32, 32, True), (1, 8, 5, 7, 13, 9, True),
                (2, 8, 6, 6, 24, 24, False), (2, 8, 16, 16, 40, 24, True), (1, 4, 12, 12, 12, 12, True),
                (2, 8, 1, 1, 6, 5, False), (2, 8, 9, 9, 5, 4, True)]


@pytest.mark.parametrize('B,C,Hi,Wi,Ho,Wo,ac', RESIZE_CASES)
def test_bilinear(ops, B, C, Hi, Wi, Ho, Wo, ac):
    key = 'resize/%d_%d_%d_%d_%d' % (C, Hi, Wi, Ho, Wo)
    x = fill.uniform(key, (B, C, Hi, Wi))
    gy = fill.uniform(key + '/g', (B, C, Ho, Wo))
    xr = x.clone().requires_grad_()
    ref = F.interpolate(xr, size=(Ho, Wo), mode='bilinear', align_corners=ac)
    ref.backward(gy)
    xa = to_act(ops, x)
    ya = ops.Act.empty(B, Ho, Wo, C, 'cuda')
    ops.bilinear_fwd(xa, ya, ac)
    assert rel(ya.to_nchw(), ref) < TOL
    out = ops.bilinear_fwd_nchw(xa, C, Ho, Wo, ac)
    assert rel(out, ref) < TOL
    dxa = xa.like()
    ops.bilinear_bwd(to_act(ops, gy), dxa, ac)
    assert rel(dxa.to_nchw(), xr.grad) < TOL
    dxb = xa.like(zero=True)
    ops.bilinear_bwd_nchw(gy.cuda(), dxb, C, ac)
    assert rel(dxb.to_nchw(), xr.grad) < TOL


def test_bilinear_nchw_padded_channels(ops):
    """21 logits channels stored in a 24-wide NHWC buffer -> [B,21,H,W] and back."""
    x = fill.uniform('rs21', (2, 21, 8, 8))
    gy = fill.uniform('rs21g', (2, 21, 32, 32))
    xr = x.clone().requires_grad_()
    ref = F.interpolate(xr, scale_factor=4, mode='bilinear', align_corners=True)
    ref.backward(gy)
    xa = to_act(ops, x, 24)
    out = ops.bilinear_fwd_nchw(xa, 21, 32, 32, True)
    assert rel(out, ref) < TOL
    dxa = xa.like(zero=True)
    ops.bilinear_bwd_nchw(gy.cuda(), dxa, 21, True)
    assert rel(dxa.to_nchw(21), xr.grad) < TOL
    assert dxa.view4()[..., 21:].abs().max().item() == 0.0


@pytest.mark.parametrize('B,C,Hi,Wi,Ho,Wo,ac', [(2, 21, 8, 8, 32, 32, True), (2, 5, 16, 16, 64, 64, False), (1, 2, 9, 7, 18, 14, True),
                                                (2, 3, 16, 12, 40, 24, True), (1, 8, 6, 6, 6, 6, False), (2, 21, 32, 32, 128, 128, True)])
def test_bilinear_bwd_nchw_both_forms(ops, B, C, Hi, Wi, Ho, Wo, ac):
    """The NCHW backward in its separable two-pass form (with scratch) and its single gather pass (no scratch), both
    against autograd, incl. accumulate, non-integer scale factors, identity size and channel counts off the 4-grid."""
    from pytorch_segmentation_amd import _lib
    x = fill.uniform('bwdnchw', (B, C, Hi, Wi)).requires_grad_()
    gy = fill.uniform('bwdnchw/g', (B, C, Ho, Wo))
    F.interpolate(x, size=(Ho, Wo), mode='bilinear', align_corners=ac).backward(gy)
    gyc = gy.cuda()
    Cp = (C + 3) // 4 * 4
    st = torch.cuda.current_stream().cuda_stream
    nbytes = _lib.query('pseg_bilinear_bwd_workspace_bytes', B, Hi, Wi, C, Ho, Wo, 1)
    assert nbytes == B * C * Ho * Wi * 4
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    for scratch in (True, False):
        dxa = ops.Act.empty(B, Hi, Wi, Cp, 'cuda', zero=True)
        for rep in range(2):   # second call accumulates
            _lib.call('pseg_bilinear_bwd', gyc.data_ptr(), 0, B, Hi, Wi, C, dxa.ptr, dxa.ld, Ho, Wo, int(ac), 1, int(rep == 1),
                      ws.data_ptr() if scratch else 0, nbytes if scratch else 0, st)
            assert rel(dxa.to_nchw(C), (rep + 1) * x.grad) < TOL, (scratch, rep)
        if Cp > C:
            assert dxa.view4()[..., C:].abs().max().item() == 0.0


def test_maxpool(ops):
    x = fill.uniform('mp', (2, 64, 33, 31)).relu()
    gy = fill.uniform('mpg', (2, 64, 17, 16))
    xr = x.clone().requires_grad_()
    ref = F.max_pool2d(xr, 3, 2, 1)
    ref.backward(gy)
    xa = to_act(ops, x)
    ya = ops.Act.empty(2, 17, 16, 64, 'cuda')
    arg = ops.maxpool_fwd(xa, ya, 3, 2, 1)
    assert torch.equal(ya.to_nchw().cpu(), ref.detach())
    dxa = xa.like()
    ops.maxpool_bwd(to_act(ops, gy), arg, dxa, 3, 2, 1)
    assert rel(dxa.to_nchw(), xr.grad) < TOL


@pytest.mark.parametrize('B,C,H,W', [(2, 21, 32, 32), (2, 2, 16, 16), (1, 5, 7, 9), (2, 8, 12, 12), (1, 30, 8, 8),
                                       (1, 40, 6, 5), (2, 21, 9, 9)])
def test_cross_entropy(ops, B, C, H, W):
    key = 'ce/%d_%d_%d' % (C, H, W)
    lg = fill.uniform(key, (B, C, H, W), 5.0)
    tg = fill.labels(key + '/t', (B, H, W), C, block=2)
    tg[0, 0, :3] = -100  # ignored pixels
    lr = lg.clone().requires_grad_()
    ref = F.cross_entropy(lr, tg)
    ref.backward()
    out, dl = ops.ce_fwd_bwd(lg.cuda(), tg.cuda())
    assert abs(out[0].item() - ref.item()) <= 1e-5 * abs(ref.item())
    assert out[1].item() == float((tg != -100).sum())
    assert rel(dl, lr.grad) < TOL
    # upstream gradient scaling
    g = torch.tensor([2.5], device='cuda')
    ops.scale_inplace(dl, g)
    assert rel(dl, 2.5 * lr.grad) < TOL
    ops.scale_inplace(dl, torch.ones(1, device='cuda'))
    assert rel(dl, 2.5 * lr.grad) < TOL
    assert out[2].item() == 0.0
    # loss only
    out2, none = ops.ce_fwd_bwd(lg.cuda(), tg.cuda(), want_grad=False)
    assert none is None and out2[0].item() == out[0].item()
    # out-of-range labels (e.g. a 255 "void" label fed without ignore_index; torch raises IndexError): divisor, sum and
    # gradient must agree -- the result equals torch's with those pixels mapped to ignore_index -- and they are reported
    tb = tg.clone()
    tb[0, 1, :5] = 255
    tb[-1, -1, -2:] = -7
    nbad = int(((tb != -100) & ((tb < 0) | (tb >= C))).sum())
    ti = tb.clone()
    ti[(tb < 0) | (tb >= C)] = -100
    lr2 = lg.clone().requires_grad_()
    ref2 = F.cross_entropy(lr2, ti)
    ref2.backward()
    out3, dl3 = ops.ce_fwd_bwd(lg.cuda(), tb.cuda())
    assert out3[1].item() == float((ti != -100).sum()) and out3[2].item() == float(nbad)
    assert abs(out3[0].item() - ref2.item()) <= 1e-5 * abs(ref2.item())
    assert rel(dl3, lr2.grad) < TOL


@pytest.mark.parametrize('B,C,h,w,scale,ac', [(2, 21, 16, 24, 4, True), (1, 21, 9, 13, 4, True), (2, 2, 8, 8, 4, True),
                                               (2, 5, 12, 20, 2, True), (2, 21, 16, 16, 4, False), (4, 21, 64, 64, 4, True)])
@pytest.mark.parametrize('form', ['scatter', 'gather'])
def test_cross_entropy_on_upsampled_logits(ops, B, C, h, w, scale, ac, form, monkeypatch):
    """pseg_ce_upsampled_fwd_bwd: CrossEntropy(interpolate(logits, x4)) and its gradient with respect to the LOW-resolution
    logits, without the full-resolution tensor -- against (1) torch CPU float64 autograd through F.interpolate +
    F.cross_entropy and (2) the library's own three-pass path (bilinear_fwd_nchw -> ce_fwd_bwd -> bilinear_bwd_nchw).
    Ignored and out-of-range labels, ragged tiles (h, w not multiples of the 4 x 8 tile), padded class channels,
    reproducibility (two runs bit-identical).  Both forms: `scatter` (round 5, the default: a block owns full-resolution pixels,
    every pixel evaluated once, per-tile gradient patches combined by a second launch) and `gather` (PSEG_CE_SCATTER=0)."""
    monkeypatch.setenv('PSEG_CE_SCATTER', '1' if form == 'scatter' else '0')
    H, W = h * scale, w * scale
    key = 'ceup/%d_%d_%d_%d_%d_%d' % (B, C, h, w, scale, ac)
    x = fill.uniform(key + '/x', (B, C, h, w), 3.0)
    t = fill.labels(key + '/t', (B, H, W), C, block=4)
    t[0, :3, :5] = -100
    t[-1, 5:9, 2:4] = C + 3          # out of range: ignored and reported
    lr = to_act(ops, x, 24)
    assert ops.ce_upsampled_ok(lr, C, H, W, ac)
    out, dlr = ops.ce_upsampled_fwd_bwd(lr, C, t.cuda(), ac)
    out2, dlr2 = ops.ce_upsampled_fwd_bwd(lr, C, t.cuda(), ac)
    assert torch.equal(out, out2) and torch.equal(dlr.t, dlr2.t)
    # (1) float64 autograd
    xr = x.double().requires_grad_()
    up = F.interpolate(xr, size=(H, W), mode='bilinear', align_corners=ac)
    tt = t.clone()
    tt[tt >= C] = -100
    loss = F.cross_entropy(up, tt, ignore_index=-100)
    loss.backward()
    assert abs(out[0].item() - loss.item()) < 1e-5 * abs(loss.item())
    assert int(out[1].item()) == int((tt != -100).sum()) and int(out[2].item()) == int((t >= C).sum())
    assert rel(dlr.to_nchw(C), xr.grad) < 2e-5
    assert torch.count_nonzero(dlr.view4()[..., C:]).item() == 0
    # (2) the three-pass path of the library
    full = ops.bilinear_fwd_nchw(lr, C, H, W, ac)
    o3, dfull = ops.ce_fwd_bwd(full, t.cuda())
    d3 = ops.Act.empty(B, h, w, 24, 'cuda', zero=True)
    ops.bilinear_bwd_nchw(dfull, d3, C, ac)
    assert abs(out[0].item() - o3[0].item()) < 2e-6 * abs(o3[0].item()) and out[1].item() == o3[1].item()
    assert rel(dlr.t, d3.t) < 2e-5


def test_cross_entropy_golden(ops, golden_dir):
    import os
    g = dict(np.load(os.path.join(golden_dir, 'loss_metrics.npz')))
    lg = fill.uniform('loss/logits', (2, 21, 32, 32), 4.0)
    tg = fill.labels('loss/target', (2, 32, 32), 21, block=4)
    out, dl = ops.ce_fwd_bwd(lg.cuda(), tg.cuda())
    assert abs(out[0].item() - float(g['ce_loss'])) <= 1e-5 * float(g['ce_loss'])
    assert rel(dl, torch.from_numpy(g['ce_dlogits'])) < TOL
    assert np.array_equal(ops.argmax(lg.cuda()).cpu().numpy(), g['ce_mask'])
    tie = torch.from_numpy(g['tie_logits']).cuda()
    assert np.array_equal(ops.argmax(tie).cpu().numpy(), g['tie_mask'])


def test_argmax_and_confusion(ops):
    from oracle import loss as oloss
    lg = fill.uniform('am', (2, 21, 17, 19), 3.0)
    lg[0, 5, 0, 0] = lg[0, 9, 0, 0] = 10.0  # tie -> first index
    tg = fill.labels('am/t', (2, 17, 19), 21, block=3)
    m = ops.argmax(lg.cuda())
    assert torch.equal(m.cpu(), lg.max(1)[1])
    cnt = torch.zeros(3, 21, dtype=torch.int64, device='cuda')
    ops.confusion(m, tg.cuda(), cnt)
    tp, fn, fp = oloss.class_counts(lg.max(1)[1], tg, 21)
    assert torch.equal(cnt[0].cpu().float(), tp) and torch.equal(cnt[1].cpu().float(), fn) and torch.equal(cnt[2].cpu().float(), fp)


@pytest.mark.parametrize('n', [1000, 4096, 12347])
def test_optimisers(ops, n):
    p0 = fill.uniform('opt/p%d' % n, (n,))
    gs = [fill.uniform('opt/g%d_%d' % (n, i), (n,)) for i in range(3)]
    # SGD momentum 0.9, wd, nesterov
    for nesterov in (False, True):
        pr = p0.clone().requires_grad_()
        opt = torch.optim.SGD([pr], lr=0.1, momentum=0.9, weight_decay=1e-2, nesterov=nesterov)
        p = p0.clone().cuda()
        mb = torch.zeros(n).cuda()
        for i, g in enumerate(gs):
            pr.grad = g.clone() * 0.5
            opt.step()
            ops.sgd_step(p, g.cuda(), mb, 0.1, 0.9, 1e-2, nesterov, 0.5, i == 0)
        assert rel(p, pr) < 1e-5
    for decoupled in (False, True):
        pr = p0.clone().requires_grad_()
        cls = torch.optim.AdamW if decoupled else torch.optim.Adam
        opt = cls([pr], lr=1e-2, weight_decay=1e-2)
        p = p0.clone().cuda()
        m, v = torch.zeros(n).cuda(), torch.zeros(n).cuda()
        for i, g in enumerate(gs):
            pr.grad = g.clone()
            opt.step()
            ops.adam_step(p, g.cuda(), m, v, 1e-2, 0.9, 0.999, 1e-8, 1e-2, decoupled, 1.0, i + 1)
        assert rel(p, pr) < 1e-5
    x = torch.empty(n, device='cuda')
    ops.fill(x, 1.5)
    assert (x == 1.5).all()


@pytest.mark.parametrize('B,C,H,W,stride', [(2, 32, 16, 16, 1), (2, 96, 17, 15, 2), (1, 144, 9, 9, 1)])
def test_depthwise(ops, B, C, H, W, stride):
    key = 'dw/%d_%d_%d' % (C, H, stride)
    x = fill.uniform(key + '/x', (B, C, H, W))
    w = fill.uniform(key + '/w', (C, 1, 3, 3), 0.5)
    Ho, Wo = ops.conv_out_size(H, 3, stride, 1, 1), ops.conv_out_size(W, 3, stride, 1, 1)
    gy = fill.uniform(key + '/g', (B, C, Ho, Wo))
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    ref = F.conv2d(xr, wr, None, stride, 1, 1, groups=C)
    ref.backward(gy)
    w_raw = w[:, 0].permute(1, 2, 0).contiguous().cuda()  # [3][3][C]
    xa = to_act(ops, x)
    ya = ops.Act.empty(B, Ho, Wo, C, 'cuda')
    ops.dwconv_fwd(xa, w_raw, ya, 3, stride, 1)
    assert rel(ya.to_nchw(), ref) < TOL
    gya = to_act(ops, gy)
    dxa = xa.like()
    ops.dwconv_dgrad(gya, w_raw, dxa, 3, stride, 1)
    assert rel(dxa.to_nchw(), xr.grad) < TOL
    dw = torch.empty_like(w_raw)
    ops.dwconv_wgrad(xa, gya, dw, 3, stride, 1)
    assert rel(dw.permute(2, 0, 1).unsqueeze(1), wr.grad) < TOL
