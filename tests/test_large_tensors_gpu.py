"""Operands above the 2 GiB buffer-descriptor cap (reference train.py:88-90: -s / -bs are free-form).

The conv kernels read their gathered operand through one buffer descriptor with 32-bit offsets (< 2 GiB); ops.conv2d_* hand a
larger operand to the library in equal batch chunks (pytorch_segmentation_amd/ops.py, "operands above 2 GiB").  Checked here at
2.68 GB operands: forward with fused BatchNorm statistics, data gradient, weight gradient -- against the same kernels run image
by image and against torch on the device in fp64 -- and the whole DeepLabV3+ in eval mode at B = 40, 1024 x 1024 (2.68 GB
stride-2 and stride-4 maps in the stem and layer 1): batch-independent, every image equal to the image evaluated alone."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()


@pytest.fixture()
def fp32_policy():
    from pytorch_segmentation_amd import ops
    before = ops.POLICY_NAME
    ops.set_conv_precision('fp32')
    yield
    ops.set_conv_precision(before)


def test_conv_ops_on_operands_above_two_gib(fp32_policy):
    from pytorch_segmentation_amd import ops
    from pytorch_segmentation_amd.ops import Act
    B, H, W, Cw, Cn = 40, 256, 256, 256, 64            # wide tensor: 40 x 256 x 256 x 256 fp32 = 2.68 GB
    g = torch.Generator(device='cuda').manual_seed(0)
    wide = Act.empty(B, H, W, Cw, 'cuda')
    wide.t.uniform_(-1, 1, generator=g)
    assert ops._span_bytes(wide) > ops._CAP_BYTES and ops._batch_chunks(wide) == 20
    # ---- forward 1x1 256 -> 64 with fused statistics: x above the cap
    w = torch.empty(Cn * Cw, device='cuda').uniform_(-0.1, 0.1, generator=g)
    y = Act.empty(B, H, W, Cn, 'cuda')
    st, rows, group = ops.conv2d_fwd(wide, w, None, y, 1, 1, 1, 0, 1, want_stats=True)
    for b in (0, 19, 20, 39):
        y1 = Act.empty(1, H, W, Cn, 'cuda')
        ops.conv2d_fwd(ops._sub(wide, b, 1), w, None, y1, 1, 1, 1, 0, 1)
        assert rel(ops._sub(y, b, 1).view4(), y1.view4()) < 1e-6, b
    yv = y.view4().reshape(-1, Cn)
    ref0 = wide.view4()[0].reshape(-1, Cw).double() @ w.view(Cn, Cw).double().t()
    assert rel(yv[:H * W], ref0) < 1e-5
    # the statistics: pivoted sums per row group -> column sums of the whole tensor
    K, S1, S2 = st[0].double(), st[1].double(), st[2].double()
    cnt = (B * H * W - group * torch.arange(rows, dtype=torch.float64, device='cuda')).clamp(min=0.0, max=float(group))
    assert rows * group == B * H * W
    colsum = (S1 + K * cnt[:, None]).sum(0)
    colsq = (S2 + 2 * K * S1 + K * K * cnt[:, None]).sum(0)
    assert rel(colsum, yv.double().sum(0)) < 1e-6 and rel(colsq, (yv.double() ** 2).sum(0)) < 1e-6
    co = ops.bn_finalize((st, rows, group), B * H * W, None, None, None, None, 0.0, 1e-5)
    # (the channel means are ~1e-4 of the spread: judged against the spread, not against themselves)
    assert (co[0].double() - yv.double().mean(0)).abs().max().item() < 1e-6 * yv.double().std().item()
    # ---- data gradient of a 64 -> 256 conv: dy (256 channels) above the cap, dx 64 channels
    wT = torch.empty(Cn * Cw, device='cuda').uniform_(-0.1, 0.1, generator=g)          # [Cin = 64][1][Cout = 256]
    dx = Act.empty(B, H, W, Cn, 'cuda')
    ops.conv2d_dgrad(wide, wT, dx, 1, 1, 1, 0, 1)
    assert dx.bnpart is None
    for b in (0, 20, 39):
        d1 = Act.empty(1, H, W, Cn, 'cuda')
        ops.conv2d_dgrad(ops._sub(wide, b, 1), wT, d1, 1, 1, 1, 0, 1)
        assert rel(ops._sub(dx, b, 1).view4(), d1.view4()) < 1e-6, b
    ref = wide.view4()[39].reshape(-1, Cw).double() @ wT.view(Cn, Cw).double().t()
    assert rel(dx.view4()[39].reshape(-1, Cn), ref) < 1e-5
    # ---- weight gradient of the 256 -> 64 conv: x above the cap; chunks add into dw
    dy = Act.empty(B, H, W, Cn, 'cuda')
    dy.t.uniform_(-1, 1, generator=g)
    dw = torch.empty(Cn * Cw, device='cuda')
    ops.conv2d_wgrad(wide, dy, dw, 1, 1, 1, 0, 1)
    ref = torch.zeros(Cn, Cw, dtype=torch.float64, device='cuda')
    for b in range(B):
        ref += dy.view4()[b].reshape(-1, Cn).double().t() @ wide.view4()[b].reshape(-1, Cw).double()
    assert rel(dw.view(Cn, Cw), ref) < 1e-5
    dw2 = dw.clone()
    ops.conv2d_wgrad(wide, dy, dw2, 1, 1, 1, 0, 1, accumulate=True)          # accumulate: the first chunk adds too
    assert rel(dw2, 2 * dw) < 1e-6
    dw3 = torch.empty_like(dw)
    ops.conv2d_wgrad(wide, dy, dw3, 1, 1, 1, 0, 1)
    assert torch.equal(dw3, dw)                                              # fixed chunk order: bit-reproducible
    # what does not chunk keeps the library's message: ONE image above the cap
    big1 = Act(wide.t, 1, 40 * H, W, Cw, Cw)
    y1 = Act.empty(1, 40 * H, W, Cn, 'cuda')
    from pytorch_segmentation_amd import _lib
    with pytest.raises(_lib.PsegError, match='2 GiB'):
        ops.conv2d_fwd(big1, w, None, y1, 1, 1, 1, 0, 1)


def test_deeplab_eval_batch40_1024_is_batch_independent(fp32_policy):
    """DeepLabV3+ (reference models/deeplabv3plus.py:28-44) in eval mode at B = 40, 1024 x 1024: the stem's stride-2 map and layer
    1's stride-4 maps are 2.68 GB each (above the descriptor cap: the convs that READ them run in two batch chunks); every probed
    image of the batch equals the same image evaluated alone."""
    from oracle import fill
    from pytorch_segmentation_amd import prepare
    from pytorch_segmentation_amd.models import DeepLabV3Plus
    m = DeepLabV3Plus(21)
    fill.fill_module_(m, 'big40')
    prepare(m, 'cuda')
    m.eval()
    x = fill.images('big40/x', (8, 3, 1024, 1024)).cuda().repeat(5, 1, 1, 1)      # 40 images (five copies of eight)
    x[39] = x[39].flip(-1)                                                        # ... the last one made different
    with torch.no_grad():
        full = m(x)
        assert tuple(full.shape) == (40, 21, 1024, 1024) and torch.isfinite(full).all()
        for b in (0, 19, 20, 39):
            one = m(x[b:b + 1].contiguous())
            assert rel(full[b:b + 1], one) < 1e-5, b
        assert rel(full[8], full[0]) < 1e-6 and rel(full[39], full[7]) > 1e-3


def test_half_conv_ops_on_operands_above_two_gib():
    """The same under the half-precision (`-mp`) storage: a 40 x 256 x 256 x 512 fp16 tensor (2.68 GB) as the gathered operand of
    the fp16 forward conv (fused statistics of the values as stored), data gradient and weight gradient (fp32 result, chunks add)."""
    from pytorch_segmentation_amd import ops
    from pytorch_segmentation_amd.ops import Act
    B, H, W, Cw, Cn = 40, 256, 256, 512, 64
    g = torch.Generator(device='cuda').manual_seed(1)
    wide = Act.empty(B, H, W, Cw, 'cuda', dtype=torch.float16)
    wide.t.copy_(torch.empty(wide.t.numel(), device='cuda').uniform_(-1, 1, generator=g))
    assert ops._span_bytes(wide) > ops._CAP_BYTES and ops._batch_chunks(wide) == 20
    w = torch.empty(Cn * Cw, device='cuda').uniform_(-0.05, 0.05, generator=g).half()
    y = Act.empty(B, H, W, Cn, 'cuda', dtype=torch.float16)
    st, rows, group = ops.conv2d_fwd(wide, w, None, y, 1, 1, 1, 0, 1, want_stats=True)
    for b in (0, 20, 39):
        y1 = Act.empty(1, H, W, Cn, 'cuda', dtype=torch.float16)
        ops.conv2d_fwd(ops._sub(wide, b, 1), w, None, y1, 1, 1, 1, 0, 1)
        assert torch.equal(ops._sub(y, b, 1).view4(), y1.view4()), b          # products of halves are exact: same fp32 sums, same rounding
    ref = (wide.view4()[39].reshape(-1, Cw).float() @ w.view(Cn, Cw).float().t())
    assert rel(y.view4()[39].reshape(-1, Cn).float(), ref) < 2e-3                # one fp16 rounding of the result
    yv = y.view4().reshape(-1, Cn).double()
    K, S1, S2 = st[0].double(), st[1].double(), st[2].double()
    assert rows * group == B * H * W
    colsum = (S1 + K * group).sum(0)
    assert rel(colsum, yv.sum(0)) < 1e-5
    # data gradient: dy = the wide tensor (512 channels), dx 64 channels
    wT = torch.empty(Cn * Cw, device='cuda').uniform_(-0.05, 0.05, generator=g).half()      # [Cin = 64][1][Cout = 512]
    dx = Act.empty(B, H, W, Cn, 'cuda', dtype=torch.float16)
    ops.conv2d_dgrad(wide, wT, dx, 1, 1, 1, 0, 1)
    for b in (0, 39):
        d1 = Act.empty(1, H, W, Cn, 'cuda', dtype=torch.float16)
        ops.conv2d_dgrad(ops._sub(wide, b, 1), wT, d1, 1, 1, 1, 0, 1)
        assert torch.equal(ops._sub(dx, b, 1).view4(), d1.view4()), b
    # weight gradient: fp32 result, the two chunks add
    dy = Act.empty(B, H, W, Cn, 'cuda', dtype=torch.float16)
    dy.t.copy_(torch.empty(dy.t.numel(), device='cuda').uniform_(-1, 1, generator=g))
    dw = torch.empty(Cn * Cw, device='cuda')
    ops.conv2d_wgrad(wide, dy, dw, 1, 1, 1, 0, 1)
    ref = torch.zeros(Cn, Cw, dtype=torch.float64, device='cuda')
    for b in range(B):
        ref += dy.view4()[b].reshape(-1, Cn).double().t() @ wide.view4()[b].reshape(-1, Cw).double()
    assert rel(dw.view(Cn, Cw), ref) < 1e-5
