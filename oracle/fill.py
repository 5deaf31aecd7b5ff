"""Deterministic closed-form tensor fill (oracle / test infrastructure).

Golden fixtures must be reproducible on any machine without shipping large
weight tensors and without depending on a torch RNG stream.  Every tensor is
a pure function of (key string, shape): splitmix64 over the flat index, mapped
to uniform [-1, 1) with 24 bits (exact in fp32), times ``scale``.
"""
import zlib

import numpy as np
import torch

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = x + np.uint64(0x9E3779B97F4A7C15)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def key_id(key):
    """Stable 32-bit id of a key string (crc32 -- identical everywhere)."""
    return zlib.crc32(key.encode('utf-8')) & 0xFFFFFFFF


def uniform(key, shape, scale=1.0, dtype=torch.float32):
    """Uniform [-scale, scale) tensor that depends only on (key, shape)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over='ignore'):
        idx = np.arange(n, dtype=np.uint64) + (np.uint64(key_id(key)) << np.uint64(32))
        z = _splitmix64(idx)
    u = (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # [0,1), 24 bits
    v = ((2.0 * u - 1.0) * scale).astype(np.float32)
    return torch.from_numpy(v).reshape(shape).to(dtype)


def labels(key, shape, num_classes, block=1):
    """int64 labels in [0, num_classes).  ``block`` > 1 makes block-constant
    (blocky) label maps like real segmentation masks."""
    if block > 1:
        small = [shape[0]] + [(s + block - 1) // block for s in shape[1:]]
        lab = labels(key, small, num_classes, 1)
        for d in range(1, len(shape)):
            lab = lab.repeat_interleave(block, dim=d)
        sl = tuple(slice(0, s) for s in shape)
        return lab[sl].contiguous()
    n = int(np.prod(shape))
    with np.errstate(over='ignore'):
        idx = np.arange(n, dtype=np.uint64) + (np.uint64(key_id(key)) << np.uint64(32))
        z = _splitmix64(idx)
    v = ((z >> np.uint64(33)) % np.uint64(num_classes)).astype(np.int64)
    return torch.from_numpy(v).reshape(shape)


def images(key, shape):
    """Synthetic normalised images: uint8 uniform [0,255] then the reference's
    mean/std normalisation (reference utils/datasets.py:199-205)."""
    n = int(np.prod(shape))
    with np.errstate(over='ignore'):
        idx = np.arange(n, dtype=np.uint64) + (np.uint64(key_id(key)) << np.uint64(32))
        z = _splitmix64(idx)
    v = ((z >> np.uint64(33)) % np.uint64(256)).astype(np.float32)
    x = torch.from_numpy(v).reshape(shape)
    mean = torch.tensor([123.675, 116.28, 103.53]).reshape(1, 3, 1, 1)
    std = torch.tensor([58.395, 57.12, 57.375]).reshape(1, 3, 1, 1)
    return (x - mean) / std


@torch.no_grad()
def fill_module_(module, prefix, bn_affine_jitter=True):
    """Fill every parameter/buffer of ``module`` in place from closed-form keys.

    conv / linear weights: uniform with Kaiming-like magnitude
    (bound = sqrt(6 / fan_in), i.e. std = sqrt(2 / fan_in)); biases small;
    BN gamma around 1, beta around 0 (jittered so dgamma/dbeta paths are
    exercised); running stats at their torch defaults (0 / 1).
    """
    for name, p in module.named_parameters():
        key = prefix + '/' + name
        if p.dim() >= 2:
            fan_in = p[0].numel()
            bound = (6.0 / fan_in) ** 0.5
            p.copy_(uniform(key, tuple(p.shape), bound))
        elif name.endswith('weight'):  # BN gamma
            if bn_affine_jitter:
                p.copy_(1.0 + uniform(key, tuple(p.shape), 0.25))
            else:
                p.fill_(1.0)
        else:  # bias / BN beta
            if bn_affine_jitter:
                p.copy_(uniform(key, tuple(p.shape), 0.1))
            else:
                p.zero_()
    for name, b in module.named_buffers():
        if name.endswith('running_mean'):
            b.zero_()
        elif name.endswith('running_var'):
            b.fill_(1.0)
        elif name.endswith('num_batches_tracked'):
            b.zero_()
    return module
