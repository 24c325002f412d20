"""Deterministic, seed-free weights for parity tests.  TEST INFRASTRUCTURE ONLY.

Both the fixture generator (which loads them into the imported reference model) and the
GPU parity tests (which load them into the HIP-backed model) regenerate the same values
from the tensor name alone, so no 30 MB state_dict has to be committed.

    w[name][i] = half_width(name) * (2 * mix32(i + crc32(name)) / 2**32 - 1)

`half_width` is the uniform bound with the same variance class as the reference's init
(see nefnet_oracle.reference_style_init); BatchNorm gamma sits near 1, beta near 0.
"""
import math
import zlib

import numpy as np
import torch

from .nefnet_oracle import buffer_shapes, param_shapes, param_shapes2


def _mix32(x):
    x = x.astype(np.uint64) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def unit_noise(name, n):
    """n doubles in [-1, 1), a pure function of (name, index)."""
    h = zlib.crc32(name.encode())
    idx = np.arange(n, dtype=np.uint64) + np.uint64(h)
    return _mix32(idx).astype(np.float64) / 2.0 ** 31 - 1.0


def _half_width(name, shp):
    if name.startswith("W_encoder."):
        return math.sqrt(3.0) * math.sqrt(2.0 / (shp[2] * shp[2] * shp[0])) * 4.0
    if len(shp) >= 2:
        fan_in = shp[1] * (shp[2] if len(shp) == 3 else 1)
        return math.sqrt(3.0 / fan_in)
    return 0.05


def hashed_params2():
    """Model_nefnet2's tensors (single-lead inventory + the two single convs)."""
    return hashed_params(1, param_shapes2())


def hashed_params(V, shapes=None):
    P = {}
    for name, shp in (shapes or param_shapes(V)).items():
        n = int(np.prod(shp))
        u = unit_noise(name, n)
        if ".double_conv.1." in name or ".double_conv.4." in name:
            vals = (1.0 + 0.2 * u) if name.endswith("weight") else 0.1 * u
        else:
            vals = _half_width(name, shp) * u
        P[name] = torch.from_numpy(vals.astype(np.float32).reshape(shp)).clone()
    return P


def hashed_buffers():
    Bf = {}
    for name, shp in buffer_shapes().items():
        if name.endswith("num_batches_tracked"):
            Bf[name] = torch.zeros((), dtype=torch.int64)
            continue
        u = unit_noise(name, int(np.prod(shp)))
        vals = (0.5 + 0.25 * (u + 1.0)) if name.endswith("running_var") else 0.2 * u
        Bf[name] = torch.from_numpy(vals.astype(np.float32).reshape(shp)).clone()
    return Bf


def hashed_masks(V, B, T, p=0.2):
    """Keep-masks (uint8 0/1) for the eight dropout sites, shape of each site's activation."""
    C = 128 * V
    shapes = {
        "W_encoder.layer1.0": (B, C, T), "W_encoder.layer1.1": (B, C, T), "W_encoder.layer1.2": (B, C, T),
        "w_conv.0": (B, C, T), "z1_conv.0": (B, C, T), "z2_conv1.0": (B, C, T),
        "z2_conv2.0": (B, 7 * C, 16), "z2_conv2.2": (B, 7 * C, 32),
    }
    out = {}
    for site, shp in shapes.items():
        u = unit_noise("mask:" + site, int(np.prod(shp)))
        out[site] = torch.from_numpy(((u + 1.0) * 0.5 >= p).astype(np.uint8).reshape(shp))
    return out
