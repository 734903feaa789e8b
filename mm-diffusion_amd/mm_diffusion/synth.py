"""Deterministic synthetic weights for benchmarks and parity fixtures.

The reference ships no weights and a freshly constructed reference model
outputs exactly zero (every output conv / projection is `zero_module`d,
/root/reference/mm_diffusion/multimodal_unet.py:377,385,609-610,1006,1011), so
both the golden fixtures and `bench.py` need weights that exercise every
branch.  Each tensor is derived from its state-dict key alone, so the GPU box
can rebuild bit-identical weights without any weight file:

    g = torch.Generator().manual_seed(crc32(key))
    weight (ndim >= 2) : randn(shape, g) / sqrt(fan_in)     (fan_in = prod(shape[1:]))
    GroupNorm weight   : 1 + 0.1 * randn(shape, g)      (= every 1-D '.weight')
    every bias / GroupNorm bias : 0.1 * randn(shape, g)

This keeps activations O(1) through the whole network, so an error in any
branch (attention, FiLM, skip) is visible at the output.
"""
import math
import zlib

import torch


def synth_tensor(key: str, shape) -> torch.Tensor:
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(key.encode("utf-8")))
    shape = tuple(int(s) for s in shape)
    x = torch.randn(shape, generator=g, dtype=torch.float32)
    if len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        return x / math.sqrt(fan_in)
    if key.endswith(".weight"):          # every 1-D weight is a GroupNorm gain ("...GroupNorm.weight", image U-Net "...norm.weight")
        return 1.0 + 0.1 * x
    return 0.1 * x


def synth_state_dict(keys_and_shapes) -> dict:
    """keys_and_shapes: iterable of (key, shape) -> {key: fp32 cpu tensor}."""
    return {k: synth_tensor(k, s) for k, s in keys_and_shapes}


@torch.no_grad()
def synth_init_(module: torch.nn.Module) -> torch.nn.Module:
    """Overwrite every parameter of `module` in place with its key-seeded value."""
    sd = module.state_dict()
    new = {k: synth_tensor(k, v.shape).to(v.dtype) for k, v in sd.items()}
    module.load_state_dict(new, strict=True)
    return module
