"""ORACLE (test infrastructure - NOT part of the product path).

CPU restatement of the reference's image super-resolution U-Net (/root/reference/mm_diffusion/image_unet.py: ResBlock 142-256,
AttentionBlock + QKVAttentionLegacy 258-358, ImageUnet 395-698, ImageSuperResModel 700-715) as one function over a state dict,
plus the tensor-valued DDPM / DDIM loops it is sampled with (gaussian_diffusion.py:400-449,499-559,626-674,748-794).
resblock_updown=True, no class conditioning, legacy attention order.

Parity status: PINNED against fixtures captured from the imported reference (tests/golden/sr_tiny_*.npz).
"""
import math

import torch
import torch.nn.functional as F

from . import diffusion_ref as dref


def _gn(x, sd, pre):
    return F.group_norm(x.float(), 32, sd[pre + ".weight"], sd[pre + ".bias"], eps=1e-5).type(x.dtype)


def _res(x, emb, sd, pre, updown, scale_shift=True):
    h = F.silu(_gn(x, sd, pre + ".in_layers.0"))
    if updown == "down":
        h, x = F.avg_pool2d(h, 2), F.avg_pool2d(x, 2)
    elif updown == "up":
        h, x = F.interpolate(h, scale_factor=2, mode="nearest"), F.interpolate(x, scale_factor=2, mode="nearest")
    h = F.conv2d(h, sd[pre + ".in_layers.2.weight"], sd[pre + ".in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), sd[pre + ".emb_layers.1.weight"], sd[pre + ".emb_layers.1.bias"])[..., None, None]
    if scale_shift:
        scale, shift = torch.chunk(e, 2, dim=1)
        h = _gn(h, sd, pre + ".out_layers.0") * (1 + scale) + shift
    else:
        h = _gn(h + e, sd, pre + ".out_layers.0")
    h = F.conv2d(F.silu(h), sd[pre + ".out_layers.3.weight"], sd[pre + ".out_layers.3.bias"], padding=1)
    if pre + ".skip_connection.weight" in sd:
        x = F.conv2d(x, sd[pre + ".skip_connection.weight"], sd[pre + ".skip_connection.bias"])
    return x + h


def _attn(x, sd, pre, heads):
    b, c, hh, ww = x.shape
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(_gn(xf, sd, pre + ".norm"), sd[pre + ".qkv.weight"], sd[pre + ".qkv.bias"])
    ch = c // heads
    q, k, v = qkv.reshape(b * heads, ch * 3, -1).split(ch, dim=1)            # legacy order: heads first, then q|k|v
    s = 1 / math.sqrt(math.sqrt(ch))
    w = torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s).float(), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, hh * ww)
    return (xf + F.conv1d(a, sd[pre + ".proj_out.weight"], sd[pre + ".proj_out.bias"])).reshape(b, c, hh, ww)


def timestep_embedding(t, dim):
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def sr_forward(sd, cfg, x, t, low_res):
    """cfg: model_channels, channel_mult, num_res_blocks, attention_ds (tuple), heads."""
    mc, mult, nrb, att, heads = cfg["model_channels"], cfg["channel_mult"], cfg["num_res_blocks"], cfg["attention_ds"], cfg["heads"]
    x = torch.cat([x, F.interpolate(low_res, x.shape[-2:], mode="bilinear")], dim=1)
    emb = F.linear(F.silu(F.linear(timestep_embedding(t, mc), sd["time_embed.0.weight"], sd["time_embed.0.bias"])),
                   sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    h = F.conv2d(x, sd["input_blocks.0.0.weight"], sd["input_blocks.0.0.bias"], padding=1)
    hs, idx, ds = [h], 1, 1
    for level in range(len(mult)):
        for _ in range(nrb):
            h = _res(h, emb, sd, f"input_blocks.{idx}.0", None)
            if ds in att:
                h = _attn(h, sd, f"input_blocks.{idx}.1", heads)
            hs.append(h)
            idx += 1
        if level != len(mult) - 1:
            h = _res(h, emb, sd, f"input_blocks.{idx}.0", "down")
            hs.append(h)
            ds *= 2
            idx += 1
    h = _res(h, emb, sd, "middle_block.0", None)
    h = _attn(h, sd, "middle_block.1", heads)
    h = _res(h, emb, sd, "middle_block.2", None)
    idx = 0
    for level in reversed(range(len(mult))):
        for i in range(nrb + 1):
            h = _res(torch.cat([h, hs.pop()], dim=1), emb, sd, f"output_blocks.{idx}.0", None)
            j = 1
            if ds in att:
                h = _attn(h, sd, f"output_blocks.{idx}.{j}", heads)
                j += 1
            if level and i == nrb:
                h = _res(h, emb, sd, f"output_blocks.{idx}.{j}", "up")
                ds //= 2
            idx += 1
    return F.conv2d(F.silu(_gn(h, sd, "out.0")), sd["out.2.weight"], sd["out.2.bias"], padding=1)


@torch.no_grad()
def p_sample_loop(S, model, shape, noise, clip=True):
    """gd:499-559 with a given start noise: the same tensor is re-used as every step's noise (gd:547-556,423-424)."""
    img = noise
    for i in reversed(range(S.T)):
        t = torch.tensor([i] * shape[0])
        mean, logvar, _ = dref.p_mean_variance(S, model(img, S.model_t(t)).float(), img, t, 1, clip)
        nz = (t != 0).float().reshape(-1, 1, 1, 1)
        img = mean + nz * torch.exp(0.5 * logvar) * noise
    return img


@torch.no_grad()
def ddim_sample_loop(S, model, shape, noise, eta=0.0, clip=True):
    img = noise
    for i in reversed(range(S.T)):
        t = torch.tensor([i] * shape[0])
        _, _, x0 = dref.p_mean_variance(S, model(img, S.model_t(t)).float(), img, t, 1, clip)
        eps = (dref._ext(S.sqrt_recip_ac, t, 4) * img - x0) / dref._ext(S.sqrt_recipm1_ac, t, 4)
        ab, ap = dref._ext(S.alphas_cumprod, t, 4), dref._ext(S.alphas_cumprod_prev, t, 4)
        sigma = eta * torch.sqrt((1 - ap) / (1 - ab)) * torch.sqrt(1 - ab / ap)
        z = torch.randn_like(img)
        img = x0 * torch.sqrt(ap) + torch.sqrt(1 - ap - sigma ** 2) * eps + (t != 0).float().reshape(-1, 1, 1, 1) * sigma * z
    return img
