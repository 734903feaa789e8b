"""ORACLE (test infrastructure - NOT part of the product path).

CPU restatement, in plain fp32 PyTorch ops, of the reference's coupled
multimodal U-Net forward.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product (mm-diffusion_amd/) never
does and fails loudly without its HIP library.

Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement
against fixtures captured from the imported reference itself
(tools/gen_golden.py -> tests/golden/*.npz): per-block outputs, full forwards
(tiny / tiny+learn_sigma / mid configs), 2- and 4-step p_sample loops and the
full-size config-1 2-step loop.

Written from the behaviour of (citations relative to /root/reference/mm_diffusion):
  multimodal_unet.py:68-131   VideoConv / AudioConv
  multimodal_unet.py:133-208  Upsample / Downsample
  multimodal_unet.py:212-287  SingleModalAtten
  multimodal_unet.py:291-495  ResBlock
  multimodal_unet.py:498-678  QKVAttention / CrossAttentionBlock (RS-MMA)
  multimodal_unet.py:697-1101 MultimodalUNet
  nn.py:16-33                 GroupNorm32
  nn.py:192-210               timestep_embedding

The structure here is NOT the reference's module tree: the network is a flat
list of block descriptors (`build_arch`) interpreted over a state dict, the
layout is channels-first [N,C,F,H,W] throughout (converted only at the API
edge), and the cross-modal windows are addressed arithmetically (frames
(i+shift+w) mod F) instead of through the reference's index matrices.
"""
import math
import random

import torch
import torch.nn.functional as F_

GN_GROUPS = 32
GN_EPS = 1e-5


# --------------------------------------------------------------------------- architecture table
def parse_cfg(flags: dict) -> dict:
    """Normalise the reference flag surface (multimodal_script_util.py:131-201) into plain ints."""
    def ints(v):
        if isinstance(v, str):
            return [int(i) for i in v.split(",")]
        return [int(i) for i in v]

    video_size = ints(flags["video_size"])
    audio_size = ints(flags["audio_size"])
    cm = flags.get("channel_mult", "")
    if cm == "" or cm is None:
        cm = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4),
              64: (1, 2, 3, 4)}[video_size[-1]]
    elif isinstance(cm, str):
        cm = tuple(int(c) for c in cm.split(","))
    learn_sigma = bool(flags.get("learn_sigma", False))
    return dict(
        video_size=video_size, audio_size=audio_size, model_channels=int(flags["num_channels"]),
        num_res_blocks=int(flags["num_res_blocks"]), channel_mult=tuple(cm),
        cross_res=ints(flags["cross_attention_resolutions"]),
        cross_win=ints(flags["cross_attention_windows"]),
        cross_shift=bool(flags["cross_attention_shift"]),
        vattn_res=ints(flags["video_attention_resolutions"]),
        aattn_res=ints(flags["audio_attention_resolutions"]),
        num_heads=int(flags["num_heads"]), num_head_channels=int(flags["num_head_channels"]),
        use_scale_shift_norm=bool(flags["use_scale_shift_norm"]),
        resblock_updown=bool(flags["resblock_updown"]),
        video_out_channels=6 if learn_sigma else 3, audio_out_channels=2 if learn_sigma else 1,
    )


def build_arch(cfg: dict):
    """Flat description of the network (multimodal_unet.py:799-1012).

    Returns (input_blocks, middle, output_blocks); each block is a list of layer dicts
    {kind: 'init'|'res'|'cross', prefix: state-dict prefix, ...}.
    """
    mc = cfg["model_channels"]
    cm = cfg["channel_mult"]
    nrb = cfg["num_res_blocks"]

    def res(prefix, cin, cout, dil, up=False, down=False, vattn=False, aattn=False):
        return dict(kind="res", prefix=prefix, cin=cin, cout=cout, dilation=2 ** (dil % 10),
                    up=up, down=down, vattn=vattn, aattn=aattn)

    def cross(prefix, ch, window, shift):
        heads = cfg["num_heads"] if cfg["num_head_channels"] == -1 else ch // cfg["num_head_channels"]
        return dict(kind="cross", prefix=prefix, ch=ch, heads=heads, window=window, shift=shift)

    ch = int(cm[0] * mc)
    chans = [ch]
    inputs = [[dict(kind="init", prefix="input_blocks.0.0", cout=ch)]]
    ds, dil = 1, 1
    for level, mult in enumerate(cm):
        for _ in range(nrb):
            i = len(inputs)
            cout = int(mult * mc)
            layers = [res(f"input_blocks.{i}.0", ch, cout, dil,
                          vattn=ds in cfg["vattn_res"], aattn=ds in cfg["aattn_res"])]
            dil += 1
            ch = cout
            if ds in cfg["cross_res"]:
                w = cfg["cross_win"][cfg["cross_res"].index(ds)]
                layers.append(cross(f"input_blocks.{i}.1", ch, w, cfg["cross_shift"]))
            inputs.append(layers)
            chans.append(ch)
        if level != len(cm) - 1:
            i = len(inputs)
            inputs.append([res(f"input_blocks.{i}.0", ch, ch, dil, down=True)])
            dil += 1
            chans.append(ch)
            ds *= 2
    if cfg["cross_win"] == [1, 4, 8]:
        middle = [res("middle_blocks.0", ch, ch, dil, vattn=True, aattn=True),
                  cross("middle_blocks.1", ch, cfg["video_size"][0], False),
                  res("middle_blocks.2", ch, ch, dil, vattn=True, aattn=True)]
    else:
        middle = [res("middle_blocks.0", ch, ch, dil, vattn=True, aattn=True),
                  res("middle_blocks.1", ch, ch, dil, vattn=True, aattn=True)]
    dil -= 1
    outputs = []
    for level, mult in list(enumerate(cm))[::-1]:
        for bid in range(nrb + 1):
            ich = chans.pop()
            i = len(outputs)
            cout = int(mc * mult)
            layers = [res(f"output_blocks.{i}.0", ch + ich, cout, dil,
                          vattn=ds in cfg["vattn_res"], aattn=ds in cfg["aattn_res"])]
            dil -= 1
            ch = cout
            if ds in cfg["cross_res"]:
                w = cfg["cross_win"][cfg["cross_res"].index(ds)]
                layers.append(cross(f"output_blocks.{i}.{len(layers)}", ch, w, cfg["cross_shift"]))
            if level and bid == nrb:
                if cfg["resblock_updown"]:
                    layers.append(res(f"output_blocks.{i}.{len(layers)}", ch, ch, dil, up=True))
                ds //= 2
            outputs.append(layers)
    return inputs, middle, outputs


# --------------------------------------------------------------------------- primitives (channels-first)
def group_norm(x, w, b):
    """GroupNorm32 (nn.py:16-33): 32 groups, eps 1e-5, statistics over (C/32, *rest) per sample."""
    return F_.group_norm(x.float(), GN_GROUPS, w, b, GN_EPS)


def timestep_embedding(t, dim, max_period=10000):
    """nn.py:192-210: [cos(t f_k) | sin(t f_k)], f_k = exp(-ln(max_period) k / half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def video_conv_2d1d(x, sd, p):
    """VideoConv('2d+1d') (multimodal_unet.py:91-99) on x[N,C,F,H,W]: 3x3 over (H,W) then k=3 over F."""
    ws, bs = sd[p + ".video_conv_spatial.weight"], sd[p + ".video_conv_spatial.bias"]
    wt, bt = sd[p + ".video_conv_temporal.weight"], sd[p + ".video_conv_temporal.bias"]
    k = ws.shape[-1]
    y = F_.conv3d(x, ws[:, :, None], bs, padding=(0, k // 2, k // 2))
    kt = wt.shape[-1]
    return F_.conv3d(y, wt[:, :, :, None, None], bt, padding=(kt // 2, 0, 0))


def video_conv_3d(x, sd, p):
    """VideoConv('3d') (multimodal_unet.py:101-104): k=1 or k=3 'same' conv3d."""
    w, b = sd[p + ".video_conv.weight"], sd[p + ".video_conv.bias"]
    return F_.conv3d(x, w, b, padding=w.shape[-1] // 2)


def audio_conv(x, sd, p, dilation=1):
    """AudioConv (multimodal_unet.py:108-131): Conv1d 'same', k in {1,3}, dilated."""
    w, b = sd[p + ".audio_conv.weight"], sd[p + ".audio_conv.bias"]
    k = w.shape[-1]
    return F_.conv1d(x, w, b, padding=dilation * (k // 2), dilation=dilation)


def _attend(q, k, v, heads):
    """softmax((q s)(k s)^T) v per head, s = ch^-1/4 (multimodal_unet.py:228-240, 535-542).
    q [B, C, Tq], k/v [B, C, Tk] -> [B, C, Tq]."""
    B, C, Tq = q.shape
    ch = C // heads
    s = 1.0 / math.sqrt(math.sqrt(ch))
    qh = (q * s).reshape(B * heads, ch, Tq)
    kh = (k * s).reshape(B * heads, ch, -1)
    vh = v.reshape(B * heads, ch, -1)
    w = torch.softmax(torch.einsum("bct,bcs->bts", qh, kh).float(), dim=-1)
    return torch.einsum("bts,bcs->bct", w, vh).reshape(B, C, Tq)


def self_attention(x, sd, p, heads):
    """SingleModalAtten (multimodal_unet.py:246-287) on x[B, C, T]."""
    h = group_norm(x, sd[p + ".norm.GroupNorm.weight"], sd[p + ".norm.GroupNorm.bias"])
    qkv = F_.conv1d(h, sd[p + ".qkv.weight"], sd[p + ".qkv.bias"])
    q, k, v = qkv.chunk(3, dim=1)
    a = _attend(q, k, v, heads)
    return x + F_.conv1d(a, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])


def cross_attention(video, audio, sd, layer, shift):
    """CrossAttentionBlock._forward (multimodal_unet.py:655-678) with arithmetic windows.

    video [N,C,F,H,W], audio [N,C,L].  Frame i's video tokens attend the audio tokens
    ((i+shift)*apf + j) mod L, j < window*apf; audio segment i attends the video tokens
    ((i+shift)*HW + j) mod (F*HW), j < window*HW  (multimodal_unet.py:624-638).  The last
    audio segment also owns the L - F*apf remainder queries (multimodal_unet.py:547-548).
    """
    p, heads, win = layer["prefix"], layer["heads"], layer["window"]
    N, C, F, H, W = video.shape
    L = audio.shape[-1]
    HW = H * W
    apf = int(L / F)
    vt = video.reshape(N, C, F * HW)
    v_qkv = F_.conv1d(group_norm(vt, sd[p + ".v_norm.GroupNorm.weight"], sd[p + ".v_norm.GroupNorm.bias"]),
                      sd[p + ".v_qkv.weight"], sd[p + ".v_qkv.bias"])
    a_qkv = F_.conv1d(group_norm(audio, sd[p + ".a_norm.GroupNorm.weight"], sd[p + ".a_norm.GroupNorm.bias"]),
                      sd[p + ".a_qkv.weight"], sd[p + ".a_qkv.bias"])
    vq, vk, vv = v_qkv.chunk(3, dim=1)
    aq, ak, av = a_qkv.chunk(3, dim=1)
    v_out = torch.empty_like(vq)
    a_out = torch.empty_like(aq)
    for i in range(F):
        a_idx = (torch.arange(win * apf) + (i + shift) * apf) % L
        v_out[:, :, i * HW:(i + 1) * HW] = _attend(vq[:, :, i * HW:(i + 1) * HW], ak[:, :, a_idx], av[:, :, a_idx], heads)
        v_idx = (torch.arange(win * HW) + (i + shift) * HW) % (F * HW)
        hi = L if i == F - 1 else (i + 1) * apf
        a_out[:, :, i * apf:hi] = _attend(aq[:, :, i * apf:hi], vk[:, :, v_idx], vv[:, :, v_idx], heads)
    vh = F_.conv3d(v_out.reshape(N, C, F, H, W), sd[p + ".video_proj_out.video_conv.weight"],
                   sd[p + ".video_proj_out.video_conv.bias"])
    ah = F_.conv1d(a_out, sd[p + ".audio_proj_out.audio_conv.weight"], sd[p + ".audio_proj_out.audio_conv.bias"])
    return video + vh, audio + ah


def res_block(video, audio, emb, sd, layer, cfg):
    """ResBlock._forward (multimodal_unet.py:434-495), eval mode (dropout = identity)."""
    p = layer["prefix"]
    N, C, F, H, W = video.shape

    def in_layers(x, mod):
        h = group_norm(x, sd[f"{p}.{mod}_in_layers.0.GroupNorm.weight"], sd[f"{p}.{mod}_in_layers.0.GroupNorm.bias"])
        h = F_.silu(h)
        if mod == "video":
            return video_conv_2d1d(h, sd, f"{p}.video_in_layers.2")
        return audio_conv(h, sd, f"{p}.audio_in_layers.2", layer["dilation"])

    vh, ah = in_layers(video, "video"), in_layers(audio, "audio")
    if layer["down"]:      # conv at input resolution THEN pool both h and x (unet:441-448)
        vh, video = F_.avg_pool3d(vh, (1, 2, 2)), F_.avg_pool3d(video, (1, 2, 2))
        ah, audio = F_.avg_pool1d(ah, 4), F_.avg_pool1d(audio, 4)
    elif layer["up"]:
        up3 = lambda t: t.repeat_interleave(2, dim=3).repeat_interleave(2, dim=4)
        vh, video = up3(vh), up3(video)
        ah, audio = ah.repeat_interleave(4, dim=2), audio.repeat_interleave(4, dim=2)
    e = F_.linear(F_.silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])

    def out_layers(h, mod):
        gw, gb = sd[f"{p}.{mod}_out_layers.0.GroupNorm.weight"], sd[f"{p}.{mod}_out_layers.0.GroupNorm.bias"]
        bshape = (N, -1, 1, 1, 1) if mod == "video" else (N, -1, 1)
        if cfg["use_scale_shift_norm"]:
            scale, shift = e.chunk(2, dim=1)
            h = group_norm(h, gw, gb) * (1 + scale.reshape(bshape)) + shift.reshape(bshape)
        else:
            h = group_norm(h + e.reshape(bshape), gw, gb)
        h = F_.silu(h)
        if mod == "video":
            return video_conv_3d(h, sd, f"{p}.video_out_layers.3")
        return audio_conv(h, sd, f"{p}.audio_out_layers.3")

    vh, ah = out_layers(vh, "video"), out_layers(ah, "audio")
    if layer["cin"] != layer["cout"]:
        video = video_conv_3d(video, sd, p + ".video_skip_connection")
        audio = audio_conv(audio, sd, p + ".audio_skip_connection")
    video, audio = video + vh, audio + ah
    if layer["vattn"]:
        Co, Hh, Ww = video.shape[1], video.shape[3], video.shape[4]
        heads = cfg["num_heads"]     # self-attention always uses num_heads (unet:410-419)
        x = video.permute(0, 2, 1, 3, 4).reshape(N * F, Co, Hh * Ww)            # (b f) c (h w)
        x = self_attention(x, sd, p + ".spatial_attention_block", heads)
        x = x.reshape(N, F, Co, Hh * Ww).permute(0, 3, 2, 1).reshape(N * Hh * Ww, Co, F)  # (b h w) c f
        x = self_attention(x, sd, p + ".temporal_attention_block", heads)
        video = x.reshape(N, Hh, Ww, Co, F).permute(0, 3, 4, 1, 2)               # b c f h w
    if layer["aattn"]:
        audio = self_attention(audio, sd, p + ".audio_attention_block", cfg["num_heads"])
    return video, audio


# --------------------------------------------------------------------------- full forward
@torch.no_grad()
def unet_forward(sd, cfg, video, audio, timesteps, shifts=None):
    """MultimodalUNet.forward (multimodal_unet.py:1058-1101).

    sd: state dict (reference key names), cfg: parse_cfg(flags),
    video [N,F,C,H,W], audio [N,C,L], timesteps [N] (int or float).
    shifts: None -> draw random.randint(0, F-window) per shifted cross block like the
    reference (multimodal_unet.py:619-620); or an iterator/list consumed in call order.
    Returns (video_out [N,F,Cv,H,W], audio_out [N,Ca,L]).
    """
    it = iter(shifts) if shifts is not None else None
    F = cfg["video_size"][0]

    def next_shift(layer):
        if not layer["shift"]:
            return 0
        if it is None:
            return random.randint(0, F - layer["window"])
        return int(next(it))

    inputs, middle, outputs = build_arch(cfg)
    mc = cfg["model_channels"]
    emb = timestep_embedding(timesteps, mc)
    emb = F_.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F_.linear(F_.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    v = video.float().permute(0, 2, 1, 3, 4).contiguous()   # [N,C,F,H,W]
    a = audio.float()

    def run(layers, v, a):
        for layer in layers:
            if layer["kind"] == "init":
                v = video_conv_2d1d(v, sd, layer["prefix"] + ".video_conv")
                a = audio_conv(a, sd, layer["prefix"] + ".audio_conv")
            elif layer["kind"] == "res":
                v, a = res_block(v, a, emb, sd, layer, cfg)
            else:
                v, a = cross_attention(v, a, sd, layer, next_shift(layer))
        return v, a

    vs, as_ = [], []
    for layers in inputs:
        v, a = run(layers, v, a)
        vs.append(v)
        as_.append(a)
    v, a = run(middle, v, a)
    for layers in outputs:
        v = torch.cat([v, vs.pop()], dim=1)
        a = torch.cat([a, as_.pop()], dim=1)
        v, a = run(layers, v, a)
    v = F_.silu(group_norm(v, sd["video_out.0.GroupNorm.weight"], sd["video_out.0.GroupNorm.bias"]))
    v = video_conv_3d(v, sd, "video_out.2")
    a = F_.silu(group_norm(a, sd["audio_out.0.GroupNorm.weight"], sd["audio_out.0.GroupNorm.bias"]))
    a = audio_conv(a, sd, "audio_out.2")
    return v.permute(0, 2, 1, 3, 4).contiguous(), a


class OracleModel:
    """Callable with the reference forward signature model(video, audio, timesteps)."""

    def __init__(self, sd, flags, shifts=None):
        self.sd = {k: v.float() for k, v in sd.items()}
        self.cfg = parse_cfg(flags)
        self.shifts = iter(shifts) if shifts is not None else None

    def __call__(self, video, audio, timesteps, **kw):
        return unet_forward(self.sd, self.cfg, video, audio, timesteps, self.shifts)
