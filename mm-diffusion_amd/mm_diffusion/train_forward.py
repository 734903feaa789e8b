"""Differentiable (training-mode) forward of the coupled U-Net: the same block walk as the inference engine, but eager
and built from the autograd Functions of train_ops.py so that loss.backward() runs the libmmd backward kernels.

Mirrors reference multimodal_unet.py:1058-1101 (forward), 434-495 (ResBlock), 655-678 (CrossAttentionBlock), including
its activation-recompute semantics for the shifted cross-attention blocks (nn.checkpoint, always on: unet:650-652).
"""
import torch
import torch.nn.functional as F_

from . import ops
from . import train_ops as T
from .ops import Geom

AUDIO_K3 = [(-1, 0, 0), (0, 0, 0), (1, 0, 0)]


def _pad_cols(x, mult=8):
    c = x.shape[1]
    return x if c % mult == 0 else F_.pad(x, (0, mult - c % mult))


def train_forward(model, video, audio, timesteps):
    P = dict(model.named_parameters())
    dt = model.dtype
    N, Fr, Cv, H, W = video.shape
    L = audio.shape[-1]
    mc = model.model_channels
    arch_in, arch_mid, arch_out = model._arch
    heads_self = model.num_heads
    ss = model.use_scale_shift_norm
    src = model.shift_source
    import random
    draw = src if src is not None else random.randint

    # ---- timestep embedding MLP (fp32)
    e0 = torch.empty(N, mc, dtype=torch.float32, device=video.device)
    ops.timestep_embedding(timesteps.contiguous(), mc, e0)
    emb = T.LinearFn.apply(T.SiluFn.apply(T.LinearFn.apply(e0, P["time_embed.0.weight"], P["time_embed.0.bias"])),
                           P["time_embed.2.weight"], P["time_embed.2.bias"])
    semb = T.SiluFn.apply(emb)

    def gn(x, prefix, geom, act, film=None):
        return T.group_norm(x, P[prefix + ".GroupNorm.weight"], P[prefix + ".GroupNorm.bias"], geom, act, film=film)

    def pw(x, prefix, residual=None):
        return T.conv(x, P[prefix + ".weight"], P[prefix + ".bias"], residual=residual)

    def self_attn(x, prefix, kind, Hh):
        rows, C = x.shape
        HW = Hh * Hh
        if kind == "spatial":
            geom = Geom.spatial(N, Fr, HW)
        elif kind == "temporal":
            geom = Geom.temporal(N, Fr, HW)
        else:
            geom = Geom.per_sample(N, rows // N)
            HW = rows // N
        qkv = pw(gn(x, prefix + ".norm", geom, False), prefix + ".qkv")
        att = T.SelfAttnFn.apply(qkv, heads_self, kind, N, Fr if kind != "audio" else 1, HW)
        return pw(att, prefix + ".proj_out", residual=x)

    def res_block(v, a, layer, Hh, Ll):
        p, cin, cout = layer["prefix"], layer["cin"], layer["cout"]
        film = T.LinearFn.apply(semb, P[p + ".emb_layers.1.weight"], P[p + ".emb_layers.1.bias"])
        down, up = layer["down"], layer["up"]
        Ho = Hh // 2 if down else (Hh * 2 if up else Hh)
        Lo = Ll // 4 if down else (Ll * 4 if up else Ll)

        def stream(x, mod):
            vid = mod == "video"
            rows_in = x.shape[0]
            h = gn(x, f"{p}.{mod}_in_layers.0", Geom.per_sample(N, rows_in // N), True)
            if vid:
                h = T.conv(h, P[f"{p}.video_in_layers.2.video_conv_spatial.weight"], P[f"{p}.video_in_layers.2.video_conv_spatial.bias"],
                           taps=ops.TAPS_SPATIAL, dims=(N * Fr, Hh, Hh))
                h = T.conv(h, P[f"{p}.video_in_layers.2.video_conv_temporal.weight"], P[f"{p}.video_in_layers.2.video_conv_temporal.bias"],
                           taps=ops.TAPS_TEMPORAL, dims=(Fr, Hh * Hh, 1))
            else:
                h = T.conv(h, P[f"{p}.audio_in_layers.2.audio_conv.weight"], P[f"{p}.audio_in_layers.2.audio_conv.bias"],
                           taps=ops.taps_audio(layer["dilation"]), dims=(Ll, 1, 1))
            xs = x
            if down or up:
                mode = 0 if down else 1
                if vid:
                    h = T.ResampleFn.apply(h, N * Fr, Hh, Hh, 2, 2, mode)
                    xs = T.ResampleFn.apply(x, N * Fr, Hh, Hh, 2, 2, mode)
                else:
                    h = T.ResampleFn.apply(h, N, 1, Ll, 1, 4, mode)
                    xs = T.ResampleFn.apply(x, N, 1, Ll, 1, 4, mode)
            rows_out = h.shape[0]
            if not ss:       # h + emb_out, then the plain norm (unet:473-477)
                h = T.RowBiasFn.apply(h, film, rows_out // N)
            h = gn(h, f"{p}.{mod}_out_layers.0", Geom.per_sample(N, rows_out // N), True, film=film if ss else None)
            if model.dropout > 0 and model.training:
                h = T.DropoutFn.apply(h, float(model.dropout))
            conv = "video_conv" if vid else "audio_conv"
            sk = xs if cin == cout else pw(xs, f"{p}.{mod}_skip_connection.{conv}")
            out = pw(h, f"{p}.{mod}_out_layers.3.{conv}", residual=sk)
            if vid and layer["vattn"]:
                out = self_attn(out, p + ".spatial_attention_block", "spatial", Ho)
                out = self_attn(out, p + ".temporal_attention_block", "temporal", Ho)
            if (not vid) and layer["aattn"]:
                out = self_attn(out, p + ".audio_attention_block", "audio", Ho)
            return out

        return stream(v, "video"), stream(a, "audio"), Ho, Lo

    def cross_block(v, a, layer, Hh, Ll):
        p, C, heads, win = layer["prefix"], layer["ch"], layer["heads"], layer["window"]
        HW = Hh * Hh
        names = [p + s for s in (".v_norm.GroupNorm.weight", ".v_norm.GroupNorm.bias", ".a_norm.GroupNorm.weight", ".a_norm.GroupNorm.bias",
                                 ".v_qkv.weight", ".v_qkv.bias", ".a_qkv.weight", ".a_qkv.bias",
                                 ".video_proj_out.video_conv.weight", ".video_proj_out.video_conv.bias",
                                 ".audio_proj_out.audio_conv.weight", ".audio_proj_out.audio_conv.bias")]

        def run(vv, aa):
            shift = draw(0, Fr - win) if layer["shift"] else 0             # drawn at EVERY evaluation (forward and recompute)
            shift = shift if torch.is_tensor(shift) else int(shift)       # device slot (graph-captured step) or python int
            vqkv = pw(gn(vv, p + ".v_norm", Geom.per_sample(N, Fr * HW), False), p + ".v_qkv")
            aqkv = pw(gn(aa, p + ".a_norm", Geom.per_sample(N, Ll), False), p + ".a_qkv")
            vatt, aatt = T.CrossAttnFn.apply(vqkv, aqkv, heads, N, Fr, HW, Ll, win, shift)
            return (pw(vatt, p + ".video_proj_out.video_conv", residual=vv), pw(aatt, p + ".audio_proj_out.audio_conv", residual=aa))

        return T.RecomputeFn.apply(run, 2, v, a, *[P[n] for n in names])

    def run_layers(layers, v, a, Hh, Ll):
        for layer in layers:
            if layer["kind"] == "res":
                v, a, Hh, Ll = res_block(v, a, layer, Hh, Ll)
            elif layer["kind"] == "cross":
                v, a = cross_block(v, a, layer, Hh, Ll)
            else:   # init: 2d+1d stem on the padded input channels, audio k=3 stem
                p = layer["prefix"]
                xv = _pad_cols(video.float().permute(0, 1, 3, 4, 2).reshape(-1, Cv)).to(dt)
                ws = P[p + ".video_conv.video_conv_spatial.weight"]
                ws = F_.pad(ws, (0, 0, 0, 0, 0, xv.shape[1] - Cv))
                v = T.conv(xv, ws, P[p + ".video_conv.video_conv_spatial.bias"], taps=ops.TAPS_SPATIAL, dims=(N * Fr, Hh, Hh))
                v = T.conv(v, P[p + ".video_conv.video_conv_temporal.weight"], P[p + ".video_conv.video_conv_temporal.bias"],
                           taps=ops.TAPS_TEMPORAL, dims=(Fr, Hh * Hh, 1))
                Ca = audio.shape[1]
                xa = _pad_cols(audio.float().permute(0, 2, 1).reshape(-1, Ca)).to(dt)
                wa = F_.pad(P[p + ".audio_conv.audio_conv.weight"], (0, 0, 0, xa.shape[1] - Ca))
                a = T.conv(xa, wa, P[p + ".audio_conv.audio_conv.bias"], taps=AUDIO_K3, dims=(Ll, 1, 1))
        return v, a, Hh, Ll

    Hh, Ll = H, L
    v = a = None
    vs, as_ = [], []
    for layers in arch_in:
        v, a, Hh, Ll = run_layers(layers, v, a, Hh, Ll)
        vs.append(v)
        as_.append(a)
    v, a, Hh, Ll = run_layers(arch_mid, v, a, Hh, Ll)
    for layers in arch_out:
        v = T.CatFn.apply(v, vs.pop())
        a = T.CatFn.apply(a, as_.pop())
        v, a, Hh, Ll = run_layers(layers, v, a, Hh, Ll)

    # ---- heads: GN -> SiLU -> conv (output channels padded to 8 for the GEMM, sliced back)
    Co_v, Co_a = model.video_out_channels, model.audio_out_channels
    hv = gn(v, "video_out.0", Geom.per_sample(N, Fr * Hh * Hh), True)
    wv = P["video_out.2.video_conv.weight"]
    wv8 = F_.pad(wv, (0, 0, 0, 0, 0, 0, 0, 0, 0, 8 - Co_v))
    bv8 = F_.pad(P["video_out.2.video_conv.bias"], (0, 8 - Co_v))
    yv = T.conv(hv, wv8, bv8, taps=ops.TAPS_3D, dims=(Fr, Hh, Hh))[:, :Co_v]
    ha = gn(a, "audio_out.0", Geom.per_sample(N, Ll), True)
    wa8 = F_.pad(P["audio_out.2.audio_conv.weight"], (0, 0, 0, 0, 0, 8 - Co_a))
    ba8 = F_.pad(P["audio_out.2.audio_conv.bias"], (0, 8 - Co_a))
    ya = T.conv(ha, wa8, ba8, taps=AUDIO_K3, dims=(Ll, 1, 1))[:, :Co_a]
    video_out = yv.float().reshape(N, Fr, Hh, Hh, Co_v).permute(0, 1, 4, 2, 3).contiguous()
    audio_out = ya.float().reshape(N, Ll, Co_a).permute(0, 2, 1).contiguous()
    return video_out, audio_out
