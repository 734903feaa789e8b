"""MI355X-native coupled multimodal U-Net behind the reference's `MultimodalUNet` surface.

Drop-in boundary (reference /root/reference/mm_diffusion/multimodal_unet.py:697-1101):
  * class name, constructor arguments, `forward(video, audio, timesteps, label=None)`
  * attributes video_size / audio_size / video_out_channels / audio_out_channels / dtype
  * state-dict keys, shapes AND registration order identical to the reference (tests/golden/state_dict_keys.json),
    so Landscape.pt / AIST++.pt load unchanged; `load_state_dict_`, `convert_to_fp16/32`

What is different by design: the module tree only HOLDS parameters.  The arithmetic is a flat launch plan of
hand-written HIP kernels (mm_diffusion/engine.py -> libmmd.so) over channels-last activations; there is no
torch-op forward and no CPU fallback - calling the model without the HIP library or with CPU tensors raises.
"""
import math
import random

import torch
import torch.nn as nn

from . import logger
from ._hip import MMDError


# --------------------------------------------------------------------------- parameter holders
class _Bag(nn.Module):
    """Named container without behaviour (mirrors the reference's nesting so state-dict keys match)."""

    def put(self, name, mod):
        self.add_module(str(name), mod)
        return mod


class _Affine(nn.Module):
    """weight/bias holder with the reference's default initialisation."""

    def __init__(self, wshape, kind="conv", zero=False):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*wshape))
        self.bias = nn.Parameter(torch.empty(wshape[0]))
        with torch.no_grad():
            if kind == "norm":
                self.weight.fill_(1.0)
                self.bias.zero_()
            elif zero:                                   # zero_module (nn.py:140-147)
                self.weight.zero_()
                self.bias.zero_()
            else:                                        # nn.Conv*/nn.Linear default: kaiming_uniform(a=sqrt(5))
                fan_in = 1
                for s in wshape[1:]:
                    fan_in *= s
                bound = 1.0 / math.sqrt(fan_in)
                self.weight.uniform_(-bound, bound)
                self.bias.uniform_(-bound, bound)


def _gn(ch):
    b = _Bag()
    b.put("GroupNorm", _Affine((ch,), "norm"))
    return b


def _video_conv(cin, cout, k, conv_type, zero=False):
    b = _Bag()
    if conv_type == "2d+1d":
        b.put("video_conv_spatial", _Affine((cout, cin, k, k), zero=zero))
        b.put("video_conv_temporal", _Affine((cout, cout, k), zero=zero))
    elif conv_type == "3d":
        b.put("video_conv", _Affine((cout, cin, k, k, k), zero=zero))
    else:
        raise NotImplementedError(conv_type)
    return b


def _audio_conv(cin, cout, k, zero=False):
    b = _Bag()
    b.put("audio_conv", _Affine((cout, cin, k), zero=zero))
    return b


def _self_attn(ch):
    b = _Bag()
    b.put("norm", _gn(ch))
    b.put("qkv", _Affine((3 * ch, ch, 1)))
    b.put("proj_out", _Affine((ch, ch, 1), zero=True))
    return b


def _res_block(cin, cout, emb_ch, scale_shift, vattn, aattn, video_type):
    b = _Bag()
    vin = b.put("video_in_layers", _Bag())
    vin.put(0, _gn(cin)), vin.put(1, nn.Identity()), vin.put(2, _video_conv(cin, cout, 3, video_type))
    ain = b.put("audio_in_layers", _Bag())
    ain.put(0, _gn(cin)), ain.put(1, nn.Identity()), ain.put(2, _audio_conv(cin, cout, 3))
    emb = b.put("emb_layers", _Bag())
    emb.put(0, nn.Identity()), emb.put(1, _Affine((2 * cout if scale_shift else cout, emb_ch), "linear"))
    vout = b.put("video_out_layers", _Bag())
    vout.put(0, _gn(cout)), vout.put(1, nn.Identity()), vout.put(2, nn.Identity())
    vout.put(3, _video_conv(cout, cout, 1, "3d", zero=True))
    aout = b.put("audio_out_layers", _Bag())
    aout.put(0, _gn(cout)), aout.put(1, nn.Identity()), aout.put(2, nn.Identity())
    aout.put(3, _audio_conv(cout, cout, 1, zero=True))
    if cin == cout:
        b.put("video_skip_connection", nn.Identity()), b.put("audio_skip_connection", nn.Identity())
    else:
        b.put("video_skip_connection", _video_conv(cin, cout, 1, "3d"))
        b.put("audio_skip_connection", _audio_conv(cin, cout, 1))
    if vattn:
        b.put("spatial_attention_block", _self_attn(cout)), b.put("temporal_attention_block", _self_attn(cout))
    if aattn:
        b.put("audio_attention_block", _self_attn(cout))
    return b


def _cross_block(ch):
    b = _Bag()
    b.put("v_norm", _gn(ch)), b.put("a_norm", _gn(ch))
    b.put("v_qkv", _Affine((3 * ch, ch, 1))), b.put("a_qkv", _Affine((3 * ch, ch, 1)))
    b.put("video_proj_out", _video_conv(ch, ch, 1, "3d", zero=True))
    b.put("audio_proj_out", _audio_conv(ch, ch, 1, zero=True))
    return b


class MultimodalUNet(nn.Module):
    """The coupled video/audio U-Net (constructor signature: reference multimodal_unet.py:737-764)."""

    def __init__(self, video_size, audio_size, model_channels, video_out_channels, audio_out_channels, num_res_blocks,
                 cross_attention_resolutions, cross_attention_windows, cross_attention_shift,
                 video_attention_resolutions, audio_attention_resolutions, video_type="2d+1d", audio_type="1d",
                 dropout=0, channel_mult=(1, 2, 3, 4), num_classes=None, use_checkpoint=False, use_fp16=False,
                 num_heads=1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False,
                 resblock_updown=True):
        super().__init__()
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        if video_type != "2d+1d" or audio_type != "1d":
            raise NotImplementedError("only video_type='2d+1d' / audio_type='1d' (the shipped configuration) are built")
        if num_classes is not None:
            raise NotImplementedError("class conditioning is dead code in the reference (num_classes is always None)")
        self.video_size, self.audio_size = list(video_size), list(audio_size)
        self.model_channels = model_channels
        self.video_out_channels, self.audio_out_channels = video_out_channels, audio_out_channels
        self.num_res_blocks = num_res_blocks
        self.cross_attention_resolutions = list(cross_attention_resolutions)
        self.cross_attention_windows = list(cross_attention_windows)
        self.cross_attention_shift = cross_attention_shift
        self.video_attention_resolutions = list(video_attention_resolutions)
        self.audio_attention_resolutions = list(audio_attention_resolutions)
        self.dropout = dropout
        self.channel_mult = tuple(channel_mult)
        self.num_classes = num_classes
        self.use_checkpoint = use_checkpoint
        # 16-bit mode on MI355X is bf16 (same exponent range as fp32: no loss scaling needed); the reference's
        # use_fp16 flag (fp16_util.py:13-21) selects it.  GroupNorm statistics / softmax / accumulation stay fp32.
        self.dtype = torch.bfloat16 if use_fp16 else torch.float32
        self.num_heads, self.num_head_channels, self.num_heads_upsample = num_heads, num_head_channels, num_heads_upsample
        self.use_scale_shift_norm = use_scale_shift_norm
        self.resblock_updown = resblock_updown
        # shift source for the random-shift cross-attention windows: callable(lo, hi) -> int.  Default = Python's
        # global `random.randint`, exactly the stream the reference consumes (multimodal_unet.py:619-620).
        self.shift_source = None
        self._engines = {}

        mc, cm, nrb = model_channels, self.channel_mult, num_res_blocks
        te = self.time_embed = _Bag()
        te.put(0, _Affine((mc, mc), "linear")), te.put(1, nn.Identity()), te.put(2, _Affine((mc, mc), "linear"))

        def res(cin, cout, dil, **kw):
            return dict(kind="res", cin=cin, cout=cout, dilation=2 ** (dil % 10), up=kw.get("up", False),
                        down=kw.get("down", False), vattn=kw.get("vattn", False), aattn=kw.get("aattn", False))

        def cross(ch, window, shift):
            heads = num_heads if num_head_channels == -1 else ch // num_head_channels
            if num_head_channels != -1 and ch % num_head_channels:
                raise ValueError(f"q,k,v channels {ch} is not divisible by num_head_channels {num_head_channels}")
            return dict(kind="cross", ch=ch, heads=heads, window=window, shift=bool(shift))

        def realise(block_layers, prefix, holder):
            for j, layer in enumerate(block_layers):
                layer["prefix"] = f"{prefix}.{j}" if holder is not None else prefix
                if layer["kind"] == "init":
                    m = _Bag()
                    m.put("video_conv", _video_conv(self.video_size[1], layer["cout"], 3, "2d+1d"))
                    m.put("audio_conv", _audio_conv(self.audio_size[0], layer["cout"], 3))
                elif layer["kind"] == "res":
                    m = _res_block(layer["cin"], layer["cout"], mc, use_scale_shift_norm, layer["vattn"], layer["aattn"], video_type)
                else:
                    m = _cross_block(layer["ch"])
                holder.put(j, m)

        ch = int(cm[0] * mc)
        chans = [ch]
        arch_in = [[dict(kind="init", cout=ch)]]
        ds, dil = 1, 1
        for level, mult in enumerate(cm):
            for _ in range(nrb):
                cout = int(mult * mc)
                layers = [res(ch, cout, dil, vattn=ds in self.video_attention_resolutions, aattn=ds in self.audio_attention_resolutions)]
                dil += 1
                ch = cout
                if ds in self.cross_attention_resolutions:
                    layers.append(cross(ch, self.cross_attention_windows[self.cross_attention_resolutions.index(ds)], cross_attention_shift))
                arch_in.append(layers)
                chans.append(ch)
            if level != len(cm) - 1:
                arch_in.append([res(ch, ch, dil, down=True)])
                dil += 1
                chans.append(ch)
                ds *= 2
        if self.cross_attention_windows == [1, 4, 8]:
            arch_mid = [res(ch, ch, dil, vattn=True, aattn=True), cross(ch, self.video_size[0], False),
                        res(ch, ch, dil, vattn=True, aattn=True)]
        else:
            arch_mid = [res(ch, ch, dil, vattn=True, aattn=True), res(ch, ch, dil, vattn=True, aattn=True)]
        dil -= 1
        arch_out = []
        for level, mult in list(enumerate(cm))[::-1]:
            for bid in range(nrb + 1):
                ich = chans.pop()
                cout = int(mc * mult)
                layers = [res(ch + ich, cout, dil, vattn=ds in self.video_attention_resolutions, aattn=ds in self.audio_attention_resolutions)]
                layers[0]["skip_ch"] = ich
                dil -= 1
                ch = cout
                if ds in self.cross_attention_resolutions:
                    layers.append(cross(ch, self.cross_attention_windows[self.cross_attention_resolutions.index(ds)], cross_attention_shift))
                if level and bid == nrb:
                    if resblock_updown:
                        layers.append(res(ch, ch, dil, up=True))
                    ds //= 2
                arch_out.append(layers)

        self.input_blocks = _Bag()
        for i, layers in enumerate(arch_in):
            realise(layers, f"input_blocks.{i}", self.input_blocks.put(i, _Bag()))
        self.middle_blocks = _Bag()
        realise(arch_mid, "middle_blocks", self.middle_blocks)
        for j, layer in enumerate(arch_mid):
            layer["prefix"] = f"middle_blocks.{j}"
        self.output_blocks = _Bag()
        for i, layers in enumerate(arch_out):
            realise(layers, f"output_blocks.{i}", self.output_blocks.put(i, _Bag()))
        input_ch = int(cm[0] * mc)
        ao = self.audio_out = _Bag()
        ao.put(0, _gn(ch)), ao.put(1, nn.Identity()), ao.put(2, _audio_conv(input_ch, audio_out_channels, 3, zero=True))
        vo = self.video_out = _Bag()
        vo.put(0, _gn(ch)), vo.put(1, nn.Identity()), vo.put(2, _video_conv(input_ch, video_out_channels, 3, "3d", zero=True))
        self._arch = (arch_in, arch_mid, arch_out)
        self._final_ch = ch

    # ----------------------------------------------------------------------- reference utility surface
    def convert_to_fp16(self):
        """Reference: conv weights -> half (multimodal_unet.py:1013-1021).  Here: switch the HIP plan to bf16
        activations / GEMM weights.  Parameters stay fp32 masters (they are re-packed to bf16 for the kernels)."""
        self.dtype = torch.bfloat16
        self._engines.clear()

    def convert_to_fp32(self):
        self.dtype = torch.float32
        self._engines.clear()

    def load_state_dict_(self, state_dict, is_strict=False):
        """Tolerant loader (multimodal_unet.py:1033-1054): drops shape-mismatched keys, logs missing/unused."""
        own = self.state_dict()
        for key, val in own.items():
            if key in state_dict:
                if val.shape != state_dict[key].shape:
                    state_dict.pop(key)
                    logger.log("{} not matchable with state_dict with shape {}".format(key, val.shape))
            else:
                logger.log("{} not exists in state_dict".format(key))
        for key in state_dict:
            if key not in own:
                logger.log("{} not used in state_dict".format(key))
        self.load_state_dict(state_dict, strict=is_strict)
        self._engines.clear()

    def load_state_dict(self, *a, **kw):
        out = super().load_state_dict(*a, **kw)
        self._engines.clear()
        return out

    def _apply(self, fn, *a, **kw):          # .to() / .cuda(): parameters moved -> packed weights are stale
        out = super()._apply(fn, *a, **kw)
        self._engines.clear()
        return out

    # ----------------------------------------------------------------------- forward
    def draw_shifts(self):
        """One shift per shifted CrossAttentionBlock, in forward call order (reference unet:619-620)."""
        src = self.shift_source or random.randint
        F = self.video_size[0]
        out = []
        for blk in self._arch[0] + [self._arch[1]] + self._arch[2]:
            for layer in blk:
                if layer["kind"] == "cross" and layer["shift"]:
                    out.append(int(src(0, F - layer["window"])))
        return out

    def engine(self, batch, device, replica=0):
        """The launch plan for `batch` samples (built on first use).  `replica` > 0 = a further, independent engine of the same
        batch size (own activation buffers and streams, shared packed weights): the batch lanes of sampler.GraphStepper."""
        from .engine import UNetEngine
        key = (int(batch), self.dtype, str(device)) + ((int(replica),) if replica else ())
        eng = self._engines.get(key)
        if eng is None or eng.stale():
            from . import _hip as H
            with H.preserve_rng(device):                  # the plan builder's autotuner must not consume the caller's noise stream
                eng = UNetEngine(self, int(batch), self.dtype, device)
            self._engines[key] = eng
        return eng

    def release_engines(self):
        """Drop every cached launch plan (activation pools, private streams, captured graphs) and the packed-weight cache: HBM goes
        back to the allocator.  The training loop calls this after its periodic sample dump so that nothing of the sampling engines
        stays resident next to the training step; the next no-grad forward rebuilds what it needs."""
        from . import _hip as H
        for eng in list(self._engines.values()):
            eng.close()
        self._engines.clear()
        self.__dict__.pop("_wcache", None)
        H.reap()

    def forward(self, video, audio, timesteps, label=None):
        """video [N,F,C,H,W], audio [N,C,L], timesteps [N] -> (video_out [N,F,Cv,H,W], audio_out [N,Ca,L]).

        Outputs are fp32 (the head kernels accumulate and store fp32 in both precisions)."""
        assert (label is not None) == (self.num_classes is not None), \
            "must specify y if and only if the model is class-conditional"
        if not video.is_cuda:
            raise MMDError("MultimodalUNet runs on the MI355X HIP path only: move the model and inputs to the GPU "
                           "(there is no CPU/torch fallback; the CPU restatement lives in oracle/ for tests)")
        if torch.is_grad_enabled() and (video.requires_grad or audio.requires_grad or
                                        any(p.requires_grad for p in self.parameters())):
            # differentiable path: eager walk over autograd Functions whose forward AND backward are libmmd kernels
            from .train_forward import train_forward
            return train_forward(self, video, audio, timesteps)
        if self.training and self.dropout > 0:
            raise NotImplementedError("dropout needs the differentiable path (call with grad enabled) or model.eval()")
        eng = self.engine(video.shape[0], video.device)
        return eng.forward(video, audio, timesteps, self.draw_shifts())
