"""64 -> 256 image super-resolution U-Net on the MI355X HIP path (reference mm_diffusion/image_unet.py:395-715:
`ImageUnet`, `ImageSuperResModel`; blocks image_unet.py:80-305,327-358).

Same surface as the reference: constructor arguments, `forward(x, timesteps, low_res=None)` on `[N, 3, H, W]` images,
`load_state_dict_`, `convert_to_fp16/32`, and the reference's state-dict keys / shapes / order (a reference
checkpoint loads unchanged).  The module tree only holds parameters; the arithmetic is a walk over the same libmmd
kernels as the multimodal U-Net on channels-last rows `[(n h w), C]`:

  ResBlock        gn_stats + gn_apply(+SiLU) -> [avg-pool / nearest x2 on h and x] -> 3x3 implicit-GEMM conv -> FiLM'd
                  GroupNorm(+SiLU) -> 3x3 conv with the skip (identity or 1x1 conv) as the GEMM residual
  AttentionBlock  GroupNorm -> 1x1 qkv GEMM (weights re-ordered ONCE at pack time from the legacy
                  [head][q|k|v][ch] channel order, image_unet.py:336-353, to [q|k|v][head][ch]) -> flash MFMA attention
                  over T = H*W tokens -> 1x1 proj GEMM with the residual
  skip `th.cat`   column-slice copies into one pre-sized buffer (mmd_copy2d)
  SuperRes input  bilinear upsample of `low_res` + channel concat in one kernel (mmd_bilinear_concat) feeding the stem conv

Inference only (the SR model is a sampling-time component of multimodal_sample_sr.py:186-253).  Not built:
`resblock_updown=False` (strided-conv Downsample / conv Upsample - the shipped SR checkpoint uses resblock_updown),
class conditioning, `use_new_attention_order` is accepted (it only changes the pack-time re-ordering).
"""
import math

import os

import torch
import torch.nn as nn

from . import _hip as H
from . import logger, ops
from ._hip import MMDError
from .multimodal_unet import _Affine, _Bag
from .ops import Geom


def _res_block(cin, cout, emb_ch, scale_shift):
    b = _Bag()
    il = b.put("in_layers", _Bag())
    il.put(0, _Affine((cin,), "norm")), il.put(1, nn.Identity()), il.put(2, _Affine((cout, cin, 3, 3)))
    el = b.put("emb_layers", _Bag())
    el.put(0, nn.Identity()), el.put(1, _Affine((2 * cout if scale_shift else cout, emb_ch), "linear"))
    ol = b.put("out_layers", _Bag())
    ol.put(0, _Affine((cout,), "norm")), ol.put(1, nn.Identity()), ol.put(2, nn.Identity()), ol.put(3, _Affine((cout, cout, 3, 3), zero=True))
    b.put("skip_connection", nn.Identity() if cin == cout else _Affine((cout, cin, 1, 1)))
    return b


def _attn_block(ch):
    b = _Bag()
    b.put("norm", _Affine((ch,), "norm")), b.put("qkv", _Affine((3 * ch, ch, 1))), b.put("proj_out", _Affine((ch, ch, 1), zero=True))
    return b


class ImageUnet(nn.Module):
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None, use_checkpoint=False, use_fp16=False,
                 num_heads=1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 use_new_attention_order=False):
        super().__init__()
        if dims != 2:
            raise NotImplementedError("ImageUnet: only dims=2 is built")
        if num_classes is not None:
            raise NotImplementedError("ImageUnet: class conditioning is not built")
        if not resblock_updown:
            raise NotImplementedError("ImageUnet: resblock_updown=False (strided-conv Downsample / conv Upsample) is not built; the "
                                      "shipped SR model uses resblock_updown=True (ssh_scripts/multimodal_sample_sr.sh:10-13)")
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        self.image_size, self.in_channels, self.model_channels, self.out_channels = image_size, in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions, self.dropout = num_res_blocks, tuple(attention_resolutions), dropout
        self.channel_mult, self.conv_resample, self.num_classes, self.use_checkpoint = tuple(channel_mult), conv_resample, num_classes, use_checkpoint
        self.dtype = torch.bfloat16 if use_fp16 else torch.float32         # the MI355X 16-bit type is bf16
        self.num_heads, self.num_head_channels, self.num_heads_upsample = num_heads, num_head_channels, num_heads_upsample
        self.use_scale_shift_norm, self.use_new_attention_order = use_scale_shift_norm, use_new_attention_order
        ted = model_channels * 4
        te = self.time_embed = _Bag()
        te.put(0, _Affine((ted, model_channels), "linear")), te.put(1, nn.Identity()), te.put(2, _Affine((ted, ted), "linear"))

        # ---- module tree (parameter holders) + the flat layer list the forward walks
        plan_in, plan_out = [], []
        ch = input_ch = int(channel_mult[0] * model_channels)
        self.input_blocks = _Bag()
        blk = self.input_blocks.put(0, _Bag())
        blk.put(0, _Affine((ch, in_channels, 3, 3)))
        plan_in.append([("stem", "input_blocks.0.0", in_channels, ch)])
        chans = [ch]
        ds, idx = 1, 1

        def heads_of(c, nh):
            return nh if num_head_channels == -1 else c // num_head_channels
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                blk = self.input_blocks.put(idx, _Bag())
                layers = []
                cout = int(mult * model_channels)
                blk.put(0, _res_block(ch, cout, ted, use_scale_shift_norm))
                layers.append(("res", f"input_blocks.{idx}.0", ch, cout, None))
                ch = cout
                if ds in self.attention_resolutions:
                    blk.put(1, _attn_block(ch))
                    layers.append(("attn", f"input_blocks.{idx}.1", ch, heads_of(ch, num_heads)))
                plan_in.append(layers)
                chans.append(ch)
                idx += 1
            if level != len(channel_mult) - 1:
                blk = self.input_blocks.put(idx, _Bag())
                blk.put(0, _res_block(ch, ch, ted, use_scale_shift_norm))
                plan_in.append([("res", f"input_blocks.{idx}.0", ch, ch, "down")])
                chans.append(ch)
                ds *= 2
                idx += 1
        self.middle_block = _Bag()
        self.middle_block.put(0, _res_block(ch, ch, ted, use_scale_shift_norm))
        self.middle_block.put(1, _attn_block(ch))
        self.middle_block.put(2, _res_block(ch, ch, ted, use_scale_shift_norm))
        plan_mid = [("res", "middle_block.0", ch, ch, None), ("attn", "middle_block.1", ch, heads_of(ch, num_heads)),
                    ("res", "middle_block.2", ch, ch, None)]
        self.output_blocks = _Bag()
        idx = 0
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                blk = self.output_blocks.put(idx, _Bag())
                cout = int(model_channels * mult)
                blk.put(0, _res_block(ch + ich, cout, ted, use_scale_shift_norm))
                layers = [("res", f"output_blocks.{idx}.0", ch + ich, cout, None)]
                ch = cout
                j = 1
                if ds in self.attention_resolutions:
                    blk.put(j, _attn_block(ch))
                    layers.append(("attn", f"output_blocks.{idx}.{j}", ch, heads_of(ch, num_heads_upsample)))
                    j += 1
                if level and i == num_res_blocks:
                    blk.put(j, _res_block(ch, ch, ted, use_scale_shift_norm))
                    layers.append(("res", f"output_blocks.{idx}.{j}", ch, ch, "up"))
                    ds //= 2
                plan_out.append(layers)
                idx += 1
        self.out = _Bag()
        self.out.put(0, _Affine((ch,), "norm")), self.out.put(1, nn.Identity()), self.out.put(2, _Affine((out_channels, input_ch, 3, 3), zero=True))
        self._plan = (plan_in, plan_mid, plan_out)
        self._packed = None
        self._graphs = {}

    # ------------------------------------------------------------------ reference surface
    def convert_to_fp16(self):
        self.dtype = torch.bfloat16
        self._packed = None

    def convert_to_fp32(self):
        self.dtype = torch.float32
        self._packed = None

    def load_state_dict_(self, state_dict, is_strict=False):
        """Tolerant loader (image_unet.py:651-672): drops shape-mismatched keys, logs missing / unused ones."""
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

    def load_state_dict(self, *a, **kw):
        out = super().load_state_dict(*a, **kw)
        self._packed = None
        return out

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._packed = None
        return out

    # ------------------------------------------------------------------ packed weights
    def _pack(self, device):
        P = {k: v.detach() for k, v in self.named_parameters()}
        dt = self.dtype
        W = {}
        f32 = lambda t: t.float().contiguous()          # noqa: E731
        plan_in, plan_mid, plan_out = self._plan
        for layers in plan_in + [plan_mid] + plan_out:
            for L in layers:
                kind, pre = L[0], L[1]
                if kind == "stem":
                    wst = P[pre + ".weight"].float()
                    cpad = (wst.shape[1] + 7) // 8 * 8            # GEMM form: input channels padded to a 16-byte row (SR model: 6 -> 8)
                    wg = torch.zeros(wst.shape[0], cpad, *wst.shape[2:], device=wst.device)
                    wg[:, :wst.shape[1]] = wst
                    W[pre] = (ops.pack_edge_weight(P[pre + ".weight"]), f32(P[pre + ".bias"]), ops.pack_conv_weight(wg, dt), cpad)
                elif kind == "res":
                    cin, cout = L[2], L[3]
                    W[pre] = dict(
                        g1=f32(P[pre + ".in_layers.0.weight"]), b1=f32(P[pre + ".in_layers.0.bias"]),
                        w_in=ops.pack_conv_weight(P[pre + ".in_layers.2.weight"].float(), dt), b_in=f32(P[pre + ".in_layers.2.bias"]),
                        w_e=f32(P[pre + ".emb_layers.1.weight"]), b_e=f32(P[pre + ".emb_layers.1.bias"]),
                        g2=f32(P[pre + ".out_layers.0.weight"]), b2=f32(P[pre + ".out_layers.0.bias"]),
                        w_out=ops.pack_conv_weight(P[pre + ".out_layers.3.weight"].float(), dt), b_out=f32(P[pre + ".out_layers.3.bias"]),
                        w_skip=None if cin == cout else ops.pack_conv_weight(P[pre + ".skip_connection.weight"].float(), dt),
                        b_skip=None if cin == cout else f32(P[pre + ".skip_connection.bias"]))
                else:
                    C, heads = L[2], L[3]
                    ch = C // heads
                    wq, bq = P[pre + ".qkv.weight"].float().reshape(3 * C, C), P[pre + ".qkv.bias"].float()
                    if not self.use_new_attention_order:     # legacy rows [head][q|k|v][ch] -> [q|k|v][head][ch]
                        perm = torch.arange(3 * C, device=wq.device).reshape(heads, 3, ch).permute(1, 0, 2).reshape(-1)
                        wq, bq = wq[perm], bq[perm]
                    W[pre] = dict(g=f32(P[pre + ".norm.weight"]), b=f32(P[pre + ".norm.bias"]), w_qkv=wq.to(dt).contiguous(), b_qkv=bq.contiguous(),
                                  w_proj=P[pre + ".proj_out.weight"].float().reshape(C, C).to(dt).contiguous(), b_proj=f32(P[pre + ".proj_out.bias"]))
        # every ResBlock's emb_layers Linear in ONE launch over the row-concatenated weights (image_unet.py:176-182 applies them one by one)
        offs, Ws, bs, off = {}, [], [], 0
        for layers in plan_in + [plan_mid] + plan_out:
            for L in layers:
                if L[0] == "res":
                    offs[L[1]] = (off, W[L[1]]["w_e"].shape[0])
                    Ws.append(W[L[1]]["w_e"])
                    bs.append(W[L[1]]["b_e"])
                    off += W[L[1]]["w_e"].shape[0]
        W["emb_all"] = (torch.cat(Ws).contiguous(), torch.cat(bs).contiguous(), offs, off)
        W["time_embed"] = tuple(f32(P[k]) for k in ("time_embed.0.weight", "time_embed.0.bias", "time_embed.2.weight", "time_embed.2.bias"))
        W["out"] = (f32(P["out.0.weight"]), f32(P["out.0.bias"]), ops.pack_edge_weight(P["out.2.weight"]), f32(P["out.2.bias"]))
        self._packed = (str(device), dt, W)
        return W

    # ------------------------------------------------------------------ forward
    def _res(self, x, semb, N, Hh, L, W, out=None):
        _, pre, cin, cout, updown = L
        w = W[pre]
        geom = Geom.per_sample(N, Hh * Hh)
        a, b = ops.gn_stats(x, w["g1"], w["b1"], geom)
        h = ops.gn_apply(x, a, b, geom, act=True)
        if updown is not None:
            Ho = Hh // 2 if updown == "down" else Hh * 2
            mode = 0 if updown == "down" else 1
            h2 = ops.alloc(N * Ho * Ho, cin, dtype=x.dtype, device=x.device)
            x2 = ops.alloc(N * Ho * Ho, cin, dtype=x.dtype, device=x.device)
            ops.resample(h, h2, N, Hh, Hh, 2, 2, mode)
            ops.resample(x, x2, N, Hh, Hh, 2, 2, mode)
            h, x, Hh = h2, x2, Ho
            geom = Geom.per_sample(N, Hh * Hh)
        h = ops.conv_gemm(h, w["w_in"], w["b_in"], taps=ops.TAPS_SPATIAL, dims=(N, Hh, Hh))
        eo, en = W["emb_all"][2][pre]
        emb_out = semb[:, eo:eo + en]            # semb = all emb_layers outputs [N, sum J], computed once per evaluation
        if self.use_scale_shift_norm:
            a, b = ops.gn_stats(h, w["g2"], w["b2"], geom, film=emb_out)
        else:
            ops.add_rowbias(h, emb_out, Hh * Hh)
            a, b = ops.gn_stats(h, w["g2"], w["b2"], geom)
        h = ops.gn_apply(h, a, b, geom, act=True)
        skip = x if w["w_skip"] is None else ops.conv_gemm(x, w["w_skip"], w["b_skip"])
        return ops.conv_gemm(h, w["w_out"], w["b_out"], taps=ops.TAPS_SPATIAL, dims=(N, Hh, Hh), residual=skip, out=out), Hh

    def _attn(self, x, N, Hh, L, W, out=None):
        _, pre, C, heads = L
        w = W[pre]
        T = Hh * Hh
        geom = Geom.per_sample(N, T)
        a, b = ops.gn_stats(x, w["g"], w["b"], geom)
        xn = ops.gn_apply(x, a, b, geom, act=False)
        qkv = ops.conv_gemm(xn, w["w_qkv"], w["b_qkv"])
        att = ops.alloc(N * T, C, dtype=x.dtype, device=x.device)
        ops.attn(qkv, qkv, att, heads, C // heads, N, 1, T, T, T, T, 1)
        return ops.conv_gemm(att, w["w_proj"], w["b_proj"], residual=x, out=out)

    def _run(self, x6, timesteps, rows=None, shape=None):
        """x6: fp32 API-layout input [N, in_channels, H, W].  SR model: x6 is None and the input arrives as channels-last rows
        [N*H*W, Cpad] in the activation dtype (`rows`, with `shape` = (N, in_channels, H, W)) - the stem then runs as an implicit GEMM."""
        src = x6 if rows is None else rows
        if not src.is_cuda:
            raise MMDError("ImageUnet runs on the MI355X HIP path only (GPU tensors); there is no CPU/torch fallback")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and src.requires_grad:
            raise NotImplementedError("ImageUnet: the SR model is inference-only on the HIP path")
        dev = src.device
        if self._packed is None or self._packed[0] != str(dev) or self._packed[1] != self.dtype:
            self._pack(dev)
        W = self._packed[2]
        dt = self.dtype
        N, Cin, Hh, Ww = x6.shape if rows is None else shape
        assert Hh == Ww, "square images only"
        mc = self.model_channels
        te = W["time_embed"]
        # time_embed MLP (fp32), then SiLU: every emb_layers Sequential starts with SiLU (image_unet.py:176-182)
        e0 = ops.alloc(N, mc, dtype=torch.float32, device=dev)
        ops.timestep_embedding(timesteps.contiguous(), mc, e0)
        e1 = ops.alloc(N, 4 * mc, dtype=torch.float32, device=dev)
        ops.linear(e0, te[0], te[1], e1)
        ops.silu(e1, None, e1)
        emb = ops.alloc(N, 4 * mc, dtype=torch.float32, device=dev)
        ops.linear(e1, te[2], te[3], emb)
        s1 = ops.alloc(N, 4 * mc, dtype=torch.float32, device=dev)
        ops.silu(emb, None, s1)
        semb = ops.alloc(N, W["emb_all"][3], dtype=torch.float32, device=dev)
        ops.linear(s1, W["emb_all"][0], W["emb_all"][1], semb)
        plan_in, plan_mid, plan_out = self._plan
        # skip concats without copies (th.cat([h, hs.pop()], dim=1), image_unet.py:689-692): output block k reads ONE buffer
        # [decoder h | skip]; input block nin-1-k writes its output straight into the right-hand column slice, the layer in front of
        # output block k into the left-hand slice (the multimodal engine does the same)
        nin = len(plan_in)
        chans_in, Hin, ch, Hc = [], [], None, Hh
        for layers in plan_in:
            for L in layers:
                if L[0] in ("stem", "res"):
                    ch = L[3]
                if L[0] == "res" and L[4] == "down":
                    Hc //= 2
            chans_in.append(ch)
            Hin.append(Hc)
        ch_dec = [plan_out[k][0][2] - chans_in[nin - 1 - k] for k in range(nin)]
        cats = [None] * nin
        h = None
        for i, layers in enumerate(plan_in):
            k = nin - 1 - i
            cats[i] = ops.alloc(N * Hin[i] * Hin[i], ch_dec[k] + chans_in[i], dtype=dt, device=dev)
            dst = cats[i][:, ch_dec[k]:]
            for j, L in enumerate(layers):
                o = dst if j == len(layers) - 1 else None
                if L[0] == "stem" and rows is not None:
                    h = ops.conv_gemm(rows, W[L[1]][2], W[L[1]][1], taps=ops.TAPS_SPATIAL, dims=(N, Hh, Hh), out=o)
                elif L[0] == "stem":
                    h = ops.alloc(N * Hh * Hh, L[3], dtype=dt, device=dev)
                    ops.stem_conv(x6.float().contiguous().view(N, 1, Cin, Hh, Hh), W[L[1]][0], W[L[1]][1], h, N, 1, Cin, Hh, Hh, ops.TAPS_SPATIAL)
                    if o is not None:
                        h = ops.copy2d(h, o)
                elif L[0] == "res":
                    h, Hh = self._res(h, semb, N, Hh, L, W, out=o)
                else:
                    h = self._attn(h, N, Hh, L, W, out=o)
        for j, L in enumerate(plan_mid):
            o = cats[nin - 1][:, :ch_dec[0]] if j == len(plan_mid) - 1 else None
            h = self._res(h, semb, N, Hh, L, W, out=o)[0] if L[0] == "res" else self._attn(h, N, Hh, L, W, out=o)
        for k, layers in enumerate(plan_out):
            h = cats[nin - 1 - k]
            for j, L in enumerate(layers):
                o = cats[nin - 2 - k][:, :ch_dec[k + 1]] if (j == len(layers) - 1 and k + 1 < nin) else None
                if L[0] == "res":
                    h, Hh = self._res(h, semb, N, Hh, L, W, out=o)
                else:
                    h = self._attn(h, N, Hh, L, W, out=o)
        g, b, w_out, b_out = W["out"]
        geom = Geom.per_sample(N, Hh * Hh)
        a, bb = ops.gn_stats(h, g, b, geom)
        h = ops.gn_apply(h, a, bb, geom, act=True)
        out = ops.alloc(N, 1, self.out_channels, Hh, Hh, dtype=torch.float32, device=dev)
        ops.head_conv(h, w_out, b_out, out, N, 1, Hh, Hh, ops.TAPS_SPATIAL)
        return out.view(N, self.out_channels, Hh, Hh)

    def forward(self, x, timesteps, y=None):
        assert (y is not None) == (self.num_classes is not None), "must specify y if and only if the model is class-conditional"
        return self._replay((x,), timesteps, lambda xs, t: self._run(xs[0], t))

    # ------------------------------------------------------------------ graph replay of the no-grad forward
    def _replay(self, inputs, timesteps, body):
        """Run body(static inputs, static timesteps) -> fp32 output.  Under no_grad the launch sequence is recorded ONCE per input
        geometry into a plan (ops.recording: per-shape tile autotune, every buffer parked in a keep-list) and captured into a hipGraph;
        every later evaluation copies the inputs into the static buffers and replays the graph: ~700 ctypes launches and as many
        torch.empty calls per evaluation become one launch (the SR stage runs 25-1000 evaluations per clip, sample_sr.py:186-253).
        OPT-IN (MMD_SR_GRAPH=1): measured on MI355X at the shipped size (16 frames of 256 x 256, 192 channels) the evaluation is
        GPU-bound - 46.6 ms replayed vs 46.9 ms eager - while the plan's keep-list (no liveness reuse) holds 27 GB against 5 GB of
        eager peak; replay pays only for small frame batches, where the host launch time shows."""
        H.require_cuda(*inputs)
        if torch.is_grad_enabled() or os.environ.get("MMD_SR_GRAPH", "0") != "1":
            return body(tuple(t.float().contiguous() for t in inputs), timesteps.contiguous())
        dev = inputs[0].device
        if self._packed is None or self._packed[0] != str(dev) or self._packed[1] != self.dtype:
            self._pack(dev)
            self._drop_graphs()
        key = (tuple(tuple(t.shape) for t in inputs), timesteps.dtype, self.dtype, str(dev))
        g = self._graphs.get(key)
        if g is None:
            H.reap()
            g = dict(ins=[torch.zeros(t.shape, dtype=torch.float32, device=dev) for t in inputs],
                     t=torch.zeros(timesteps.shape, dtype=timesteps.dtype, device=dev), plan=[], keep=[], stream=H.Stream(dev))
            with H.preserve_rng(dev), ops.recording(g["plan"], keep=g["keep"]):
                g["out"] = body(tuple(g["ins"]), g["t"])
            side = g["stream"].torch
            side.wait_stream(torch.cuda.current_stream(dev))
            ops.run_plan(g["plan"], side.cuda_stream)                   # warm-up: one-time function attributes, lazy module load
            torch.cuda.synchronize(dev)
            with H.capture(side.cuda_stream) as cap:
                ops.run_plan(g["plan"], side.cuda_stream)
            g["exec"] = cap.exec
            self._graphs[key] = g
        for dst, src in zip(g["ins"], inputs):
            dst.copy_(src)
        g["t"].copy_(timesteps)
        H.call("mmd_graph_launch", g["exec"], H.stream_handle())
        return g["out"].clone()

    def _drop_graphs(self):
        for g in getattr(self, "_graphs", {}).values():
            if g.get("exec") is not None:
                H.retire("graph", g["exec"])
            g["stream"].close()
        self._graphs = {}

    def __del__(self):
        try:
            self._drop_graphs()
        except Exception:
            pass


class ImageSuperResModel(ImageUnet):
    """U-Net conditioned on a low-resolution image: bilinear upsample + channel concat (image_unet.py:700-715)."""

    def __init__(self, image_size, in_channels, *args, **kwargs):
        super().__init__(image_size, in_channels * 2, *args, **kwargs)

    def forward(self, x, timesteps, low_res=None, **kwargs):
        if low_res is None:
            raise MMDError("ImageSuperResModel.forward needs low_res")
        assert (kwargs.get("y") is not None) == (self.num_classes is not None), "must specify y if and only if the model is class-conditional"

        def body(xs, t):
            xx, low = xs
            N, C, Hh, Ww = xx.shape
            cpad = (2 * C + 7) // 8 * 8
            rows = ops.alloc(N * Hh * Ww, cpad, dtype=self.dtype, device=xx.device)
            ops.bilinear_concat_rows(xx, low, rows)           # [x | bilinear(low) | 0] as channels-last rows: the stem is a K = 9 * cpad GEMM
            return self._run(None, t, rows=rows, shape=(N, 2 * C, Hh, Ww))
        return self._replay((x, low_res), timesteps, body)
