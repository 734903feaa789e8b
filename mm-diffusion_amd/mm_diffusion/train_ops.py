"""Differentiable wrappers (torch.autograd.Function) of the libmmd ops for the training step.

torch's autograd engine only ORDERS the backward calls and accumulates gradients at fan-out points; every forward
and backward computation is a libmmd kernel (mmd_conv_gemm / mmd_conv_wgrad / mmd_gn_bwd / mmd_attn_bwd / ...).
Activations are channels-last rows [rows, C] as in the inference engine; parameters stay in the reference layouts so
their .grad lands on the nn.Parameters the optimizer / DDP see.
"""
import os

import torch
from torch.autograd import Function

from . import _hip as H
from . import ops
from .ops import Geom


def _neg(taps):
    return [(-a, -b, -c) for a, b, c in taps]


def _grad_slot(*params):
    """The parameters' .grad buffers when the backward kernels may accumulate straight into them (fp32, contiguous, already
    allocated - optim.FlatAdamW re-homes every .grad into one flat buffer), else None.  Skips the per-parameter zeros /
    permute / AccumulateGrad kernels: ~2000 launches per training step of the base model."""
    for p in params:
        if not p.is_leaf:
            return None
        g = p.grad
        if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape or not p.requires_grad:
            return None
    # bucketed gradient reduction (optim.FlatAdamW.arm_overlap): first launch what EARLIER kernels completed, then mark this kernel's
    # parameters - all of them before anything is launched again (one kernel writes the gradients of the whole group)
    opts = [getattr(p, "_mmd_opt", None) for p in params]
    for o in {id(o[0]): o[0] for o in opts if o is not None}.values():
        o._flush_ready()
    for o in opts:
        if o is not None:
            o[0]._param_done(o[1])
    return [p.grad for p in params]


class ConvFn(Function):
    """Y = conv(X) (+ R): implicit-GEMM forward, dgrad = same kernel on the transposed weight with mirrored taps,
    wgrad/bias grad = mmd_conv_wgrad.  weight in torch conv layout [Cout, Cin, *k]."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, taps, dims):
        x = x.contiguous() if x.stride(1) != 1 else x
        pk = getattr(weight, "_mmd_packed", None)           # (fwd, bwd) kept current by optim.FlatAdamW's batched re-pack
        wp = pk[0] if pk is not None and pk[0].dtype == x.dtype else ops.pack_conv_weight(weight.detach().float(), x.dtype)
        y = ops.conv_gemm(x, wp, bias.detach().float().contiguous(), taps=taps, dims=dims,
                          residual=None if residual is None else residual)
        ctx.save_for_backward(x, weight, bias)
        ctx.taps, ctx.dims, ctx.has_res = taps, dims, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        dy = dy.contiguous()
        taps, dims = ctx.taps, ctx.dims
        Cout, Cin = weight.shape[0], weight.shape[1]
        nt = len(taps)
        dx = None
        if ctx.needs_input_grad[0]:
            pk = getattr(weight, "_mmd_packed", None)
            wt = pk[1] if pk is not None and pk[1].dtype == x.dtype else \
                weight.detach().float().reshape(Cout, Cin, nt).permute(1, 2, 0).reshape(Cin, nt * Cout).to(x.dtype).contiguous()
            dx = ops.conv_gemm(dy, wt, None, taps=_neg(taps), dims=dims)
        if not (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):      # frozen weights (gradient-guided sampling)
            return dx, None, None, (dy if ctx.has_res else None), None, None
        slot = _grad_slot(weight, bias)
        if slot is not None:
            acc = getattr(weight, "_mmd_wgrad", None)         # packed fp32 accumulator (coalesced atomics), folded into .grad once per step
            if acc is not None:
                ops.conv_wgrad(dy, x, acc, slot[1], taps, dims)
            else:                                             # accumulate in the parameter's own layout, straight into .grad
                ops.conv_wgrad(dy, x, slot[0], slot[1], taps, dims, torch_layout=True)
            return dx, None, None, (dy if ctx.has_res else None), None, None
        dW32 = torch.zeros(Cout, nt * Cin, dtype=torch.float32, device=x.device)
        db32 = torch.zeros(Cout, dtype=torch.float32, device=x.device)
        ops.conv_wgrad(dy, x, dW32, db32, taps, dims)
        dW = dW32.reshape(Cout, nt, Cin).permute(0, 2, 1).reshape(weight.shape).to(weight.dtype)
        return dx, dW, db32.to(weight.dtype), (dy if ctx.has_res else None), None, None


def conv(x, weight, bias, taps=ops.TAPS_1, dims=(1, 1, 1), residual=None):
    return ConvFn.apply(x, weight, bias, residual, taps, dims)


class GroupNormFn(Function):
    """y = act(GroupNorm32(x) [* (1 + scale) + shift]) on slices `geom`; film = [S, 2C] (scale | shift) or None."""

    @staticmethod
    def forward(ctx, x, gamma, beta, film, geom, act):
        x = x.contiguous() if x.stride(1) != 1 else x
        C = x.shape[1]
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        f32 = None if film is None else film.detach().float()
        mr = torch.empty(geom.S, 32, 2, dtype=torch.float32, device=x.device)
        a, b = ops.gn_stats(x, g32, b32, geom, film=f32, mr=mr)
        y = ops.gn_apply(x, a, b, geom, act=act)
        ctx.save_for_backward(x, a, b, mr, g32, b32, f32 if f32 is not None else torch.empty(0, device=x.device))
        ctx.params = (gamma, beta)
        ctx.geom, ctx.act, ctx.has_film = geom, act, film is not None
        ctx.film_shape = None if film is None else tuple(film.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, a, b, mr, g32, b32, f32 = ctx.saved_tensors
        geom = ctx.geom
        dy = dy.contiguous()
        C = x.shape[1]
        dx = torch.empty_like(x)
        dfilm = torch.empty(geom.S, 2 * C, dtype=torch.float32, device=x.device) if ctx.has_film else None
        slot = _grad_slot(*ctx.params)
        if slot is not None:
            ops.gn_bwd(x, dy, dx, geom, a, b, mr, g32, b32, f32 if ctx.has_film else None, ctx.act, slot[0], slot[1], dfilm)
            return dx, None, None, dfilm, None, None
        dgamma = torch.zeros(C, dtype=torch.float32, device=x.device)
        dbeta = torch.zeros(C, dtype=torch.float32, device=x.device)
        ops.gn_bwd(x, dy, dx, geom, a, b, mr, g32, b32, f32 if ctx.has_film else None, ctx.act, dgamma, dbeta, dfilm)
        return dx, dgamma, dbeta, dfilm, None, None


def group_norm(x, gamma, beta, geom, act, film=None):
    return GroupNormFn.apply(x, gamma, beta, film, geom, act)


class SelfAttnFn(Function):
    """softmax(q k^T / sqrt(ch)) v on qkv rows [rows, 3C]; kind: 'spatial' (units = (n, f), T = HW), 'temporal'
    (units = (n, pixel), rows strided by HW, T = F) or 'audio' (units = n)."""

    @staticmethod
    def forward(ctx, qkv, heads, kind, N, F, HW):
        qkv = qkv.contiguous()
        rows, C3 = qkv.shape
        C = C3 // 3
        out = torch.empty(rows, C, dtype=qkv.dtype, device=qkv.device)
        if kind == "temporal":
            ops.attn_small(qkv, out, C, heads, Geom.temporal(N, F, HW))
            desc = dict(nb=N * HW, inner=HW, outer=F * HW, istride=1, tstride=HW, T=F, geom=Geom.temporal(N, F, HW))
        else:
            T = HW if kind == "spatial" else rows // N
            nb = N * F if kind == "spatial" else N
            desc = dict(nb=nb, inner=1, outer=T, istride=1, tstride=1, T=T)
            if ops.attn_mfma_ok(qkv, C // heads):      # bf16: MFMA forward (+ log-sum-exp) and MFMA backward
                lse = torch.empty(rows * heads, dtype=torch.float32, device=qkv.device)
                ops.attn_lse(qkv, qkv, out, lse, heads, C // heads, nb, 1, T, T, T, T, 1)
                ctx.save_for_backward(qkv, out, lse)
                ctx.heads, ctx.desc, ctx.mfma = heads, desc, True
                return out
            ops.attn(qkv, qkv, out, heads, C // heads, nb, 1, T, T, T, T, 1)
        ctx.save_for_backward(qkv, out)
        ctx.heads, ctx.desc, ctx.mfma = heads, desc, False
        return out

    @staticmethod
    def backward(ctx, dout):
        d, heads = ctx.desc, ctx.heads
        if ctx.mfma:
            qkv, out, lse = ctx.saved_tensors
            C = out.shape[1]
            dqkv = torch.empty_like(qkv)
            ops.attn_bwd_mfma(qkv, qkv, out, dout.contiguous(), dqkv, 0, dqkv, C, 2 * C, lse, heads, C // heads, d["nb"], 1,
                              d["T"], d["T"], d["T"], d["T"], 1)
            return dqkv, None, None, None, None, None
        qkv, out = ctx.saved_tensors
        C = out.shape[1]
        dqkv = torch.empty_like(qkv)
        if "geom" in d and C // heads in ops.MFMA_HEADS:       # temporal: one wave per (pixel, head)
            ops.attn_small_bwd(qkv, dout.contiguous(), dqkv, C, heads, d["geom"])
            return dqkv, None, None, None, None, None
        ops.attn_bwd(qkv, 0, qkv, C, 2 * C, out, dout.contiguous(), dqkv, 0, dqkv, C, 2 * C, heads, C // heads, d["nb"], 1,
                     (d["inner"], d["outer"], d["istride"], d["tstride"]), d["T"], d["T"],
                     (d["inner"], d["outer"], d["istride"], d["tstride"]), d["T"], d["T"], 1, None)
        return dqkv, None, None, None, None, None


class CrossAttnFn(Function):
    """Random-shift windowed cross-modal attention, both directions (reference unit QKVAttention.forward):
    video queries attend the audio window, audio queries the video window.  shift: python int (this call's draw) or a
    1-element int32 device tensor (graph-captured training: the value is refreshed in place before every replay)."""

    @staticmethod
    def forward(ctx, vqkv, aqkv, heads, N, F, HW, L, win, shift):
        vqkv, aqkv = vqkv.contiguous(), aqkv.contiguous()
        C = vqkv.shape[1] // 3
        apf = int(L / F)
        sh = shift if torch.is_tensor(shift) else torch.full((1,), int(shift), dtype=torch.int32, device=vqkv.device)   # fill kernel: capturable
        vatt = torch.empty(N * F * HW, C, dtype=vqkv.dtype, device=vqkv.device)
        aatt = torch.empty(N * L, C, dtype=vqkv.dtype, device=vqkv.device)
        ctx.cfg = (heads, N, F, HW, L, apf, win)
        ctx.mfma = ops.attn_mfma_ok(vqkv, C // heads) and ops.attn_mfma_ok(aqkv, C // heads)
        if ctx.mfma:
            vlse = torch.empty(N * F * HW * heads, dtype=torch.float32, device=vqkv.device)
            alse = torch.empty(N * L * heads, dtype=torch.float32, device=vqkv.device)
            ops.attn_lse(vqkv, aqkv, vatt, vlse, heads, C // heads, N, F, F * HW, HW, L, apf, win, shift_dev=sh)
            ops.attn_lse(aqkv, vqkv, aatt, alse, heads, C // heads, N, F, L, apf, F * HW, HW, win, shift_dev=sh)
            ctx.save_for_backward(vqkv, aqkv, vatt, aatt, sh, vlse, alse)
            return vatt, aatt
        ops.attn(vqkv, aqkv, vatt, heads, C // heads, N, F, F * HW, HW, L, apf, win, shift_dev=sh)
        ops.attn(aqkv, vqkv, aatt, heads, C // heads, N, F, L, apf, F * HW, HW, win, shift_dev=sh)
        ctx.save_for_backward(vqkv, aqkv, vatt, aatt, sh)
        return vatt, aatt

    @staticmethod
    def backward(ctx, dvatt, daatt):
        heads, N, F, HW, L, apf, win = ctx.cfg
        if ctx.mfma:
            vqkv, aqkv, vatt, aatt, sh, vlse, alse = ctx.saved_tensors
            C = vatt.shape[1]
            dv, da = torch.empty_like(vqkv), torch.empty_like(aqkv)
            ops.attn_bwd_mfma(vqkv, aqkv, vatt, dvatt.contiguous(), dv, 0, da, C, 2 * C, vlse, heads, C // heads, N, F, F * HW, HW,
                              L, apf, win, shift_dev=sh)
            ops.attn_bwd_mfma(aqkv, vqkv, aatt, daatt.contiguous(), da, 0, dv, C, 2 * C, alse, heads, C // heads, N, F, L, apf,
                              F * HW, HW, win, shift_dev=sh)
            return dv, da, None, None, None, None, None, None, None
        vqkv, aqkv, vatt, aatt, sh = ctx.saved_tensors
        C = vatt.shape[1]
        dv, da = torch.empty_like(vqkv), torch.empty_like(aqkv)
        vgeo, ageo = (1, F * HW, 1, 1), (1, L, 1, 1)
        # video queries <- audio keys: dQ -> dv[:, :C], dK/dV -> da[:, C:]
        ops.attn_bwd(vqkv, 0, aqkv, C, 2 * C, vatt, dvatt.contiguous(), dv, 0, da, C, 2 * C, heads, C // heads, N, F,
                     vgeo, F * HW, HW, ageo, L, apf, win, sh)
        # audio queries <- video keys: dQ -> da[:, :C], dK/dV -> dv[:, C:]
        ops.attn_bwd(aqkv, 0, vqkv, C, 2 * C, aatt, daatt.contiguous(), da, 0, dv, C, 2 * C, heads, C // heads, N, F,
                     ageo, L, apf, vgeo, F * HW, HW, win, sh)
        return dv, da, None, None, None, None, None, None, None


class RowBiasFn(Function):
    """h + emb_out broadcast over the rows of each sample: the non-FiLM ResBlock (use_scale_shift_norm=False, unet:473-477).
    Forward mmd_copy2d + mmd_add_rowbias; backward dh = dy, d emb_out = per-sample column sums of dy (mmd_colsum_slices)."""

    @staticmethod
    def forward(ctx, h, e, rows_per_sample):
        out = torch.empty(h.shape, dtype=h.dtype, device=h.device)
        ops.copy2d(h, out)
        ops.add_rowbias(out, e.detach().float().contiguous(), rows_per_sample)
        ctx.rps, ctx.eshape = rows_per_sample, e.shape
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = dy if dy.stride(1) == 1 else dy.contiguous()
        de = None
        if ctx.needs_input_grad[1]:
            de = torch.zeros(ctx.eshape, dtype=torch.float32, device=dy.device)
            ops.colsum_slices(dy, de, ctx.rps)
        return (dy if ctx.needs_input_grad[0] else None), de, None


class CatFn(Function):
    """[h | skip] along the channel axis of channels-last rows (the U-Net skip concat, unet:1093-1094) with libmmd copies: forward two
    strided row copies into one buffer, backward two strided copies out of the gradient (the inference engine has no copy at all -
    producers write column slices of the consumer's buffer; under autograd the concat is a node of the graph)."""

    @staticmethod
    def forward(ctx, a, b):
        out = torch.empty(a.shape[0], a.shape[1] + b.shape[1], dtype=a.dtype, device=a.device)
        ops.copy2d(a, out[:, :a.shape[1]])
        ops.copy2d(b, out[:, a.shape[1]:])
        ctx.ca = a.shape[1]
        return out

    @staticmethod
    def backward(ctx, g):
        ca = ctx.ca
        g = g if g.stride(1) == 1 else g.contiguous()
        ga = torch.empty(g.shape[0], ca, dtype=g.dtype, device=g.device) if ctx.needs_input_grad[0] else None
        gb = torch.empty(g.shape[0], g.shape[1] - ca, dtype=g.dtype, device=g.device) if ctx.needs_input_grad[1] else None
        if ga is not None:
            ops.copy2d(g[:, :ca], ga)
        if gb is not None:
            ops.copy2d(g[:, ca:], gb)
        return ga, gb


class ResampleFn(Function):
    """avg-pool (mode 0) / nearest upsample (mode 1) by (1, fh, fw); the backward of one is the other, rescaled."""

    @staticmethod
    def forward(ctx, x, NF, Hh, Ww, fh, fw, mode):
        x = x.contiguous() if x.stride(1) != 1 else x
        rows = NF * (Hh // fh) * (Ww // fw) if mode == 0 else NF * Hh * fh * Ww * fw
        y = torch.empty(rows, x.shape[1], dtype=x.dtype, device=x.device)
        ops.resample(x, y, NF, Hh, Ww, fh, fw, mode)
        ctx.cfg = (NF, Hh, Ww, fh, fw, mode, x.shape[0])
        return y

    @staticmethod
    def backward(ctx, dy):
        NF, Hh, Ww, fh, fw, mode, rows_in = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty(rows_in, dy.shape[1], dtype=dy.dtype, device=dy.device)
        if mode == 0:      # d avg-pool = nearest upsample / (fh*fw)
            ops.resample(dy, dx, NF, Hh // fh, Ww // fw, fh, fw, 1, scale=1.0 / (fh * fw))
        else:              # d nearest-upsample = sum over the replicated cells = avg-pool * (fh*fw)
            ops.resample(dy, dx, NF, Hh * fh, Ww * fw, fh, fw, 0, scale=float(fh * fw))
        return dx, None, None, None, None, None, None


class LinearFn(Function):
    """fp32 y = x W^T + b (emb layers / time_embed); backward through the same small kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.float().contiguous()
        y = torch.empty(x.shape[0], weight.shape[0], dtype=torch.float32, device=x.device)
        ops.linear(x, weight.detach().float().contiguous(), bias.detach().float().contiguous(), y)
        ctx.save_for_backward(x, weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        dy = dy.float().contiguous()
        J, K = weight.shape
        dx = torch.empty_like(x)
        ops.linear(dy, weight.detach().float().t().contiguous(), None, dx)
        slot = _grad_slot(weight, bias)
        if slot is not None:
            for j0 in range(0, J, 1024):
                j1 = min(j0 + 1024, J)
                ops.conv_wgrad(dy[:, j0:j1].contiguous(), x, slot[0][j0:j1], slot[1][j0:j1], ops.TAPS_1, (1, 1, 1))
            return dx, None, None
        dW = torch.zeros(J, K, dtype=torch.float32, device=x.device)
        db = torch.zeros(J, dtype=torch.float32, device=x.device)
        Jp = (J + 3) // 4 * 4
        for j0 in range(0, J, 1024):        # column sums handle <= 1024 channels per call
            j1 = min(j0 + 1024, J)
            ops.conv_wgrad(dy[:, j0:j1].contiguous(), x, dW[j0:j1], db[j0:j1], ops.TAPS_1, (1, 1, 1))
        return dx, dW.to(weight.dtype), db.to(weight.dtype)


class SiluFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        ops.silu(x, None, y)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        ops.silu(x, dy.contiguous(), dx)
        return dx


class MseLossFn(Function):
    """per-sample mean((target - out)^2) on API-layout fp32 tensors; grad = 2 (out - target) / per * dloss[n]."""

    @staticmethod
    def forward(ctx, out, target):
        out, target = out.float().contiguous(), target.float().contiguous()
        N = out.shape[0]
        per = out[0].numel()
        dummy = torch.zeros(7, 1, dtype=torch.float32, device=out.device)
        mse, _ = ops.loss_terms(out.reshape(N, 1, 1, per), target.reshape(N, 1, 1, per), dummy,
                                torch.zeros(N, dtype=torch.int64, device=out.device), 1, 1, per, 0)
        ctx.save_for_backward(out, target)
        return mse

    @staticmethod
    def backward(ctx, dloss):
        out, target = ctx.saved_tensors
        g = torch.empty_like(out)
        ops.mse_grad(out, target, dloss.float().contiguous(), g)
        return g, None


class LossTermsFn(Function):
    """(mse [N], vb [N]) of one stream with the learned-range variance (gd:1143-1203): model_out [N,F,2C,HW] API layout fp32.
    The vb term sees the mean prediction detached (gd:1147-1151), so its gradient reaches the variance channels only."""

    @staticmethod
    def forward(ctx, model_out, target, x0, xt, tables, t, geom, flags, vb_scale):
        mo = model_out.float().contiguous()
        F, C, HW = geom
        mse, vb = ops.loss_terms(mo, target, tables, t, F, C, HW, flags, x0=x0, xt=xt, vb_scale=vb_scale)
        ctx.save_for_backward(mo, target, x0, xt, tables, t)
        ctx.cfg = (geom, flags, vb_scale)
        return mse, vb

    @staticmethod
    def backward(ctx, dmse, dvb):
        mo, target, x0, xt, tables, t = ctx.saved_tensors
        (F, C, HW), flags, vb_scale = ctx.cfg
        g = torch.empty_like(mo)
        ops.loss_terms_bwd(mo, target, tables, t, F, C, HW, flags, dmse.float().contiguous(), dvb.float().contiguous(), g, x0=x0, xt=xt,
                           vb_scale=vb_scale)
        return g, None, None, None, None, None, None, None, None


class DdpmUpdateFn(Function):
    """Differentiable p_sample update of one stream (gd:231-343,415-474; fixed variance): returns (sample, pred_xstart);
    gradients flow through the posterior mean to x_t and to the model output (mmd_ddpm_update_bwd)."""

    @staticmethod
    def forward(ctx, x, mo, noise, tab, t, flags, geom):
        if flags & 4:
            raise NotImplementedError("gradient-guided sampling with learned variance is not built")
        x, mo = x.contiguous(), mo.contiguous()
        F, C, HW = geom
        sample, x0 = torch.empty_like(x), torch.empty_like(x)
        ops.ddpm_update(x, mo, noise, sample, tab, t, F, C, HW, flags, x0_out=x0)
        ctx.save_for_backward(x, mo, tab, t)
        ctx.flags = flags
        ctx.mark_non_differentiable(x0)
        return sample, x0

    @staticmethod
    def backward(ctx, dsample, _dx0):
        x, mo, tab, t = ctx.saved_tensors
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dmo = torch.empty_like(mo) if ctx.needs_input_grad[1] else None
        ops.ddpm_update_bwd(x, mo, dsample.float().contiguous(), dx, dmo, tab, t, ctx.flags)
        return dx, dmo, None, None, None, None, None


class DropoutFn(Function):
    """nn.Dropout(p) (reference unet:376,384): the Bernoulli mask is drawn by torch's generator (noise input), the scaling
    runs in libmmd; backward re-applies the same mask."""

    @staticmethod
    def forward(ctx, x, p):
        x = x.contiguous()
        # keep with probability 1 - p: ONE Philox kernel writing the byte mask (graph-safe like torch.rand).  `(torch.rand(...) >= p).to(uint8)`
        # was three launches and 13 bytes of traffic per element - 6 ms of a 120 ms step (rand 24.6 us + compare 19.6 us + copy 38 us, x 75)
        # same-call A/B (profiles/r06_train_small_kernels.txt): training step 121.4 / 121.7 -> 118.5 / 118.5 ms; keep rate 0.90001 at p = 0.1
        mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device).bernoulli_(1.0 - p)
        y = torch.empty_like(x)
        ops.dropout(x, mask, 1.0 / (1.0 - p), y)
        ctx.save_for_backward(mask)
        ctx.p = p
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dx = torch.empty_like(dy)
        ops.dropout(dy.contiguous(), mask, 1.0 / (1.0 - ctx.p), dx)
        return dx, None


class RecomputeFn(Function):
    """Activation recompute with the reference's semantics (nn.py:233-279): forward under no_grad, backward re-runs
    `run` (which re-draws the cross-attention window shift, quirk Q3) and differentiates the recomputed graph."""

    @staticmethod
    def forward(ctx, run, n_in, *args):
        ctx.run = run
        ctx.inputs = list(args[:n_in])
        ctx.params = list(args[n_in:])
        with torch.no_grad():
            out = run(*ctx.inputs)
        return out

    @staticmethod
    def backward(ctx, *gouts):
        need = ctx.needs_input_grad[2:]
        n_in = len(ctx.inputs)
        ins = [x.detach().requires_grad_(bool(need[i])) for i, x in enumerate(ctx.inputs)]
        with torch.enable_grad():
            outs = ctx.run(*ins)
        cand = ins + ctx.params
        wrt = [t for i, t in enumerate(cand) if need[i]]          # frozen parameters / constant inputs are skipped
        pairs = [(o, g) for o, g in zip(outs, gouts) if o.requires_grad and g is not None]
        got = iter(torch.autograd.grad([o for o, _ in pairs], wrt, [g for _, g in pairs], allow_unused=True))
        grads = tuple(next(got) if need[i] else None for i in range(len(cand)))
        assert len(grads) == n_in + len(ctx.params)
        return (None, None) + grads
