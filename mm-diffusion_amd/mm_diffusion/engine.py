"""Launch-plan engine: turns a MultimodalUNet parameter tree into a flat list of libmmd kernel launches.

Built once per (batch, dtype, device):
  * weights are packed for the kernels (GEMM operands [Cout][tap*Cin] in the activation dtype, biases /
    GroupNorm affine / emb Linear in fp32, all ResBlock emb_layers concatenated into ONE Linear)
  * every activation buffer is pre-allocated from a plan-time pool (liveness-based reuse, so the working set
    stays small and L2/MALL resident); skip-connection concats are free: each input block writes its output
    straight into the right-hand column slice of the buffer its output block will read, and the previous
    output block writes the left-hand slice
  * the window shifts / timesteps live in device buffers, so the identical plan can be captured into a
    hipGraph (mmd_graph_*) and replayed with new shifts and timesteps every denoising step
Layout: video rows (n, f, h, w) x C, audio rows (n, l) x C; API-layout conversion happens only inside the
stem (InitialBlock) and head kernels.
"""
import os

import torch

from . import _hip as H
from . import ops
from .ops import Geom


_EMB_ON_AUDIO_STREAM = os.environ.get("MMD_EMB_AUX", "1") != "0"
_POOL_NOREUSE = os.environ.get("MMD_POOL_NOREUSE") == "1"      # diagnostics (tools/determinism_graph.py): every tensor keeps its own buffer
# round 5: the out layers of the up ResBlocks at the input resolution (one upsample of the block's result instead of two of its operands);
# MMD_UP_LOWRES=0: the reference's order of operations (A/B)
_UP_LOWRES = os.environ.get("MMD_UP_LOWRES", "1") != "0"
# round 5: a cross-attention block's audio-side attention runs behind its video-side attention and the video stream does not wait for it;
# MMD_CROSS_SERIAL=0: both start together and each stream waits for the other's (A/B)
_CROSS_SERIAL = os.environ.get("MMD_CROSS_SERIAL", "1") != "0"
# round 5: GroupNorm statistics of resampled tensors from the resample launch's epilogue (mmd_resample_stats); =0: statistics passes (A/B)
_RESAMPLE_STATS = os.environ.get("MMD_RESAMPLE_STATS", "1") != "0"


class _Pool:
    """Plan-time buffer pool with reuse by liveness (run-time order == plan order on one stream)."""

    def __init__(self, device):
        self.device = device
        self.free = []
        self.all = []

    def get(self, nbytes):
        nbytes = (int(nbytes) + 255) // 256 * 256
        best = None
        for i, raw in enumerate(self.free):
            sz = raw.numel()
            if sz >= nbytes and sz <= 2 * nbytes + 65536 and (best is None or sz < self.free[best].numel()):
                best = i
        if best is not None:
            return self.free.pop(best)
        raw = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self.all.append(raw)
        return raw

    def put(self, raw):
        if not _POOL_NOREUSE:
            self.free.append(raw)

    def total_bytes(self):
        return sum(r.numel() for r in self.all)


class UNetEngine:
    def __init__(self, model, N, dtype, device):
        if not torch.cuda.is_available():
            raise H.MMDError("no HIP device visible: the MI355X path cannot run (no CPU fallback)")
        H.lib()
        self.model, self.N, self.dtype, self.device = model, N, dtype, torch.device(device)
        self.F, self.Cv_in, self.H0, self.W0 = model.video_size
        self.Ca_in, self.L0 = model.audio_size
        self.mc = model.model_channels
        self.params = {k: v for k, v in model.named_parameters()}
        for k, v in self.params.items():
            if v.device.type != "cuda":
                raise H.MMDError(f"parameter {k} is on {v.device}: move the model to the GPU first (model.to('cuda'))")
        self._sig = self._signature()
        self.pools = [_Pool(self.device), _Pool(self.device)]   # one per launch stream (video / audio run concurrently)
        # GroupNorm statistics from the producer GEMM's epilogue: bf16 mode only (the records carry plain sums, no pivot - the fp32
        # mode keeps the pivoted statistics pass and its 2e-6 parity).  MMD_GN_EPILOGUE=0 restores the statistics pass everywhere
        # (A/B runs), =2 also uses the epilogue statistics in fp32 mode (tests: isolates the mechanism from bf16 rounding noise).
        mode = os.environ.get("MMD_GN_EPILOGUE", "1")
        self.rec_enabled = mode == "2" or (mode != "0" and dtype == torch.bfloat16)
        self._recs = {}
        # (Round 3's in-launch statistics + affine - "tails": integer accumulators in the producers, the last block of the last producer
        # finalises - measured slower in rounds 3 and 5 and were removed from the library in round 6; DESIGN.md section 5 has the numbers.)
        self._gn_small = os.environ.get("MMD_GN_SMALL", "1") != "0"
        self._vconv_fused = self._tattn_fused = self._tconv = self._aconv = dtype == torch.bfloat16
        self._deferred = []           # video-stream buffers a launch of the AUDIO stream still reads (see _cross): released at the next sync
        H.reap()
        self._aux, self._side = H.Stream(self.device), H.Stream(self.device)
        self.aux = self._aux.torch          # audio-chain launches
        self.side = self._side.torch        # capture stream (graphs of this engine and of the samplers driving it)
        self.keep = []          # packed weights etc. (kept alive)
        self.plan = []
        self.graph = None
        self._build()

    # ------------------------------------------------------------------ staleness
    def _signature(self):
        return tuple((p.data_ptr(), p._version) for p in self.params.values())

    def stale(self):
        return self._sig != self._signature()

    # ------------------------------------------------------------------ helpers
    def _alloc(self, rows, C, dtype=None, stats=False, unit=64):
        """stats: the tensor is written by GEMMs only and normalised by a GroupNorm afterwards -> give it a record buffer for the
        producers' epilogue statistics (mmd_conv_gemm_stats); unit = rows of the smallest slice a consumer norm will use."""
        dtype = dtype or self.dtype
        es = torch.empty(0, dtype=dtype).element_size()
        pool = self.pools[ops.cur_sid]
        raw = pool.get(rows * C * es)
        t = raw[: rows * C * es].view(dtype).view(rows, C)
        t._raw, t._pool = raw, pool
        if stats and self.rec_enabled and rows % 64 == 0 and unit % 64 == 0 and C % 4 == 0:
            rec = self._alloc(rows // 64, C // 2, torch.float32)            # one (sum, sum of squares) record per 64 rows and channel QUAD
            self._recs[raw.untyped_storage().data_ptr()] = dict(rec=rec, view=rec.view(rows // 64, C // 4, 2), ptr=t.data_ptr(), es=es,
                                                                rows=rows, C=C, cover=[])
        return t

    def _release(self, *ts):
        for t in ts:
            if t is not None and hasattr(t, "_raw"):
                ent = self._recs.pop(t._raw.untyped_storage().data_ptr(), None)
                if ent is not None:
                    self._release(ent["rec"])
                t._pool.put(t._raw)
                del t._raw

    # ------------------------------------------------------------------ producer-side GroupNorm statistics
    def _rec_slice(self, t):
        """(registry entry, first column) of a tensor that lives in a buffer with a record buffer, else (None, 0)."""
        ent = self._recs.get(t.untyped_storage().data_ptr())
        if ent is None or t.stride(0) != ent["C"] or t.shape[0] != ent["rows"]:
            return None, 0
        c0 = (t.data_ptr() - ent["ptr"]) // ent["es"]
        return (ent, c0) if 0 <= c0 and c0 + t.shape[1] <= ent["C"] else (None, 0)

    def _stats_for(self, out, perm_unit=0):
        """The record view a GEMM writing `out` should fill (None: the buffer has no record buffer).  perm_unit: the producer's
        records are 64-row groups in ITS OWN order inside every perm_unit rows (the fused VideoConv: patches x frames), so only norms
        whose slices are whole multiples of perm_unit may finalize from them."""
        ent, c0 = self._rec_slice(out)
        if ent is None or c0 % 4 or out.shape[1] % 4:
            return None
        ent["perm_unit"] = max(ent.get("perm_unit", 0), perm_unit)
        ent["cover"].append((c0, c0 + out.shape[1]))
        return ent["view"][:, c0 // 4:(c0 + out.shape[1]) // 4, :]

    def _has_stats(self, out):
        return self._rec_slice(out)[0] is not None

    def _resample_stats(self, out):
        """The record view a resample writing `out` fills (None: no record buffer, or a column slice its 16-byte record stores cannot
        address; MMD_RESAMPLE_STATS=0: the statistics pass - A/B)."""
        if not _RESAMPLE_STATS or self.dtype != torch.bfloat16:       # (MMD_GN_EPILOGUE=2 gives fp32 plans record buffers: bf16 kernel only)
            return None
        ent, c0 = self._rec_slice(out)
        if ent is None or c0 % 8 or out.shape[1] % 8 or ent["C"] % 8:
            return None
        return self._stats_for(out)

    def _stats_kw(self, out):
        """Keyword arguments for the GEMM that writes `out`: {"stats": record view}, or {} when its buffer has no record buffer."""
        rec = self._stats_for(out)
        return {} if rec is None else {"stats": rec}

    def _rec_ready(self, x, geom):
        """The record view a GroupNorm over x can finalize from: every column of x written by a statistics-emitting GEMM, contiguous
        slices that are multiples of 64 rows.  None -> the classic statistics pass."""
        ent, c0 = self._rec_slice(x)
        if (ent is None or geom.inner != 1 or geom.tstride != 1 or geom.outer_stride != geom.Tn or geom.Tn % 64 or geom.S * geom.Tn != x.shape[0]
                or c0 % 4 or x.shape[1] % 128):                    # quad records: groups must be whole quads
            return None
        if ent.get("perm_unit", 0) and geom.Tn % ent["perm_unit"]:
            return None
        need, pos = c0 + x.shape[1], c0
        for lo, hi in sorted(ent["cover"]):
            if lo > pos:
                break
            pos = max(pos, hi)
        return ent["view"][:, c0 // 4:need // 4, :] if pos >= need else None

    def _static(self, shape, dtype):
        t = torch.zeros(shape, dtype=dtype, device=self.device)
        self.keep.append(t)
        return t

    def _packed(self, kind, key, make):
        """Packed kernel operands are shared by every engine of the model that was built from the same parameter versions
        (the batch lanes of a sampler, engines of different batch sizes): one copy in HBM / MALL, not one per engine."""
        cache = self.model.__dict__.setdefault("_wcache", {})
        if cache.get("sig") != self._sig:
            cache.clear()
            cache["sig"] = self._sig
        k = (kind, key, self.dtype if kind == "gemm" else None, str(self.device))
        t = cache.get(k)
        if t is None:
            t = cache[k] = make()
        self.keep.append(t)
        return t

    def _f32(self, key):
        return self._packed("f32", key, lambda: self.params[key].detach().float().contiguous())

    def _gemm_w(self, key):
        return self._packed("gemm", key, lambda: ops.pack_conv_weight(self.params[key].detach().float(), self.dtype))

    def _edge_w(self, key):
        return self._packed("edge", key, lambda: ops.pack_edge_weight(self.params[key].detach()))

    def _temporal(self, Hh):
        """taps / dims of the k=3 conv along frames: (F, HW, 1) with taps (df, 0, 0); the equivalent (N, F, HW) form with taps
        (0, df, 0) when the halo-tile main loop is a candidate (it wants the taps inside the (D1, D2) plane)."""
        if ops.HALO_CANDIDATE:
            return dict(taps=ops.TAPS_TEMPORAL_D1, dims=(self.N, self.F, Hh * Hh))
        return dict(taps=ops.TAPS_TEMPORAL, dims=(self.F, Hh * Hh, 1))

    def _gn(self, x, prefix, geom, act, film=None, out=None):
        """GroupNorm32(+FiLM)(+SiLU): stats -> fused affine -> apply.  Returns the normalised tensor."""
        C = x.shape[1]
        if film is None and self._gn_small and ops.gn_small_ok(x, geom):
            # short slices (the temporal-attention norm: 16 frames of a pixel): one launch, one read of the tensor
            y = self._alloc(x.shape[0], C) if out is None else out
            return ops.gn_small(x, self._f32(prefix + ".GroupNorm.weight"), self._f32(prefix + ".GroupNorm.bias"), geom, act=act, out=y)
        if ops.gn_group_ok(x, geom) and self._rec_ready(x, geom) is None:
            # no producer records and a (group, slice) fits one block: statistics + apply in ONE launch (was partial | finalize | apply)
            y = self._alloc(x.shape[0], C) if out is None else out
            return ops.gn_group(x, self._f32(prefix + ".GroupNorm.weight"), self._f32(prefix + ".GroupNorm.bias"), geom, film=film, out=y, act=act)
        a, b = self._gn_affine(x, prefix, geom, film)
        y = self._alloc(x.shape[0], C) if out is None else out
        ops.gn_apply(x, a, b, geom, act=act, out=y)
        self._release(a, b)
        return y

    def _gn_affine(self, x, prefix, geom, film):
        """Fused per-(slice, channel) affine of GroupNorm32(+FiLM): from the producers' epilogue statistics when x has them, else
        by the statistics pass over x."""
        C = x.shape[1]
        a = self._alloc(geom.S, C, torch.float32)
        b = self._alloc(geom.S, C, torch.float32)
        gamma, beta = self._f32(prefix + ".GroupNorm.weight"), self._f32(prefix + ".GroupNorm.bias")
        rec = self._rec_ready(x, geom)
        if rec is not None:
            ops.gn_finalize_stats(rec, gamma, beta, geom, film=film, a=a, b=b)
        elif ops.gn_group_ok(x, geom):
            ops.gn_group(x, gamma, beta, geom, film=film, a=a, b=b)        # one launch (gn_stats: partial | finalize on multi-block slices)
        else:
            ws = self._alloc(ops.gn_workspace_bytes(x, geom) // 8, 1, torch.float64)
            ops.gn_stats(x, gamma, beta, geom, film=film, a=a, b=b, ws=ws)
            self._release(ws)
        return a, b

    def _gn_pw(self, x, gn_prefix, geom, act, wkey, bkey, film=None, residual=None, out=None):
        """GroupNorm32(+FiLM)(+SiLU) -> 1x1 conv with the normalisation applied inside the GEMM loader."""
        C = x.shape[1]
        Cout = self.params[wkey].shape[0]
        will_emit = True if (out is not None and self._has_stats(out)) else None    # the launch below emits out's statistics
        if not ops.gn_fusable(geom, C, Cout, x, will_emit, act):
            # wide outputs (qkv, deep levels) outside the row-strip kernel's shapes: every column tile would redo the normalisation
            # in its loader (measured 2x slower than materialising once), so normalise once and run the plain GEMM; likewise a strip
            # GEMM whose few rows force a deep column split (ops.gn_fusable)
            n1 = self._gn(x, gn_prefix, geom, act, film=film)
            y = self._pw(n1, wkey, bkey, residual=residual, out=out)
            self._release(n1)
            return y
        a, b = self._gn_affine(x, gn_prefix, geom, film)
        y = self._alloc(x.shape[0], Cout) if out is None else out
        ops.gn_conv1x1(x, a, b, geom, act, self._gemm_w(wkey), self._f32(bkey), residual=residual, out=y, **self._stats_kw(y))
        self._release(a, b)
        return y

    def _pw(self, x, wkey, bkey, residual=None, out=None):
        y = self._alloc(x.shape[0], self.params[wkey].shape[0]) if out is None else out
        return ops.conv_gemm(x, self._gemm_w(wkey), self._f32(bkey), residual=residual, out=y, **self._stats_kw(y))

    # ------------------------------------------------------------------ blocks
    def _self_attn(self, x, prefix, kind, Hh, out, pre=None, no_proj=False):
        """SingleModalAtten on rows x [rows, C]: kind in {'spatial','temporal','audio'} (unet:246-287,485-493).  no_proj: stop after the
        attention and return its output (the caller fuses proj_out + residual into the next launch); pre = (att, spatial prefix): this
        temporal block's input is x + proj_out_spatial(att) (the fused kernel's front stage)."""
        N, F, C = self.N, self.F, x.shape[1]
        heads = self.model.num_heads
        ch = C // heads
        rows = x.shape[0]
        if kind == "spatial":
            geom = Geom.spatial(N, F, Hh * Hh)
        elif kind == "temporal":
            geom = Geom.temporal(N, F, Hh * Hh)
        else:
            geom = Geom.per_sample(N, rows // N)
        if kind == "temporal" and self._tattn_fused and ops.tattn_fused_ok(x, heads, N, F, Hh * Hh):
            # the whole block in one launch: norm over a pixel's frames, qkv, attention, proj_out + residual (mmd_tattn_block) - and,
            # with `pre`, the spatial block's proj_out + residual in front of it
            pk = None
            if pre is not None:
                att, sp = pre
                mid = self._alloc(rows, C)
                pk = (att, self._f32(sp + ".proj_out.bias"), mid)
            wf = self._packed("tattn_pre" if pre is not None else "tattn", prefix, lambda: ops.tattn_pack(
                self._gemm_w(prefix + ".qkv.weight"), self._gemm_w(prefix + ".proj_out.weight"),
                wpre=self._gemm_w(pre[1] + ".proj_out.weight") if pre is not None else None))
            ops.tattn_block(x, wf, self._f32(prefix + ".qkv.bias"), self._f32(prefix + ".proj_out.bias"),
                            self._f32(prefix + ".norm.GroupNorm.weight"), self._f32(prefix + ".norm.GroupNorm.bias"), heads, N, F, Hh * Hh,
                            out=out, stats=self._stats_for(out, perm_unit=rows // N), pre=pk)
            if pk is not None:
                self._release(pk[2])
            return out
        qkv = self._gn_pw(x, prefix + ".norm", geom, False, prefix + ".qkv.weight", prefix + ".qkv.bias")
        att = self._alloc(rows, C)
        if kind == "temporal":
            if F > 32:
                raise H.MMDError(f"temporal attention over {F} frames: the short-sequence kernel handles <= 32")
            ops.attn_small(qkv, att, C, heads, geom)
        else:
            T = geom.Tn
            G = F if kind == "spatial" else 1
            ops.attn(qkv, qkv, att, heads, ch, N, G, G * T, T, G * T, T, 1)
        self._release(qkv)
        if no_proj:
            return att
        self._pw(att, prefix + ".proj_out.weight", prefix + ".proj_out.bias", residual=x, out=out)
        self._release(att)
        return out

    def _res(self, v, a, layer, Hh, L, out_v=None, out_a=None):
        """ResBlock (unet:434-495).  v [N*F*H*H, cin], a [N*L, cin].  Returns (v', a', H', L')."""
        N, F = self.N, self.F
        p, cin, cout = layer["prefix"], layer["cin"], layer["cout"]
        ss = self.model.use_scale_shift_norm
        film = self.emb_all[:, layer["emb_off"]: layer["emb_off"] + (2 * cout if ss else cout)]
        fh = 2 if (layer["up"] or layer["down"]) else 1
        Ho = Hh // 2 if layer["down"] else (Hh * 2 if layer["up"] else Hh)
        Lo = L // 4 if layer["down"] else (L * 4 if layer["up"] else L)
        mode = 0 if layer["down"] else 1

        def stream(x, mod, rows_in, rows_out, out):
            vid = mod == "video"
            gin = Geom.per_sample(N, rows_in // N)
            # h feeds the out_layers GroupNorm directly unless it is resampled first (up / down blocks) or shifted by the embedding
            # (non-FiLM blocks): then its producer's epilogue statistics would describe a different tensor
            # Up blocks (round 5): everything behind the nearest upsample is pointwise - the out_layers norm (the statistics of a tensor whose
            # every element is repeated fh * fw times ARE the statistics of the tensor), FiLM, SiLU, the 1x1x1 out conv, the skip
            # connection (identity or 1x1x1 conv) and their sum - so up(skip(x)) + out_conv(act(norm(up(h)))) == up(skip(x) + out_conv(act(norm(h)))):
            # the out layers run at the INPUT resolution (a quarter of the rows, h's producer statistics instead of a statistics pass) and ONE
            # upsample writes the block's output (unet:441-448, 457-476; equal up to the rounding of the statistics sums)
            low_out = layer["up"] and ss and _UP_LOWRES
            if low_out and (layer["vattn"] or layer["aattn"]):
                # the low-resolution form returns before the attention branch below: an up block WITH attention (no shipped architecture
                # builds one) must fail loudly, not be computed without its attention
                raise H.MMDError(f"{p}: an upsampling ResBlock with self-attention is not supported by the low-resolution out layers")
            hstats = (fh == 1 or low_out) and ss
            t0 = t1 = h = None
            if vid and self._vconv_fused and ops.vconv_fused_ok(x, cout, N, F, Hh, Hh):
                # in_layers norm + SiLU, spatial 3x3 and temporal k=3 in ONE launch (ds1 level): the intermediate stays in LDS
                ga, gb = self._gn_affine(x, f"{p}.{mod}_in_layers.0", gin, None)
                h = self._alloc(rows_in, cout, stats=hstats, unit=rows_in // N)
                wkey = f"{p}.video_in_layers.2.video_conv_spatial.weight"
                wf = self._packed("vconv", wkey, lambda: ops.vconv_pack(
                    self._gemm_w(wkey), self._gemm_w(f"{p}.video_in_layers.2.video_conv_temporal.weight")))
                ops.vconv2d1d(x, wf, self._f32(f"{p}.video_in_layers.2.video_conv_spatial.bias"),
                              self._f32(f"{p}.video_in_layers.2.video_conv_temporal.bias"), N, F, Hh, Hh, a=ga, b=gb, geom=gin, act=True,
                              out=h, stats=self._stats_for(h, perm_unit=rows_in // N))
                self._release(ga, gb)
            elif vid and ops.halo_gn_ok(x, ops.TAPS_SPATIAL, (N * F, Hh, Hh), gin):
                # in_layers norm + SiLU inside the 3x3 conv's halo stage: no normalised tensor in HBM (ds1 / ds2 levels)
                ga, gb = self._gn_affine(x, f"{p}.{mod}_in_layers.0", gin, None)
                t1 = ops.gn_conv_gemm(x, ga, gb, gin, True, self._gemm_w(f"{p}.video_in_layers.2.video_conv_spatial.weight"),
                                      self._f32(f"{p}.video_in_layers.2.video_conv_spatial.bias"), ops.TAPS_SPATIAL, (N * F, Hh, Hh),
                                      out=self._alloc(rows_in, cout))
                self._release(ga, gb)
                t0 = None
            elif not vid and self._aconv and ops.aconv_ok(x, cout, N, L):
                # in_layers norm + SiLU + dilated k = 3 conv in ONE launch (round 6): no normalised tensor, no gn_apply launch
                ga, gb = self._gn_affine(x, f"{p}.{mod}_in_layers.0", gin, None)
                h = self._alloc(rows_in, cout, stats=hstats, unit=rows_in // N)
                ops.aconv(x, ga, gb, self._gemm_w(f"{p}.audio_in_layers.2.audio_conv.weight"), self._f32(f"{p}.audio_in_layers.2.audio_conv.bias"),
                          N, L, layer["dilation"], act=True, out=h, stats=self._stats_for(h))
                self._release(ga, gb)
            else:
                t0 = self._gn(x, f"{p}.{mod}_in_layers.0", gin, act=True)
            if h is not None:
                pass
            elif vid:
                if t0 is not None:
                    t1 = ops.conv_gemm(t0, self._gemm_w(f"{p}.video_in_layers.2.video_conv_spatial.weight"),
                                       self._f32(f"{p}.video_in_layers.2.video_conv_spatial.bias"), taps=ops.TAPS_SPATIAL,
                                       dims=(N * F, Hh, Hh), out=self._alloc(rows_in, cout))
                    self._release(t0)
                h = self._alloc(rows_in, cout, stats=hstats, unit=rows_in // N)
                wtk = f"{p}.video_in_layers.2.video_conv_temporal.weight"
                if self._tconv and ops.tconv_ok(t1, cout, N, F, Hh * Hh):
                    # the k = 3 conv along the frames with stationary activations (tap shift = DPP lane shift): bitwise the GEMM below
                    ops.tconv(t1, self._packed("tconv", wtk, lambda: ops.tconv_pack(self._gemm_w(wtk))),
                              self._f32(f"{p}.video_in_layers.2.video_conv_temporal.bias"), cout, N, F, Hh * Hh, out=h,
                              stats=self._stats_for(h, perm_unit=rows_in // N))
                else:
                    ops.conv_gemm(t1, self._gemm_w(wtk), self._f32(f"{p}.video_in_layers.2.video_conv_temporal.bias"), **self._temporal(Hh),
                                  out=h, **self._stats_kw(h))
                self._release(t1)
            else:
                h = self._alloc(rows_in, cout, stats=hstats, unit=rows_in // N)
                ops.conv_gemm(t0, self._gemm_w(f"{p}.audio_in_layers.2.audio_conv.weight"),
                              self._f32(f"{p}.audio_in_layers.2.audio_conv.bias"), taps=ops.taps_audio(layer["dilation"]),
                              dims=(L, 1, 1), out=h, **self._stats_kw(h))
                self._release(t0)
            xs = x
            rs = (N * F, Hh, Hh, 2, 2, mode) if vid else (N, 1, L, 1, 4, mode)
            conv = "video_conv" if vid else "audio_conv"
            if low_out:
                sk = x if cin == cout else self._pw(x, f"{p}.{mod}_skip_connection.{conv}.weight", f"{p}.{mod}_skip_connection.{conv}.bias")
                ylow = self._alloc(rows_in, cout)
                self._gn_pw(h, f"{p}.{mod}_out_layers.0", gin, True, f"{p}.{mod}_out_layers.3.{conv}.weight",
                            f"{p}.{mod}_out_layers.3.{conv}.bias", film=film, residual=sk, out=ylow)
                self._release(h)
                if sk is not x:
                    self._release(sk)
                dest = self._alloc(rows_out, cout, stats=True, unit=rows_out // N) if out is None else out
                ops.resample(ylow, dest, *rs, stats=self._resample_stats(dest))
                self._release(ylow)
                return dest
            if fh != 1:        # conv at the input resolution, THEN resample both h and x (unet:441-448)
                hp, xp = self._alloc(rows_out, cout, stats=ss, unit=rows_out // N), self._alloc(rows_out, cin)
                ops.resample(h, hp, *rs, stats=self._resample_stats(hp))     # (the out_layers norm finalizes from the resample's records)
                ops.resample(x, xp, *rs)
                self._release(h)
                h, xs = hp, xp
            geom = Geom.per_sample(N, rows_out // N)
            if not ss:
                ops.add_rowbias(h, film, rows_out // N)
            if cin != cout:
                sk = self._pw(xs, f"{p}.{mod}_skip_connection.{conv}.weight", f"{p}.{mod}_skip_connection.{conv}.bias")
            else:
                sk = xs
            attn_here = layer["vattn"] if vid else layer["aattn"]
            # consumer of dest: the spatial-attention norm (per-frame slices), the audio-attention norm or the next block's norm (per sample)
            dest = self._alloc(rows_out, cout, stats=True, unit=(Ho * Ho if (attn_here and vid) else rows_out // N)) \
                if (attn_here or out is None) else out
            self._gn_pw(h, f"{p}.{mod}_out_layers.0", geom, True, f"{p}.{mod}_out_layers.3.{conv}.weight",
                        f"{p}.{mod}_out_layers.3.{conv}.bias", film=film if ss else None, residual=sk, out=dest)
            self._release(h)
            if sk is not xs:
                self._release(sk)
            if xs is not x:
                self._release(xs)
            if attn_here:
                if vid and self._tattn_fused and ops._TATTN_PRE and ops.tattn_fused_ok(dest, self.model.num_heads, N, F, Ho * Ho):
                    # spatial block up to its attention; its proj_out + residual ride in the fused temporal block's launch
                    att = self._self_attn(dest, p + ".spatial_attention_block", "spatial", Ho, None, no_proj=True)
                    fin = out if out is not None else self._alloc(rows_out, cout, stats=True, unit=rows_out // N)
                    self._self_attn(dest, p + ".temporal_attention_block", "temporal", Ho, fin, pre=(att, p + ".spatial_attention_block"))
                    self._release(att, dest)
                elif vid:
                    mid = self._self_attn(dest, p + ".spatial_attention_block", "spatial", Ho, self._alloc(rows_out, cout))
                    self._release(dest)
                    fin = out if out is not None else self._alloc(rows_out, cout, stats=True, unit=rows_out // N)
                    self._self_attn(mid, p + ".temporal_attention_block", "temporal", Ho, fin)
                    self._release(mid)
                else:
                    fin = out if out is not None else self._alloc(rows_out, cout, stats=True, unit=rows_out // N)
                    self._self_attn(dest, p + ".audio_attention_block", "audio", Ho, fin)
                    self._release(dest)
                dest = fin
            return dest

        ops.cur_sid, ops.cur_tag = 0, p + ":video"
        vo = stream(v, "video", N * F * Hh * Hh, N * F * Ho * Ho, out_v)
        ops.cur_sid, ops.cur_tag = 1, p + ":audio"
        ao = stream(a, "audio", N * L, N * Lo, out_a)
        ops.cur_sid, ops.cur_tag = 0, ""
        return vo, ao, Ho, Lo

    def _cross(self, v, a, layer, Hh, L, out_v=None, out_a=None):
        """CrossAttentionBlock (unet:655-678) with arithmetic random-shift windows."""
        N, F = self.N, self.F
        p, C, heads, win = layer["prefix"], layer["ch"], layer["heads"], layer["window"]
        ch = C // heads
        HW, apf = Hh * Hh, int(L / F)
        if apf < 1:
            raise H.MMDError(f"cross attention needs at least one audio token per frame (L={L}, F={F})")
        ops.cur_tag = p + ":cross"
        ops.cur_sid = 0
        vqkv = self._gn_pw(v, p + ".v_norm", Geom.per_sample(N, F * HW), False, p + ".v_qkv.weight", p + ".v_qkv.bias")
        ops.cur_sid = 1
        aqkv = self._gn_pw(a, p + ".a_norm", Geom.per_sample(N, L), False, p + ".a_qkv.weight", p + ".a_qkv.bias")
        ops.record_sync(1, 0)      # video queries need the audio k/v ...
        ops.record_sync(0, 1)      # ... and vice versa
        # the video stream has now waited for everything recorded on the audio stream (the record_sync(1, 0) two lines up - the releases
        # below depend on it): the video qkv buffers that earlier blocks' AUDIO attentions were still reading go back to the video pool
        self._release(*self._deferred)
        self._deferred = []
        sh = self.shift_dev[layer["shift_idx"]: layer["shift_idx"] + 1] if layer["shift"] else None
        ops.cur_sid = 0
        vatt = self._alloc(N * F * HW, C)
        ops.attn(vqkv, aqkv, vatt, heads, ch, N, F, F * HW, HW, L, apf, win, shift_dev=sh)
        if _CROSS_SERIAL:
            # Round 5.  The two attentions of a block used to start together and each stream then waited for the OTHER's (buffer recycling):
            # two MFMA-bound kernels shared the chip and the video chain - the critical path - paid for both.  Now the audio stream's
            # attention starts BEHIND the video stream's (the audio chain has ~6 ms of slack per step), the video stream never waits for it
            # (its proj_out follows its own attention at once), and the video qkv buffer the audio attention reads is parked until the
            # next point where the video stream has waited for the audio stream anyway (the next block's first sync)
            ops.record_sync(0, 1)
            ops.cur_sid = 1
            aatt = self._alloc(N * L, C)
            ops.attn(aqkv, vqkv, aatt, heads, ch, N, F, L, apf, F * HW, HW, win, shift_dev=sh)
            self._release(aqkv)    # audio pool: its readers are audio-stream launches and the video attention this stream has waited for
            self._deferred.append(vqkv)
        else:
            ops.cur_sid = 1
            aatt = self._alloc(N * L, C)
            ops.attn(aqkv, vqkv, aatt, heads, ch, N, F, L, apf, F * HW, HW, win, shift_dev=sh)
            ops.record_sync(0, 1)      # both attentions retired before either stream recycles the other's qkv buffer
            ops.record_sync(1, 0)
            self._release(vqkv, aqkv)
        ops.cur_sid = 0
        vo = out_v if out_v is not None else self._alloc(N * F * HW, C, stats=True, unit=F * HW)
        self._pw(vatt, p + ".video_proj_out.video_conv.weight", p + ".video_proj_out.video_conv.bias", residual=v, out=vo)
        self._release(vatt)
        ops.cur_sid = 1
        ao = out_a if out_a is not None else self._alloc(N * L, C, stats=True, unit=L)
        self._pw(aatt, p + ".audio_proj_out.audio_conv.weight", p + ".audio_proj_out.audio_conv.bias", residual=a, out=ao)
        self._release(aatt)
        ops.cur_sid, ops.cur_tag = 0, ""
        return vo, ao

    # ------------------------------------------------------------------ plan
    def _build(self):
        m, N, F, dt = self.model, self.N, self.F, self.dtype
        self._deferred = []
        arch_in, arch_mid, arch_out = m._arch
        if self.H0 != self.W0:
            raise H.MMDError("square frames only (the reference's avg-pool/upsample path is exercised on H == W)")
        # ---- static I/O buffers
        self.x_video = self._static((N, F, self.Cv_in, self.H0, self.W0), torch.float32)
        self.x_audio = self._static((N, self.Ca_in, self.L0), torch.float32)
        self.t_i64 = self._static((N,), torch.int64)
        self.t_f32 = self._static((N,), torch.float32)
        self.out_video = self._static((N, F, m.video_out_channels, self.H0, self.W0), torch.float32)
        self.out_audio = self._static((N, m.audio_out_channels, self.L0), torch.float32)
        # ---- shifts + emb Linear concatenation
        nshift, off = 0, 0
        Ws, bs = [], []
        for blk in arch_in + [arch_mid] + arch_out:
            for layer in blk:
                if layer["kind"] == "cross" and layer["shift"]:
                    layer["shift_idx"] = nshift
                    nshift += 1
                if layer["kind"] == "res":
                    layer["emb_off"] = off
                    W = self.params[layer["prefix"] + ".emb_layers.1.weight"]
                    Ws.append(W.detach().float())
                    bs.append(self.params[layer["prefix"] + ".emb_layers.1.bias"].detach().float())
                    off += W.shape[0]
        self.nshift = nshift
        self.shift_dev = self._static((max(nshift, 1),), torch.int32)
        self._shift_up = H.Staged(self.shift_dev)
        self.emb_W = torch.cat(Ws).contiguous()
        self.emb_b = torch.cat(bs).contiguous()
        self.emb_silu = self._static((N, self.mc), torch.float32)
        self.emb_all = self._static((N, off), torch.float32)

        def record(t_tensor):
            plan = []
            with ops.recording(plan):
                self._record(t_tensor, arch_in, arch_mid, arch_out)
            return plan

        # two plans that differ only in the timestep dtype read by the first kernel
        self.plan = record(self.t_i64)
        self.plan_f32 = [(fn, (self.t_f32.data_ptr(), 2) + args[2:], name, meta, sid, tag) if name == "mmd_temb_fwd"
                         else (fn, args, name, meta, sid, tag) for fn, args, name, meta, sid, tag in self.plan]

    def _record(self, t_tensor, arch_in, arch_mid, arch_out):
        m, N, F, dt = self.model, self.N, self.F, self.dtype
        mc = self.mc
        # fork first: the audio stream (which has ~3x less work per step than the video stream) computes the timestep embedding and the
        # FiLM table of every ResBlock, while the video stream starts on its stem convs at once; the video stream picks the table up
        # behind its stem (the first consumer is the first ResBlock's out-norm) - ~30 us off the step's critical path
        ops.cur_sid = 0
        ops.record_sync(0, 1)      # the audio stream starts behind the host-side input copies
        ops.cur_sid = 1 if _EMB_ON_AUDIO_STREAM else 0
        ops.temb(t_tensor, mc, self._f32("time_embed.0.weight"), self._f32("time_embed.0.bias"),
                 self._f32("time_embed.2.weight"), self._f32("time_embed.2.bias"), self.emb_silu)
        ops.linear(self.emb_silu, self.emb_W, self.emb_b, self.emb_all)
        ops.cur_sid = 0
        film_joined = not _EMB_ON_AUDIO_STREAM
        if not _EMB_ON_AUDIO_STREAM:
            ops.record_sync(0, 1)  # (A/B switch: embedding on the video stream, the audio stream waits for the table)

        # consumer channel split of every skip: output block k reads [h (ch_prev) | skip (ich)]
        skip_cols = []
        ch = arch_mid[-1]["cout"]
        for layers in arch_out:
            ich = layers[0]["skip_ch"]
            skip_cols.append((ch, ich))
            ch = layers[-1]["cout"] if layers[-1]["kind"] == "res" else layers[-1]["ch"]

        Hh, L = self.H0, self.L0
        cat_bufs = []   # per input block: (video cat buffer, audio cat buffer) for its consumer
        v = a = None
        nin = len(arch_in)
        for i, layers in enumerate(arch_in):
            chp, ich = skip_cols[nin - 1 - i]
            # geometry of this block's OUTPUT
            Ho, Lo = Hh, L
            for layer in layers:
                if layer["kind"] == "res" and layer["down"]:
                    Ho, Lo = Hh // 2, L // 4
            ops.cur_sid = 0
            vcat = self._alloc(N * F * Ho * Ho, chp + ich, stats=True, unit=F * Ho * Ho)
            ops.cur_sid = 1
            acat = self._alloc(N * Lo, chp + ich, stats=True, unit=Lo)
            ops.cur_sid = 0
            cat_bufs.append((vcat, acat))
            ov, oa = vcat[:, chp:], acat[:, chp:]
            for j, layer in enumerate(layers):
                last = j == len(layers) - 1
                tv, ta = (ov, oa) if last else (None, None)
                if layer["kind"] == "init":
                    C0 = layer["cout"]
                    p = layer["prefix"]
                    s1 = self._alloc(N * F * Hh * Hh, C0)
                    ops.stem_conv(self.x_video, self._edge_w(p + ".video_conv.video_conv_spatial.weight"),
                                  self._f32(p + ".video_conv.video_conv_spatial.bias"), s1, N, F, self.Cv_in, Hh, Hh,
                                  ops.TAPS_SPATIAL)
                    nv = tv if tv is not None else self._alloc(N * F * Hh * Hh, C0, stats=True, unit=F * Hh * Hh)
                    ops.conv_gemm(s1, self._gemm_w(p + ".video_conv.video_conv_temporal.weight"),
                                  self._f32(p + ".video_conv.video_conv_temporal.bias"), **self._temporal(Hh), out=nv,
                                  **self._stats_kw(nv))
                    self._release(s1)
                    ops.record_sync(1, 0)      # the FiLM table is ready (recorded behind the two embedding launches only)
                    film_joined = True
                    ops.cur_sid = 1
                    na = ta if ta is not None else self._alloc(N * L, C0)
                    ops.stem_conv(self.x_audio, self._edge_w(p + ".audio_conv.audio_conv.weight"),
                                  self._f32(p + ".audio_conv.audio_conv.bias"), na, N, 1, self.Ca_in, 1, L,
                                  [(0, 0, -1), (0, 0, 0), (0, 0, 1)])
                    ops.cur_sid = 0
                elif layer["kind"] == "res":
                    if not film_joined:        # (an architecture without an init layer in front of its first ResBlock)
                        ops.record_sync(1, 0)
                        film_joined = True
                    nv, na, Hh, L = self._res(v, a, layer, Hh, L, tv, ta)
                else:
                    nv, na = self._cross(v, a, layer, Hh, L, tv, ta)
                if j > 0:            # intermediates inside a block are ours; block inputs belong to the skip stack
                    self._release(v, a)
                v, a = nv, na

        # ---- middle: last layer writes the left slice of the first output block's concat buffer
        def run_block(layers, v, a, Hh, L, ov, oa, own_input):
            for j, layer in enumerate(layers):
                last = j == len(layers) - 1
                tv, ta = (ov, oa) if last else (None, None)
                if layer["kind"] == "res":
                    nv, na, Hh, L = self._res(v, a, layer, Hh, L, tv, ta)
                else:
                    nv, na = self._cross(v, a, layer, Hh, L, tv, ta)
                if j > 0 or own_input:
                    self._release(v, a)
                v, a = nv, na
            return v, a, Hh, L

        vcat, acat = cat_bufs[-1]
        chp, _ = skip_cols[0]
        v, a, Hh, L = run_block(arch_mid, v, a, Hh, L, vcat[:, :chp], acat[:, :chp], own_input=False)

        # ---- output blocks
        for k, layers in enumerate(arch_out):
            vcat, acat = cat_bufs[nin - 1 - k]
            if k + 1 < len(arch_out):
                nvcat, nacat = cat_bufs[nin - 2 - k]
                chn, _ = skip_cols[k + 1]
                ov, oa = nvcat[:, :chn], nacat[:, :chn]
            else:
                ov = oa = None
            v, a, Hh, L = run_block(layers, vcat, acat, Hh, L, ov, oa, own_input=False)
            self._release(vcat, acat)

        # ---- heads: GN -> SiLU -> conv (unet:1003-1012), fp32 API-layout outputs
        ops.cur_sid = 0
        gh = Geom.per_sample(N, F * Hh * Hh)
        hw = self._edge_w("video_out.2.video_conv.weight")
        hv = None
        if ops.head_gemm_ok(v, hw, gh):
            # norm + SiLU in the operand registers of a GEMM over (tap, channel) outputs, then a gather over the taps (round 5)
            ga, gb = self._gn_affine(v, "video_out.0", gh, None)
            NO = hw.shape[0] * hw.shape[2]
            P = self._alloc(NO, v.shape[0], torch.float32)
            ops.head_gemm(v, ga, gb, gh, True, self._packed("head_gemm", "video_out.2.video_conv.weight", lambda: ops.head_gemm_pack(hw)), P, NO)
            ops.head_gather(P, self._f32("video_out.2.video_conv.bias"), self.out_video, N, F, Hh, Hh, hw.shape[2], ops.TAPS_3D)
            self._release(ga, gb, P)
        else:
            hv = self._gn(v, "video_out.0", gh, act=True)
            ops.head_conv(hv, hw, self._f32("video_out.2.video_conv.bias"), self.out_video, N, F, Hh, Hh, ops.TAPS_3D)
        ops.cur_sid = 1
        ha = self._gn(a, "audio_out.0", Geom.per_sample(N, L), act=True)
        ops.head_conv(ha, self._edge_w("audio_out.2.audio_conv.weight"), self._f32("audio_out.2.audio_conv.bias"),
                      self.out_audio, N, 1, 1, L, [(0, 0, -1), (0, 0, 0), (0, 0, 1)])
        ops.cur_sid = 0
        self._release(hv, ha, v, a)
        if self._deferred:
            # the last cross block's video qkv buffer (parked while the AUDIO stream's attention read it, see _cross): back to the video
            # pool behind an explicit audio -> video sync, so that no release depends on the caller's join
            ops.record_sync(1, 0)
            self._release(*self._deferred)
            self._deferred = []
        # NOTE: no join here - the caller appends per-stream work (DDPM update) and then joins (join_plan)

    # ------------------------------------------------------------------ execution
    def set_inputs(self, video, audio, timesteps, shifts):
        N = self.N
        if tuple(video.shape) != tuple(self.x_video.shape) or tuple(audio.shape) != tuple(self.x_audio.shape):
            raise H.MMDError(f"input shapes {tuple(video.shape)}/{tuple(audio.shape)} do not match the model's "
                             f"{tuple(self.x_video.shape)}/{tuple(self.x_audio.shape)}")
        if video.data_ptr() != self.x_video.data_ptr():
            self.x_video.copy_(video)
        if audio.data_ptr() != self.x_audio.data_ptr():
            self.x_audio.copy_(audio)
        use_f32 = timesteps.dtype.is_floating_point
        tgt = self.t_f32 if use_f32 else self.t_i64
        if timesteps.data_ptr() != tgt.data_ptr():
            tgt.copy_(timesteps)
        self.set_shifts(shifts)
        return use_f32

    def set_shifts(self, shifts):
        if self.nshift:
            if len(shifts) != self.nshift:
                raise H.MMDError(f"expected {self.nshift} window shifts, got {len(shifts)}")
            host = self._shift_up.host()
            for i, s in enumerate(shifts):
                host[i] = int(s)
            self._shift_up.push()

    def join_plan(self):
        """Plan tail: the main stream waits for the audio stream (everything after it sees both outputs)."""
        tail = []
        with ops.recording(tail):
            ops.record_sync(1, 0)
        return tail

    def run(self, use_f32=False):
        if not hasattr(self, "_join"):
            self._join = self.join_plan()
        st = H.stream_handle()
        self.aux.wait_stream(torch.cuda.current_stream(self.device))   # aux starts clean behind the inputs
        ops.run_plan((self.plan_f32 if use_f32 else self.plan) + self._join, st, self.aux.cuda_stream)

    def _graph(self, use_f32):
        """hipGraph of the whole forward plan (captured on first use, one per timestep dtype): every no-grad `model(x, t)` call -
        DPM-Solver, eager DDIM / conditional loops - is then ONE launch instead of ~1100 ctypes launches from Python."""
        if not hasattr(self, "_graphs"):
            self._graphs = {}
        if use_f32 not in self._graphs:
            if not hasattr(self, "_join"):
                self._join = self.join_plan()
            plan = (self.plan_f32 if use_f32 else self.plan) + self._join
            side = self.side
            side.wait_stream(torch.cuda.current_stream(self.device))
            ops.run_plan(plan, side.cuda_stream, self.aux.cuda_stream)      # warm-up: one-time function attributes
            torch.cuda.synchronize(self.device)
            with H.capture(side.cuda_stream) as cap:
                ops.run_plan(plan, side.cuda_stream, self.aux.cuda_stream)
            torch.cuda.current_stream(self.device).wait_stream(side)
            self._graphs[use_f32] = cap.exec
        return self._graphs[use_f32]

    def forward(self, video, audio, timesteps, shifts, use_graph=True):
        use_f32 = self.set_inputs(video, audio, timesteps, shifts)
        if use_graph:
            H.call("mmd_graph_launch", self._graph(use_f32), H.stream_handle())
        else:
            self.run(use_f32)
        return self.out_video.clone(), self.out_audio.clone()

    def close(self):
        """Give up the HIP handles of this engine (graph execs, fork/join events, the two private streams).  Nothing is destroyed
        here - a finaliser may run in the middle of another engine's capture - the handles are retired and H.reap() frees them
        at the next safe point."""
        try:
            graphs, self._graphs = getattr(self, "_graphs", {}), {}
            for g in graphs.values():
                H.retire("graph", g)
            plans = [p for p in (getattr(self, "plan", None), getattr(self, "plan_f32", None), getattr(self, "_join", None)) if p]
            seen = set()
            for ev in H.plan_events(*plans):
                if ev.value not in seen:
                    seen.add(ev.value)
                    H.retire("event", ev)
            self.plan, self.plan_f32, self._join = [], [], []
            for s in (getattr(self, "_aux", None), getattr(self, "_side", None)):
                if s is not None:
                    s.close()
        except Exception:
            pass

    __del__ = close
