"""Host-side logic of the product (no GPU): flag surface, state-dict parity, diffusion tables, respacing,
loud failure without a device."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import DEFAULT_FLAGS, GOLD, flags, gold, gold_keys


def test_flag_surface_matches_reference_defaults():
    from mm_diffusion import multimodal_script_util as msu
    assert msu.model_and_diffusion_defaults() == DEFAULT_FLAGS
    import argparse
    p = argparse.ArgumentParser()
    msu.add_dict_to_argparser(p, msu.model_and_diffusion_defaults())
    a = p.parse_args(["--use_fp16", "True", "--num_channels", "64", "--resblock_updown", "yes"])
    assert a.use_fp16 is True and a.num_channels == 64 and a.resblock_updown is True
    assert msu.args_to_dict(a, ["num_channels"]) == {"num_channels": 64}
    with pytest.raises(argparse.ArgumentTypeError):
        msu.str2bool("maybe")


@pytest.mark.parametrize("name,cfg,over", [("tiny", "tiny", {}), ("full", "full", {}), ("tiny_learn_sigma", "tiny", dict(learn_sigma=True))])
def test_state_dict_keys_shapes_and_order(name, cfg, over):
    from mm_diffusion import multimodal_script_util as msu
    model, _ = msu.create_model_and_diffusion(**flags(cfg, **over))
    want = gold_keys()[name]
    assert [[k, list(v.shape)] for k, v in model.state_dict().items()] == want
    assert [k for k, _ in model.named_parameters()] == [k for k, _ in want]
    assert model.video_size == flags(cfg)["video_size"] and model.audio_size == flags(cfg)["audio_size"]
    assert model.video_out_channels == (6 if over else 3) and model.audio_out_channels == (2 if over else 1)
    assert model.dtype == torch.float32
    # fresh model: zero-initialised output convs like the reference (zero_module)
    sd = model.state_dict()
    assert float(sd["video_out.2.video_conv.weight"].abs().max()) == 0
    assert float(sd["input_blocks.1.0.video_out_layers.3.video_conv.weight"].abs().max()) == 0


def test_load_state_dict_tolerant():
    from mm_diffusion import logger, multimodal_script_util as msu
    logger.set_quiet(True)
    model, _ = msu.create_model_and_diffusion(**flags("tiny"))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    sd["time_embed.0.weight"] = torch.zeros(3, 3)       # shape mismatch -> dropped
    sd.pop("time_embed.0.bias")                          # missing -> tolerated
    sd["bogus"] = torch.zeros(1)
    with pytest.raises(RuntimeError):                   # strict=True would still reject the unexpected key
        model.load_state_dict_(dict(sd), is_strict=True)
    sd.pop("bogus")
    model.load_state_dict_(sd, is_strict=False)


def test_diffusion_tables_match_reference():
    from mm_diffusion import multimodal_script_util as msu
    from mm_diffusion import multimodal_gaussian_diffusion as gd
    from mm_diffusion.multimodal_respace import space_timesteps
    g = gold("tables")
    for sched in ("linear", "cosine"):
        for steps in (1000, 50):
            np.testing.assert_array_equal(gd.get_named_beta_schedule(sched, steps), g[f"betas_{sched}_{steps}"])
    with open(os.path.join(GOLD, "space_timesteps.json")) as f:
        for k, v in json.load(f).items():
            steps, sc = k.split("|")
            assert sorted(space_timesteps(int(steps), sc)) == v
    for resp in ("", "2", "250", "ddim25"):
        d = msu.create_gaussian_diffusion(steps=1000, timestep_respacing=resp)
        tag = resp or "full"
        assert d.timestep_map == list(g[f"{tag}.timestep_map"])
        for attr in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                     "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                     "posterior_mean_coef1", "posterior_mean_coef2", "sqrt_alphas_cumprod",
                     "sqrt_one_minus_alphas_cumprod"):
            np.testing.assert_array_equal(getattr(d, attr), g[f"{tag}.{attr}"])
    d = msu.create_gaussian_diffusion(learn_sigma=True, predict_xstart=True, rescale_learned_sigmas=True)
    assert d.model_var_type == gd.ModelVarType.LEARNED_RANGE and d.model_mean_type == gd.ModelMeanType.START_X
    assert d.loss_type == gd.LossType.RESCALED_MSE
    with pytest.raises(ValueError):
        space_timesteps(10, "20")


def test_shift_draws_follow_reference_order():
    import random
    from mm_diffusion import multimodal_script_util as msu
    model, _ = msu.create_model_and_diffusion(**flags("tiny"))
    g = gold("tiny_forward")
    random.seed(int(g["seed"]))
    assert model.draw_shifts() == list(g["shifts"])        # same global `random` stream, same call order
    model.shift_source = lambda lo, hi: hi
    F = 8
    assert model.draw_shifts() == [F - 1, F - 4, F - 8, F - 8, F - 8, F - 4, F - 4, F - 1, F - 1]


def test_no_cpu_fallback():
    """The product never computes on the CPU: CPU tensors / missing device raise."""
    from mm_diffusion import multimodal_script_util as msu
    from mm_diffusion._hip import MMDError
    fl = flags("tiny")
    model, diff = msu.create_model_and_diffusion(**fl)
    v, a = torch.zeros(1, *fl["video_size"]), torch.zeros(1, *fl["audio_size"])
    with torch.no_grad(), pytest.raises(MMDError):
        model(v, a, torch.zeros(1, dtype=torch.int64))
    with pytest.raises(MMDError):
        diff.p_sample_loop(model, {"video": (1, *fl["video_size"]), "audio": (1, *fl["audio_size"])},
                           device=torch.device("cpu"), progress=False)
    if not torch.cuda.is_available():
        with pytest.raises(MMDError):
            diff.q_sample(v, torch.zeros(1, dtype=torch.int64))


def test_shard_batch_partition():
    from mm_diffusion.dist_util import shard_batch
    for gb, w in ((32, 8), (10, 4), (3, 8), (7, 1)):
        parts = [shard_batch(gb, w, r) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == gb
        assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        assert max(hi - lo for lo, hi in parts) - min(hi - lo for lo, hi in parts) <= 1


def test_sr_model_surface_matches_reference_keys():
    """image_sr_create_model_and_diffusion: same default flag set, and the tiny SR model registers exactly the reference's
    state-dict keys / shapes / order (tests/golden/sr_state_dict_keys.json, captured from the reference)."""
    import json
    import os
    from helpers import GOLD
    from mm_diffusion import script_util as su
    d = su.image_sr_model_and_diffusion_defaults()
    assert d["sr_learn_sigma"] is True and d["large_size"] == 256 and "diffusion_steps" not in d and "sr_diffusion_steps" in d
    d.update(large_size=64, small_size=16, sr_num_channels=32, sr_num_res_blocks=1, sr_attention_resolutions="2,4", sr_num_heads=2,
             sr_resblock_updown=True, sr_timestep_respacing="ddim4")
    model, diff = su.image_sr_create_model_and_diffusion(**d)
    with open(os.path.join(GOLD, "sr_state_dict_keys.json")) as f:
        keys = json.load(f)["tiny"]
    assert [(k, list(v.shape)) for k, v in model.state_dict().items()] == [(k, s) for k, s in keys]
    assert diff.num_timesteps == 4 and len(diff.timestep_map) == 4
    import pytest
    import torch
    from mm_diffusion._hip import MMDError
    with pytest.raises(MMDError):                      # no CPU fallback
        model(torch.zeros(1, 3, 64, 64), torch.zeros(1, dtype=torch.int64), low_res=torch.zeros(1, 3, 16, 16))


def test_script_shims_common_and_datasets(tmp_path):
    """The thin counterparts the training / sampling scripts import: run set-up, wav / png writers, clip loader contract."""
    import types
    import wave
    import numpy as np
    import torch
    from mm_diffusion import common, logger
    from mm_diffusion.multimodal_datasets import load_data
    logger.set_quiet(True)
    args = types.SimpleNamespace(output_dir=str(tmp_path / "out"), seed=3)
    assert common.set_seed_logger(args) is args and (tmp_path / "out").is_dir()
    common.save_audio(np.sin(np.arange(1600) / 10)[None], str(tmp_path / "a.wav"), 16000)
    with wave.open(str(tmp_path / "a.wav")) as w:
        assert w.getframerate() == 16000 and w.getnframes() == 1600
    common.save_one_video(torch.zeros(2, 3, 3, 8, 8), str(tmp_path / "v.gif"), row=2)
    assert (tmp_path / "v.png").exists()
    for i in range(3):
        np.savez(tmp_path / f"clip{i}.npz", video=np.full((4, 8, 8, 3), 10 * i, dtype=np.uint8), audio=np.zeros(64, dtype=np.float32))
    it = load_data(data_dir=str(tmp_path), batch_size=2, video_size=[4, 3, 8, 8], audio_size=[1, 64], deterministic=True)
    b = next(it)
    assert b["video"].shape == (2, 4, 3, 8, 8) and b["audio"].shape == (2, 1, 64) and float(b["video"].min()) == -1.0
    syn = next(load_data(data_dir="synthetic", batch_size=3, video_size=[4, 3, 8, 8], audio_size=[1, 64]))
    assert syn["video"].shape == (3, 4, 3, 8, 8) and float(syn["video"].abs().max()) <= 1.0


def test_halo_tile_addressing_model():
    """numpy model of conv_gemm tile 130's addressing (tools/halo_index_check.py): the DMA lane mapping + chunk swizzle must put
    every (pixel + tap, channel chunk) where the fragment reads look for it - zero rows on the frame border - with no two lanes of
    a 16-lane ds_read_b128 group in the same 16-byte bank slot."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("halo_index_check", os.path.join(os.path.dirname(__file__), "..", "tools", "halo_index_check.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.run(2, 16, 32, 128, m.taps, 1, 0, 0)            # top-left patch: halo rows / columns outside the frame
    assert m.run(2, 16, 32, 64, m.taps, 0, 8, 16)            # bottom-right patch
    assert m.run(2, 16, 48, 64, [(-1, 0), (0, 0), (1, 0)], 1, 8, 16)   # temporal form: taps along D1 only


def test_halo_tile_block_flow_model():
    """Sequential model of a tile-130 block (tools/halo_flow_check.py): double-buffered weight stages, the halo of the next
    channel chunk issued across the taps of the current one, chunk-major K offsets and the patch -> row mapping of the epilogue
    reproduce a direct convolution (3x3 with several chunks and a ragged Cout, temporal form, 1x1)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("halo_flow_check", os.path.join(os.path.dirname(__file__), "..", "tools", "halo_flow_check.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.check(1, 8, 32, 128, 136, m.sp) < 1e-10
    assert m.check(1, 16, 16, 64, 64, [(0, -1, 0), (0, 0, 0), (0, 1, 0)]) < 1e-10
    assert m.check(1, 8, 16, 64, 8, [(0, 0, 0)]) < 1e-10


def test_halo_tile_is_pinned_by_layer_geometry_only():
    """The halo tiles (130 / 133) are chosen for spatial 3x3 convs on frames of >= 1024 pixels whatever the batch
    (ops.halo_tile_pinned): the choice may never depend on the batch size, or a batch-4 plan and a batch-1 plan would differ in the
    last bit; between 130 and 133 (bitwise equal) the channel count decides."""
    import torch
    from mm_diffusion import ops
    for n in (1, 2, 4, 8):
        x = torch.zeros(n * 16 * 32 * 32, 256, dtype=torch.bfloat16)
        assert ops.halo_tile_pinned(x, ops.TAPS_SPATIAL, (n * 16, 32, 32))                     # ds2 frames: 1024 pixels
        x = torch.zeros(n * 16 * 16 * 16, 384, dtype=torch.bfloat16)
        assert not ops.halo_tile_pinned(x, ops.TAPS_SPATIAL, (n * 16, 16, 16))                 # ds4 frames: 256 pixels (measured a loss)
        assert ops.halo_tile_code(x, ops.TAPS_SPATIAL, (n * 16, 16, 16)) == 133
    assert ops.halo_tile_code(torch.zeros(16 * 64 * 64, 128, dtype=torch.bfloat16), ops.TAPS_SPATIAL, (16, 64, 64)) == 130
    x = torch.zeros(16 * 64 * 64, 128, dtype=torch.bfloat16)
    assert not ops.halo_tile_pinned(x, ops.TAPS_TEMPORAL, (16, 4096, 1))                      # temporal k=3: not a 9-tap conv
    assert not ops.halo_tile_pinned(x.float(), ops.TAPS_SPATIAL, (16, 64, 64))                # fp32 mode keeps the bitwise-equal tiles
    assert not ops.halo_tile_pinned(torch.zeros(16 * 64 * 64, 8, dtype=torch.bfloat16), ops.TAPS_SPATIAL, (16, 64, 64))   # Cin below one K step


def test_strip_kernel_lane_model():
    """Lane-level numpy model of conv_gemm tile 131 (tools/strip_model.py: weight DMA image + swizzle, fragment reads, the
    activation fragments loaded straight from global memory, GroupNorm slice selection, v_permlane32_swap epilogue, recursive-halving
    statistics, column split and XCD remap, ragged last strip): every output element is written exactly once and equals
    X W^T + bias (+ R) to bf16 rounding; the records equal the sums over the stored values."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("strip_model", os.path.join(os.path.dirname(__file__), "..", "tools", "strip_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for (M, K, Cout, res, gn, st) in [(512, 128, 128, True, (256, True), True),        # two strips x two column ranges, statistics
                                      (320, 256, 192, False, None, False),             # ragged second strip, three column ranges
                                      (384, 384, 96, False, (128, False), False),      # one row fragment per wave, 32-channel chunks
                                      (800, 128, 64, True, (400, True), True)]:        # a strip that straddles two GroupNorm slices
        Y, ref, stats, sref, touched, nsplit = m.run(M, K, Cout, residual=res, gn=gn, want_stats=st and M % 64 == 0)
        assert (touched == 1).all() and not np.isnan(Y).any()
        assert np.abs(Y - ref).max() / np.abs(ref).max() < 4e-3
        if st and M % 64 == 0:
            assert np.abs(stats - sref).max() / np.abs(sref).max() < 1e-6
    # one row fragment per wave with statistics (the even wave of a pair adds its partner's half-record after the chunk's barrier),
    # K = 512, and the k=3 convs at 128 channels (a 64-channel plane of the fragment = the row shifted by the tap, or the zero row)
    for (M, K, Cout, res, taps, dims) in [(384, 384, 96, False, None, (1, 1, 1)), (256, 512, 96, True, None, (1, 1, 1)),
                                          (512, 384, 64, True, [(-1, 0, 0), (0, 0, 0), (1, 0, 0)], (4, 128, 1)),
                                          (640, 384, 32, False, [(-4, 0, 0), (0, 0, 0), (4, 0, 0)], (320, 1, 1))]:
        Y, ref, stats, sref, touched, nsplit = m.run(M, K, Cout, residual=res, want_stats=True, taps=taps, dims=dims)
        assert (touched == 1).all() and not np.isnan(Y).any() and not np.isnan(stats).any()
        assert np.abs(Y - ref).max() / np.abs(ref).max() < 4e-3
        assert np.abs(stats - sref).max() / np.abs(sref).max() < 1e-6
    # the column split only fills the chip: it never splits below one chunk and always divides the chunk count
    for M, Cout, BR, CC in [(65536, 768, 256, 64), (16384, 1152, 128, 32), (262144, 128, 256, 64), (1600, 256, 256, 64), (64, 64, 256, 64)]:
        n = m.pick_nsplit(M, Cout, BR, CC)
        assert (Cout // CC) % n == 0 and 1 <= n <= 16


def test_strip_tile_is_pinned_by_layer_geometry_only(monkeypatch):
    """Tile 131 is chosen from dtype, taps and channel counts (and, when GroupNorm is fused, the slice length) - never from M or a
    timing: its statistics are folded in its own order, so a layer must run it at every batch size or at none."""
    import torch
    from mm_diffusion import ops
    monkeypatch.setattr(ops, "_STRIP_MODE", "pin")
    for n in (1, 2, 4, 8):
        x = torch.zeros(n * 16 * 1024, 256, dtype=torch.bfloat16)
        assert ops.strip_tile_pinned(x, 768) and ops.strip_tile_pinned(x, 256, stats=torch.zeros(n * 256, 256, 2))
        assert ops.strip_tile_pinned(x, 768, geom=ops.Geom.spatial(n, 16, 1024))
        assert ops.gn_fusable(ops.Geom.spatial(n, 16, 1024), 256, 768, x) and not ops.gn_fusable(ops.Geom.spatial(n, 16, 1024), 256, 768)
        assert not ops.strip_tile_pinned(x, 768, geom=ops.Geom.temporal(n, 16, 1024))          # strided slices: gn_apply + plain strip GEMM
        x3 = torch.zeros(n * 16 * 256, 384, dtype=torch.bfloat16)
        assert ops.strip_tile_pinned(x3, 1152) and ops.strip_tile_pinned(x3, 384, stats=torch.zeros(n * 64, 384, 2))
        g3 = ops.Geom.per_sample(n, 16 * 256)
        assert ops.gn_fusable(g3, 384, 1152, x3) and ops.gn_fusable(g3, 384, 384, x3, True)
        assert not ops.strip_tile_ok(x3, 384, stats=True, base=True) and not ops.gn_fusable(ops.Geom.per_sample(n * 64, 64), 384, 384, x3, True)
        x1 = torch.zeros(n * 16 * 4096, 128, dtype=torch.bfloat16)
        assert ops.strip_tile_pinned(x1, 128, taps=ops.TAPS_TEMPORAL, stats=True) and ops.strip_tile_pinned(x1[: n * 25600], 128, taps=ops.taps_audio(16))
    # fusing GroupNorm + SiLU into the strip is a speed choice between two bitwise-equal paths: not when few rows force a deep column split
    big, small = torch.zeros(262144, 128, dtype=torch.bfloat16), torch.zeros(4096, 512, dtype=torch.bfloat16)
    assert ops.gn_fusable(ops.Geom.per_sample(4, 65536), 128, 128, big, True, True) and ops.strip_column_split(262144, 128, 128) == 1
    assert not ops.gn_fusable(ops.Geom.per_sample(4, 1024), 512, 512, small, True, True) and ops.strip_column_split(4096, 512, 512) == 16
    assert ops.gn_fusable(ops.Geom.per_sample(4, 1024), 512, 1536, small, None, False)         # no SiLU (qkv norms): always fused
    assert not ops.gn_fusable(ops.Geom.per_sample(4, 6400), 256, 256, torch.zeros(25600, 256, dtype=torch.bfloat16), True, True)
    x = torch.zeros(4096, 256, dtype=torch.bfloat16)
    assert not ops.strip_tile_pinned(x.float(), 256)                                           # fp32 mode keeps the exact-fp32 tiles
    assert not ops.strip_tile_pinned(x, 256, taps=ops.TAPS_TEMPORAL)                           # K = 768
    assert ops.strip_tile_pinned(torch.zeros(4096, 512, dtype=torch.bfloat16), 512)
    assert not ops.strip_tile_pinned(torch.zeros(4096, 640, dtype=torch.bfloat16), 512)
    assert not ops.strip_tile_pinned(x, 96 + 8)                                                # Cout not a multiple of the 64-channel chunk


def test_default_lanes_and_bucket_partition():
    import torch
    from mm_diffusion.optim import FlatAdamW
    from mm_diffusion.sampler import default_lanes
    assert default_lanes(4) == 2 and default_lanes(8) == 2 and default_lanes(1) == 1 and default_lanes(2) == 1 and default_lanes(5) == 1
    g = torch.Generator().manual_seed(0)
    for nb in (1, 2, 4, 9):
        sizes = [int(v) for v in torch.randint(1, 5000, (23,), generator=g)]
        params = [torch.nn.Parameter(torch.zeros(s)) for s in sizes]
        opt = FlatAdamW(params, lr=1e-3, grad_buckets=nb)
        assert 1 <= len(opt.buckets) <= nb and opt.buckets[0][0] == 0 and opt.buckets[-1][1] == sum(sizes)
        assert all(a[1] == b[0] for a, b in zip(opt.buckets, opt.buckets[1:])) and sum(b[2] for b in opt.buckets) == len(params)
        assert opt.bucket_of == sorted(opt.bucket_of) and len(set(opt.bucket_of)) == len(opt.buckets)
        off = 0
        for i, p in enumerate(params):                        # every parameter lies inside the bucket it is assigned to
            lo, hi, _ = opt.buckets[opt.bucket_of[i]]
            assert lo <= off and off + p.numel() <= hi
            off += p.numel()


def test_recording_keep_list_parks_op_allocations():
    """ops.recording(plan, keep=[...]): buffers an op allocates while recording must outlive the plan (it replays raw pointers)."""
    import torch
    from mm_diffusion import ops
    keep = []
    with ops.recording([], keep=keep):
        t = ops.alloc(3, 4, dtype=torch.float32, device="cpu")
    assert len(keep) == 1 and keep[0] is t
    u = ops.alloc(2, dtype=torch.float32, device="cpu")
    assert len(keep) == 1 and u.shape == (2,)


def test_retired_handles_are_only_destroyed_by_reap():
    from mm_diffusion import _hip
    n0 = len(_hip._retired)
    _hip.retire("event", None)
    _hip.retire("graph", 0)
    assert len(_hip._retired) == n0                            # null handles are ignored
    _hip.retire("event", 1234)
    assert _hip._retired[-1] == ("event", 1234)
    _hip._retired.pop()


def test_transposing_read_tile_model():
    """tools/tr_read_model.py: the row-major, two-row-swizzled LDS plane written by the descriptor DMA and read with ds_read_b64_tr_b16
    (attention V operand, both operands of the 9-tap weight gradient): every lane of every fragment receives exactly the eight rows of
    its channel that the MFMA k-slots stand for, and the two half-waves cover the sixteen rows of a k-step once."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("tr_read_model", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "tr_read_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.check()


def test_vconv_index_model():
    """The lane-level numpy model of the fused VideoConv kernel's index maps (weight image, halo pieces, swizzles, fragment addresses,
    MFMA layouts, T image, epilogue rows and records - tools/vconv_model.py restates the kernel's expressions) against two direct
    convolutions, exact on integer data: with / without the fused input norm, two chunk counts, several patches and samples."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("vconv_model", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "vconv_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.check(N=1, H=8, W=4, Cin=64, with_gn=True, seed=1)
    assert m.check(N=2, H=4, W=4, Cin=32, with_gn=False, seed=2)


def test_vconv_dma_schedule_model():
    """The counted s_waitcnt vmcnt(N) of the fused VideoConv kernel against its DMA issue order (tools/vconv_dma_model.py): at the
    top of every step everything the step reads - weights, the halo rows of its taps, the slots and affine rows of its norm pieces -
    has landed, for 1 .. 12 channel chunks with and without the fused norm."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("vconv_dma_model", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "vconv_dma_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.check()


def test_noise_schedule_vp_matches_reference_fixture():
    """NoiseScheduleVP (round 4: own structure - one knot-table object used in both directions, one small class per schedule family)
    against the reference's class evaluated on fixed times (tools/gen_golden.py: gen_noise_schedule): every marginal_* function and
    inverse_lambda, all three families.  Bitwise where torch's CPU kernels are deterministic; asserted to 1 ulp-level tolerance."""
    import numpy as np
    import torch
    from helpers import gold
    from mm_diffusion.multimodal_dpm_solver_plus import NoiseScheduleVP
    g = gold("noise_schedule_vp")
    betas = np.linspace(1e-4, 0.02, 1000, dtype=np.float64)
    cases = {"discrete_betas": dict(schedule="discrete", betas=torch.tensor(betas)),
             "discrete_acp": dict(schedule="discrete", alphas_cumprod=torch.tensor(np.cumprod(1 - betas), dtype=torch.float32)),
             "linear": dict(schedule="linear"), "cosine": dict(schedule="cosine")}
    for tag, kw in cases.items():
        ns = NoiseScheduleVP(**kw)
        assert ns.T == float(g[f"{tag}_T"]) and ns.total_N == int(g[f"{tag}_N"])
        t = torch.tensor(g[f"{tag}_t"])
        for name, fn in (("log_alpha", ns.marginal_log_mean_coeff), ("alpha", ns.marginal_alpha), ("std", ns.marginal_std), ("lambda", ns.marginal_lambda)):
            out = fn(t)
            assert tuple(out.shape) == g[f"{tag}_{name}"].shape
            np.testing.assert_allclose(out.numpy(), g[f"{tag}_{name}"], rtol=2e-7, atol=1e-9, err_msg=f"{tag} {name}")
        inv = ns.inverse_lambda(torch.tensor(g[f"{tag}_lambda"]))
        assert tuple(inv.shape) == g[f"{tag}_inv"].shape
        np.testing.assert_allclose(inv.numpy(), g[f"{tag}_inv"], rtol=2e-7, atol=1e-9, err_msg=f"{tag} inverse_lambda")
    with pytest.raises(ValueError):
        NoiseScheduleVP("quadratic")
    # the read-only counterparts of the reference class's public attributes (dpm:113-118)
    ns = NoiseScheduleVP(schedule="discrete", betas=torch.tensor(betas))
    assert tuple(ns.t_array.shape) == (1, 1000) and tuple(ns.log_alpha_array.shape) == (1, 1000)
    assert float(ns.t_array[0, 0]) == pytest.approx(1e-3) and float(ns.t_array[0, -1]) == 1.0
    np.testing.assert_allclose(ns.log_alpha_array[0].numpy(), 0.5 * np.log(1 - betas).cumsum(), rtol=1e-6)
    lin = NoiseScheduleVP("linear", continuous_beta_0=0.2, continuous_beta_1=15.)
    assert lin.beta_0 == 0.2 and lin.beta_1 == pytest.approx(15.)
    with pytest.raises(AttributeError):
        lin.t_array


def test_tattn_register_model():
    """tools/tattn_model.py: the lane-level model of the fused temporal-attention kernel's register algebra (accumulators packed straight
    into the next MFMA's operands, v^T by swapping the operands, the K-permuted proj_out weight) reproduces the block's math."""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "tattn_model.py")
    spec = importlib.util.spec_from_file_location("tattn_model", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.check(seed=1) < 1e-12


def test_tconv_register_model():
    """tools/tconv_model.py: the temporal conv's tap shift as a DPP lane shift of the stationary operand fragments (zeros shifted in =
    the zero padding in time), in the kernel's K order, against a float64 conv1d."""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "tconv_model.py")
    spec = importlib.util.spec_from_file_location("tconv_model", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.check(seed=2) < 1e-12


def test_profile_summarisers_on_a_synthetic_trace(tmp_path):
    """tools/kt_by_shape.py (kernel trace per launch shape and HW queue over the last replayed steps) and tools/clock_summary.py (GRBM_GUI_ACTIVE /
    dispatch wall time per kernel) on a hand-made rocprofv3 csv: three replayed steps of two kernels on two queues."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = ('"Kind","Agent_Id","Queue_Id","Stream_Id","Thread_Id","Dispatch_Id","Kernel_Id","Kernel_Name","Correlation_Id","Start_Timestamp",'
           '"End_Timestamp","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Workgroup_Size_X","Workgroup_Size_Y",'
           '"Workgroup_Size_Z","Grid_Size_X","Grid_Size_Y","Grid_Size_Z"')
    rows, disp, t, did = [hdr], [], 1_000_000, 0
    for step in range(4):                                   # 4 x ddpm_update marks = 3 complete steps behind the first mark
        for q, name, grid, dur in ((4, "void conv1x1_strip_kernel<8, 1, 32, 0, 0>(ConvGemmParams, int)", 131072, 12_000),
                                   (4, "void conv1x1_strip_kernel<8, 1, 32, 0, 0>(ConvGemmParams, int)", 65536, 8_000),
                                   (1, "gn_finalize_rec_kernel(float const*, long)", 8192, 4_000),
                                   (4, "ddpm_update_kernel(DdpmParams)", 4096, 2_000)):
            did += 1
            rows.append(f'"KERNEL_DISPATCH","Agent 2",{q},0,1,{did},7,"{name}",{did},{t},{t + dur},0,0,64,0,32,256,1,1,{grid},1,1')
            disp.append((did, name, t, t + dur))
            t += dur + 1_000
        t += 3_000_000                                      # the gap between replayed steps
    trace = tmp_path / "kt_kernel_trace.csv"
    trace.write_text("\n".join(rows) + "\n")
    out = tmp_path / "by_shape.txt"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "kt_by_shape.py"), str(trace), str(out), "3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    txt = out.read_text()
    assert "last 3 replayed steps" in txt and "queue 4" in txt and "queue 1" in txt
    line = [ln for ln in txt.splitlines() if "conv1x1_strip_kernel<8, 1, 32, 0, 0> | 131072x1x1" in ln][0].split()
    assert float(line[0]) == 1.0 and float(line[2]) == 12.0 and float(line[3]) == 12.0          # one launch per step, 12 us average and minimum
    assert any("conv1x1_strip_kernel<8, 1, 32, 0, 0> | 65536x1x1" in ln for ln in txt.splitlines())   # the same kernel, another shape: its own row
    # two batch lanes whose updates end 3 ms apart (round 6): the step is delimited by the noise draw, not by clustered update end times
    rows2, t, did = [hdr], 1_000_000, 0
    for step in range(5):
        seq = [(1, "void at::native::(anonymous namespace)::distribution_elementwise_grid_stride_kernel<float, 4, normal>(long)", 524288, 5_000, 0),
               (1, "void at::native::(anonymous namespace)::distribution_elementwise_grid_stride_kernel<float, 4, normal>(long)", 102400, 3_000, 0)]
        for lane, q in ((0, 1), (1, 2)):
            seq += [(q, "gn_finalize_rec_kernel(float const*, long)", 8192, 4_000, 0), (q, "ddpm_update_kernel(DdpmParams)", 4096, 2_000, 3_000_000 * lane)]
        for q, name, grid, dur, wait in seq:
            did += 1
            t += wait
            rows2.append(f'"KERNEL_DISPATCH","Agent 2",{q},0,1,{did},7,"{name}",{did},{t},{t + dur},0,0,64,0,32,256,1,1,{grid},1,1')
            t += dur + 1_000
        t += 500_000
    trace2 = tmp_path / "kt2_kernel_trace.csv"
    trace2.write_text("\n".join(rows2) + "\n")
    out3 = tmp_path / "by_shape2.txt"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "kt_by_shape.py"), str(trace2), str(out3), "3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    txt2 = out3.read_text()
    fin = [ln for ln in txt2.splitlines() if "gn_finalize_rec_kernel" in ln][0].split()
    assert "last 3 replayed steps" in txt2 and float(fin[0]) == 2.0                          # both lanes' launches belong to ONE step
    # clock summary: 1.7 GHz x 8 XCCs x the dispatch's wall time in the counter
    pmc = tmp_path / "pmc"
    pmc.mkdir()
    (pmc / "p_kernel_trace.csv").write_text("\n".join(rows) + "\n")
    cc = ['"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name","Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value","Start_Timestamp","End_Timestamp"']
    for d, name, s, e in disp:
        cc.append(f'{d},{d},"Agent 2",4,1,1,4096,7,"{name}",256,0,0,64,0,32,"GRBM_GUI_ACTIVE",{8 * 1.7 * (e - s)},{s},{e}')
    (pmc / "p_counter_collection.csv").write_text("\n".join(cc) + "\n")
    out2 = tmp_path / "clocks.txt"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "clock_summary.py"), str(pmc), str(out2), "strip"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    row = [ln for ln in out2.read_text().splitlines() if "conv1x1_strip_kernel" in ln][0].split()
    assert abs(float(row[4]) - 1.7) < 1e-3                  # GHz(sum / 8)


def test_strip_residual_registers_are_not_the_compilers():
    """tools/strip_asm_check.py on the built object: the pipelined row-strip GEMM keeps its in-flight residual pieces in v[248:255] of the
    kernels that are limited to 248 allocatable VGPRs; any other instruction naming one of them (a copy of a register whose load has not
    landed: the failure mode measured in round 6, profiles/r06_strip_pipeline.txt) fails the build check.  Also on a synthetic listing."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("strip_asm_check", os.path.join(root, "tools", "strip_asm_check.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    good = """0000000000001000 <_Z24conv1x1_strip_res_kernelILi6ELi1ELi32ELi0ELi0EEv14ConvGemmParamsi>:
\tglobal_load_dwordx4 v[248:251], v[2:3], off                // 000000001000: DC5C8000 F87F0002
\tv_add_f32_e32 v1, v2, v3                                   // 000000001008: 02020702
\ts_waitcnt vmcnt(2)                                         // 00000000100C: BF8C0F72
\tv_lshlrev_b32_e32 v4, 16, v248                             // 000000001010: 2409F090
\tv_and_b32_e32 v5, 0xffff0000, v248                         // 000000001014: 260BF0FF FFFF0000
"""
    n, bad = chk.check(good)
    assert n == 1 and bad == []
    n, bad = chk.check(good + "\tv_mov_b64_e32 v[6:7], v[248:249]                           // 00000000101C: 7E0C71F8\n")
    assert n == 1 and len(bad) == 1 and "touches" in bad[0]
    n, bad = chk.check(good.replace("\ts_waitcnt vmcnt(2)", "\ts_nop 0"))
    assert len(bad) == 2 and all("without the wait" in b for b in bad)
    obj = os.path.join(root, "mm-diffusion_amd", "lib", "mmd_gemm.o")
    if os.path.exists(obj) and os.path.exists(os.path.join(chk.LLVM, "llvm-objdump")):
        n, bad = chk.check(chk.disassemble(obj))
        assert n == 12 and bad == [], bad[:3]


def test_preserve_rng_restores_the_device_generator(monkeypatch):
    """_hip.preserve_rng (around plan building: the tile autotuner draws N(0,1) scratch data from torch's default device generator, which a
    seeded sampling loop must not see - profiles/r05_rccl_world1_and_rng.txt): a no-op for CPU devices, and for a device generator the state
    saved on entry is put back on exit, also when the body raises.  The GPU half (seeded trajectories bitwise equal whether or not the plan
    was built after seeding) is tools/rccl_world1_check.py."""
    from mm_diffusion import _hip as H
    calls = []
    monkeypatch.setattr(torch.cuda, "get_rng_state", lambda dev: calls.append(("get", str(dev))) or "STATE")
    monkeypatch.setattr(torch.cuda, "set_rng_state", lambda st, dev: calls.append(("set", st, str(dev))))
    with H.preserve_rng("cpu"):
        pass
    assert calls == []
    with H.preserve_rng(torch.device("cuda", 0)):
        calls.append("body")
    assert calls == [("get", "cuda:0"), "body", ("set", "STATE", "cuda:0")]
    calls.clear()
    with pytest.raises(RuntimeError):
        with H.preserve_rng("cuda:0"):
            raise RuntimeError("plan building failed")
    assert calls == [("get", "cuda:0"), ("set", "STATE", "cuda:0")]
    import inspect
    from mm_diffusion import image_unet, multimodal_unet
    assert "preserve_rng" in inspect.getsource(multimodal_unet.MultimodalUNet.engine)
    assert "preserve_rng" in inspect.getsource(image_unet)


def test_head_gemm_register_model():
    """tools/head_gemm_model.py: lane-level model of head_gemm_kernel and of the (hi, lo) weight image of ops.head_gemm_pack - fragment
    addresses in LDS, the activation fragment with the GroupNorm slice table, accumulator register -> (output, row) -> P[o][m], every
    element written exactly once - against (hi + lo) . act(x a + b) in float64 (exact: bf16 products)."""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "head_gemm_model.py")
    spec = importlib.util.spec_from_file_location("head_gemm_model", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.check(seed=3) < 1e-12


def test_attn_pipe_stream_generator():
    """tools/gen_attn_pipe.py: the committed instruction stream of attn_pipe_kernel<64> is what the generator writes today, and the
    generator's own replay holds - every register read still holds the temporary it names (linear-scan allocation), the hazard
    distances (v_exp -> use, v_cvt_pk / rescale -> MFMA operand) hold on the final order, no MFMA reads an LDS destination in front
    of its counted wait, 16 MFMAs / 24 LDS reads per iteration."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_attn_pipe", os.path.join(root, "tools", "gen_attn_pipe.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    text, order, nops, mf, top = mod.generate()
    assert open(mod.OUT).read() == text, "mmd_attn_pipe_body.inc is stale: run python tools/gen_attn_pipe.py"
    assert len(mf) == 16 and top <= 255 and nops <= 4
    assert min(b - a for a, b in zip(mf, mf[1:])) >= mod.MFMA_GAP
    # the two variants differ only in the score-set registers and the stage offsets
    a = text[text.index("#define ATTN_PIPE_ASM_A"):text.index("#define ATTN_PIPE_ASM_B")].count("v_mfma")
    b = text[text.index("#define ATTN_PIPE_ASM_B"):text.index("#define ATTN_PIPE_CLOBBERS")].count("v_mfma")
    assert a == b == 16
