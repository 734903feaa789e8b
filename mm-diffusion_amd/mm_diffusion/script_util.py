"""Factories of the image super-resolution stage (reference mm_diffusion/script_util.py:10-24,70-106,326-472), imported
by multimodal_sample_sr.py:20-26 as `image_sr_model_and_diffusion_defaults`, `image_sr_create_model_and_diffusion`,
`args_to_dict`, `add_dict_to_argparser`.  Same flag names and defaults."""
import inspect

from . import gaussian_diffusion as gd
from .image_unet import ImageSuperResModel
from .multimodal_script_util import add_dict_to_argparser, args_to_dict, str2bool  # noqa: F401
from .respace import SpacedDiffusion, space_timesteps

NUM_CLASSES = 1000


def diffusion_defaults():
    return dict(learn_sigma=False, diffusion_steps=1000, noise_schedule="linear", timestep_respacing="", use_kl=False, predict_xstart=False,
                rescale_timesteps=False, rescale_learned_sigmas=False)


def image_sr_model_defaults():
    return dict(sr_num_channels=128, sr_num_res_blocks=2, sr_num_heads=4, sr_num_heads_upsample=-1, sr_num_head_channels=-1,
                sr_attention_resolutions="16,8", sr_channel_mult="", sr_dropout=0.0, sr_class_cond=False, use_checkpoint=False,
                sr_use_scale_shift_norm=True, sr_resblock_updown=False, use_fp16=False, sr_use_new_attention_order=False)


def image_sr_model_and_diffusion_defaults():
    res = image_sr_model_defaults()
    res.update(diffusion_defaults())
    res["sr_learn_sigma"] = True
    res["large_size"] = 256
    res["small_size"] = 128
    res["sr_diffusion_steps"] = 1000
    res["sr_timestep_respacing"] = ""
    res.pop("diffusion_steps")
    res.pop("timestep_respacing")
    arg_names = inspect.getfullargspec(image_sr_create_model_and_diffusion)[0]
    for k in res.copy().keys():
        if k not in arg_names:
            del res[k]
    return res


def image_sr_create_model_and_diffusion(large_size, small_size, sr_class_cond, sr_learn_sigma, sr_num_channels, sr_num_res_blocks, sr_num_heads,
                                        sr_num_head_channels, sr_num_heads_upsample, sr_attention_resolutions, sr_dropout, sr_diffusion_steps,
                                        noise_schedule, sr_timestep_respacing, use_kl, predict_xstart, rescale_timesteps, rescale_learned_sigmas,
                                        use_checkpoint, sr_use_scale_shift_norm, sr_resblock_updown, use_fp16):
    model = image_sr_create_model(large_size, small_size, sr_num_channels, sr_num_res_blocks, learn_sigma=sr_learn_sigma, class_cond=sr_class_cond,
                                  use_checkpoint=use_checkpoint, attention_resolutions=sr_attention_resolutions, num_heads=sr_num_heads,
                                  num_head_channels=sr_num_head_channels, num_heads_upsample=sr_num_heads_upsample,
                                  use_scale_shift_norm=sr_use_scale_shift_norm, dropout=sr_dropout, resblock_updown=sr_resblock_updown,
                                  use_fp16=use_fp16)
    diffusion = create_gaussian_diffusion(diffusion_steps=sr_diffusion_steps, learn_sigma=sr_learn_sigma, noise_schedule=noise_schedule, use_kl=use_kl,
                                          predict_xstart=predict_xstart, rescale_timesteps=rescale_timesteps,
                                          rescale_learned_sigmas=rescale_learned_sigmas, timestep_respacing=sr_timestep_respacing)
    return model, diffusion


def image_sr_create_model(large_size, small_size, num_channels, num_res_blocks, learn_sigma, class_cond, use_checkpoint, attention_resolutions,
                          num_heads, num_head_channels, num_heads_upsample, use_scale_shift_norm, dropout, resblock_updown, use_fp16):
    _ = small_size
    if large_size in (512, 256):
        channel_mult = (1, 1, 2, 2, 4, 4)
    elif large_size == 64:
        channel_mult = (1, 2, 3, 4)
    else:
        raise ValueError(f"unsupported large size: {large_size}")
    attention_ds = [int(res) for res in attention_resolutions.split(",") if res != ""]
    return ImageSuperResModel(image_size=large_size, in_channels=3, model_channels=num_channels, out_channels=(3 if not learn_sigma else 6),
                              num_res_blocks=num_res_blocks, attention_resolutions=tuple(attention_ds), dropout=dropout, channel_mult=channel_mult,
                              num_classes=(NUM_CLASSES if class_cond else None), use_checkpoint=use_checkpoint, num_heads=num_heads,
                              num_head_channels=num_head_channels, num_heads_upsample=num_heads_upsample,
                              use_scale_shift_norm=use_scale_shift_norm, resblock_updown=resblock_updown, use_fp16=use_fp16)


def create_gaussian_diffusion(*, diffusion_steps=1000, learn_sigma=False, sigma_small=False, noise_schedule="linear", use_kl=False,
                              predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=False, timestep_respacing=""):
    betas = gd.get_named_beta_schedule(noise_schedule, diffusion_steps)
    if use_kl:
        loss_type = gd.LossType.RESCALED_KL
    elif rescale_learned_sigmas:
        loss_type = gd.LossType.RESCALED_MSE
    else:
        loss_type = gd.LossType.MSE
    if not timestep_respacing:
        timestep_respacing = [diffusion_steps]
    return SpacedDiffusion(
        use_timesteps=space_timesteps(diffusion_steps, timestep_respacing), betas=betas,
        model_mean_type=(gd.ModelMeanType.EPSILON if not predict_xstart else gd.ModelMeanType.START_X),
        model_var_type=((gd.ModelVarType.FIXED_LARGE if not sigma_small else gd.ModelVarType.FIXED_SMALL) if not learn_sigma
                        else gd.ModelVarType.LEARNED_RANGE),
        loss_type=loss_type, rescale_timesteps=rescale_timesteps)
