"""MI355X-native MM-Diffusion denoising hot path behind the reference's `mm_diffusion` module names."""
