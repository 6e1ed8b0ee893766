"""synt_isic_amd -- MI355X-native DDPM reverse-diffusion sampler.

A from-scratch gfx950 implementation of the one data-parallel hot path of
fims9000/SYNT_ISIC: the T-step loop of core/generator/image_generator.py:395-403
(``eps = model(x, t).sample; x = scheduler.step(eps, t, x).prev_sample``) behind
the same duck-typed call surface.  All arithmetic on the path runs in the
hand-written HIP library ``libsisic_hip.so`` (``synt_isic_amd/csrc``) reached
through the C ABI declared in ``include/sisic.h``; there is no CPU fallback --
importing the compute classes without the built library raises.
"""
from .arch import UNetConfig, unet_param_spec, unet_num_params  # noqa: F401

__all__ = ["UNetConfig", "unet_param_spec", "unet_num_params"]
__version__ = "0.1.0"
