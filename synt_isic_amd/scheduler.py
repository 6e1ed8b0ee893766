"""HipDDPMScheduler -- drop-in for the ``diffusers.DDPMScheduler`` the reference steps with.

Call surface mirrored:

    DDPMScheduler(num_train_timesteps=1000, beta_schedule="squaredcos_cap_v2")   model_manager.py:199-202
    DDPMScheduler(..., beta_schedule="linear", beta_start=1e-4, beta_end=0.02)    diffusion_generator.py:123-128
    scheduler.set_timesteps(steps)                                                model_manager.py:209
    for t in scheduler.timesteps: ...                                             image_generator.py:395
    latents = scheduler.step(noise_pred, t, latents).prev_sample                  image_generator.py:403

The beta / alphas_cumprod tables, the integer timestep grid and the per-step scalars
are host data built with the same fp32 torch operations the published algorithm uses
(SURVEY.md Appendix B); the elementwise update itself runs in the fused HIP kernel
``sisic_ddpm_step``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from types import SimpleNamespace
from typing import List, Optional, Tuple, Union

import numpy as np
import torch

from . import ops


@dataclass
class DDPMSchedulerOutput:
    """Mirror of ``diffusers.schedulers.scheduling_ddpm.DDPMSchedulerOutput``."""
    prev_sample: torch.Tensor
    pred_original_sample: Optional[torch.Tensor] = None


def _betas_for_alpha_bar(n: int, max_beta: float = 0.999) -> torch.Tensor:
    def alpha_bar(t):
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    return torch.tensor([min(1 - alpha_bar((i + 1) / n) / alpha_bar(i / n), max_beta) for i in range(n)],
                        dtype=torch.float32)


class HipDDPMScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02,
                 beta_schedule: str = "linear", variance_type: str = "fixed_small", clip_sample: bool = True,
                 prediction_type: str = "epsilon", clip_sample_range: float = 1.0,
                 timestep_spacing: str = "leading", steps_offset: int = 0, **unsupported):
        if unsupported:
            raise NotImplementedError(f"unsupported DDPMScheduler arguments: {sorted(unsupported)}")
        if prediction_type != "epsilon":
            raise NotImplementedError("only prediction_type='epsilon' (what the reference trains and samples with)")
        if variance_type != "fixed_small":
            raise NotImplementedError("only variance_type='fixed_small' (the diffusers default the reference uses)")
        if timestep_spacing != "leading":
            raise NotImplementedError("only timestep_spacing='leading' (the diffusers default the reference uses)")
        if beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "squaredcos_cap_v2":
            self.betas = _betas_for_alpha_bar(num_train_timesteps)
        else:
            raise NotImplementedError(f"beta_schedule '{beta_schedule}' is not used by the reference")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                                      beta_end=beta_end, beta_schedule=beta_schedule, variance_type=variance_type,
                                      clip_sample=clip_sample, prediction_type=prediction_type,
                                      clip_sample_range=clip_sample_range, timestep_spacing=timestep_spacing,
                                      steps_offset=steps_offset)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())

    def __len__(self) -> int:
        return self.config.num_train_timesteps

    def scale_model_input(self, sample: torch.Tensor, timestep=None) -> torch.Tensor:
        return sample

    def set_timesteps(self, num_inference_steps: int, device=None) -> None:
        n_train = self.config.num_train_timesteps
        if num_inference_steps > n_train:
            raise ValueError(f"num_inference_steps {num_inference_steps} > num_train_timesteps {n_train}")
        if num_inference_steps < 1:
            raise ValueError("num_inference_steps must be >= 1")
        self.num_inference_steps = num_inference_steps
        step_ratio = n_train // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
        ts += self.config.steps_offset
        # kept on the host: the loop only needs int(t); device= is accepted for API compatibility
        self.timesteps = torch.from_numpy(ts)

    def previous_timestep(self, timestep: int) -> int:
        n = self.num_inference_steps if self.num_inference_steps else self.config.num_train_timesteps
        return int(timestep) - self.config.num_train_timesteps // n

    def step_coefficients(self, timestep) -> Tuple[float, float, float, float, float]:
        """(sqrt(1-abar_t), sqrt(abar_t), c0, c1, sigma) as fp32 values, sigma = 0 at t == 0."""
        t = int(timestep)
        prev_t = self.previous_timestep(t)
        alpha_prod_t = self.alphas_cumprod[t]
        alpha_prod_t_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        beta_prod_t = 1 - alpha_prod_t
        beta_prod_t_prev = 1 - alpha_prod_t_prev
        current_alpha_t = alpha_prod_t / alpha_prod_t_prev
        current_beta_t = 1 - current_alpha_t
        c0 = (alpha_prod_t_prev ** (0.5) * current_beta_t) / beta_prod_t
        c1 = current_alpha_t ** (0.5) * beta_prod_t_prev / beta_prod_t
        sigma = 0.0
        if t > 0:
            variance = (1 - alpha_prod_t_prev) / (1 - alpha_prod_t) * current_beta_t
            variance = torch.clamp(variance, min=1e-20)
            sigma = float(variance ** 0.5)
        return (float(beta_prod_t ** (0.5)), float(alpha_prod_t ** (0.5)), float(c0), float(c1), sigma)

    def coefficient_table(self) -> torch.Tensor:
        """[T,5] fp32 host table for ``sisic_sample``."""
        return torch.tensor([self.step_coefficients(t) for t in self.timesteps], dtype=torch.float32)

    @torch.no_grad()
    def step(self, model_output: torch.Tensor, timestep: Union[int, torch.Tensor], sample: torch.Tensor,
             generator: Optional[torch.Generator] = None, return_dict: bool = True,
             variance_noise: Optional[torch.Tensor] = None):
        """prev_sample = DDPM ancestral update on the GPU.  Noise: ``variance_noise`` if given, else
        ``torch.randn`` on the sample's device with ``generator`` (the reference passes none)."""
        if model_output.device.type != "cuda":
            raise RuntimeError("HipDDPMScheduler.step runs on MI355X tensors only (no CPU path)")
        coef = self.step_coefficients(timestep)
        z = None
        if int(timestep) > 0:
            if variance_noise is not None:
                z = variance_noise.to(device=sample.device, dtype=torch.float32).contiguous()
            elif generator is not None and generator.device.type == "cpu":
                z = torch.randn(model_output.shape, generator=generator, dtype=torch.float32).to(sample.device)
            else:
                z = torch.randn(model_output.shape, generator=generator, device=sample.device, dtype=torch.float32)
        clip = self.config.clip_sample_range if self.config.clip_sample else 0.0
        prev = ops.ddpm_step(model_output.to(torch.float32).contiguous(), sample.to(torch.float32).contiguous(), z,
                             coef, clip)
        if not return_dict:
            return (prev,)
        return DDPMSchedulerOutput(prev_sample=prev)

    def add_noise_coefficients(self, timesteps: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(alphas_cumprod[t] ** 0.5, (1 - alphas_cumprod[t]) ** 0.5) as fp32 host rows, the scalars of ``add_noise``."""
        t = torch.as_tensor(timesteps).detach().to("cpu").to(torch.int64).reshape(-1)
        acp = self.alphas_cumprod[t]
        return (acp ** 0.5).contiguous(), ((1 - acp) ** 0.5).contiguous()

    @torch.no_grad()
    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        """``DDPMScheduler.add_noise`` (diffusion/train_diffusion.py:217):
        noisy = sqrt(abar_t) * x0 + sqrt(1 - abar_t) * noise with one timestep per sample, on the GPU (sisic_add_noise)."""
        if original_samples.device.type != "cuda":
            raise RuntimeError("HipDDPMScheduler.add_noise runs on MI355X tensors only (no CPU path)")
        from . import _lib
        from ._lib import check
        import ctypes as C
        x0 = original_samples.to(torch.float32).contiguous()
        nz = noise.to(device=x0.device, dtype=torch.float32).contiguous()
        B = x0.shape[0]
        a, c = self.add_noise_coefficients(timesteps)
        if a.numel() != B:
            raise ValueError(f"{a.numel()} timesteps for a batch of {B}")
        a, c = a.to(x0.device), c.to(x0.device)
        out = torch.empty_like(x0)
        check(_lib.load().sisic_add_noise(ops.context(x0.device), x0.data_ptr(), nz.data_ptr(), a.data_ptr(), c.data_ptr(),
                                          out.data_ptr(), B, x0[0].numel(),
                                          C.c_void_p(torch.cuda.current_stream(x0.device).cuda_stream)))
        return out
