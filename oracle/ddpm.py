"""CPU oracle: the DDPMScheduler the reference steps with.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates the published
``diffusers.DDPMScheduler`` algorithm (``diffusers>=0.21.0``, requirements.txt:6,
not vendored) for the configurations the reference constructs:

  * core/generator/model_manager.py:199-209 --
    ``DDPMScheduler(num_train_timesteps=1000, beta_schedule="squaredcos_cap_v2")``
    followed by ``set_timesteps(steps)``  (prediction_type="epsilon" explicit at
    core/generator/image_generator.py:292-296);
  * diffusion/diffusion_generator.py:123-128 -- ``beta_schedule="linear"``,
    beta in [1e-4, 0.02], no ``set_timesteps`` (all 1000 steps).

Defaults in play: variance_type="fixed_small",
clip_sample=True, clip_sample_range=1.0, timestep_spacing="leading",
steps_offset=0, thresholding=False.

Every scalar is a 0-dim fp32 torch tensor and the operation order follows the
published ``step`` so results are bit-identical to running it with torch on CPU.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch


def betas_for_alpha_bar(n: int, max_beta: float = 0.999) -> torch.Tensor:
    """``squaredcos_cap_v2``: python float64 per element, stored as fp32 (Appendix B)."""
    def alpha_bar(t: float) -> float:
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2

    betas = []
    for i in range(n):
        t1 = i / n
        t2 = (i + 1) / n
        betas.append(min(1 - alpha_bar(t2) / alpha_bar(t1), max_beta))
    return torch.tensor(betas, dtype=torch.float32)


@dataclass
class StepCoefficients:
    """The per-step scalars of ``DDPMScheduler.step`` (all fp32)."""
    sqrt_beta_prod_t: float      # (1 - abar_t) ** 0.5
    sqrt_alpha_prod_t: float     # abar_t ** 0.5
    pred_original_coeff: float   # abar_prev ** 0.5 * beta_cur / (1 - abar_t)
    current_sample_coeff: float  # alpha_cur ** 0.5 * (1 - abar_prev) / (1 - abar_t)
    sigma: float                 # clamp(variance, 1e-20) ** 0.5   (0.0 when t == 0)
    add_noise: bool              # t > 0


class DDPMSchedulerOracle:
    def __init__(self, num_train_timesteps: int = 1000, beta_schedule: str = "squaredcos_cap_v2",
                 beta_start: float = 1e-4, beta_end: float = 0.02, prediction_type: str = "epsilon",
                 clip_sample: bool = True, clip_sample_range: float = 1.0):
        if prediction_type != "epsilon":
            raise NotImplementedError("the reference only uses prediction_type='epsilon'")
        self.num_train_timesteps = num_train_timesteps
        if beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "squaredcos_cap_v2":
            self.betas = betas_for_alpha_bar(num_train_timesteps)
        else:
            raise NotImplementedError(beta_schedule)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.clip_sample = clip_sample
        self.clip_sample_range = clip_sample_range
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())

    def set_timesteps(self, num_inference_steps: int, device=None) -> None:
        """``timestep_spacing="leading"``: (arange(T) * (1000 // T)).round()[::-1] as int64."""
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError("num_inference_steps > num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        step_ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)

    def previous_timestep(self, t: int) -> int:
        n = self.num_inference_steps if self.num_inference_steps else self.num_train_timesteps
        return t - self.num_train_timesteps // n

    def coefficients(self, timestep) -> StepCoefficients:
        t = int(timestep)
        prev_t = self.previous_timestep(t)
        alpha_prod_t = self.alphas_cumprod[t]
        alpha_prod_t_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        beta_prod_t = 1 - alpha_prod_t
        beta_prod_t_prev = 1 - alpha_prod_t_prev
        current_alpha_t = alpha_prod_t / alpha_prod_t_prev
        current_beta_t = 1 - current_alpha_t
        pred_original_coeff = (alpha_prod_t_prev ** (0.5) * current_beta_t) / beta_prod_t
        current_sample_coeff = current_alpha_t ** (0.5) * beta_prod_t_prev / beta_prod_t
        sigma = 0.0
        if t > 0:
            variance = (1 - alpha_prod_t_prev) / (1 - alpha_prod_t) * current_beta_t
            variance = torch.clamp(variance, min=1e-20)
            sigma = float(variance ** 0.5)
        return StepCoefficients(float(beta_prod_t ** (0.5)), float(alpha_prod_t ** (0.5)),
                                float(pred_original_coeff), float(current_sample_coeff), sigma, t > 0)

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor,
             noise: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """Returns ``prev_sample``.  ``noise`` (same shape as model_output) overrides the RNG draw."""
        t = int(timestep)
        prev_t = self.previous_timestep(t)
        alpha_prod_t = self.alphas_cumprod[t]
        alpha_prod_t_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        beta_prod_t = 1 - alpha_prod_t
        beta_prod_t_prev = 1 - alpha_prod_t_prev
        current_alpha_t = alpha_prod_t / alpha_prod_t_prev
        current_beta_t = 1 - current_alpha_t

        pred_original_sample = (sample - beta_prod_t ** (0.5) * model_output) / alpha_prod_t ** (0.5)
        if self.clip_sample:
            pred_original_sample = pred_original_sample.clamp(-self.clip_sample_range, self.clip_sample_range)
        pred_original_sample_coeff = (alpha_prod_t_prev ** (0.5) * current_beta_t) / beta_prod_t
        current_sample_coeff = current_alpha_t ** (0.5) * beta_prod_t_prev / beta_prod_t
        pred_prev_sample = pred_original_sample_coeff * pred_original_sample + current_sample_coeff * sample

        if t > 0:
            if noise is None:
                noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype)
            variance = (1 - alpha_prod_t_prev) / (1 - alpha_prod_t) * current_beta_t
            variance = torch.clamp(variance, min=1e-20)
            pred_prev_sample = pred_prev_sample + (variance ** 0.5) * noise
        return pred_prev_sample

    def all_coefficients(self) -> List[StepCoefficients]:
        return [self.coefficients(t) for t in self.timesteps]


    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        """Published ``DDPMScheduler.add_noise`` (called at diffusion/train_diffusion.py:217): per-sample
        sqrt(abar_t) * x0 + sqrt(1 - abar_t) * noise, coefficients broadcast over the trailing dimensions."""
        acp = self.alphas_cumprod.to(dtype=original_samples.dtype)
        t = timesteps.to(torch.int64)
        sqrt_alpha_prod = (acp[t] ** 0.5).flatten()
        sqrt_one_minus = ((1 - acp[t]) ** 0.5).flatten()
        while sqrt_alpha_prod.dim() < original_samples.dim():
            sqrt_alpha_prod = sqrt_alpha_prod.unsqueeze(-1)
            sqrt_one_minus = sqrt_one_minus.unsqueeze(-1)
        return sqrt_alpha_prod * original_samples + sqrt_one_minus * noise
