"""The training loop of diffusion/train_diffusion.py:201-266 on the HIP kernels.

The reference's loop body, and what stands in for each object here (same names, arguments and call order, so that the loop
reads like the reference's):

    model = create_model().to(DEVICE)                               HipUNet2DModel(...).to("cuda")           :201
    scheduler = DDPMScheduler(1000, "squaredcos_cap_v2")            HipDDPMScheduler(...)                    :202
    optimizer = Adam(model.parameters(), lr=LR)                     HipAdam(model, lr=LR)                    :203
    scaler = amp.GradScaler()                                       HipGradScaler()                          :204
    model.train()                                                   model.train()                            :209
    noise = torch.randn_like(images)                                (torch RNG: plumbing)                    :215
    timesteps = torch.randint(0, TIMESTEPS, (B,), device=DEVICE)                                             :216
    noisy_images = scheduler.add_noise(images, noise, timesteps)    sisic_add_noise                          :217
    noise_pred = model(noisy_images, timesteps).sample              sisic_unet_train_forward                 :218
    loss = torch.nn.functional.mse_loss(noise_pred, noise)          mse_loss(noise_pred, noise)              :219
    optimizer.zero_grad(set_to_none=True)                           HipAdam.zero_grad                        :230
    scaler.scale(loss).backward()                                   sisic_mse_loss + sisic_unet_backward     :231
    scaler.step(optimizer); scaler.update()                         sisic_unet_optimizer_step                :232-233
    epoch_loss += loss.item()                                       HipLoss.item()                           :235

No torch.autograd anywhere: the backward pass is explicit HIP kernels (csrc/train.cpp).  Arithmetic is fp32; the
GradScaler protocol (scale, unscale, inf check, skip, growth/backoff) is implemented, the autocast-to-fp16 is not.
``train_step_fused`` runs the whole loop body in ONE C call (sisic_unet_train_step).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Iterable, Optional

import torch

from . import _lib
from ._lib import check
from .scheduler import HipDDPMScheduler
from .unet import HipUNet2DModel

LR = 1e-4            # train_diffusion.py:62
TIMESTEPS = 1000     # train_diffusion.py:60


def _stream(dev) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class HipLoss:
    """What ``F.mse_loss(noise_pred, noise)`` returns in the reference's loop: ``.item()`` and ``.backward()``."""

    def __init__(self, model: HipUNet2DModel, pred: torch.Tensor, target: torch.Tensor, scale: float = 1.0):
        self.model, self.pred, self.target, self.scale = model, pred, target, float(scale)
        self._loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        check(_lib.load().sisic_mse_loss(model.handle, pred.data_ptr(), target.data_ptr(), pred.numel(), 1.0,
                                         self._loss.data_ptr(), None, _stream(pred.device)))

    def item(self) -> float:
        return float(self._loss.item())

    def __float__(self) -> float:
        return self.item()

    def detach(self) -> torch.Tensor:
        return self._loss.detach().clone()

    def backward(self) -> None:
        """d(scale * loss)/d(parameters) into the model's gradient arena (the tape of the last forward is consumed)."""
        dpred = torch.empty_like(self.pred)
        lib = _lib.load()
        check(lib.sisic_mse_loss(self.model.handle, self.pred.data_ptr(), self.target.data_ptr(), self.pred.numel(),
                                 self.scale, None, dpred.data_ptr(), _stream(self.pred.device)))
        check(lib.sisic_unet_backward(self.model.handle, dpred.data_ptr(), _stream(self.pred.device)))


def mse_loss(noise_pred: torch.Tensor, noise: torch.Tensor) -> HipLoss:
    """``torch.nn.functional.mse_loss(noise_pred, noise)`` for the output of a training-mode ``HipUNet2DModel`` call."""
    model = getattr(noise_pred, "_sisic_model", None)
    if model is None:
        raise RuntimeError("mse_loss expects the .sample of a HipUNet2DModel called in training mode")
    return HipLoss(model, noise_pred.contiguous(), noise.to(device=noise_pred.device, dtype=torch.float32).contiguous())


class HipAdam:
    """``torch.optim.Adam(model.parameters(), lr)`` for a HipUNet2DModel: the state (m, v, step) lives in the library."""

    def __init__(self, model_or_params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 amsgrad: bool = False):
        if weight_decay != 0.0 or amsgrad:
            raise NotImplementedError("the reference uses plain Adam (train_diffusion.py:203)")
        model = model_or_params if isinstance(model_or_params, HipUNet2DModel) else getattr(model_or_params, "_sisic_model", None)
        if model is None:
            raise RuntimeError("HipAdam needs the HipUNet2DModel (or its .parameters())")
        self.model, self.lr, self.betas, self.eps = model, float(lr), (float(betas[0]), float(betas[1])), float(eps)
        model._ensure_training()

    def zero_grad(self, set_to_none: bool = True) -> None:
        check(_lib.load().sisic_unet_zero_grad(self.model.handle, _stream(self.model.device)))

    def step(self, inv_scale: float = 1.0, check_inf: bool = False) -> bool:
        """One Adam update; returns False when it was skipped because a gradient was inf/nan (GradScaler semantics)."""
        found = C.c_int(0)
        check(_lib.load().sisic_unet_optimizer_step(self.model.handle, self.lr, self.betas[0], self.betas[1], self.eps,
                                                    float(inv_scale), C.byref(found) if check_inf else None,
                                                    _stream(self.model.device)))
        self.model._params_stale = True
        return found.value == 0


class _ScaledLoss:
    def __init__(self, loss: HipLoss, scale: float):
        self.loss, self.scale = loss, scale

    def backward(self) -> None:
        self.loss.scale = self.scale
        self.loss.backward()


class HipGradScaler:
    """``torch.cuda.amp.GradScaler()`` (train_diffusion.py:204): defaults init_scale 65536, growth 2 every 2000 clean steps,
    backoff 0.5 after a step with inf/nan gradients (that step is skipped)."""

    def __init__(self, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5,
                 growth_interval: int = 2000, enabled: bool = True):
        self._scale = float(init_scale) if enabled else 1.0
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self.enabled = enabled
        self._growth_tracker = 0
        self._found_inf = False

    def get_scale(self) -> float:
        return self._scale

    def scale(self, loss: HipLoss) -> _ScaledLoss:
        return _ScaledLoss(loss, self._scale)

    def step(self, optimizer: HipAdam) -> bool:
        ok = optimizer.step(inv_scale=1.0 / self._scale, check_inf=self.enabled)
        self._found_inf = not ok
        return ok

    def update(self) -> None:
        if not self.enabled:
            return
        if self._found_inf:
            self._scale *= self.backoff_factor
            self._growth_tracker = 0
        else:
            self._growth_tracker += 1
            if self._growth_tracker == self.growth_interval:
                self._scale *= self.growth_factor
                self._growth_tracker = 0
        self._found_inf = False


def train_step_fused(model: HipUNet2DModel, scheduler: HipDDPMScheduler, images: torch.Tensor, noise: torch.Tensor,
                     timesteps: torch.Tensor, optimizer: HipAdam, scaler: Optional[HipGradScaler] = None):
    """The loop body of train_diffusion.py:215-233 in one library call (sisic_unet_train_step); returns
    (loss, step_taken)."""
    model._ensure_training()
    x0 = images.to(device=model.device, dtype=torch.float32).contiguous()
    nz = noise.to(device=model.device, dtype=torch.float32).contiguous()
    B, _, H, W = x0.shape
    t = torch.as_tensor(timesteps).detach().to("cpu").to(torch.int64).reshape(-1).contiguous()
    a, c = scheduler.add_noise_coefficients(t)
    loss, found = C.c_float(0.0), C.c_int(0)
    scale = scaler.get_scale() if scaler is not None else 1.0
    check(_lib.load().sisic_unet_train_step(model.handle, x0.data_ptr(), nz.data_ptr(), C.cast(t.data_ptr(), _lib.c_int64_p),
                                            C.cast(a.data_ptr(), _lib.c_float_p), C.cast(c.data_ptr(), _lib.c_float_p), B, H, W,
                                            optimizer.lr, optimizer.betas[0], optimizer.betas[1], optimizer.eps, float(scale),
                                            C.byref(loss), C.byref(found) if scaler is not None and scaler.enabled else None,
                                            _stream(model.device)))
    model._params_stale = True
    if scaler is not None:
        scaler._found_inf = bool(found.value)
        scaler.update()
    return loss.value, found.value == 0


def train_class(model: HipUNet2DModel, loader: Iterable[torch.Tensor], class_name: str, epochs: int = 50, lr: float = LR,
                checkpoint_dir: Optional[str] = None, fused: bool = True, generator: Optional[torch.Generator] = None,
                log: Optional[Callable[[str], None]] = print):
    """``train_class`` of train_diffusion.py:187-266 for one class: epochs over ``loader`` (batches of images in [-1,1],
    [B,3,H,W]), best-loss checkpoint ``unet_{class}_best.pth`` and a checkpoint every 5 epochs.  Returns the per-epoch
    average losses.  ``generator`` seeds noise / timestep draws (the reference uses the global RNG)."""
    dev = model.device
    scheduler = HipDDPMScheduler(num_train_timesteps=TIMESTEPS, beta_schedule="squaredcos_cap_v2")
    optimizer = HipAdam(model, lr=lr)
    scaler = HipGradScaler()
    best_loss = float("inf")
    history = []
    for epoch in range(epochs):
        model.train()
        epoch_loss, n_batches = 0.0, 0
        for images in loader:
            images = images.to(dev, non_blocking=True)
            noise = torch.randn(images.shape, generator=generator, device=generator.device if generator is not None else dev)
            timesteps = torch.randint(0, TIMESTEPS, (images.size(0),), generator=generator,
                                      device=generator.device if generator is not None else dev).long()
            if fused:
                value, _ = train_step_fused(model, scheduler, images, noise, timesteps, optimizer, scaler)
            else:
                noisy_images = scheduler.add_noise(images, noise, timesteps)
                noise_pred = model(noisy_images, timesteps).sample
                loss = mse_loss(noise_pred, noise)
                optimizer.zero_grad(set_to_none=True)
                scaler.scale(loss).backward()
                scaler.step(optimizer)
                scaler.update()
                value = loss.item()
            epoch_loss += value
            n_batches += 1
        avg_loss = epoch_loss / max(1, n_batches)
        history.append(avg_loss)
        if log:
            log(f"Loss: {avg_loss:.5f}")
        if checkpoint_dir:
            os.makedirs(checkpoint_dir, exist_ok=True)
            if avg_loss < best_loss:
                best_loss = avg_loss
                torch.save(model.state_dict(), os.path.join(checkpoint_dir, f"unet_{class_name}_best.pth"))
            if (epoch + 1) % 5 == 0:
                torch.save(model.state_dict(), os.path.join(checkpoint_dir, f"unet_{class_name}_epoch_{epoch + 1:02d}.pth"))
        elif avg_loss < best_loss:
            best_loss = avg_loss
    return history
