"""CPU oracle: one training step of the reference's loop (diffusion/train_diffusion.py:201-266).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The reference trains with torch.autograd over diffusers'
UNet2DModel; here torch.autograd runs over the functional restatement of oracle/unet.py (same "parity unpinned" status
for the network itself), with the rest of the loop body as the reference writes it:

    noisy = scheduler.add_noise(images, noise, timesteps)                   :217
    loss = F.mse_loss(model(noisy, timesteps).sample, noise)                :218-219
    loss.backward(); Adam(lr=1e-4).step()                                   :230-233  (the GradScaler is an identity in fp32)

The reference runs the forward under torch.cuda.amp.autocast (fp16 matmuls); the oracle and the HIP path are fp32.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

from . import ddpm as oddpm
from . import unet as ounet

LR = 1e-4          # train_diffusion.py:62


def loss_and_grads(sd: Dict[str, torch.Tensor], images: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor,
                   dtype=torch.float32) -> Tuple[float, "OrderedDict[str, torch.Tensor]", torch.Tensor]:
    """(loss, {name: d loss / d parameter}, noise_pred) for one batch; dtype=float64 gives a rounding-free reference."""
    params = OrderedDict((k, v.detach().to(dtype).clone().requires_grad_(True)) for k, v in sd.items())
    sched = oddpm.DDPMSchedulerOracle()
    noisy = sched.add_noise(images.to(dtype), noise.to(dtype), timesteps)
    with torch.enable_grad():
        if dtype == torch.float64:
            pred = _forward64(params, noisy, timesteps)
        else:
            pred = ounet.unet_forward(params, noisy, timesteps)
        loss = F.mse_loss(pred, noise.to(dtype))
        grads = torch.autograd.grad(loss, list(params.values()))
    return float(loss.detach()), OrderedDict((k, g.detach()) for k, g in zip(params, grads)), pred.detach()


def _forward64(params, noisy, timesteps):
    """oracle/unet.py computes its sinusoid in fp32; for the float64 reference the embedding is rebuilt in float64 from
    the same fp32 frequency table so that only rounding, not the definition, differs."""
    orig = ounet.timestep_embedding

    def emb64(t, dim=ounet.BLOCK_OUT_CHANNELS[0]):
        e = t[:, None].double() * ounet.timestep_frequencies(dim)[None, :].double()
        return torch.cat([torch.cos(e), torch.sin(e)], dim=-1)

    ounet.timestep_embedding = emb64
    try:
        return ounet.unet_forward(params, noisy, timesteps)
    finally:
        ounet.timestep_embedding = orig


def adam_step(sd: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], state=None, lr: float = LR):
    """torch.optim.Adam(model.parameters(), lr=LR).step() (train_diffusion.py:203, :232) on copies of the tensors.
    Returns (new state dict, optimizer) -- pass the optimizer back in to take further steps."""
    if state is None:
        params = OrderedDict((k, torch.nn.Parameter(v.detach().clone())) for k, v in sd.items())
        opt = torch.optim.Adam(list(params.values()), lr=lr)
        state = (params, opt)
    params, opt = state
    for k, p in params.items():
        p.grad = grads[k].detach().clone().to(p.dtype)
    opt.step()
    return OrderedDict((k, p.detach().clone()) for k, p in params.items()), state
