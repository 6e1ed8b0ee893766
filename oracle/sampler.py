"""CPU oracle: the T-step reverse-diffusion loop and its glue.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows
core/generator/image_generator.py:

  * seed policy           :586-592, 626-637   (md5 class offset, 31-bit seeds)
  * initial noise / hash  :369-389            (per-image torch.Generator, sha256[:16])
  * the loop              :395-403            (eps = model(x, t); x = step(eps, t, x))
  * trajectory capture    :406-407            (x.clone() after every step)
  * de-normalise          :441-447            (clamp((x+1)/2,0,1)*255 -> uint8 TRUNCATED, HWC)

Noise contract (a defined extension -- the reference
seeds only x_T and lets ``scheduler.step`` draw from the global RNG): image b
owns one CPU ``torch.Generator().manual_seed(seed_b)``; x_T[b] is drawn first,
then z_t[b] for every step with t > 0 in loop order.  Results therefore do not
depend on batch composition or on how images are sharded over GPUs.
"""
from __future__ import annotations

import hashlib
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .ddpm import DDPMSchedulerOracle
from .unet import unet_forward

ISIC_CLASSES = ("MEL", "NV", "BCC", "AKIEC", "BKL", "DF", "VASC")     # xai/XAI.py:196


def class_seed_offset(class_name: str) -> int:
    """image_generator.py:586-592."""
    h = hashlib.md5(class_name.encode("utf-8")).hexdigest()
    return int(h[:8], 16) & 0x7FFFFFFF


def image_seed(base_seed: int, class_name: str, index: int) -> int:
    """image_generator.py:626-631."""
    return (int(base_seed) + class_seed_offset(class_name) + index) & 0x7FFFFFFF


def initial_noise(seed: int, shape: Tuple[int, ...]) -> torch.Tensor:
    """image_generator.py:369-381 on device='cpu'."""
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return torch.randn(*shape, generator=g)


def noise_hash(x_T: torch.Tensor) -> str:
    """image_generator.py:383-389."""
    return hashlib.sha256(x_T.detach().to("cpu").numpy().tobytes()).hexdigest()[:16]


def draw_noise(seeds: Sequence[int], n_noise_steps: int, chw: Tuple[int, int, int]):
    """Per-image streams: returns x_T [B,C,H,W] and z [n_noise_steps,B,C,H,W]."""
    B = len(seeds)
    x_T = torch.empty((B,) + tuple(chw), dtype=torch.float32)
    z = torch.empty((n_noise_steps, B) + tuple(chw), dtype=torch.float32)
    for b, s in enumerate(seeds):
        g = torch.Generator(device="cpu")
        g.manual_seed(int(s))
        x_T[b] = torch.randn((1,) + tuple(chw), generator=g)[0]
        for i in range(n_noise_steps):
            z[i, b] = torch.randn((1,) + tuple(chw), generator=g)[0]
    return x_T, z


def denormalize_to_uint8(x: torch.Tensor) -> np.ndarray:
    """image_generator.py:441-447 batched: [B,3,H,W] in [-1,1] -> uint8 [B,H,W,3] (truncation)."""
    img = x.permute(0, 2, 3, 1)
    img = (img + 1) / 2
    img = torch.clamp(img, 0, 1)
    return (img.cpu().numpy() * 255).astype(np.uint8)


def denormalize_to_uint8_generate_test(x: torch.Tensor) -> np.ndarray:
    """diffusion/generate_test.py:94-97, batched: clamp(-1,1), (x+1)*0.5, HWC, *255, astype(uint8)."""
    image = x.clamp(-1, 1)
    image = (image + 1) * 0.5
    image = image.cpu().permute(0, 2, 3, 1).numpy()
    return (image * 255).astype(np.uint8)


def denormalize_to_uint8_diffusion_generator(x: torch.Tensor) -> np.ndarray:
    """diffusion/diffusion_generator.py:231-232: numpy fp32 ((images + 1) * 127.5).clip(0, 255).astype(uint8)."""
    images = x.permute(0, 2, 3, 1).cpu().numpy()
    return ((images + 1) * 127.5).clip(0, 255).astype(np.uint8)


def color_postprocess(image: np.ndarray, stats: Optional[dict]) -> np.ndarray:
    """image_generator.py:502-545 for one uint8 [H,W,3] image: per-channel mean/std matching towards the class
    statistics (``color_statistics.json`` entry ``{"rgb": {"mean": [...], "std": [...]}}``) with the scale clipped
    to [0.6, 1.4], blended 35 % into the original, clipped to [0,255] and truncated to uint8.  An absent class entry or
    an entry without rgb.mean leaves the image untouched."""
    if not stats or "rgb" not in stats or "mean" not in stats["rgb"]:
        return image
    target_mean = np.array(stats["rgb"].get("mean", [128, 128, 128]), dtype=np.float32)
    target_std = np.array(stats["rgb"].get("std", [50, 50, 50]), dtype=np.float32)
    cur_mean = np.mean(image, axis=(0, 1)).astype(np.float32)
    cur_std = np.std(image, axis=(0, 1)).astype(np.float32)
    scale = np.clip(target_std / np.maximum(cur_std, 1e-6), 0.6, 1.4)
    shifted = (image.astype(np.float32) - cur_mean) * scale + target_mean
    out = 0.35 * shifted + (1.0 - 0.35) * image.astype(np.float32)
    return np.clip(out, 0, 255).astype(np.uint8)


@torch.no_grad()
def sample(sd: Dict[str, torch.Tensor], seeds: Sequence[int], T: int, size: Tuple[int, int] = (64, 64),
           beta_schedule: str = "squaredcos_cap_v2", return_trajectory: bool = False,
           keep_steps: Optional[Sequence[int]] = None,
           x_T: Optional[torch.Tensor] = None, z: Optional[torch.Tensor] = None):
    """Run the loop for a batch of per-image seeds.  Returns (uint8 images, final latents, trajectory)."""
    H, W = size
    sched = DDPMSchedulerOracle(beta_schedule=beta_schedule)
    sched.set_timesteps(T)
    ts = [int(t) for t in sched.timesteps]
    n_noise = sum(1 for t in ts if t > 0)
    if x_T is None:
        x_T, z = draw_noise(seeds, n_noise, (3, H, W))
    x = x_T.clone()
    traj: List[torch.Tensor] = []
    zi = 0
    for step_idx, t in enumerate(ts):
        eps = unet_forward(sd, x, t)
        noise = None
        if t > 0:
            noise = z[zi]
            zi += 1
        x = sched.step(eps, t, x, noise=noise)
        if return_trajectory and (keep_steps is None or step_idx in keep_steps):
            traj.append(x.clone())
    return denormalize_to_uint8(x), x, traj
