"""Generates the committed golden fixtures under tests/golden/ from the CPU oracle.

Run from the repo root:  python tests/golden/make_golden.py

The reference itself cannot produce vectors here (its hot-path modules need ``diffusers``,
which is absent offline -- SURVEY.md section 8c), so these fixtures pin the ORACLE against
drift (torch version, refactors) and give the GPU tests a CPU-independent target:

  anchors.json              reference-derived known answers: class seed offsets
                            (image_generator.py:586-592), noise_hash of x_T (:383-389), integer
                            timestep grids, schedule-table endpoints, parameter
                            count (cache_metadata.json sizes), synthetic-weight fingerprint
  unet_forward_b2_64.npz    one UNet forward, B=2, 3x64x64, per-sample timesteps
  sample_T50_seed0_64.npz   BASELINE config 1: B=1, 3x64x64, T=50, seed 0: x after steps
                            0, 1, 25, 49 + the final uint8 image
  sample_T1000_seed3_32.npz the full 1000-step chain of one 3x32x32 image (seed 3): x after steps
                            0, 99, 499, 899, 999 + the final uint8 image (error accumulation over T=1000)
  unet_forward_b1_128.npz   one UNet forward at 3x128x128 (BASELINE config 4's resolution: attention
                            over 1024 and 256 tokens)
  sample_T1000_seed0_64.npz   (--long) the full 1000-step chain at BASELINE config 2's own resolution: B=1,
                            3x64x64, seed 0 (image_generator.py:395-403 loop), x after steps 0, 99, 499, 899, 999
                            + the final uint8 image
  sample_T1000_seed5_128.npz  (--long) the same at config 4's resolution, 3x128x128, seed 5

``--long`` writes ONLY the two full-length chains (minutes of CPU time each); without it only the
short fixtures are (re)written.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ddpm, sampler, unet  # noqa: E402
from synt_isic_amd.weights import DEFAULT_WEIGHT_SEED, state_dict_sha256, synthetic_unet_state_dict  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def long_chains(sd):
    """Full T=1000 chains at the headline resolutions (error accumulation at 64x64 and 128x128)."""
    keep = (0, 99, 499, 899, 999)
    for name, seed, size in (("sample_T1000_seed0_64.npz", 0, (64, 64)), ("sample_T1000_seed5_128.npz", 5, (128, 128))):
        img, x0, traj = sampler.sample(sd, [seed], 1000, size, return_trajectory=True, keep_steps=keep)
        np.savez_compressed(os.path.join(OUT, name), steps=np.array(keep), traj=torch.stack(traj).numpy(), image=img,
                            final=x0.numpy(), seed=np.array(seed))
        print(name, os.path.getsize(os.path.join(OUT, name)), flush=True)


def main():
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    sd = synthetic_unet_state_dict(DEFAULT_WEIGHT_SEED)
    if "--long" in sys.argv[1:]:
        long_chains(sd)
        return

    cos = ddpm.DDPMSchedulerOracle(beta_schedule="squaredcos_cap_v2")
    lin = ddpm.DDPMSchedulerOracle(beta_schedule="linear")
    cos50 = ddpm.DDPMSchedulerOracle(); cos50.set_timesteps(50)
    cos1000 = ddpm.DDPMSchedulerOracle(); cos1000.set_timesteps(1000)
    anchors = {
        "class_seed_offsets": {c: sampler.class_seed_offset(c) for c in sampler.ISIC_CLASSES},
        "noise_hash": {
            "seed0_1x3x128x128": sampler.noise_hash(sampler.initial_noise(0, (1, 3, 128, 128))),
            "seed0_1x3x64x64": sampler.noise_hash(sampler.initial_noise(0, (1, 3, 64, 64))),
            "seed42_1x3x128x128": sampler.noise_hash(sampler.initial_noise(42, (1, 3, 128, 128))),
            "seed42_1x3x64x64": sampler.noise_hash(sampler.initial_noise(42, (1, 3, 64, 64))),
        },
        "x_T_seed0_first3": [float(v) for v in sampler.initial_noise(0, (1, 3, 128, 128))[0, 0, 0, :3]],
        "timesteps_T50": [int(t) for t in cos50.timesteps],
        "timesteps_T1000_head_tail": [int(t) for t in cos1000.timesteps[:3]] + [int(t) for t in cos1000.timesteps[-3:]],
        "cosine_betas": {"0": float(cos.betas[0]), "999": float(cos.betas[999])},
        "cosine_alphas_cumprod": {str(i): float(cos.alphas_cumprod[i]) for i in (0, 20, 500, 980, 999)},
        "linear_alphas_cumprod": {str(i): float(lin.alphas_cumprod[i]) for i in (0, 500, 999)},
        "unet_num_params": unet.num_params(),
        "unet_num_tensors": len(unet.param_spec()),
        "synthetic_weights": {"seed": DEFAULT_WEIGHT_SEED, "sha256": state_dict_sha256(sd)},
        "torch_version": torch.__version__,
    }
    with open(os.path.join(OUT, "anchors.json"), "w") as f:
        json.dump(anchors, f, indent=1, sort_keys=True)

    # one forward, per-sample timesteps
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 3, 64, 64, generator=g)
    t = torch.tensor([500, 37], dtype=torch.int64)
    with torch.no_grad():
        y, inter = unet.unet_forward(sd, x, t, return_intermediates=True)
    np.savez_compressed(os.path.join(OUT, "unet_forward_b2_64.npz"), x=x.numpy(), t=t.numpy(), y=y.numpy(),
                        temb=inter["temb"].numpy(),
                        inter_absmean=np.array([float(inter[k].abs().mean()) for k in sorted(inter)]),
                        inter_names=np.array(sorted(inter)))

    # BASELINE config 1
    keep = (0, 1, 25, 49)
    img, x0, traj = sampler.sample(sd, [0], 50, (64, 64), return_trajectory=True, keep_steps=keep)
    np.savez_compressed(os.path.join(OUT, "sample_T50_seed0_64.npz"), steps=np.array(keep),
                        traj=torch.stack(traj).numpy(), image=img, final=x0.numpy())
    # the full T=1000 chain at a size the oracle finishes in seconds
    keep = (0, 99, 499, 899, 999)
    img, x0, traj = sampler.sample(sd, [3], 1000, (32, 32), return_trajectory=True, keep_steps=keep)
    np.savez_compressed(os.path.join(OUT, "sample_T1000_seed3_32.npz"), steps=np.array(keep),
                        traj=torch.stack(traj).numpy(), image=img, final=x0.numpy())

    # BASELINE config 4's resolution
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 3, 128, 128, generator=g)
    with torch.no_grad():
        y = unet.unet_forward(sd, x, 321)
    np.savez_compressed(os.path.join(OUT, "unet_forward_b1_128.npz"), x=x.numpy(), t=np.array(321), y=y.numpy())
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
