"""The oracle reproduces its own committed fixtures (guards against drift of torch or of the oracle)."""
import os

import numpy as np
import torch

from oracle import sampler, unet
from synt_isic_amd.weights import state_dict_sha256


def test_synthetic_weight_fingerprint(anchors, synthetic_sd):
    assert state_dict_sha256(synthetic_sd) == anchors["synthetic_weights"]["sha256"]
    assert sum(v.numel() for v in synthetic_sd.values()) == anchors["unet_num_params"]


def test_unet_forward_golden(golden_dir, synthetic_sd):
    g = np.load(os.path.join(golden_dir, "unet_forward_b2_64.npz"))
    with torch.no_grad():
        y, inter = unet.unet_forward(synthetic_sd, torch.from_numpy(g["x"]), torch.from_numpy(g["t"]),
                                     return_intermediates=True)
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(inter["temb"].numpy(), g["temb"], rtol=0, atol=1e-5)
    assert y.shape == (2, 3, 64, 64)
    # timesteps are applied per sample: swapping them changes the result
    with torch.no_grad():
        y2 = unet.unet_forward(synthetic_sd, torch.from_numpy(g["x"]), torch.from_numpy(g["t"][::-1].copy()))
    assert not np.allclose(y2.numpy(), g["y"], atol=1e-3)


def test_sample_T50_golden(golden_dir, synthetic_sd):
    """BASELINE config 1: 1 image, 3x64x64, T=50, seed 0 on the CPU path."""
    g = np.load(os.path.join(golden_dir, "sample_T50_seed0_64.npz"))
    keep = tuple(int(s) for s in g["steps"])
    img, x0, traj = sampler.sample(synthetic_sd, [0], 50, (64, 64), return_trajectory=True, keep_steps=keep)
    assert img.shape == (1, 64, 64, 3) and img.dtype == np.uint8
    np.testing.assert_allclose(torch.stack(traj).numpy(), g["traj"], rtol=0, atol=2e-4)
    assert np.mean(np.abs(img.astype(np.int32) - g["image"].astype(np.int32)) <= 1) >= 0.999
    assert sampler.noise_hash(sampler.initial_noise(0, (1, 3, 64, 64))) == "ce480957dd270985"


def test_forward_is_batch_independent(synthetic_sd):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 3, 32, 32, generator=g)
    with torch.no_grad():
        full = unet.unet_forward(synthetic_sd, x, 100)
        one = unet.unet_forward(synthetic_sd, x[1:2], 100)
    torch.testing.assert_close(full[1:2], one, rtol=0, atol=1e-5)


def test_unet_forward_128_golden(golden_dir, synthetic_sd):
    """BASELINE config 4's resolution: attention over 1024 (32x32) and 256 (16x16) tokens."""
    g = np.load(os.path.join(golden_dir, "unet_forward_b1_128.npz"))
    with torch.no_grad():
        y = unet.unet_forward(synthetic_sd, torch.from_numpy(g["x"]), int(g["t"]))
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=0, atol=2e-5)


def test_sample_T1000_golden_prefix(golden_dir, synthetic_sd):
    """The 1000-step fixture on the CPU path: its first step always; its first 100 steps (to the fixture's second kept frame)
    with SISIC_SLOW_TESTS=1 -- five minutes on eight shared cores, and the whole chain is replayed by the GPU test
    (tests/test_gpu_sampler.py) against the same file."""
    from oracle import ddpm
    n_replay = 100 if os.environ.get("SISIC_SLOW_TESTS") == "1" else 1
    g = np.load(os.path.join(golden_dir, "sample_T1000_seed3_32.npz"))
    assert [int(s) for s in g["steps"]] == [0, 99, 499, 899, 999]
    sched = ddpm.DDPMSchedulerOracle()
    sched.set_timesteps(1000)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(1, 3, 32, 32, generator=gen)
    with torch.no_grad():
        for i, t in enumerate(sched.timesteps[:n_replay]):
            eps = unet.unet_forward(synthetic_sd, x, int(t))
            z = torch.randn(1, 3, 32, 32, generator=gen)
            x = sched.step(eps, int(t), x, noise=z)
            if i == 0:
                np.testing.assert_allclose(x.numpy(), g["traj"][0], rtol=0, atol=1e-5)
    if n_replay == 100:
        np.testing.assert_allclose(x.numpy(), g["traj"][1], rtol=0, atol=5e-4)
