"""UNet2DModel forward parity through the drop-in object (libsisic_hip.so sisic_unet_forward) vs the oracle.

Stated tolerance: one forward <= 2e-4 max-abs on O(1) outputs.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
FWD_TOL = 2e-4          # the stated tolerance
FWD_GUARD = 4e-5        # regression guard: what the library measures is ~4e-6 (profiles/r03/test_errors.txt); a change that costs an
                        # order of magnitude of accuracy fails here long before the stated tolerance


def _log(err, what):
    if os.environ.get("SISIC_TEST_ERRLOG"):
        with open(os.environ["SISIC_TEST_ERRLOG"], "a") as f:
            f.write(f"{err:.3e}\t{FWD_TOL:.1e}\t{what}\n")


@pytest.fixture(scope="module")
def model(synthetic_sd):
    from synt_isic_amd.unet import HipUNet2DModel
    # exactly the reference's constructor call, model_manager.py:175-194
    m = HipUNet2DModel(
        sample_size=128, in_channels=3, out_channels=3, layers_per_block=2,
        block_out_channels=(64, 128, 256, 256),
        down_block_types=("DownBlock2D", "DownBlock2D", "AttnDownBlock2D", "DownBlock2D"),
        up_block_types=("UpBlock2D", "AttnUpBlock2D", "UpBlock2D", "UpBlock2D"),
        class_embed_type=None,
    )
    m.load_state_dict(synthetic_sd)           # model_manager.py:138-139: load first, move second
    m = m.to(torch.device(DEV))
    m.eval()
    return m


def test_module_surface(model):
    assert str(model.device).startswith("cuda")
    p = next(model.parameters())
    assert p.device.type == "cuda" and not p.requires_grad
    assert sum(q.numel() for q in model.parameters()) == 25_304_963
    assert model.training is False
    assert model.to(torch.device(DEV)) is model


def test_forward_matches_golden_and_oracle(model, golden_dir, synthetic_sd):
    from oracle import unet as ounet
    g = np.load(os.path.join(golden_dir, "unet_forward_b2_64.npz"))
    x, t = torch.from_numpy(g["x"]), torch.from_numpy(g["t"])
    out = model(x.to(DEV), t.to(DEV)).sample                      # per-sample int64[B] timesteps on the GPU
    assert out.shape == (2, 3, 64, 64) and out.device.type == "cuda" and out.dtype == torch.float32
    err = (out.cpu() - torch.from_numpy(g["y"])).abs().max().item()
    _log(err, "unet forward B2 64x64 vs golden")
    assert err <= FWD_TOL and err <= FWD_GUARD, f"forward vs golden: {err:.3e}"
    with torch.no_grad():
        ref = ounet.unet_forward(synthetic_sd, x, t)
    assert (out.cpu() - ref).abs().max().item() <= FWD_GUARD


def test_forward_128_matches_golden(model, golden_dir):
    """BASELINE config 4's resolution (3x128x128): attention over 1024 and 256 tokens, five 64-row tile rounds."""
    g = np.load(os.path.join(golden_dir, "unet_forward_b1_128.npz"))
    y = model(torch.from_numpy(g["x"]).to(DEV), int(g["t"])).sample.cpu().numpy()
    err = float(np.abs(y - g["y"]).max())
    _log(err, "unet forward B1 128x128 vs golden")
    assert err <= FWD_TOL and err <= FWD_GUARD, err


def test_timestep_argument_forms(model, synthetic_sd):
    """model(latents, t): t is a 0-dim int64 tensor from scheduler.timesteps (image_generator.py:395-400);
    XAI.py:805-807 passes t.unsqueeze(0); an int must work too."""
    from oracle import unet as ounet
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 32, 32, generator=g)
    with torch.no_grad():
        ref = ounet.unet_forward(synthetic_sd, x, 980)
    xd = x.to(DEV)
    forms = [980, torch.tensor(980), torch.tensor(980).unsqueeze(0), torch.tensor([980, 980]),
             torch.tensor(980, device=DEV), torch.tensor(980.0)]
    outs = [model(xd, t).sample.cpu() for t in forms]
    for o in outs:
        assert (o - ref).abs().max().item() <= FWD_TOL
        assert torch.equal(o, outs[0])
    assert isinstance(model(xd, 980, return_dict=False), tuple)
    with pytest.raises(ValueError):
        model(xd, torch.tensor([1, 2, 3]))
    with pytest.raises(ValueError):
        model(torch.zeros(1, 4, 32, 32, device=DEV), 1)


@pytest.mark.parametrize("B,H,W,t", [(1, 128, 128, 999), (1, 72, 40, 0), (3, 8, 8, 500), (1, 64, 64, 20),
                                     (6, 32, 32, 321),      # 8x8 level on the four-image K-split tiling, batch not a multiple of 4
                                     (5, 40, 24, 77),       # ragged tiles at every level
                                     (1, 256, 256, 7)])     # attention over 4096 and 1024 tokens
def test_forward_other_resolutions(model, synthetic_sd, B, H, W, t):
    """fully convolutional: any H, W divisible by 8 (128x128 is the reference's native size)."""
    from oracle import unet as ounet
    g = torch.Generator().manual_seed(H * 1000 + W)
    x = torch.randn(B, 3, H, W, generator=g)
    with torch.no_grad():
        ref = ounet.unet_forward(synthetic_sd, x, t)
    out = model(x.to(DEV), t).sample.cpu()
    err = (out - ref).abs().max().item()
    assert err <= FWD_TOL, f"{B}x{H}x{W} t={t}: {err:.3e}"


def test_forward_rejects_bad_resolution(model):
    from synt_isic_amd._lib import SisicError
    with pytest.raises(SisicError, match="multiples of 8"):
        model(torch.zeros(1, 3, 36, 36, device=DEV), 1)


def test_forward_is_deterministic_and_batch_independent(model):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(5, 3, 64, 64, generator=g).to(DEV)
    a = model(x, 400).sample
    b = model(x, 400).sample
    assert torch.equal(a, b)
    one = model(x[3:4].contiguous(), 400).sample
    assert torch.equal(one, a[3:4])
    mixed = model(x, torch.tensor([400, 1, 400, 400, 7])).sample
    assert torch.equal(mixed[3], a[3]) and not torch.equal(mixed[1], a[1])


def test_reload_weights_and_second_instance(synthetic_sd):
    from oracle import unet as ounet
    from synt_isic_amd.unet import HipUNet2DModel
    from synt_isic_amd.weights import synthetic_unet_state_dict
    sd2 = synthetic_unet_state_dict(seed=99)
    m = HipUNet2DModel().to(DEV)
    m.load_state_dict(sd2)                    # load AFTER the move must work as well
    m.eval()
    x = torch.randn(1, 3, 32, 32, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref2 = ounet.unet_forward(sd2, x, 50)
        ref1 = ounet.unet_forward(synthetic_sd, x, 50)
    assert (m(x.to(DEV), 50).sample.cpu() - ref2).abs().max().item() <= FWD_TOL
    m.load_state_dict(synthetic_sd)
    assert (m(x.to(DEV), 50).sample.cpu() - ref1).abs().max().item() <= FWD_TOL
    cpu = m.to("cpu")
    assert str(cpu.device) == "cpu"
    with pytest.raises(RuntimeError, match="MI355X only"):
        cpu(x, 50)


def test_groupnorm_from_conv_epilogues_matches_standalone_pass(model, synthetic_sd, monkeypatch):
    """the executor takes GroupNorm statistics from the producing convolutions' epilogues where it can
    (sisic_conv_args.stats_out + sisic_groupnorm_finalize); SISIC_FUSED_GN=0 forces the stand-alone statistics
    pass everywhere.  Both are the same network to fp32 rounding."""
    from synt_isic_amd.unet import HipUNet2DModel
    monkeypatch.setenv("SISIC_FUSED_GN", "0")
    plain = HipUNet2DModel()
    plain.load_state_dict(synthetic_sd)
    plain = plain.to(DEV).eval()
    monkeypatch.delenv("SISIC_FUSED_GN")
    for shape, t in (((2, 3, 64, 64), 321), ((1, 3, 40, 24), 7)):
        x = torch.randn(*shape, generator=torch.Generator().manual_seed(5)).to(DEV)
        a, b = model(x, t).sample, plain(x, t).sample
        assert (a - b).abs().max().item() <= 2e-5


def test_latency_mode(synthetic_sd, golden_dir):
    """sisic_unet_set_latency_mode: the single-image kernel choices (input channels of the Winograd convolutions K-split
    over workgroups, 64-pixel 1x1 tiles).  Same tolerance against the goldens as the default mode, batch-independent and
    deterministic within the mode, and within rounding of the default mode's result."""
    from synt_isic_amd.unet import HipUNet2DModel
    m = HipUNet2DModel().set_latency_mode(True)
    m.load_state_dict(synthetic_sd)
    m = m.to(DEV).eval()
    g = np.load(os.path.join(golden_dir, "unet_forward_b1_128.npz"))
    x128 = torch.from_numpy(g["x"]).to(DEV)
    y = m(x128, int(g["t"])).sample
    assert np.abs(y.cpu().numpy() - g["y"]).max() <= FWD_TOL
    assert torch.equal(y, m(x128, int(g["t"])).sample)
    xb = torch.cat([torch.randn(2, 3, 128, 128, generator=torch.Generator().manual_seed(3)).to(DEV), x128])
    assert torch.equal(m(xb, int(g["t"])).sample[2], y[0])                 # an image's bits do not depend on its batch
    g64 = np.load(os.path.join(golden_dir, "unet_forward_b2_64.npz"))
    y64 = m(torch.from_numpy(g64["x"]).to(DEV), torch.from_numpy(g64["t"]).to(DEV)).sample
    assert np.abs(y64.cpu().numpy() - g64["y"]).max() <= FWD_TOL
    d = HipUNet2DModel()
    d.load_state_dict(synthetic_sd)
    d = d.to(DEV).eval()
    assert (d(x128, int(g["t"])).sample - y).abs().max().item() <= 2e-5     # the two modes differ by rounding only
    # a non-square resolution with ragged tiles at every level (40x56 -> 20x28 -> 10x14 -> 5x7), against the oracle
    from oracle import unet as ounet
    xr = torch.randn(2, 3, 40, 56, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        ref = ounet.unet_forward(synthetic_sd, xr, torch.tensor([3, 977]))
    assert (m(xr.to(DEV), torch.tensor([3, 977])).sample.cpu() - ref).abs().max().item() <= FWD_TOL
    m.set_latency_mode(False)
    assert torch.equal(m(x128, int(g["t"])).sample, d(x128, int(g["t"])).sample)
