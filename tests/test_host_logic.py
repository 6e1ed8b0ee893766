"""Host-side logic of synt_isic_amd (no GPU): schedule tables, seed policy, noise streams,
state-dict handling, and the 'no CPU fallback' rule."""
import pytest
import torch

from oracle import ddpm as oddpm
from oracle import sampler as osampler
from oracle import unet as ounet
from synt_isic_amd import arch, sampler, scheduler, unet, weights


def test_param_spec_matches_oracle():
    assert list(arch.unet_param_spec().items()) == list(ounet.param_spec().items())
    assert arch.unet_num_params() == 25_304_963


@pytest.mark.parametrize("schedule", ["squaredcos_cap_v2", "linear"])
def test_scheduler_tables_bit_exact(schedule):
    a = scheduler.HipDDPMScheduler(num_train_timesteps=1000, beta_schedule=schedule)
    b = oddpm.DDPMSchedulerOracle(beta_schedule=schedule)
    assert torch.equal(a.betas, b.betas)
    assert torch.equal(a.alphas_cumprod, b.alphas_cumprod)
    assert torch.equal(a.timesteps, b.timesteps)
    for T in (1000, 50, 7, 1):
        a.set_timesteps(T)
        b.set_timesteps(T)
        assert a.timesteps.dtype == torch.int64
        assert torch.equal(a.timesteps, b.timesteps)
        for t in a.timesteps[:: max(1, T // 10)].tolist() + [int(a.timesteps[-1])]:
            ca = a.step_coefficients(t)
            cb = b.coefficients(t)
            assert ca == (cb.sqrt_beta_prod_t, cb.sqrt_alpha_prod_t, cb.pred_original_coeff,
                          cb.current_sample_coeff, cb.sigma)
    tab = a.coefficient_table()
    assert tab.shape == (1, 5) and tab.dtype == torch.float32


def test_scheduler_reference_constructor_forms():
    # model_manager.py:199-202
    s = scheduler.HipDDPMScheduler(num_train_timesteps=1000, beta_schedule="squaredcos_cap_v2")
    s.set_timesteps(50)
    assert [int(t) for t in s.timesteps][:3] == [980, 960, 940]
    assert float(s.timesteps[0]) == 980.0 and int(s.timesteps[-1]) == 0      # XAI.py:744,764 use float()/int()
    # image_generator.py:292-296
    scheduler.HipDDPMScheduler(num_train_timesteps=1000, beta_schedule="squaredcos_cap_v2", prediction_type="epsilon")
    # diffusion_generator.py:123-128
    lin = scheduler.HipDDPMScheduler(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear")
    assert len(lin) == 1000 and len(lin.timesteps) == 1000 and int(lin.timesteps[0]) == 999
    with pytest.raises(NotImplementedError):
        scheduler.HipDDPMScheduler(prediction_type="v_prediction")
    with pytest.raises(ValueError):
        s.set_timesteps(1001)


def test_scheduler_step_refuses_cpu():
    s = scheduler.HipDDPMScheduler(beta_schedule="squaredcos_cap_v2")
    x = torch.zeros(1, 3, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU path"):
        s.step(x, 10, x)


def test_seed_policy_and_noise_hash():
    for c in sampler.ISIC_CLASSES:
        assert sampler.class_seed_offset(c) == osampler.class_seed_offset(c)
    assert sampler.ISIC_CLASSES == osampler.ISIC_CLASSES
    assert sampler.image_seed(42, "NV", 5) == osampler.image_seed(42, "NV", 5) == (42 + 1396962837 + 5) & 0x7FFFFFFF
    x = osampler.initial_noise(0, (1, 3, 64, 64))
    assert sampler.noise_hash(x) == "ce480957dd270985"


def test_noise_streams_match_oracle_and_are_batch_independent():
    seeds = [0, 7, 123456]
    x1, z1 = sampler.draw_noise(seeds, 5, (3, 16, 16))
    x2, z2 = osampler.draw_noise(seeds, 5, (3, 16, 16))
    assert torch.equal(x1, x2) and torch.equal(z1, z2)
    xs, zs = sampler.draw_noise([7], 5, (3, 16, 16))
    assert torch.equal(xs[0], x1[1]) and torch.equal(zs[:, 0], z1[:, 1])
    assert torch.equal(x1[0:1], osampler.initial_noise(0, (1, 3, 16, 16)))


def test_state_dict_strictness_and_legacy_keys():
    sd = weights.synthetic_unet_state_dict()
    m = unet.HipUNet2DModel()          # reference's constructor defaults == model_manager.py:175-194
    m.load_state_dict(sd)
    assert sum(p.numel() for p in m.parameters()) == 25_304_963
    assert str(m.device) == "cpu" and m.training is True
    assert m.eval() is m and m.training is False
    assert all(not p.requires_grad for p in m.parameters())
    # strict: a missing key or a wrong shape is an error
    bad = dict(sd); bad.pop("conv_in.bias")
    with pytest.raises(RuntimeError, match="missing"):
        unet.HipUNet2DModel().load_state_dict(bad)
    bad = dict(sd); bad["extra.weight"] = torch.zeros(1)
    with pytest.raises(RuntimeError, match="unexpected"):
        unet.HipUNet2DModel().load_state_dict(bad)
    bad = dict(sd); bad["conv_in.weight"] = torch.zeros(64, 3, 1, 1)
    with pytest.raises(RuntimeError, match="size mismatch"):
        unet.HipUNet2DModel().load_state_dict(bad)
    # checkpoints from older diffusers spell the attention projections query/key/value/proj_attn
    legacy = {}
    for k, v in sd.items():
        for new, old in (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn")):
            if ".attentions." in k and f".{new}." in k:
                k = k.replace(f".{new}.", f".{old}.")
        legacy[k] = v
    assert any(".query." in k for k in legacy)
    m2 = unet.HipUNet2DModel()
    m2.load_state_dict(legacy)
    assert list(m2.state_dict()) == list(sd)


def test_no_cpu_fallback():
    m = unet.HipUNet2DModel()
    m.load_state_dict(weights.synthetic_unet_state_dict())
    with pytest.raises(RuntimeError, match="MI355X only"):
        m(torch.zeros(1, 3, 64, 64), 10)
    # training mode exists but, like everything else, only on the GPU: no optimizer on a CPU model
    assert m.train() is m and m.training and m.eval().training is False
    from synt_isic_amd.train import HipAdam
    with pytest.raises(RuntimeError, match="MI355X only"):
        HipAdam(m.parameters(), lr=1e-4)
    with pytest.raises(NotImplementedError):
        unet.HipUNet2DModel(class_embed_type="timestep")


def test_frequency_table_matches_oracle():
    assert torch.equal(unet.timestep_frequencies(64), ounet.timestep_frequencies(64))


def test_classifier_spec_and_no_cpu_fallback():
    from oracle import resnet18 as ores
    from synt_isic_amd.classifier import HipMelanomaClassifier
    assert list(arch.resnet18_param_spec(7).items()) == list(ores.param_spec(7).items())
    assert ores.num_trainable_params(7) == 11_180_103
    assert len(arch.resnet18_param_spec(7)) == 102
    sd = weights.synthetic_resnet18_state_dict()
    m = HipMelanomaClassifier(num_classes=7, pretrained=False)
    m.load_state_dict(sd)
    assert sum(p.numel() for p in m.parameters()) == 11_180_103
    with pytest.raises(RuntimeError, match="MI355X only"):
        m(torch.zeros(1, 3, 64, 64))
    with pytest.raises(NotImplementedError):
        HipMelanomaClassifier(pretrained=True)               # a network download in the reference (XAI.py:389)
    # non-strict load keeps only name+shape matches (XAI.py:518-527): an 8-way fc is skipped, the rest is taken
    other = dict(weights.synthetic_resnet18_state_dict(seed=1, num_classes=8))
    m.load_state_dict(other, strict=False)
    got = m.state_dict()
    assert torch.equal(got["model.fc.weight"], sd["model.fc.weight"])
    assert torch.equal(got["model.conv1.weight"], other["model.conv1.weight"])
    with pytest.raises(RuntimeError):
        m.load_state_dict(other, strict=True)


def test_oracle_classifier_matches_torch_modules():
    """The functional restatement equals the same network assembled from torch.nn modules (BasicBlock form)."""
    import torch.nn as nn
    import torch.nn.functional as F
    from oracle import resnet18 as ores
    sd = weights.synthetic_resnet18_state_dict()

    def conv_bn(prefix_c, prefix_b, cin, cout, k, stride):
        c = nn.Conv2d(cin, cout, k, stride, k // 2, bias=False)
        b = nn.BatchNorm2d(cout)
        c.weight.data = sd[prefix_c + ".weight"]
        b.weight.data, b.bias.data = sd[prefix_b + ".weight"], sd[prefix_b + ".bias"]
        b.running_mean, b.running_var = sd[prefix_b + ".running_mean"], sd[prefix_b + ".running_var"]
        return nn.Sequential(c, b).eval()

    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        h = ores.preprocess_for_classifier(x)
        h = F.max_pool2d(F.relu(conv_bn("model.conv1", "model.bn1", 3, 64, 7, 2)(h)), 3, 2, 1)
        cin = 64
        for l, width in enumerate((64, 128, 256, 512)):
            for j in range(2):
                stride = 2 if (l > 0 and j == 0) else 1
                base = f"model.layer{l + 1}.{j}"
                out = F.relu(conv_bn(base + ".conv1", base + ".bn1", cin, width, 3, stride)(h))
                out = conv_bn(base + ".conv2", base + ".bn2", width, width, 3, 1)(out)
                idn = h if (stride == 1 and cin == width) else conv_bn(base + ".downsample.0", base + ".downsample.1",
                                                                     cin, width, 1, stride)(h)
                h = F.relu(out + idn)
                cin = width
        logits = F.linear(F.adaptive_avg_pool2d(h, 1).flatten(1), sd["model.fc.weight"], sd["model.fc.bias"])
        ref = ores.classifier_forward(sd, x)
    torch.testing.assert_close(logits, ref, rtol=1e-5, atol=1e-5)
    assert ref.shape == (2, 7) and torch.isfinite(ref).all()


def test_oracle_classifier_matches_transformers_resnet():
    """Independent third-party pin of oracle/resnet18.py (xai/XAI.py:389-394 builds torchvision's resnet18, which is not
    installed here): the ResNet that ships with the installed `transformers` package, configured as ResNet-18
    (layer_type="basic", depths 2-2-2-2, widths 64..512, 7x7/2 stem + 3x3/2 max-pool, fc -> 7), written by other people
    from the same paper, gives the same logits for the same weights -- stem, stride placement (first 3x3 of a stage),
    1x1/2 projection shortcuts, BatchNorm in eval mode, global average pool and the classifier head all agree."""
    transformers = pytest.importorskip("transformers")
    from transformers import ResNetConfig, ResNetForImageClassification
    from oracle import resnet18 as ores
    sd = weights.synthetic_resnet18_state_dict()
    cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[64, 128, 256, 512], depths=[2, 2, 2, 2],
                       layer_type="basic", hidden_act="relu", downsample_in_first_stage=False, num_labels=7)
    hf = ResNetForImageClassification(cfg).eval()

    def bn_map(src, dst, out):
        for s in ("weight", "bias", "running_mean", "running_var"):
            out[f"{dst}.normalization.{s}"] = sd[f"{src}.{s}"]

    m = {"resnet.embedder.embedder.convolution.weight": sd["model.conv1.weight"]}
    bn_map("model.bn1", "resnet.embedder.embedder", m)
    for l in range(4):
        for j in range(2):
            src, dst = f"model.layer{l + 1}.{j}", f"resnet.encoder.stages.{l}.layers.{j}"
            for c in (1, 2):
                m[f"{dst}.layer.{c - 1}.convolution.weight"] = sd[f"{src}.conv{c}.weight"]
                bn_map(f"{src}.bn{c}", f"{dst}.layer.{c - 1}", m)
            if f"{src}.downsample.0.weight" in sd:
                m[f"{dst}.shortcut.convolution.weight"] = sd[f"{src}.downsample.0.weight"]
                bn_map(f"{src}.downsample.1", f"{dst}.shortcut", m)
    m["classifier.1.weight"], m["classifier.1.bias"] = sd["model.fc.weight"], sd["model.fc.bias"]
    res = hf.load_state_dict(m, strict=False)
    assert not res.unexpected_keys and all(k.endswith("num_batches_tracked") for k in res.missing_keys)
    assert sum(p.numel() for p in hf.parameters()) == ores.EXPECTED_NUM_PARAMS_7     # 11 180 103
    x = torch.randn(3, 3, 224, 224, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        theirs = hf(pixel_values=x).logits
        ours = ores.resnet18_features(sd, x)
    assert torch.isfinite(ours).all() and ours.abs().max() > 1e-3
    torch.testing.assert_close(ours, theirs, rtol=1e-5, atol=1e-5 * float(theirs.abs().max().clamp(min=1.0)))
    # and through the reference's pre-processing at a non-native resolution (XAI.py:399-431)
    x64 = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(6)).clamp(-1.5, 1.5)
    with torch.no_grad():
        torch.testing.assert_close(ores.classifier_forward(sd, x64),
                                   hf(pixel_values=ores.preprocess_for_classifier(x64)).logits, rtol=1e-5, atol=2e-5)


def test_oracle_attention_block_matches_torch_multihead_attention():
    """Independent pin of the attention block's arithmetic (oracle/unet.py attention_block; diffusers' Attention as configured at
    model_manager.py:173-194: 32 heads x d = 8, GroupNorm in front, residual behind): torch's own nn.MultiheadAttention --
    packed in-projection, contiguous channel runs per head, softmax(q k^T / sqrt(d)) v, out-projection -- assembled from the
    same state-dict entries, with nn.GroupNorm in front.  Not the oracle's code path: a different implementation of the same
    published block."""
    import torch.nn as nn
    from oracle import unet as ounet
    sd = weights.synthetic_unet_state_dict()
    for p, hw in (("mid_block.attentions.0", 8), ("down_blocks.2.attentions.1", 16)):
        assert f"{p}.to_q.weight" in sd
        C = sd[f"{p}.to_q.weight"].shape[0]
        mha = nn.MultiheadAttention(C, C // ounet.HEAD_DIM, bias=True, batch_first=True).eval()
        gn = nn.GroupNorm(ounet.NORM_GROUPS, C, eps=ounet.NORM_EPS).eval()
        with torch.no_grad():
            mha.in_proj_weight.copy_(torch.cat([sd[f"{p}.to_{n}.weight"] for n in "qkv"]))
            mha.in_proj_bias.copy_(torch.cat([sd[f"{p}.to_{n}.bias"] for n in "qkv"]))
            mha.out_proj.weight.copy_(sd[f"{p}.to_out.0.weight"])
            mha.out_proj.bias.copy_(sd[f"{p}.to_out.0.bias"])
            gn.weight.copy_(sd[f"{p}.group_norm.weight"])
            gn.bias.copy_(sd[f"{p}.group_norm.bias"])
            x = torch.randn(2, C, hw, hw, generator=torch.Generator().manual_seed(3))
            tokens = gn(x).flatten(2).transpose(1, 2)                       # [B, N, C]
            o, _ = mha(tokens, tokens, tokens, need_weights=False)
            ref = o.transpose(1, 2).reshape(x.shape) + x
            got = ounet.attention_block(sd, p, x)
        torch.testing.assert_close(got, ref, rtol=1e-5, atol=2e-5)


def test_product_does_not_import_oracle():
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "synt_isic_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{fn} imports the oracle"


def test_color_postprocessing_matches_oracle_bit_exact(tmp_path):
    """image_generator.py:502-545: the batched host function == the per-image restatement, including the clip of the
    scale, an all-constant image (zero variance) and the no-op cases."""
    import json
    import numpy as np
    from oracle import sampler as osampler
    from synt_isic_amd import sampler
    rng = np.random.default_rng(0)
    imgs = rng.integers(0, 256, size=(5, 32, 24, 3), dtype=np.uint8)
    imgs[3] = 77                                               # zero variance: scale clips at 1.4, mean moves
    imgs[4, :, :, 0] = rng.integers(100, 104, size=(32, 24))   # tiny variance in one channel
    stats = {"rgb": {"mean": [180.5, 120.25, 90.0], "std": [40.0, 55.5, 10.0]}}
    got = sampler.apply_color_statistics(imgs, stats)
    assert got.dtype == np.uint8 and got.shape == imgs.shape
    for b in range(imgs.shape[0]):
        assert np.array_equal(got[b], osampler.color_postprocess(imgs[b], stats))
    assert not np.array_equal(got, imgs)
    # documented no-ops: unknown class / entry without rgb.mean
    assert sampler.apply_color_statistics(imgs, None) is imgs
    assert sampler.apply_color_statistics(imgs, {"rgb": {"std": [1, 2, 3]}}) is imgs
    # defaults when std is absent
    only_mean = {"rgb": {"mean": [10, 20, 30]}}
    assert np.array_equal(sampler.apply_color_statistics(imgs[:1], only_mean)[0], osampler.color_postprocess(imgs[0], only_mean))
    # a known answer worked by hand: constant image 100, target mean 200 -> 0.35*(0*1.4+200) + 0.65*100 = 135
    const = np.full((1, 4, 4, 3), 100, dtype=np.uint8)
    assert np.all(sampler.apply_color_statistics(const, {"rgb": {"mean": [200, 200, 200], "std": [50, 50, 50]}}) == 135)
    # the JSON loader (checkpoints/color_statistics.json); a missing file is not an error
    s = sampler.Sampler.__new__(sampler.Sampler)
    s.color_statistics = {}
    assert s.load_color_statistics(str(tmp_path / "missing.json")) == 0
    (tmp_path / "color_statistics.json").write_text(json.dumps({"NV": stats, "MEL": only_mean}))
    assert s.load_color_statistics(str(tmp_path / "color_statistics.json")) == 2 and "NV" in s.color_statistics


def test_streamed_segments_cover_the_run_and_ramp_when_a_segment_is_a_lot_of_rng():
    """sampler.segment_bounds: contiguous segments over [0, T], none longer than the stream's buffers; a run whose segment
    is more than a million normals (one 128x128 image: 3 M per 64 steps) starts with 4, 8, 16, ... steps, one 64x64 image
    (0.8 M) does not."""
    from synt_isic_amd.sampler import segment_bounds
    for T in (1, 3, 4, 50, 64, 65, 130, 1000):
        for seg in (1, 8, 64):
            for per in (3 * 128 * 128, 64 * 3 * 64 * 64):
                b = segment_bounds(T, seg, per)
                assert b[0] == 0 and b[-1] == T and all(0 < y - x <= seg for x, y in zip(b[:-1], b[1:]))
    assert segment_bounds(1000, 64, 64 * 3 * 64 * 64)[:6] == [0, 4, 12, 28, 60, 124]
    assert segment_bounds(50, 64, 3 * 128 * 128) == [0, 4, 12, 28, 50]
    assert segment_bounds(50, 64, 3 * 64 * 64) == [0, 50]
    assert segment_bounds(130, 64, 3 * 64 * 64) == [0, 64, 128, 130]



def _reference_save_indices(timesteps, save_every):
    """xai/XAI.py:751-777 and :815-822 restated step by step as that function runs (the set built before the loop, then the
    per-step test inside it) -- the test's own statement of the reference's behaviour"""
    n = len(timesteps)
    save_indices = set(range(0, n, save_every))
    if (n - 1) not in save_indices:
        save_indices.add(n - 1)
    by_t = save_every >= n
    if by_t:
        t_list = [int(float(t)) for t in timesteps]
        desired = {0, max(t_list)}
        k = 0
        while k <= 1000:
            desired.add(k)
            k += max(1, int(save_every))
        for dt in desired:
            save_indices.add(min(range(len(t_list)), key=lambda i: abs(t_list[i] - dt)))
    kept = []
    for i, t in enumerate(timesteps):
        save = i in save_indices
        if not save and by_t:
            t_int = int(float(t))
            save = (t_int % max(1, save_every) == 0) or t_int == 0
        if save:
            kept.append(i)
    return kept


def test_trajectory_save_indices_follow_the_reference():
    from synt_isic_amd.sampler import trajectory_save_indices
    from synt_isic_amd.scheduler import HipDDPMScheduler
    for T in (1, 2, 13, 50, 100, 1000):
        s = HipDDPMScheduler()
        s.set_timesteps(T)
        ts = [int(t) for t in s.timesteps]
        for every in (1, 2, 5, 10, 49, 50, 100, 250, 1000):
            assert trajectory_save_indices(ts, every) == _reference_save_indices(ts, every), (T, every)
    assert trajectory_save_indices(list(range(980, -1, -20)), 10) == [0, 10, 20, 30, 40, 49]       # 50 steps, every 10th + the last
    import pytest
    with pytest.raises(ValueError):
        trajectory_save_indices([3, 2, 1], 0)
