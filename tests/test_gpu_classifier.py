"""ResNet18 classifier forward, the as-coded Time-SHAP / patch-SHAP passes and the README-form
permutation Time-SHAP, through the C ABI, against the CPU oracle (oracle/resnet18.py).

Tolerance: logits max-abs <= 2e-4 * max(1, |ref|_inf) (18 fp32 conv layers, BatchNorm folded in float64);
scores follow from the logits; everything integer / mask-related is exact.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"
NV = 1          # target class id of "NV" (xai/XAI.py:196)


def _close(got, ref, tol, what=""):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    bound = tol * max(1.0, ref.abs().max().item())
    err = (got - ref).abs().max().item()
    assert got.shape == ref.shape and err <= bound, f"{what}: err {err:.3e} > {bound:.3e} (shape {tuple(got.shape)})"


@pytest.fixture(scope="module")
def clf_sd():
    from synt_isic_amd.weights import synthetic_resnet18_state_dict
    return synthetic_resnet18_state_dict()


@pytest.fixture(scope="module")
def clf(clf_sd):
    from synt_isic_amd.classifier import HipMelanomaClassifier
    m = HipMelanomaClassifier(num_classes=7, pretrained=False)
    sd = dict(clf_sd)
    sd["model.bn1.num_batches_tracked"] = torch.tensor(0)       # integer buffers of a real checkpoint are ignored
    m.load_state_dict(sd)
    return m.to(DEV).eval()


def _x(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, 3, H, W, generator=g) * 0.8          # some values leave [-1,1]: the clamp is exercised


@pytest.mark.parametrize("cfg,H,W", [(0, 224, 224), (41, 64, 48), (41, 30, 70)])
def test_conv7x7_stride2(cfg, H, W):
    from synt_isic_amd import ops
    x = _x(2, H, W, 1)
    w = torch.randn(64, 3, 7, 7, generator=torch.Generator().manual_seed(2)) * 0.1
    b = torch.randn(64, generator=torch.Generator().manual_seed(3))
    got = ops.conv2d(x.to(DEV), ops.pack_conv_weight(w.to(DEV)), 64, 7, bias=b.to(DEV), stride=2, relu=True, tile_cfg=cfg)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=3))
    _close(got, ref, 1e-5, f"conv7x7 s2 {H}x{W}")


@pytest.mark.parametrize("cfg,H,W,cin,cout", [(0, 56, 56, 64, 128), (31, 56, 56, 64, 128), (32, 28, 28, 128, 256),
                                              (33, 14, 14, 256, 512), (31, 30, 50, 24, 70), (0, 14, 14, 256, 512)])
def test_conv1x1_stride2(cfg, H, W, cin, cout):
    from synt_isic_amd import ops
    x = _x(2, H, W, 4)[:, :1].repeat(1, cin, 1, 1) * torch.randn(1, cin, 1, 1, generator=torch.Generator().manual_seed(5))
    w = torch.randn(cout, cin, 1, 1, generator=torch.Generator().manual_seed(6)) * 0.1
    b = torch.randn(cout, generator=torch.Generator().manual_seed(7))
    got = ops.conv2d(x.contiguous().to(DEV), ops.pack_conv_weight(w.to(DEV)), cout, 1, bias=b.to(DEV), stride=2, tile_cfg=cfg)
    _close(got, F.conv2d(x.double(), w.double(), b.double(), stride=2), 1e-5, f"conv1x1 s2 {H}x{W}")


@pytest.mark.parametrize("B,H,W", [(3, 64, 64), (2, 128, 128), (1, 224, 224), (2, 96, 40)])
def test_classifier_logits(clf, clf_sd, B, H, W):
    from oracle import resnet18 as ores
    x = _x(B, H, W, 10 + H)
    ref = ores.classifier_forward(clf_sd, x)
    got = clf(x)                                        # a CPU tensor is moved to the model's device (XAI.py:407-409)
    assert got.shape == (B, 7) and got.device.type == "cuda"
    _close(got, ref, 2e-4, f"logits {B}x{H}x{W}")


def test_classifier_preprocessed_input_and_bounds(clf, clf_sd):
    from oracle import resnet18 as ores
    from synt_isic_amd._lib import SisicError
    x = _x(2, 64, 64, 20)
    pre = ores.preprocess_for_classifier(x)
    _close(clf.forward(pre.to(DEV), preprocessed=True), ores.resnet18_features(clf_sd, pre), 2e-4, "preprocessed")
    with pytest.raises(SisicError, match="down-scale"):
        clf(torch.zeros(1, 3, 256, 256))


def test_scores_and_module_surface(clf, clf_sd):
    from oracle import resnet18 as ores
    x = _x(4, 64, 64, 30)
    p_ref, s_ref = ores.class_scores(clf_sd, x, NV)
    _close(clf.get_confidence(x, NV), p_ref, 2e-4, "confidence")
    _close(clf.get_per_class_score(x, NV), s_ref, 5e-4, "log score")
    probs = clf.get_probabilities(x)
    assert probs.shape == (4, 7)
    torch.testing.assert_close(probs.sum(1).cpu(), torch.ones(4), rtol=0, atol=1e-5)      # XAI.py:543-555 self-check
    assert torch.equal(clf.predict(x).cpu(), ores.classifier_forward(clf_sd, x).argmax(1))
    assert sum(p.numel() for p in clf.parameters()) == 11_180_103
    assert next(clf.parameters()).device.type == "cuda" and clf.training is False


def test_time_shap_as_coded(clf, clf_sd):
    """compute_time_shap (XAI.py:1179-1234): N frames of a trajectory -> min-max normalised log-scores."""
    from oracle import resnet18 as ores
    from synt_isic_amd import xai
    g = torch.Generator().manual_seed(40)
    frames = [torch.randn(1, 3, 64, 64, generator=g) * (1.0 - 0.08 * i) for i in range(10)]
    timesteps = list(range(900, -1, -100))
    imp_ref, raw_ref = ores.time_shap_as_coded(clf_sd, frames, timesteps, NV)
    imp, raw = xai.compute_time_shap(clf, [f.to(DEV) for f in frames], timesteps, NV)
    assert imp.shape == (10,) and raw["timesteps"] == timesteps
    np.testing.assert_allclose(raw["confidence_scores"], raw_ref["confidence_scores"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(raw["probability_scores"], raw_ref["probability_scores"], rtol=0, atol=2e-4)
    rng = raw_ref["confidence_scores"].max() - raw_ref["confidence_scores"].min()
    np.testing.assert_allclose(imp, imp_ref, rtol=0, atol=2e-3 / max(rng, 1e-3) + 1e-6)
    assert imp.min() == 0.0 and imp.max() == 1.0
    # flat scores -> uniform importance (the reference's else-branch)
    same = [frames[0].to(DEV)] * 4
    imp2, _ = xai.compute_time_shap(clf, same, [3, 2, 1, 0], NV)
    np.testing.assert_allclose(imp2, np.full(4, 0.25))
    # a [T,B,3,H,W] trajectory tensor from the sampler is accepted too
    imp3, _ = xai.compute_time_shap(clf, torch.stack([f.to(DEV) for f in frames]), timesteps, NV)
    np.testing.assert_array_equal(imp3, imp)


def test_patch_shap_matches_oracle(clf, clf_sd):
    """compute_shap_approximation (XAI.py:1111-1177) with the same patch masks on both sides."""
    from oracle import resnet18 as ores
    from synt_isic_amd import xai
    image = _x(1, 64, 64, 50)
    masks = xai.draw_patch_masks(12, 4, 4, generator=torch.Generator().manual_seed(51))
    ref = ores.shap_approximation(clf_sd, image, NV, n_samples=12, patch_size=16, patch_masks=masks)
    got = xai.compute_shap_approximation(clf, image, NV, n_samples=12, patch_size=16, patch_masks=masks, chunk=5)
    assert got.shape == (1, 3, 64, 64)
    _close(got, ref, 1e-3, "patch SHAP attribution")
    # masked copies are built exactly: kept pixels equal the image, dropped pixels are zero
    import ctypes as C
    from synt_isic_amd import _lib, ops
    out = torch.empty((12, 3, 64, 64), device=DEV)
    m8 = masks.to(torch.uint8).to(DEV)
    img = image.to(DEV)
    _lib.check(_lib.load().sisic_mask_patches(ops.context(img.device), img.data_ptr(), m8.data_ptr(), out.data_ptr(),
                                              12, 3, 64, 64, 16, None))
    for s in range(12):
        full = ores.expand_patch_mask(masks[s], 64, 64, 16)
        want = image[0].clone()
        want[:, ~full] = 0
        assert torch.equal(out[s].cpu(), want)
    # the default path draws its masks from the global CPU RNG like the reference (XAI.py:1147)
    torch.manual_seed(7)
    a = xai.compute_shap_approximation(clf, image, NV, n_samples=8)
    torch.manual_seed(7)
    b = xai.compute_shap_approximation(clf, image, NV, n_samples=8)
    assert torch.equal(a, b)


def test_permutation_time_shap_properties(clf, clf_sd, synthetic_sd):
    """README.md:171-207 form: efficiency axiom (exact by telescoping), determinism, and v(S) against the oracle."""
    from oracle import ddpm as oddpm, resnet18 as ores, sampler as osampler, unet as ounet
    from synt_isic_amd import xai
    from synt_isic_amd.sampler import Sampler, draw_noise
    s = Sampler(DEV)
    s.add_model("NV", synthetic_sd)
    T = 6
    r1 = xai.time_shap_permutation(s, clf, "NV", [0, 1], T, NV, n_permutations=2, size=(32, 32),
                                   generator=torch.Generator().manual_seed(1))
    r2 = xai.time_shap_permutation(s, clf, "NV", [0, 1], T, NV, n_permutations=2, size=(32, 32),
                                   generator=torch.Generator().manual_seed(1))
    assert r1["phi"].shape == (T,) and np.array_equal(r1["phi"], r2["phi"])
    assert abs(r1["phi"].sum() - (r1["v_full"] - r1["v_empty"])) < 1e-9
    # v(S) for S = {steps 0, 2, 5} vs the oracle: skip the other steps, standard coefficients on the kept ones
    x_T, z = draw_noise([0, 1], T - 1, (3, 32, 32))
    v = xai.coalition_value(s, clf, "NV", x_T.to(DEV), z.to(DEV), T, [[0, 2, 5], [], list(range(T))], NV)
    sched = oddpm.DDPMSchedulerOracle(); sched.set_timesteps(T)
    ts = [int(t) for t in sched.timesteps]
    x = x_T.clone()
    with torch.no_grad():
        for i in (0, 2, 5):
            eps = ounet.unet_forward(synthetic_sd, x, ts[i])
            x = sched.step(eps, ts[i], x, noise=z[i] if ts[i] > 0 else None)
    v_ref = ores.classifier_forward(clf_sd, x)[:, NV].mean().item()
    assert abs(v[0].item() - v_ref) <= 2e-3 * max(1.0, abs(v_ref))
    v_empty_ref = ores.classifier_forward(clf_sd, x_T)[:, NV].mean().item()
    assert abs(v[1].item() - v_empty_ref) <= 2e-4 * max(1.0, abs(v_empty_ref))
    assert abs(v[2].item() - r1["v_full"]) <= 1e-6 * max(1.0, abs(r1["v_full"]))


# ---------------------------------------------------------------- backward to the input
def test_zero_insertion_conv_is_the_transposed_stride2_conv():
    """sisic_conv_args.upsample = 2: a stride-1 convolution over the zero-inserted 2x grid with transposed, tap-flipped
    filters == conv_transpose2d(stride 2, padding 1, output_padding 1) == the gradient of a 3x3 stride-2 convolution."""
    import torch.nn.functional as F
    from synt_isic_amd import ops
    g = torch.Generator().manual_seed(3)
    B, cin, cout, H = 2, 24, 40, 12                       # forward conv: cin -> cout, 12x12 -> 6x6
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    dy = torch.randn(B, cout, H // 2, H // 2, generator=g)
    wt = w.flip(2, 3).transpose(0, 1).contiguous()        # [cin, cout, 3, 3]: W'[ci][co][a][b] = W[co][ci][2-a][2-b]
    got = ops.conv2d(dy.to(DEV), ops.pack_conv_weight(wt.to(DEV)), cin, 3, upsample=2)
    ref = F.conv_transpose2d(dy.double(), w.double(), stride=2, padding=1, output_padding=1)
    assert got.shape == ref.shape == (B, cin, H, H)
    assert (got.cpu().double() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    # and it is what autograd computes for the forward convolution
    x = torch.randn(B, cin, H, H, generator=g, dtype=torch.float64, requires_grad=True)
    F.conv2d(x, w.double(), stride=2, padding=1).backward(dy.double())
    assert (got.cpu().double() - x.grad).abs().max().item() <= 1e-5 * max(1.0, x.grad.abs().max().item())


@pytest.mark.parametrize("B,H,W,target", [(2, 64, 64, 1), (3, 128, 128, 4), (1, 40, 56, 0), (2, 224, 224, 6), (1, 96, 96, 2)])
def test_input_gradient_matches_autograd_of_the_oracle(clf, clf_sd, B, H, W, target):
    """d log(softmax[c] + 1e-8) / dx through pre-processing, stem, max-pool, the 8 residual blocks and the head, with
    no autograd graph on the GPU side, against torch.autograd over the CPU restatement.

    Where both passes take the same max-pool routes the agreement is ~1e-6 of the largest gradient.  The stem's
    max-pool has ~2e5 windows per image; when the two largest values of one window differ by less than the fp32
    rounding of the stem convolution, the CPU and GPU forwards pick different arg-maxima and the gradient of that
    window lands on the neighbouring pixel -- a legitimate, local difference (about every second image has one).
    Stated tolerance therefore: per image, >= 98 % of the elements within 2e-4 of the largest gradient, cosine
    similarity >= 0.999, and at least one image of the batch (the first) within 2e-4 everywhere for the seeds used;
    test_input_gradient_is_the_directional_derivative below checks the gradient against the GPU forward itself."""
    from oracle import resnet18 as ores
    g = torch.Generator().manual_seed(100 + H)
    x = torch.rand(B, 3, H, W, generator=g) * 1.6 - 0.8
    grad, logits = clf.input_gradient(x.to(DEV), target)
    ref_g, ref_l = ores.score_input_gradient(clf_sd, x, target)
    assert (logits.cpu() - ref_l).abs().max().item() <= 2e-4 * max(1.0, ref_l.abs().max().item())
    assert torch.equal(logits, clf(x.to(DEV)))                  # the forward it reports is the ordinary forward
    strict = 0
    for b in range(B):
        gb, rb = grad[b].cpu(), ref_g[b]
        scale = rb.abs().max().item()
        assert scale > 0
        d = (gb - rb).abs()
        assert (d <= 2e-4 * scale).float().mean().item() >= 0.98, f"image {b}"
        assert F.cosine_similarity(gb.flatten(), rb.flatten(), dim=0).item() >= 0.999, f"image {b}"
        strict += int(d.max().item() <= 2e-4 * scale)
    assert strict >= 1


REPLAY_TOL = 1e-3       # one ReLU of the 17 flipping at a pre-activation within rounding of zero moves the gradient by ~4e-4
STRICT_TOL = 1e-5


@pytest.mark.parametrize("B,H,W,target", [(2, 64, 64, 1), (3, 128, 128, 4), (2, 224, 224, 6)])
def test_input_gradient_max_abs_when_the_pool_routes_are_replayed(clf, clf_sd, B, H, W, target):
    """The arg-max-tie explanation of the statistical test above, demonstrated.  The CPU autograd pass is run twice: with
    its own max-pool routes ("free"), and with the GPU's own stem activation substituted (values only; the gradient still
    flows through the CPU graph), so that its max-pool takes the routes the GPU took ("replayed").  Measured (r02, errlog):
    the images whose free-route error is 7e-3 / 1e-1 of the largest gradient drop to 2e-6 / 4e-6 once the routes are
    replayed -- the whole discrepancy was the route -- and on every image at least one of the two CPU passes agrees with the
    GPU EVERYWHERE to <= 1e-5 of the largest gradient (max-abs, no fraction, no cosine).  The replayed pass itself is bounded
    by 1e-3: substituting the stem perturbs every later activation by ~1e-6, which now and then flips one ReLU whose
    pre-activation is within rounding of zero (one such flip measured: 3.7e-4)."""
    from oracle import resnet18 as ores
    g = torch.Generator().manual_seed(100 + H)                 # the same inputs as the statistical test
    x = torch.rand(B, 3, H, W, generator=g) * 1.6 - 0.8
    grad, logits = clf.input_gradient(x.to(DEV), target)
    stem = clf.stem_activation(x.to(DEV)).cpu()
    with torch.no_grad():
        h = ores.preprocess_for_classifier(x)
        h = F.relu(ores._bn(clf_sd, "model.bn1", F.conv2d(h, clf_sd["model.conv1.weight"], None, stride=2, padding=3)))
    assert (stem - h).abs().max().item() <= 1e-5 * max(1.0, h.abs().max().item())
    ref_g, ref_l = ores.score_input_gradient(clf_sd, x, target, stem_override=stem)
    free_g, _ = ores.score_input_gradient(clf_sd, x, target)                  # the CPU's own routes
    assert (logits.cpu() - ref_l).abs().max().item() <= 2e-4 * max(1.0, ref_l.abs().max().item())
    for b in range(B):
        scale = ref_g[b].abs().max().item()
        err = (grad[b].cpu() - ref_g[b]).abs().max().item()
        err_free = (grad[b].cpu() - free_g[b]).abs().max().item()
        if os.environ.get("SISIC_TEST_ERRLOG"):
            with open(os.environ["SISIC_TEST_ERRLOG"], "a") as f:
                f.write(f"{err / scale:.3e}\t{err_free / scale:.3e}\tinput_gradient replayed/free routes B{B} {H}x{W} image {b}\n")
        assert err <= REPLAY_TOL * scale, f"image {b}: replayed routes max|dgrad| = {err:.3e} vs scale {scale:.3e}"
        assert min(err, err_free) <= STRICT_TOL * scale, f"image {b}: replayed {err / scale:.3e}, free {err_free / scale:.3e}"


def test_input_gradient_is_the_directional_derivative(clf):
    """Independent of any CPU pass: <grad, v> against the central difference of the GPU forward's own score along a
    random direction v.  The score is piecewise smooth (ReLU / max-pool kinks along the segment) and the fp32 forward
    is noisy at small steps, so this is a coarse check of sign and size (10 % + 0.015), not a parity statement."""
    g = torch.Generator().manual_seed(77)
    x = (torch.rand(4, 3, 64, 64, generator=g) * 1.6 - 0.8).to(DEV)
    v = torch.randn(4, 3, 64, 64, generator=g).to(DEV)
    grad, _ = clf.input_gradient(x, 2)
    eps = 1e-3
    fd = (clf.get_per_class_score(x + eps * v, 2) - clf.get_per_class_score(x - eps * v, 2)) / (2 * eps)
    an = (grad * v).sum(dim=(1, 2, 3))
    assert torch.allclose(an.cpu(), fd.cpu(), rtol=0.1, atol=1.5e-2), (an.cpu(), fd.cpu())


def test_input_gradient_with_saturated_pixels(clf, clf_sd):
    """pixels outside [-1,1]: clamp((x+1)/2, 0, 1) is flat there, so their gradient is exactly zero.  Saturated
    neighbours make exactly constant patches in the 224x224 image, hence near-ties in the stem's max-pool windows whose
    argmax (and with it the route of the gradient) depends on the last bit of the convolution: a handful of elements
    may legitimately differ from the CPU pass (here: thousands of tied windows, ~8 % of the elements), so this case is
    checked in norm, by fraction and by direction, not by max-abs."""
    from oracle import resnet18 as ores
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(164)) * 2.4 - 1.2
    grad, _ = clf.input_gradient(x.to(DEV), 1)
    ref, _ = ores.score_input_gradient(clf_sd, x, 1)
    out = x.abs() > 1.0
    assert out.float().mean().item() > 0.1 and torch.all(grad.cpu()[out] == 0) and torch.all(ref[out] == 0)
    d = grad.cpu() - ref
    assert (d.norm() / ref.norm()).item() <= 5e-2
    assert (d.abs() <= 2e-4 * ref.abs().max()).float().mean().item() >= 0.85
    assert F.cosine_similarity(grad.cpu().flatten(), ref.flatten(), dim=0).item() >= 0.998


def test_integrated_gradients_matches_oracle_and_completeness(clf, clf_sd):
    """XAI.py:1039-1084: captum IntegratedGradients, n_steps = 50, riemann_right, noise baseline * 0.1.  Parity with
    the restatement, and the completeness axiom: sum of attributions ~ score(x) - score(baseline) up to the Riemann
    error of 50 steps."""
    from oracle import resnet18 as ores
    from synt_isic_amd import xai
    g = torch.Generator().manual_seed(9)
    x = torch.rand(2, 3, 64, 64, generator=g) * 1.6 - 0.8
    base = xai.make_baseline(x, "noise", torch.Generator().manual_seed(10))
    assert torch.equal(base, torch.randn(x.shape, generator=torch.Generator().manual_seed(10)) * 0.1)
    ig = xai.compute_integrated_gradients(clf, x.to(DEV), 1, n_steps=50, baseline=base, max_batch=64)
    ref = ores.integrated_gradients(clf_sd, x, 1, base, n_steps=50)
    scale = ref.abs().max().item()
    d = (ig.cpu() - ref).abs()                    # arg-max route flips (see above) at single Riemann points, diluted 1/50
    assert d.max().item() <= 1e-2 * scale and (d <= 3e-4 * scale).float().mean().item() >= 0.99
    assert F.cosine_similarity(ig.cpu().flatten(), ref.flatten(), dim=0).item() >= 0.9999
    fx = clf.get_per_class_score(x.to(DEV), 1).cpu()
    fb = clf.get_per_class_score(base.to(DEV), 1).cpu()
    total = ig.cpu().sum(dim=(1, 2, 3))
    assert torch.allclose(total, fx - fb, atol=0.05 * (fx - fb).abs().max().item() + 1e-3)
    # the plain-gradient fallback (XAI.py:1086-1109) is the same gradient at the image itself
    ga = xai.compute_gradient_attribution(clf, x.to(DEV), 1)
    assert torch.equal(ga, clf.input_gradient(x.to(DEV), 1)[0])
    # zero and blur baselines of _get_baseline
    assert torch.count_nonzero(xai.make_baseline(x, "zero")) == 0
    assert xai.make_baseline(x, "blur").shape == x.shape


def test_grad_cam_matches_oracle(clf, clf_sd):
    """pytorch_grad_cam-style Grad-CAM on layer4[-1].conv2 (XAI.py:2945-3035) against the autograd restatement: maps in
    [0,1] with their extremes attained, <= 2e-3 apart (a flipped ReLU mask at the 7x7 level moves a channel weight
    by 1/49 of its value), logits as the ordinary forward."""
    from oracle import resnet18 as ores
    from synt_isic_amd import xai
    g = torch.Generator().manual_seed(31)
    x = torch.rand(5, 3, 64, 64, generator=g) * 1.8 - 0.9
    cam, logits = clf.grad_cam(x.to(DEV), NV)
    ref = ores.grad_cam(clf_sd, x, NV)
    assert cam.shape == (5, 224, 224) and torch.equal(logits, clf(x.to(DEV)))
    c = cam.cpu()
    assert c.min().item() >= 0.0 and c.max().item() <= 1.0
    assert torch.all(c.flatten(1).min(1).values == 0) and torch.all(c.flatten(1).max(1).values > 0.999)
    assert (c - ref).abs().max().item() <= 2e-3, (c - ref).abs().max().item()
    # 128x128 frames (the reference's trajectory size) and the per-trajectory summary
    frames = [torch.rand(1, 3, 128, 128, generator=g) * 1.6 - 0.8 for _ in range(4)]
    res = xai.compute_grad_cam(clf, frames, [980, 600, 300, 0], NV)
    assert sorted(res) == ["summary", "t_0", "t_300", "t_600", "t_980"]
    ref4 = ores.grad_cam(clf_sd, torch.cat(frames), NV).numpy()
    assert np.abs(res["t_600"] - ref4[1]).max() <= 2e-3
    mean = ref4.mean(0)
    assert np.abs(res["summary"] - (mean - mean.min()) / (mean.max() - mean.min() + 1e-8)).max() <= 4e-3
