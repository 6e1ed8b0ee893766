"""CPU oracle: the ResNet18 classifier and the XAI forward passes built on it.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- "parity unpinned": the network comes from
``torchvision.models.resnet18`` (``torchvision>=0.13.0``, requirements.txt:5, not vendored, absent
here), so this restates the published architecture with ``torch.nn.functional``
primitives, wired the way xai/XAI.py uses it:

  * MelanomaClassifierAdaptive._create_builtin_model  XAI.py:385-397  (fc -> num_classes, keys "model.*")
  * preprocess_for_classifier                          XAI.py:399-431
  * get_probabilities / get_per_class_score / get_confidence   XAI.py:438-471
  * ModernXAIAnalyzer.compute_shap_approximation      XAI.py:1111-1177  (patch coalitions)
  * ModernXAIAnalyzer.compute_time_shap               XAI.py:1179-1234  (as coded: per-frame scores, min-max)
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

CLASSIFIER_IMAGE_SIZE = 224
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
BN_EPS = 1e-5
EXPECTED_NUM_PARAMS_7 = 11_180_103        # SURVEY.md Appendix C (fc -> 7)


def param_spec(num_classes: int = 7) -> "OrderedDict[str, Tuple[int, ...]]":
    """Float tensors of the state dict (weights, biases, BatchNorm running statistics), 'model.' prefix."""
    spec: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def conv_bn(conv, bn, cout, cin, k):
        spec[f"model.{conv}.weight"] = (cout, cin, k, k)
        for s in ("weight", "bias", "running_mean", "running_var"):
            spec[f"model.{bn}.{s}"] = (cout,)

    conv_bn("conv1", "bn1", 64, 3, 7)
    in_ch = 64
    for l, width in enumerate((64, 128, 256, 512)):
        for j in range(2):
            stride = 2 if (l > 0 and j == 0) else 1
            base = f"layer{l + 1}.{j}"
            conv_bn(f"{base}.conv1", f"{base}.bn1", width, in_ch, 3)
            conv_bn(f"{base}.conv2", f"{base}.bn2", width, width, 3)
            if stride != 1 or in_ch != width:
                conv_bn(f"{base}.downsample.0", f"{base}.downsample.1", width, in_ch, 1)
            in_ch = width
    spec["model.fc.weight"] = (num_classes, 512)
    spec["model.fc.bias"] = (num_classes,)
    return spec


def num_trainable_params(num_classes: int = 7) -> int:
    return sum(int(np.prod(s)) for k, s in param_spec(num_classes).items() if "running_" not in k)


def preprocess_for_classifier(x: torch.Tensor) -> torch.Tensor:
    """XAI.py:399-431."""
    x = torch.clamp((x + 1.0) / 2.0, 0, 1)
    if x.shape[-1] != CLASSIFIER_IMAGE_SIZE or x.shape[-2] != CLASSIFIER_IMAGE_SIZE:
        x = F.interpolate(x, size=(CLASSIFIER_IMAGE_SIZE, CLASSIFIER_IMAGE_SIZE), mode="bilinear",
                          align_corners=False, antialias=True)
    mean = torch.tensor(IMAGENET_MEAN, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=x.dtype).view(1, 3, 1, 1)
    return (x - mean) / std


def _bn(sd, name, x):
    return F.batch_norm(x, sd[f"{name}.running_mean"], sd[f"{name}.running_var"], sd[f"{name}.weight"],
                        sd[f"{name}.bias"], training=False, eps=BN_EPS)


def resnet18_features(sd: Dict[str, torch.Tensor], x: torch.Tensor,
                      stem_override: Optional[torch.Tensor] = None) -> torch.Tensor:
    """torchvision resnet18 in eval mode on an already normalised input -> logits.

    stem_override (tests of a backward pass): VALUES of relu(bn1(conv1(x))) to use in place of the ones computed here,
    while the gradient still flows through this graph (straight-through: y + (override - y).detach()).  The max-pool
    that follows then takes the arg-maxima of the override, i.e. the routes of the implementation under test."""
    p = "model."
    x = F.conv2d(x, sd[p + "conv1.weight"], None, stride=2, padding=3)
    x = F.relu(_bn(sd, p + "bn1", x))
    if stem_override is not None:
        x = x + (stem_override.to(x.dtype) - x).detach()
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    in_ch = 64
    for l, width in enumerate((64, 128, 256, 512)):
        for j in range(2):
            stride = 2 if (l > 0 and j == 0) else 1
            base = f"{p}layer{l + 1}.{j}"
            identity = x
            out = F.conv2d(x, sd[base + ".conv1.weight"], None, stride=stride, padding=1)
            out = F.relu(_bn(sd, base + ".bn1", out))
            out = F.conv2d(out, sd[base + ".conv2.weight"], None, stride=1, padding=1)
            out = _bn(sd, base + ".bn2", out)
            if base + ".downsample.0.weight" in sd:
                identity = F.conv2d(x, sd[base + ".downsample.0.weight"], None, stride=stride)
                identity = _bn(sd, base + ".downsample.1", identity)
            x = F.relu(out + identity)
            in_ch = width
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)
    return F.linear(x, sd[p + "fc.weight"], sd[p + "fc.bias"])


@torch.no_grad()
def classifier_forward(sd, x: torch.Tensor) -> torch.Tensor:
    """MelanomaClassifierAdaptive.forward: preprocess + resnet18.  x: [B,3,H,W] in [-1,1]."""
    return resnet18_features(sd, preprocess_for_classifier(x))


@torch.no_grad()
def class_scores(sd, x: torch.Tensor, target_class: int):
    """(get_confidence, get_per_class_score): softmax(logits)[:,c] and log(. + 1e-8)."""
    probs = F.softmax(classifier_forward(sd, x), dim=1)
    p = probs[:, target_class]
    return p, torch.log(p + 1e-8)


def score_input_gradient(sd, x: torch.Tensor, target_class: int, stem_override: Optional[torch.Tensor] = None):
    """d/dx of get_per_class_score = log(softmax(forward(x))[:, c] + 1e-8) by autograd over the restatement above
    (what ``_compute_gradient_attribution``, XAI.py:1086-1109, returns per sample, and what captum's
    IntegratedGradients differentiates at every Riemann point).  Returns (grad [B,3,H,W], logits [B,n])."""
    with torch.enable_grad():
        xg = x.detach().clone().requires_grad_(True)
        logits = resnet18_features(sd, preprocess_for_classifier(xg), stem_override)
        score = torch.log(F.softmax(logits, dim=1)[:, target_class] + 1e-8)
        (grad,) = torch.autograd.grad(score.sum(), xg)
    return grad.detach(), logits.detach()


def integrated_gradients(sd, image: torch.Tensor, target_class: int, baseline: torch.Tensor, n_steps: int = 50):
    """captum IntegratedGradients.attribute(image, baselines=baseline, n_steps, method='riemann_right') for the
    per-class score (XAI.py:1039-1084): (x - x') * mean_k grad(x' + k/n (x - x')), k = 1..n."""
    total = torch.zeros_like(image)
    for k in range(1, n_steps + 1):
        point = baseline + (k / n_steps) * (image - baseline)
        total += score_input_gradient(sd, point, target_class)[0]
    return (image - baseline) * total / n_steps


def _scale_cam_image(cam: torch.Tensor, target_size=None) -> torch.Tensor:
    """pytorch_grad_cam.utils.image.scale_cam_image per image: (img - min) / (1e-7 + max), then (optionally) the bilinear
    resize cv2.resize performs (INTER_LINEAR on float32 = half-pixel centres, i.e. align_corners=False)."""
    out = []
    for img in cam:
        img = img - img.min()
        img = img / (1e-7 + img.max())
        if target_size is not None:
            img = F.interpolate(img[None, None], size=target_size, mode="bilinear", align_corners=False)[0, 0]
        out.append(img)
    return torch.stack(out)


def grad_cam(sd, x: torch.Tensor, target_class: int, size: int = CLASSIFIER_IMAGE_SIZE) -> torch.Tensor:
    """pytorch_grad_cam.GradCAM(model, target_layers=[model.layer4[-1].conv2]) with ClassifierOutputTarget(target_class)
    on the pre-processed image, as called at XAI.py:2992-3020: activations A = the conv2 output (before bn2), gradients
    of the raw class logit w.r.t. A, weights = their spatial mean, cam = relu(sum_k w_k A_k); scale_cam_image with the
    resize to (224, 224), and the aggregation's second scale_cam_image.  Returns [B, 224, 224] in [0, 1]."""
    p = "model."
    with torch.enable_grad():
        h = preprocess_for_classifier(x)
        h = F.conv2d(h, sd[p + "conv1.weight"], None, stride=2, padding=3)
        h = F.relu(_bn(sd, p + "bn1", h))
        h = F.max_pool2d(h, kernel_size=3, stride=2, padding=1)
        A = None
        for l in range(4):
            for j in range(2):
                stride = 2 if (l > 0 and j == 0) else 1
                base = f"{p}layer{l + 1}.{j}"
                identity = h
                out = F.conv2d(h, sd[base + ".conv1.weight"], None, stride=stride, padding=1)
                out = F.relu(_bn(sd, base + ".bn1", out))
                out = F.conv2d(out, sd[base + ".conv2.weight"], None, stride=1, padding=1)
                if l == 3 and j == 1:
                    A = out.detach().clone().requires_grad_(True)       # the hooked activation
                    out = A
                out = _bn(sd, base + ".bn2", out)
                if base + ".downsample.0.weight" in sd:
                    identity = F.conv2d(h, sd[base + ".downsample.0.weight"], None, stride=stride)
                    identity = _bn(sd, base + ".downsample.1", identity)
                h = F.relu(out + identity)
        feat = F.adaptive_avg_pool2d(h, 1).flatten(1)
        logits = F.linear(feat, sd[p + "fc.weight"], sd[p + "fc.bias"])
        (G,) = torch.autograd.grad(logits[:, target_class].sum(), A)
    weights = G.mean(dim=(2, 3), keepdim=True)
    cam = torch.clamp((weights * A.detach()).sum(dim=1), min=0)                 # [B, 7, 7]
    cam = _scale_cam_image(cam, (size, size))                                   # compute_cam_per_layer
    cam = torch.clamp(cam, min=0)
    return _scale_cam_image(cam)                                                # aggregate_multi_layers


def expand_patch_mask(patch_mask: torch.Tensor, H: int, W: int, patch: int) -> torch.Tensor:
    """XAI.py:1149-1157: boolean patch grid -> boolean pixel mask (pixels beyond the grid stay False)."""
    full = torch.zeros(H, W, dtype=torch.bool)
    nh, nw = patch_mask.shape
    for i in range(nh):
        for j in range(nw):
            if patch_mask[i, j]:
                full[i * patch:(i + 1) * patch, j * patch:(j + 1) * patch] = True
    return full


@torch.no_grad()
def shap_approximation(sd, image: torch.Tensor, target_class: int, n_samples: int = 512, patch_size: int = 16,
                       patch_masks: Optional[torch.Tensor] = None) -> torch.Tensor:
    """compute_shap_approximation (XAI.py:1111-1177).  ``patch_masks`` (bool [n_samples, nh, nw]) replaces the
    reference's ``torch.rand(nh, nw) > 0.5`` draws from the global CPU RNG so both sides see the same masks."""
    B, C, H, W = image.shape
    nh, nw = H // patch_size, W // patch_size
    attribution = torch.zeros_like(image)
    baseline_score = class_scores(sd, torch.zeros_like(image), target_class)[1].item()
    for s in range(n_samples):
        pm = patch_masks[s] if patch_masks is not None else (torch.rand(nh, nw) > 0.5)
        full = expand_patch_mask(pm, H, W, patch_size)
        masked = image.clone()
        masked[:, :, ~full] = 0
        score = class_scores(sd, masked, target_class)[1].item()
        attribution += (score - baseline_score) * full.unsqueeze(0).unsqueeze(0).float()
    return attribution / n_samples


@torch.no_grad()
def time_shap_as_coded(sd, trajectory: Sequence[torch.Tensor], timesteps: Sequence[float], target_class: int):
    """compute_time_shap (XAI.py:1179-1234): per-frame log-score, min-max normalised (uniform if flat)."""
    conf, prob = [], []
    for image in trajectory:
        p, s = class_scores(sd, image, target_class)
        prob.append(p.item())
        conf.append(s.item())
    conf = np.array(conf)
    prob = np.array(prob)
    if len(conf) > 1 and (conf.max() - conf.min()) > 1e-6:
        imp = (conf - conf.min()) / (conf.max() - conf.min())
    else:
        imp = np.ones_like(conf) / len(conf)
    return imp, {"confidence_scores": conf, "probability_scores": prob, "timesteps": list(timesteps)}
