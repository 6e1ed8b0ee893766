"""Batched forward passes of the reference's explainability analyses (xai/XAI.py), on the HIP kernels.

* ``compute_time_shap``            -- XAI.py:1179-1234 as coded: per-frame classifier scores, min-max
                                     normalised.  The reference runs 2 batch-1 forwards per frame; here the N
                                     frames are ONE [N,3,H,W] batch.
* ``compute_shap_approximation``   -- XAI.py:1111-1177: 512 random 16x16-patch coalitions x classifier score.
                                     The reference runs 513 batch-1 forwards per image; here the coalition
                                     images are built on the GPU (``sisic_mask_patches``) and scored in batches.
* ``compute_integrated_gradients`` -- XAI.py:1039-1084: captum IntegratedGradients (n_steps = 50, ``riemann_right``) of
                                     the per-class score; all Riemann points go through ONE batched backward-to-input
                                     pass of the HIP classifier (``sisic_resnet_input_gradient``), no autograd.
* ``compute_grad_cam``             -- XAI.py:2945-3134: Grad-CAM on ``layer4[-1].conv2`` for every trajectory frame (ONE
                                     batch instead of a GradCAM call per frame) + the normalised mean map.
* ``compute_gradient_attribution`` -- XAI.py:1086-1109: the plain input gradient (the reference's fallback).
* ``time_shap_permutation``        -- README.md:171-207: Shapley values of the denoising STEPS with
                                     v(S) = F(Dec(x_T; S)) (transitions applied only on the steps in S, the other
                                     steps are skipped) and the unbiased permutation estimator.  The reference
                                     documents this form but does not implement it.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, ops
from ._lib import check
from .classifier import HipMelanomaClassifier
from .sampler import Sampler, draw_noise, run_sampling_loop

SHAP_N_SAMPLES = 512          # xai/XAI.py SHAP_N_SAMPLES


def _as_batch(trajectory) -> torch.Tensor:
    if torch.is_tensor(trajectory):
        t = trajectory
        return t.reshape((-1,) + tuple(t.shape[-3:])) if t.dim() == 5 else t
    return torch.cat([f if f.dim() == 4 else f.unsqueeze(0) for f in trajectory], dim=0)


@torch.no_grad()
def compute_time_shap(classifier: HipMelanomaClassifier, trajectory, timesteps: Sequence[float], target_class: int):
    """(normalized_importance, raw_data) exactly as XAI.py:1179-1234 returns them."""
    frames = _as_batch(trajectory)
    if frames.shape[0] != len(timesteps):
        raise ValueError(f"{frames.shape[0]} frames but {len(timesteps)} timesteps")
    prob, logscore = classifier._scores(frames, target_class)
    confidence_scores = logscore.cpu().numpy().astype(np.float64)
    prob_scores = prob.cpu().numpy().astype(np.float64)
    if len(confidence_scores) > 1 and (confidence_scores.max() - confidence_scores.min()) > 1e-6:
        normalized = (confidence_scores - confidence_scores.min()) / (confidence_scores.max() - confidence_scores.min())
    else:
        normalized = np.ones_like(confidence_scores) / len(confidence_scores)
    raw = {"confidence_scores": confidence_scores, "probability_scores": prob_scores, "timesteps": list(timesteps)}
    return normalized, raw


IG_N_STEPS = 50               # xai/XAI.py:240


def make_baseline(image: torch.Tensor, baseline_type: str = "noise", generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """XAI.py:1010-1037: 'noise' = randn_like(image) * 0.1 (drawn on the CPU here so that it can be seeded), 'zero',
    'blur' = avg_pool2d(kernel 31, stride 1, padding 15); anything else = zero."""
    if baseline_type == "noise":
        return (torch.randn(image.shape, generator=generator, dtype=torch.float32) * 0.1).to(image.device)
    if baseline_type == "blur":
        return torch.nn.functional.avg_pool2d(image, kernel_size=31, stride=1, padding=15)
    return torch.zeros_like(image)


@torch.no_grad()
def compute_gradient_attribution(classifier: HipMelanomaClassifier, image: torch.Tensor, target_class: int) -> torch.Tensor:
    """XAI.py:1086-1109: d get_per_class_score / d image."""
    return classifier.input_gradient(image, target_class)[0]


@torch.no_grad()
def compute_integrated_gradients(classifier: HipMelanomaClassifier, image: torch.Tensor, target_class: int,
                                 n_steps: int = IG_N_STEPS, baseline: Optional[torch.Tensor] = None,
                                 baseline_type: str = "noise", generator: Optional[torch.Generator] = None,
                                 max_batch: int = 128) -> torch.Tensor:
    """XAI.py:1039-1084 (captum ``IntegratedGradients.attribute(image, baselines, n_steps, method='riemann_right')``
    with ``forward_func = get_per_class_score``):
        IG(x) = (x - x') * (1/n) * sum_{k=1..n} grad score(x' + (k/n)(x - x'))
    image: [B,3,H,W]; the n*B Riemann points are differentiated in batches of ``max_batch`` images."""
    image = image.to(classifier.device).to(torch.float32)
    if baseline is None:
        baseline = make_baseline(image, baseline_type, generator)
    baseline = baseline.to(image.device).to(torch.float32)
    B = image.shape[0]
    alphas = torch.arange(1, n_steps + 1, dtype=torch.float32, device=image.device) / n_steps
    diff = image - baseline
    total = torch.zeros_like(image)
    per = max(1, max_batch // B)                       # Riemann points per pass
    for k0 in range(0, n_steps, per):
        a = alphas[k0:k0 + per].view(-1, 1, 1, 1, 1)
        pts = (baseline.unsqueeze(0) + a * diff.unsqueeze(0)).reshape((-1,) + tuple(image.shape[1:]))
        g = classifier.input_gradient(pts, target_class)[0]
        total += g.view((-1, B) + tuple(image.shape[1:])).sum(0)
    return diff * total / n_steps


@torch.no_grad()
def compute_grad_cam(classifier: HipMelanomaClassifier, trajectory, timesteps: Sequence[float], target_class: int) -> Dict:
    """XAI.py:2945-3134: {"t_<timestep>": cam (224,224) numpy in [0,1]} for every frame, and "summary": the mean of the
    frames' maps min-max normalised with eps 1e-8 (``gradcam_summary``, :3102-3104)."""
    frames = _as_batch(trajectory)
    if frames.shape[0] != len(timesteps):
        raise ValueError(f"{frames.shape[0]} frames but {len(timesteps)} timesteps")
    cams = classifier.grad_cam(frames, target_class)[0].cpu().numpy()
    out = {f"t_{float(t):.0f}": cams[i] for i, t in enumerate(timesteps)}
    mean = cams.mean(axis=0)
    out["summary"] = (mean - mean.min()) / (mean.max() - mean.min() + 1e-8)
    return out


def draw_patch_masks(n_samples: int, nh: int, nw: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """The reference's per-sample ``torch.rand(nh, nw) > 0.5`` draws (XAI.py:1147), in order, as one bool tensor."""
    return torch.stack([torch.rand(nh, nw, generator=generator) > 0.5 for _ in range(n_samples)])


@torch.no_grad()
def compute_shap_approximation(classifier: HipMelanomaClassifier, image: torch.Tensor, target_class: int,
                               n_samples: int = SHAP_N_SAMPLES, patch_size: int = 16,
                               patch_masks: Optional[torch.Tensor] = None, chunk: int = 256) -> torch.Tensor:
    """Attribution map [1,3,H,W] of XAI.py:1111-1177 for ONE image [1,3,H,W] in [-1,1]."""
    dev = classifier.device
    image = image.to(dev, torch.float32).contiguous()
    if image.dim() != 4 or image.shape[0] != 1:
        raise ValueError("compute_shap_approximation takes one image [1,3,H,W]")
    _, Cc, H, W = image.shape
    nh, nw = H // patch_size, W // patch_size
    if patch_masks is None:
        patch_masks = draw_patch_masks(n_samples, nh, nw)          # global CPU RNG, like the reference
    if tuple(patch_masks.shape) != (n_samples, nh, nw):
        raise ValueError(f"patch_masks must be [{n_samples},{nh},{nw}]")
    masks_u8 = patch_masks.to(torch.uint8).contiguous().to(dev)
    lib = _lib.load()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    baseline = classifier.get_per_class_score(torch.zeros_like(image), target_class)       # [1]
    scores = torch.empty(n_samples, dtype=torch.float32, device=dev)
    for s0 in range(0, n_samples, chunk):
        s1 = min(n_samples, s0 + chunk)
        batch = torch.empty((s1 - s0, Cc, H, W), dtype=torch.float32, device=dev)
        check(lib.sisic_mask_patches(ops.context(dev), image.data_ptr(), masks_u8[s0:s1].data_ptr(), batch.data_ptr(),
                                     s1 - s0, Cc, H, W, patch_size, stream))
        scores[s0:s1] = classifier.get_per_class_score(batch, target_class)
    # attribution = (1/n) * sum_s (score_s - baseline) * mask_s  -- a [n] x [n,nh,nw] contraction of ~10^4 values
    contrib = (scores - baseline).double().cpu()
    grid = torch.einsum("s,sij->ij", contrib, patch_masks.double()) / n_samples
    full = torch.zeros(H, W, dtype=torch.float64)
    full[: nh * patch_size, : nw * patch_size] = grid.repeat_interleave(patch_size, 0).repeat_interleave(patch_size, 1)
    return full.float().to(dev).expand(1, Cc, H, W).contiguous()


@torch.no_grad()
def coalition_value(sampler: Sampler, classifier: HipMelanomaClassifier, class_name: str, x_T: torch.Tensor,
                    z: torch.Tensor, T: int, coalitions: Sequence[Sequence[int]], target_class: int) -> torch.Tensor:
    """v(S) for every coalition S (step indices into the T-step grid): the mean over the batch of the target
    logit after a denoising run that applies only the steps in S.  x_T [B,3,H,W], z [n_noise,B,3,H,W] on the
    GPU; the same x_T / z_t serve every coalition, so v is a deterministic set function."""
    model = sampler.models[class_name]
    sched_full = sampler.create_scheduler(T)
    ts_full = sched_full.timesteps.clone()
    noise_index = {}
    k = 0
    for i, t in enumerate(ts_full.tolist()):
        if t > 0:
            noise_index[i] = k
            k += 1
    finals = []
    for S in coalitions:
        steps = sorted(set(int(s) for s in S))
        if steps:
            sched = sampler.create_scheduler(T)
            sched.timesteps = ts_full[steps]                       # descending t, standard coefficients per step
            zi = [noise_index[i] for i in steps if i in noise_index]
            zs = z[zi].contiguous() if zi else None
            finals.append(run_sampling_loop(model, sched, x_T, zs).latents)
        else:
            finals.append(x_T)
    # ONE classifier batch for all coalitions ([n_coalitions * B, 3, H, W]: 16 x 32 = 512 forwards in BASELINE config 5)
    B = x_T.shape[0]
    logits = classifier.forward(torch.cat(finals, dim=0))
    return logits[:, target_class].view(len(finals), B).mean(dim=1)


@torch.no_grad()
def time_shap_permutation(sampler: Sampler, classifier: HipMelanomaClassifier, class_name: str, seeds: Sequence[int],
                          T: int, target_class: int, n_permutations: int = 4, size: Tuple[int, int] = (64, 64),
                          generator: Optional[torch.Generator] = None) -> Dict[str, np.ndarray]:
    """Permutation estimator of README.md:198-207: phi_t = mean_m [ v(Pref_pi_m(t) u {t}) - v(Pref_pi_m(t)) ].
    Per permutation the T+1 nested prefixes give every marginal, so efficiency holds exactly:
    sum_t phi_t = v({all steps}) - v({})."""
    H, W = size
    sched = sampler.create_scheduler(T)
    n_noise = sum(1 for t in sched.timesteps if int(t) > 0)
    x_T, z = draw_noise(seeds, n_noise, (3, H, W))
    x_T, z = x_T.to(sampler.device), z.to(sampler.device)
    phi = np.zeros(T, dtype=np.float64)
    v_full = v_empty = None
    for _ in range(n_permutations):
        perm = torch.randperm(T, generator=generator).tolist()
        coalitions = [perm[:k] for k in range(T + 1)]
        v = coalition_value(sampler, classifier, class_name, x_T, z, T, coalitions, target_class).double().cpu().numpy()
        for k, step in enumerate(perm):
            phi[step] += v[k + 1] - v[k]
        v_empty, v_full = v[0], v[T]
    phi /= n_permutations
    return {"phi": phi, "v_full": np.float64(v_full), "v_empty": np.float64(v_empty),
            "timesteps": np.array([int(t) for t in sched.timesteps])}
