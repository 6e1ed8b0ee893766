"""Seeded synthetic weights (host-side data generation only).

The reference's trained checkpoints ``unet_{CLASS}_best.pth`` are fetched from
Google Drive by download_models.py:15-59 and are not available offline, so
tests, the smoke run and the benchmark use random-initialised weights of the
reference architecture.  The recipe is fixed here once ("fixture descriptor"):

  * one CPU ``torch.Generator`` seeded with ``seed`` walks the tensors in
    ``unet_param_spec`` order;
  * conv / linear weights and biases ~ U(-1/sqrt(fan_in), +1/sqrt(fan_in))
    (the bound torch's default ``kaiming_uniform_(a=sqrt(5))`` produces);
  * GroupNorm weight = 1 + 0.1*N(0,1), bias = 0.1*N(0,1) so the affine part of
    every normalisation is exercised (a default-initialised 1/0 would hide it).

``state_dict_sha256`` fingerprints the result so fixtures can detect drift.
"""
from __future__ import annotations

import hashlib
import math
from collections import OrderedDict
from typing import Dict

import torch

from .arch import UNetConfig, unet_param_spec

DEFAULT_WEIGHT_SEED = 1234


def _fill(spec, seed: int) -> "OrderedDict[str, torch.Tensor]":
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    fan_in_of_layer: Dict[str, int] = {}
    for name, shape in spec.items():
        layer, kind = name.rsplit(".", 1)
        if len(shape) >= 2:                                   # conv / linear weight
            fan_in = math.prod(shape[1:])
            fan_in_of_layer[layer] = fan_in
            bound = 1.0 / math.sqrt(fan_in)
            sd[name] = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * bound
        elif layer in fan_in_of_layer:                        # bias of the conv / linear just seen
            bound = 1.0 / math.sqrt(fan_in_of_layer[layer])
            sd[name] = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * bound
        elif kind == "weight":                                # norm scale
            sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
        else:                                                 # norm shift
            sd[name] = 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
    return sd


def synthetic_unet_state_dict(seed: int = DEFAULT_WEIGHT_SEED,
                              cfg: UNetConfig = UNetConfig()) -> "OrderedDict[str, torch.Tensor]":
    return _fill(unet_param_spec(cfg), seed)


def state_dict_sha256(sd: Dict[str, torch.Tensor]) -> str:
    h = hashlib.sha256()
    for name, t in sd.items():
        h.update(name.encode())
        h.update(t.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def synthetic_resnet18_state_dict(seed: int = 4321, num_classes: int = 7) -> "OrderedDict[str, torch.Tensor]":
    """Seeded random classifier (``IMAGENET1K_V1`` weights are a network download, XAI.py:389, unavailable
    offline).  Conv/fc ~ U(+-sqrt(3/fan_in)) (unit-gain so activations stay O(1) through 18 layers),
    BatchNorm weight 1+0.1N, bias 0.1N, running_mean 0.1N, running_var 1+0.2|N|."""
    from .arch import resnet18_param_spec
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in resnet18_param_spec(num_classes).items():
        if len(shape) >= 2:
            bound = math.sqrt(3.0 / math.prod(shape[1:]))
            sd[name] = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * bound
        elif name.endswith("running_var"):
            sd[name] = 1.0 + 0.2 * torch.randn(shape, generator=g, dtype=torch.float32).abs()
        elif name.endswith("running_mean") or name.endswith(".bias"):
            sd[name] = 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
        else:
            sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
    return sd
