"""HipUNet2DModel -- drop-in for the ``diffusers.UNet2DModel`` object the reference samples with.

It mirrors exactly what the reference's callers touch:

    model = UNet2DModel(sample_size=128, in_channels=3, ...)      model_manager.py:173-194
    model.load_state_dict(torch.load(path, map_location=dev))     model_manager.py:138-139 (strict)
    model = model.to(dev); model.eval()                           model_manager.py:142-143
    str(model.device); next(model.parameters()).device; model.training      image_generator.py:347-358,
                                                                            model_manager.py:286-313
    noise_pred = model(latents, t).sample                         image_generator.py:400

All arithmetic runs in libsisic_hip.so (``sisic_unet_forward``); this class only owns
the handle, keeps the state dict for ``state_dict()/parameters()`` and translates
arguments.  It raises if the model is asked to run anywhere but on an MI355X.
"""
from __future__ import annotations

import ctypes as C
import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Iterator, Optional, Sequence, Union

import torch

from . import _lib
from ._lib import UNetConfigC, check
from .arch import UNetConfig, normalize_state_dict_keys, unet_param_spec
from . import ops


@dataclass
class UNet2DOutput:
    """Mirror of ``diffusers.models.unets.unet_2d.UNet2DOutput``."""
    sample: torch.Tensor


def timestep_frequencies(dim: int) -> torch.Tensor:
    """fp32 frequency table of ``get_timestep_embedding`` (max_period 10000, shift 0), computed with
    the same torch CPU ops the reference's diffusers code uses, then handed to the library."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(start=0, end=half, dtype=torch.float32)
    exponent = exponent / (half - 0)
    return torch.exp(exponent)


class _ParameterView:
    """``model.parameters()``: an iterator over the tensors (``next(model.parameters()).device``,
    image_generator.py:347-358) that also knows its model (``Adam(model.parameters(), lr)`` in the reference's training
    loop becomes ``HipAdam(model.parameters(), lr)``)."""

    def __init__(self, model, tensors):
        self._it = iter(tensors)
        self._sisic_model = model

    def __iter__(self):
        return self

    def __next__(self):
        return next(self._it)


class HipUNet2DModel:
    def __init__(self, sample_size: int = 128, in_channels: int = 3, out_channels: int = 3,
                 layers_per_block: int = 2, block_out_channels: Sequence[int] = (64, 128, 256, 256),
                 down_block_types: Sequence[str] = ("DownBlock2D", "DownBlock2D", "AttnDownBlock2D", "DownBlock2D"),
                 up_block_types: Sequence[str] = ("UpBlock2D", "AttnUpBlock2D", "UpBlock2D", "UpBlock2D"),
                 class_embed_type=None, norm_num_groups: int = 32, norm_eps: float = 1e-5,
                 attention_head_dim: int = 8, **unsupported):
        if class_embed_type is not None:
            raise NotImplementedError("class_embed_type must be None (the reference's models are unconditional)")
        if unsupported:
            raise NotImplementedError(f"unsupported UNet2DModel arguments: {sorted(unsupported)}")
        self.config = UNetConfig(sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
                                 layers_per_block=layers_per_block, block_out_channels=tuple(block_out_channels),
                                 down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
                                 norm_num_groups=norm_num_groups, norm_eps=norm_eps,
                                 attention_head_dim=attention_head_dim)
        self.config.validate()
        if attention_head_dim != 8:
            raise NotImplementedError("attention_head_dim must be 8 (the HIP attention kernel is built for d=8)")
        self._spec = unet_param_spec(self.config)
        self._params: "OrderedDict[str, torch.Tensor]" = OrderedDict()   # on self._device
        self._device = torch.device("cpu")
        self._handle: Optional[C.c_void_p] = None
        self._uploaded = False
        self.training = True                  # nn.Module default until .eval()
        self._train_begun = False             # gradient / Adam arenas exist in the library (HipAdam creates them)
        self._tape_input = None               # input of the last training-mode forward (the tape points into it)
        self._latency_mode = False
        self._params_stale = False            # the library's weights have moved on (optimizer steps) since _params was read

    # ------------------------------------------------------------------ nn.Module surface
    @property
    def device(self) -> torch.device:
        return self._device

    @property
    def dtype(self) -> torch.dtype:
        return torch.float32

    def eval(self) -> "HipUNet2DModel":
        self.training = False
        return self

    def train(self, mode: bool = True) -> "HipUNet2DModel":
        """``model.train()`` (diffusion/train_diffusion.py:209).  The network has no dropout or batch statistics, so the
        mode only decides whether a call records the tape for ``loss.backward()`` (once an optimizer exists, synt_isic_amd.train)."""
        self.training = bool(mode)
        return self

    def requires_grad_(self, flag: bool = True) -> "HipUNet2DModel":
        return self

    def set_latency_mode(self, on: bool = True) -> "HipUNet2DModel":
        """Kernel choices for single-image latency (sisic_unet_set_latency_mode): the reference samples one image at a
        time (image_generator.py:379).  Results are batch-independent within a mode and differ in the last bits between
        modes, so a sampler should stay in one mode."""
        self._latency_mode = bool(on)
        if self._handle is not None:
            check(_lib.load().sisic_unet_set_latency_mode(self._handle, int(self._latency_mode)))
        return self

    def set_graph_mode(self, mode: int = 1) -> "HipUNet2DModel":
        """sisic_sample as a replayed hipGraph: 1 on, 0 off, -1 follow the latency mode (the default)."""
        check(_lib.load().sisic_unet_set_graph_mode(self.handle, int(mode)))
        return self

    def _ensure_training(self) -> None:
        """allocate the gradient and Adam arenas in the library (sisic_unet_train_begin), once"""
        if not self._train_begun:
            check(_lib.load().sisic_unet_train_begin(self.handle))
            self._train_begun = True

    def _refresh_params(self) -> None:
        """after optimizer steps: read the current weights back from the library"""
        if not self._params_stale or self._handle is None:
            return
        new = OrderedDict()
        for name, t in self._read_all(0).items():
            new[name] = t.to(self._device)
        self._params = new
        self._params_stale = False

    def _read_all(self, what: int) -> "OrderedDict[str, torch.Tensor]":
        lib = _lib.load()
        h = self.handle
        out = OrderedDict()
        index = {lib.sisic_unet_tensor_name(h, i).decode(): i for i in range(lib.sisic_unet_num_tensors(h))}
        for name, shape in self._spec.items():
            t = torch.empty(tuple(shape), dtype=torch.float32)
            check(lib.sisic_unet_read(h, what, index[name], C.cast(t.data_ptr(), _lib.c_float_p), t.numel()))
            out[name] = t
        return out

    def grads(self) -> "OrderedDict[str, torch.Tensor]":
        """{name: d loss / d parameter} of the last backward pass, on the host (what ``p.grad`` holds in the reference)."""
        if not self._train_begun:
            raise RuntimeError("no backward pass has run: create a HipAdam for this model first")
        return self._read_all(1)

    def optimizer_state(self) -> Dict[str, "OrderedDict[str, torch.Tensor]"]:
        return {"exp_avg": self._read_all(2), "exp_avg_sq": self._read_all(3),
                "step": int(_lib.load().sisic_unet_train_steps(self.handle))}

    def parameters(self):
        self._refresh_params()
        return _ParameterView(self, list(self._params.values()))

    def named_parameters(self):
        self._refresh_params()
        return iter(self._params.items())

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        self._refresh_params()
        return OrderedDict((k, v) for k, v in self._params.items())

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        """Strict load of a flat ``{name: tensor}`` dict with diffusers key names."""
        sd = normalize_state_dict_keys(dict(state_dict))
        missing = [k for k in self._spec if k not in sd]
        unexpected = [k for k in sd if k not in self._spec]
        if missing or (unexpected and strict):
            raise RuntimeError(f"Error(s) in loading state_dict for HipUNet2DModel: missing keys {missing[:5]}"
                               f"{'...' if len(missing) > 5 else ''} ({len(missing)}), unexpected keys "
                               f"{unexpected[:5]}{'...' if len(unexpected) > 5 else ''} ({len(unexpected)})")
        new = OrderedDict()
        for name, shape in self._spec.items():
            t = sd[name]
            if tuple(t.shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {name}: checkpoint {tuple(t.shape)} vs model {tuple(shape)}")
            new[name] = t.detach().to(device=self._device, dtype=torch.float32).contiguous().clone()
        self._params = new
        self._uploaded = False
        self._params_stale = False
        if self._device.type == "cuda":
            self._upload()
        return self

    def to(self, device=None, *args, **kwargs) -> "HipUNet2DModel":
        if device is None:
            return self
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if device == self._device:
            return self
        self._refresh_params()
        self._release()
        self._params = OrderedDict((k, v.to(device)) for k, v in self._params.items())
        self._device = device
        if device.type == "cuda" and self._params:
            self._upload()
        return self

    def cuda(self, index: Optional[int] = None) -> "HipUNet2DModel":
        return self.to(torch.device("cuda", index if index is not None else torch.cuda.current_device()))

    def cpu(self) -> "HipUNet2DModel":
        return self.to("cpu")

    # ------------------------------------------------------------------ library handle
    def _create_handle(self) -> None:
        lib = _lib.load()
        cfg = self.config
        c = UNetConfigC()
        c.in_channels, c.out_channels = cfg.in_channels, cfg.out_channels
        c.layers_per_block = cfg.layers_per_block
        c.n_blocks = len(cfg.block_out_channels)
        for i, ch in enumerate(cfg.block_out_channels):
            c.block_out_channels[i] = ch
            c.down_attn[i] = int(cfg.down_has_attn[i])
            c.up_attn[i] = int(cfg.up_has_attn[i])
        c.norm_groups, c.norm_eps, c.head_dim = cfg.norm_num_groups, cfg.norm_eps, cfg.attention_head_dim
        freqs = timestep_frequencies(cfg.block_out_channels[0]).contiguous()
        c.n_freqs = freqs.numel()
        c.freqs = C.cast(freqs.data_ptr(), _lib.c_float_p)
        h = C.c_void_p()
        check(lib.sisic_unet_create(ops.context(self._device), C.byref(c), C.byref(h)))
        self._handle = h
        if self._latency_mode:
            check(lib.sisic_unet_set_latency_mode(h, 1))
        # the library's own view of the expected keys must agree with ours
        n = lib.sisic_unet_num_tensors(h)
        names = [lib.sisic_unet_tensor_name(h, i).decode() for i in range(n)]
        if sorted(names) != sorted(self._spec):
            raise RuntimeError("libsisic_hip.so and synt_isic_amd.arch disagree on the state-dict keys")

    def _upload(self) -> None:
        if self._handle is None:
            self._create_handle()
        lib = _lib.load()
        host = [(k, v.detach().to("cpu", torch.float32).contiguous()) for k, v in self._params.items()]
        n = len(host)
        names = (C.c_char_p * n)(*[k.encode() for k, _ in host])
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for _, t in host])
        numels = (C.c_int64 * n)(*[t.numel() for _, t in host])
        check(lib.sisic_unet_load(self._handle, n, names, ptrs, numels))
        self._uploaded = True
        if self._train_begun:                 # new weights under an existing optimizer: fresh moments, like a new Adam
            check(lib.sisic_unet_train_begin(self._handle))

    def _release(self) -> None:
        if self._handle is not None:
            _lib.load().sisic_unet_destroy(self._handle)
            self._handle = None
            self._uploaded = False
            self._train_begun = False

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    @property
    def handle(self) -> C.c_void_p:
        if self._device.type != "cuda":
            raise RuntimeError("HipUNet2DModel runs on MI355X only: call .to('cuda') first (there is no CPU path)")
        if not self._params:
            raise RuntimeError("HipUNet2DModel has no weights: call load_state_dict() first")
        if not self._uploaded:
            self._upload()
        return self._handle

    # ------------------------------------------------------------------ forward
    def _timesteps_host(self, timestep, batch: int) -> torch.Tensor:
        if not torch.is_tensor(timestep):
            t = torch.tensor([timestep], dtype=torch.int64)
        else:
            t = timestep.detach().to("cpu").to(torch.int64).reshape(-1)
        if t.numel() == 1:
            t = t.expand(batch)
        elif t.numel() != batch:
            raise ValueError(f"timestep has {t.numel()} entries for a batch of {batch}")
        return t.contiguous()

    @torch.no_grad()
    def __call__(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int],
                 class_labels=None, return_dict: bool = True):
        if class_labels is not None:
            raise NotImplementedError("class conditioning is not part of the reference's models")
        h = self.handle
        if sample.device != self._device:
            raise RuntimeError(f"sample is on {sample.device} but the model is on {self._device}")
        if sample.dim() != 4 or sample.shape[1] != self.config.in_channels:
            raise ValueError(f"sample must be [B,{self.config.in_channels},H,W], got {tuple(sample.shape)}")
        x = sample.to(torch.float32).contiguous()
        B, _, H, W = x.shape
        t = self._timesteps_host(timestep, B)
        out = torch.empty((B, self.config.out_channels, H, W), dtype=torch.float32, device=x.device)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        if self.training and self._train_begun:
            # training mode with an optimizer: the same kernels, every activation kept for loss.backward()
            check(_lib.load().sisic_unet_train_forward(h, x.data_ptr(), C.cast(t.data_ptr(), _lib.c_int64_p),
                                                       out.data_ptr(), B, H, W, stream))
            out._sisic_model = self
            # the tape refers to the input by address (conv_in's weight gradient reads it in the backward pass): keep the
            # tensor alive until the next forward, as autograd's graph would
            self._tape_input = x
        else:
            check(_lib.load().sisic_unet_forward(h, x.data_ptr(), C.cast(t.data_ptr(), _lib.c_int64_p), out.data_ptr(),
                                                 B, H, W, stream))
        if not return_dict:
            return (out,)
        return UNet2DOutput(sample=out)

    forward = __call__
