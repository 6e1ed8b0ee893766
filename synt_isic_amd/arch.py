"""Architecture description of the reference's UNet2DModel (host-side data only).

The reference builds its network at core/generator/model_manager.py:173-194
(``UNet2DModel(sample_size=128, in_channels=3, out_channels=3, layers_per_block=2,
block_out_channels=(64,128,256,256), down=(Down,Down,AttnDown,Down),
up=(Up,AttnUp,Up,Up))``).  This module turns that configuration into

  * the ordered ``{name: shape}`` table of the 330 checkpoint tensors
    (diffusers key names) -- the format a strict
    ``load_state_dict`` expects (model_manager.py:135-143);
  * the flat list of layer records the HIP library is configured with.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Tuple


@dataclass(frozen=True)
class UNetConfig:
    sample_size: int = 128
    in_channels: int = 3
    out_channels: int = 3
    layers_per_block: int = 2
    block_out_channels: Tuple[int, ...] = (64, 128, 256, 256)
    down_block_types: Tuple[str, ...] = ("DownBlock2D", "DownBlock2D", "AttnDownBlock2D", "DownBlock2D")
    up_block_types: Tuple[str, ...] = ("UpBlock2D", "AttnUpBlock2D", "UpBlock2D", "UpBlock2D")
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    attention_head_dim: int = 8
    class_embed_type: None = None

    @property
    def time_embed_dim(self) -> int:
        return 4 * self.block_out_channels[0]

    @property
    def down_has_attn(self) -> Tuple[bool, ...]:
        return tuple(t == "AttnDownBlock2D" for t in self.down_block_types)

    @property
    def up_has_attn(self) -> Tuple[bool, ...]:
        return tuple(t == "AttnUpBlock2D" for t in self.up_block_types)

    def validate(self) -> None:
        n = len(self.block_out_channels)
        if len(self.down_block_types) != n or len(self.up_block_types) != n:
            raise ValueError("block type tuples must match block_out_channels")
        for t in self.down_block_types:
            if t not in ("DownBlock2D", "AttnDownBlock2D"):
                raise ValueError(f"unsupported down block type {t}")
        for t in self.up_block_types:
            if t not in ("UpBlock2D", "AttnUpBlock2D"):
                raise ValueError(f"unsupported up block type {t}")
        for c in self.block_out_channels:
            if c % self.norm_num_groups:
                raise ValueError("channels must be divisible by norm_num_groups")


@dataclass
class ResnetDesc:
    name: str
    cin: int
    cout: int
    skip_ch: int = 0          # channels of the concatenated skip tensor (up path), 0 on the down path


def unet_param_spec(cfg: UNetConfig = UNetConfig()) -> "OrderedDict[str, Tuple[int, ...]]":
    cfg.validate()
    spec: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    boc = cfg.block_out_channels
    temb = cfg.time_embed_dim

    def conv(name, cout, cin, k):
        spec[name + ".weight"] = (cout, cin, k, k)
        spec[name + ".bias"] = (cout,)

    def linear(name, cout, cin):
        spec[name + ".weight"] = (cout, cin)
        spec[name + ".bias"] = (cout,)

    def norm(name, c):
        spec[name + ".weight"] = (c,)
        spec[name + ".bias"] = (c,)

    def resnet(name, cin, cout):
        norm(name + ".norm1", cin)
        conv(name + ".conv1", cout, cin, 3)
        linear(name + ".time_emb_proj", cout, temb)
        norm(name + ".norm2", cout)
        conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(name + ".conv_shortcut", cout, cin, 1)

    def attention(name, c):
        norm(name + ".group_norm", c)
        for proj in ("to_q", "to_k", "to_v", "to_out.0"):
            linear(f"{name}.{proj}", c, c)

    conv("conv_in", boc[0], cfg.in_channels, 3)
    linear("time_embedding.linear_1", temb, boc[0])
    linear("time_embedding.linear_2", temb, temb)
    out_ch = boc[0]
    for i, ch in enumerate(boc):
        in_ch, out_ch = out_ch, ch
        for j in range(cfg.layers_per_block):
            resnet(f"down_blocks.{i}.resnets.{j}", in_ch if j == 0 else out_ch, out_ch)
            if cfg.down_has_attn[i]:
                attention(f"down_blocks.{i}.attentions.{j}", out_ch)
        if i != len(boc) - 1:
            conv(f"down_blocks.{i}.downsamplers.0.conv", out_ch, out_ch, 3)
    mid = boc[-1]
    resnet("mid_block.resnets.0", mid, mid)
    attention("mid_block.attentions.0", mid)
    resnet("mid_block.resnets.1", mid, mid)
    rev = tuple(reversed(boc))
    out_ch = rev[0]
    for i in range(len(rev)):
        prev_out, out_ch = out_ch, rev[i]
        in_ch = rev[min(i + 1, len(rev) - 1)]
        n_layers = cfg.layers_per_block + 1
        for j in range(n_layers):
            skip_ch = in_ch if j == n_layers - 1 else out_ch
            res_in = prev_out if j == 0 else out_ch
            resnet(f"up_blocks.{i}.resnets.{j}", res_in + skip_ch, out_ch)
            if cfg.up_has_attn[i]:
                attention(f"up_blocks.{i}.attentions.{j}", out_ch)
        if i != len(rev) - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", out_ch, out_ch, 3)
    norm("conv_norm_out", boc[0])
    conv("conv_out", cfg.out_channels, boc[0], 3)
    return spec


def unet_num_params(cfg: UNetConfig = UNetConfig()) -> int:
    return sum(math.prod(s) for s in unet_param_spec(cfg).values())


# Older diffusers releases spell the attention projections differently
# (SURVEY.md Appendix A.6); checkpoints saved with them are remapped on load.
LEGACY_ATTENTION_KEYS = {
    "query": "to_q",
    "key": "to_k",
    "value": "to_v",
    "proj_attn": "to_out.0",
}


def normalize_state_dict_keys(sd: Dict[str, object]) -> Dict[str, object]:
    out = {}
    for k, v in sd.items():
        parts = k.split(".")
        if ".attentions." in k and len(parts) >= 2 and parts[-2] in LEGACY_ATTENTION_KEYS:
            parts[-2] = LEGACY_ATTENTION_KEYS[parts[-2]]
            k = ".".join(parts)
        out[k] = v
    return out


def resnet18_param_spec(num_classes: int = 7) -> "OrderedDict[str, Tuple[int, ...]]":
    """Float tensors of the reference classifier's state dict (xai/XAI.py:385-397: torchvision resnet18
    held as ``self.model`` => keys prefixed ``model.``, fc replaced by Linear(512, num_classes))."""
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
