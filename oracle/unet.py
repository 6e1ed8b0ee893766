"""CPU oracle: the UNet2DModel forward pass the reference samples with.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- "parity unpinned": the
reference obtains this network from the third-party ``diffusers`` package
(``diffusers>=0.21.0``, requirements.txt:6 -- un-pinned, not vendored, absent
from this container), so this file restates the published ``UNet2DModel``
algorithm for exactly the configuration the reference passes at
core/generator/model_manager.py:173-194 (identical copies at
core/generator/image_generator.py:268-287, diffusion/train_diffusion.py:118-138,
xai/XAI.py:313-339):

    sample_size=128, in_channels=3, out_channels=3, layers_per_block=2,
    block_out_channels=(64, 128, 256, 256),
    down_block_types=(Down, Down, AttnDown, Down),
    up_block_types=(Up, AttnUp, Up, Up), class_embed_type=None

with every other constructor argument at its diffusers default (SURVEY.md
Appendix A): act_fn="silu", norm_num_groups=32, norm_eps=1e-5,
attention_head_dim=8, resnet_time_scale_shift="default" (ADDITIVE time
embedding, not FiLM -- Appendix A.4), flip_sin_to_cos=True, freq_shift=0,
downsample_padding=1, dropout=0, add_attention=True.

The network is expressed as a pure function of a flat ``{name: fp32 tensor}``
state dict that uses the diffusers key names (Appendix A.6), which is the
checkpoint format the reference loads with a strict ``load_state_dict``
(core/generator/model_manager.py:135-143).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

# Configuration fixed by core/generator/model_manager.py:175-194.
BLOCK_OUT_CHANNELS: Tuple[int, ...] = (64, 128, 256, 256)
LAYERS_PER_BLOCK = 2
DOWN_HAS_ATTN = (False, False, True, False)   # DownBlock2D, DownBlock2D, AttnDownBlock2D, DownBlock2D
UP_HAS_ATTN = (False, True, False, False)     # UpBlock2D, AttnUpBlock2D, UpBlock2D, UpBlock2D
IN_CHANNELS = 3
OUT_CHANNELS = 3
NORM_GROUPS = 32
NORM_EPS = 1e-5
HEAD_DIM = 8
TIME_EMBED_DIM = 4 * BLOCK_OUT_CHANNELS[0]     # 256
EXPECTED_NUM_PARAMS = 25_304_963               # SURVEY.md A.7 (matches cache_metadata.json sizes)
EXPECTED_NUM_TENSORS = 330


def param_spec() -> "OrderedDict[str, Tuple[int, ...]]":
    """Names and shapes of the 330 state-dict tensors (diffusers naming, Appendix A.6)."""
    spec: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    boc = BLOCK_OUT_CHANNELS
    temb = TIME_EMBED_DIM

    def conv(name, cout, cin, k):
        spec[f"{name}.weight"] = (cout, cin, k, k)
        spec[f"{name}.bias"] = (cout,)

    def linear(name, cout, cin):
        spec[f"{name}.weight"] = (cout, cin)
        spec[f"{name}.bias"] = (cout,)

    def norm(name, c):
        spec[f"{name}.weight"] = (c,)
        spec[f"{name}.bias"] = (c,)

    def resnet(name, cin, cout):
        norm(f"{name}.norm1", cin)
        conv(f"{name}.conv1", cout, cin, 3)
        linear(f"{name}.time_emb_proj", cout, temb)
        norm(f"{name}.norm2", cout)
        conv(f"{name}.conv2", cout, cout, 3)
        if cin != cout:
            conv(f"{name}.conv_shortcut", cout, cin, 1)

    def attention(name, c):
        norm(f"{name}.group_norm", c)
        linear(f"{name}.to_q", c, c)
        linear(f"{name}.to_k", c, c)
        linear(f"{name}.to_v", c, c)
        linear(f"{name}.to_out.0", c, c)

    conv("conv_in", boc[0], IN_CHANNELS, 3)
    linear("time_embedding.linear_1", temb, boc[0])
    linear("time_embedding.linear_2", temb, temb)

    out_ch = boc[0]
    for i, ch in enumerate(boc):
        in_ch, out_ch = out_ch, ch
        for j in range(LAYERS_PER_BLOCK):
            resnet(f"down_blocks.{i}.resnets.{j}", in_ch if j == 0 else out_ch, out_ch)
            if DOWN_HAS_ATTN[i]:
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
        prev_out = out_ch
        out_ch = rev[i]
        in_ch = rev[min(i + 1, len(rev) - 1)]
        n_layers = LAYERS_PER_BLOCK + 1
        for j in range(n_layers):
            skip_ch = in_ch if j == n_layers - 1 else out_ch
            res_in = prev_out if j == 0 else out_ch
            resnet(f"up_blocks.{i}.resnets.{j}", res_in + skip_ch, out_ch)
            if UP_HAS_ATTN[i]:
                attention(f"up_blocks.{i}.attentions.{j}", out_ch)
        if i != len(rev) - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", out_ch, out_ch, 3)

    norm("conv_norm_out", boc[0])
    conv("conv_out", OUT_CHANNELS, boc[0], 3)
    return spec


def num_params() -> int:
    return sum(math.prod(s) for s in param_spec().values())


def timestep_frequencies(dim: int = BLOCK_OUT_CHANNELS[0]) -> torch.Tensor:
    """fp32 frequency table of diffusers' ``get_timestep_embedding`` (max_period 1e4, shift 0)."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(start=0, end=half, dtype=torch.float32)
    exponent = exponent / (half - 0)
    return torch.exp(exponent)


def timestep_embedding(timesteps: torch.Tensor, dim: int = BLOCK_OUT_CHANNELS[0]) -> torch.Tensor:
    """Sinusoidal embedding, cos half first (flip_sin_to_cos=True). SURVEY.md A.1."""
    emb = timesteps[:, None].float() * timestep_frequencies(dim)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def _broadcast_t(timestep, batch: int) -> torch.Tensor:
    t = torch.as_tensor(timestep)
    if t.dim() == 0:
        t = t[None]
    return t.to(torch.int64).expand(batch) if t.numel() == 1 else t.to(torch.int64).reshape(batch)


def group_norm(x, w, b, silu: bool):
    y = F.group_norm(x, NORM_GROUPS, w, b, eps=NORM_EPS)
    return F.silu(y) if silu else y


def resnet_block(sd: Dict[str, torch.Tensor], p: str, x: torch.Tensor, temb_act: torch.Tensor) -> torch.Tensor:
    """ResnetBlock2D, time_embedding_norm="default" (additive).  SURVEY.md A.3."""
    h = group_norm(x, sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], True)
    h = F.conv2d(h, sd[f"{p}.conv1.weight"], sd[f"{p}.conv1.bias"], padding=1)
    t = F.linear(temb_act, sd[f"{p}.time_emb_proj.weight"], sd[f"{p}.time_emb_proj.bias"])
    h = h + t[:, :, None, None]
    h = group_norm(h, sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], True)
    h = F.conv2d(h, sd[f"{p}.conv2.weight"], sd[f"{p}.conv2.bias"], padding=1)
    if f"{p}.conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[f"{p}.conv_shortcut.weight"], sd[f"{p}.conv_shortcut.bias"])
    return x + h            # output_scale_factor = 1.0


def attention_block(sd: Dict[str, torch.Tensor], p: str, x: torch.Tensor) -> torch.Tensor:
    """Self-attention over the H*W tokens, 32 heads x d=8.  SURVEY.md A.5."""
    B, C, H, W = x.shape
    heads = C // HEAD_DIM
    h = group_norm(x.reshape(B, C, H * W), sd[f"{p}.group_norm.weight"], sd[f"{p}.group_norm.bias"], False)
    h = h.transpose(1, 2)                                           # [B, N, C]
    q = F.linear(h, sd[f"{p}.to_q.weight"], sd[f"{p}.to_q.bias"])
    k = F.linear(h, sd[f"{p}.to_k.weight"], sd[f"{p}.to_k.bias"])
    v = F.linear(h, sd[f"{p}.to_v.weight"], sd[f"{p}.to_v.bias"])
    split = lambda z: z.reshape(B, H * W, heads, HEAD_DIM).transpose(1, 2)   # [B, heads, N, d]
    q, k, v = split(q), split(k), split(v)
    scores = torch.matmul(q, k.transpose(-1, -2)) * (HEAD_DIM ** -0.5)
    probs = torch.softmax(scores if scores.dtype == torch.float64 else scores.float(), dim=-1)   # fp32 softmax (float64 only for rounding-free test references)
    o = torch.matmul(probs, v)                                      # [B, heads, N, d]
    o = o.transpose(1, 2).reshape(B, H * W, C)
    o = F.linear(o, sd[f"{p}.to_out.0.weight"], sd[f"{p}.to_out.0.bias"])
    o = o.transpose(1, 2).reshape(B, C, H, W)
    return o + x            # residual_connection=True, rescale_output_factor=1.0


def unet_forward(sd: Dict[str, torch.Tensor], sample: torch.Tensor, timestep,
                 return_intermediates: bool = False):
    """``UNet2DModel.__call__(sample, timestep).sample``.  SURVEY.md section 3.2 / Appendix A.2.

    sample: fp32 [B,3,H,W] (H, W divisible by 8); timestep: int | 0-dim | [B] int64.
    """
    B = sample.shape[0]
    t = _broadcast_t(timestep, B)
    inter: Dict[str, torch.Tensor] = {}

    # 1. time embedding
    temb = timestep_embedding(t)
    temb = F.linear(temb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    temb = F.silu(temb)
    temb = F.linear(temb, sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
    temb_act = F.silu(temb)       # every ResBlock consumes Linear(SiLU(temb))
    inter["temb"] = temb

    # 2. conv_in
    x = F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    inter["conv_in"] = x
    skips: List[torch.Tensor] = [x]

    # 3. down
    n = len(BLOCK_OUT_CHANNELS)
    for i in range(n):
        for j in range(LAYERS_PER_BLOCK):
            x = resnet_block(sd, f"down_blocks.{i}.resnets.{j}", x, temb_act)
            if DOWN_HAS_ATTN[i]:
                x = attention_block(sd, f"down_blocks.{i}.attentions.{j}", x)
            skips.append(x)
        if i != n - 1:
            p = f"down_blocks.{i}.downsamplers.0.conv"
            x = F.conv2d(x, sd[f"{p}.weight"], sd[f"{p}.bias"], stride=2, padding=1)
            skips.append(x)
        inter[f"down{i}"] = x

    # 4. mid
    x = resnet_block(sd, "mid_block.resnets.0", x, temb_act)
    x = attention_block(sd, "mid_block.attentions.0", x)
    x = resnet_block(sd, "mid_block.resnets.1", x, temb_act)
    inter["mid"] = x

    # 5. up
    for i in range(n):
        for j in range(LAYERS_PER_BLOCK + 1):
            skip = skips.pop()
            x = torch.cat([x, skip], dim=1)           # hidden first, skip second
            x = resnet_block(sd, f"up_blocks.{i}.resnets.{j}", x, temb_act)
            if UP_HAS_ATTN[i]:
                x = attention_block(sd, f"up_blocks.{i}.attentions.{j}", x)
        if i != n - 1:
            p = f"up_blocks.{i}.upsamplers.0.conv"
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, sd[f"{p}.weight"], sd[f"{p}.bias"], padding=1)
        inter[f"up{i}"] = x
    assert not skips

    # 6. out
    x = group_norm(x, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], True)
    x = F.conv2d(x, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)
    if return_intermediates:
        return x, inter
    return x


def flops_per_image(H: int, W: int) -> Dict[str, float]:
    """Algorithmic FLOPs (2*MAC) of one forward at HxW, split as SURVEY.md section 8d does."""
    out = {"conv3x3": 0.0, "conv1x1": 0.0, "attention": 0.0}
    boc = BLOCK_OUT_CHANNELS
    spec = param_spec()
    # resolution of every conv is determined by its block index
    def res_of(name: str) -> Tuple[int, int]:
        if name.startswith("down_blocks."):
            i = int(name.split(".")[1]); d = 2 ** i
            if ".downsamplers." in name:
                d *= 2
        elif name.startswith("mid_block."):
            d = 2 ** (len(boc) - 1)
        elif name.startswith("up_blocks."):
            i = int(name.split(".")[1]); d = 2 ** (len(boc) - 1 - i)
            if ".upsamplers." in name:
                d //= 2
        else:
            d = 1
        return H // d, W // d
    for name, shape in spec.items():
        if not name.endswith(".weight"):
            continue
        h, w = res_of(name)
        if len(shape) == 4:
            co, ci, k, _ = shape
            key = "conv3x3" if k == 3 else "conv1x1"
            out[key] += 2.0 * co * ci * k * k * h * w
        elif ".attentions." in name and len(shape) == 2:
            out["attention"] += 2.0 * shape[0] * shape[1] * h * w
        if name.endswith("group_norm.weight"):
            n_tok = h * w
            out["attention"] += 2 * 2.0 * n_tok * n_tok * shape[0]      # QK^T and PV
    out["total"] = out["conv3x3"] + out["conv1x1"] + out["attention"]
    return out
