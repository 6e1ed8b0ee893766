"""Thin torch-tensor wrappers over the single-operator entry points of libsisic_hip.so.

torch is plumbing here (device memory + the current HIP stream); every function
passes raw device pointers to the C ABI.  Tensors must be contiguous fp32 on an
MI355X (``cuda``) device.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import ConvArgs, check

_ctx_by_device = {}


def context(device: torch.device) -> C.c_void_p:
    """One ``sisic_ctx`` per device, created on first use."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError(f"synt_isic_amd runs on MI355X (torch device 'cuda'); got '{device}'. There is no CPU path.")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _ctx_by_device:
        lib = _lib.load()
        h = C.c_void_p()
        check(lib.sisic_create(idx, C.byref(h)))
        _ctx_by_device[idx] = h
    return _ctx_by_device[idx]


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t: Optional[torch.Tensor], name: str = "tensor") -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError(f"{name} must be a contiguous fp32 tensor on the GPU (got {t.dtype}, {t.device}, "
                         f"contiguous={t.is_contiguous()})")
    return t.data_ptr()


def pack_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """OIHW -> the kernel's [Cin_pad][k*k][Cout_pad] layout (sisic_conv_pack_weights)."""
    lib = _lib.load()
    cout, cin, k, k2 = w.shape
    assert k == k2
    n = lib.sisic_conv_packed_numel(cout, cin, k)
    if n < 0:
        raise ValueError(f"unsupported conv weight shape {tuple(w.shape)}")
    out = torch.empty(n, dtype=torch.float32, device=w.device)
    check(lib.sisic_conv_pack_weights(context(w.device), _ptr(w, "weight"), cout, cin, k, out.data_ptr(),
                                      _stream(w.device)))
    return out


def pack_winograd_weight(w: torch.Tensor) -> torch.Tensor:
    """OIHW 3x3 -> Winograd-domain filters G g G^T, [Cin_pad][16][Cout_pad] (sisic_conv_winograd_pack)."""
    lib = _lib.load()
    cout, cin, k, k2 = w.shape
    if k != 3 or k2 != 3:
        raise ValueError("Winograd F(2x2,3x3) needs a 3x3 weight")
    out = torch.empty(lib.sisic_conv_winograd_numel(cout, cin), dtype=torch.float32, device=w.device)
    check(lib.sisic_conv_winograd_pack(context(w.device), _ptr(w, "weight"), cout, cin, out.data_ptr(),
                                       _stream(w.device)))
    return out


def conv2d(x: torch.Tensor, w_packed: torch.Tensor, cout: int, ksize: int, *, bias=None, x2=None, stride=1,
           upsample=False, gn_scale=None, gn_shift=None, gn_silu=False, chan_bias=None, residual=None,
           relu=False, tile_cfg=0, w_winograd=None, with_stats=False, out=None, finalize=None):
    """sisic_conv2d.  with_stats=True also returns the GroupNorm partials the epilogue wrote, as a
    [B, Cout, slots, 4] tensor of (count, sum, centred M2, 0), or None when this launch cannot produce them
    (sisic_conv_stats_slots() == 0).  ``out``: the output tensor to write (it may be ``residual`` itself: include/sisic.h).
    ``finalize=(gamma, beta, groups, eps)``: ask the launch to finalize the GroupNorm over its own output as well; the result
    then ends with ``(scale, shift)`` [B, Cout] tensors, or ``None`` where sisic_conv_finalizes() says this launch does not."""
    lib = _lib.load()
    B, c0, H, W = x.shape
    c1 = 0 if x2 is None else x2.shape[1]
    Hc, Wc = (2 * H, 2 * W) if upsample else (H, W)
    pad = ksize // 2
    Ho = (Hc + 2 * pad - ksize) // stride + 1
    Wo = (Wc + 2 * pad - ksize) // stride + 1
    if out is None:
        out = torch.empty((B, cout, Ho, Wo), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (B, cout, Ho, Wo) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device:
        raise ValueError(f"out must be a contiguous fp32 {(B, cout, Ho, Wo)} tensor on {x.device}")
    a = ConvArgs()
    a.in0 = _ptr(x, "x"); a.in1 = _ptr(x2, "x2"); a.c0 = c0; a.c1 = c1
    a.B = B; a.Hin = H; a.Win = W
    a.upsample = int(upsample); a.ksize = ksize; a.stride = stride
    a.w_packed = _ptr(w_packed, "w_packed"); a.bias = _ptr(bias, "bias"); a.Cout = cout
    a.gn_scale = _ptr(gn_scale, "gn_scale"); a.gn_shift = _ptr(gn_shift, "gn_shift"); a.gn_silu = int(gn_silu)
    a.chan_bias = _ptr(chan_bias, "chan_bias"); a.chan_bias_stride = cout if chan_bias is not None and chan_bias.dim() == 2 and chan_bias.shape[0] == B else 0
    a.residual = _ptr(residual, "residual"); a.relu = int(relu)
    a.out = out.data_ptr(); a.tile_cfg = tile_cfg
    a.w_winograd = _ptr(w_winograd, "w_winograd")
    stats = None
    if with_stats:
        slots = lib.sisic_conv_stats_slots(C.byref(a))
        if slots > 0:
            stats = torch.empty((B, cout, slots, 4), dtype=torch.float32, device=x.device)
            a.stats_out = stats.data_ptr()
    fin = None
    if finalize is not None:
        gamma, beta, groups, eps = finalize
        a.fin_gamma = _ptr(gamma, "gamma"); a.fin_beta = _ptr(beta, "beta"); a.fin_groups = int(groups); a.fin_eps = float(eps)
        if lib.sisic_conv_finalizes(C.byref(a)):
            fin = (torch.empty((B, cout), dtype=torch.float32, device=x.device), torch.empty((B, cout), dtype=torch.float32, device=x.device))
            a.fin_scale, a.fin_shift = fin[0].data_ptr(), fin[1].data_ptr()
        else:
            a.fin_gamma = None
    check(lib.sisic_conv2d(context(x.device), C.byref(a), _stream(x.device)))
    res = (out, stats) if with_stats else out
    if finalize is not None:
        res = (res + (fin,)) if with_stats else (res, fin)
    return res


def groupnorm_finalize(stats: torch.Tensor, hw: int, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float,
                       stats2: Optional[torch.Tensor] = None):
    """scale/shift of GroupNorm over the (concatenated) producers' outputs from their epilogue partials."""
    lib = _lib.load()
    B, c0, slots0, _ = stats.shape
    c1, slots1 = (0, 0) if stats2 is None else (stats2.shape[1], stats2.shape[2])
    scale = torch.empty((B, c0 + c1), dtype=torch.float32, device=stats.device)
    shift = torch.empty_like(scale)
    check(lib.sisic_groupnorm_finalize(context(stats.device), _ptr(stats, "stats"), c0, slots0, _ptr(stats2, "stats2"),
                                       c1, slots1, B, hw, groups, float(eps), _ptr(gamma, "gamma"), _ptr(beta, "beta"),
                                       scale.data_ptr(), shift.data_ptr(), _stream(stats.device)))
    return scale, shift


def groupnorm_stats(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float,
                    x2: Optional[torch.Tensor] = None):
    lib = _lib.load()
    B, c0 = x.shape[0], x.shape[1]
    c1 = 0 if x2 is None else x2.shape[1]
    HW = x[0, 0].numel()
    scale = torch.empty((B, c0 + c1), dtype=torch.float32, device=x.device)
    shift = torch.empty_like(scale)
    check(lib.sisic_groupnorm_stats(context(x.device), _ptr(x, "x"), c0, _ptr(x2, "x2"), c1, B, HW, groups,
                                    float(eps), _ptr(gamma, "gamma"), _ptr(beta, "beta"), scale.data_ptr(),
                                    shift.data_ptr(), _stream(x.device)))
    return scale, shift


def attention(qkv: torch.Tensor, head_dim: int = 8) -> torch.Tensor:
    """qkv [B,3C,N] -> [B,C,N]."""
    lib = _lib.load()
    B, C3, N = qkv.shape
    Cc = C3 // 3
    out = torch.empty((B, Cc, N), dtype=torch.float32, device=qkv.device)
    check(lib.sisic_attention(context(qkv.device), _ptr(qkv, "qkv"), out.data_ptr(), B, Cc, N, head_dim,
                              _stream(qkv.device)))
    return out


def ddpm_step(eps: torch.Tensor, x: torch.Tensor, z: Optional[torch.Tensor], coef, clip: float = 1.0,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """coef = (sqrt_beta_prod, sqrt_alpha_prod, c0, c1, sigma)."""
    lib = _lib.load()
    if out is None:
        out = torch.empty_like(x)
    sb, sa, c0, c1, sigma = (float(v) for v in coef)
    check(lib.sisic_ddpm_step(context(x.device), _ptr(eps, "eps"), _ptr(x, "x"), _ptr(z, "z"), _ptr(out, "out"),
                              x.numel(), sb, sa, c0, c1, sigma, float(clip), _stream(x.device)))
    return out


def conv2d_wgrad(x: torch.Tensor, dy: torch.Tensor, ksize: int, *, x2=None, stride=1, upsample=False, gn_scale=None,
                 gn_shift=None, gn_silu=False) -> torch.Tensor:
    """d/dW of ``conv2d`` with the same prologue / index maps: dW [Cout, Cin, k, k] from the forward input(s) and the
    gradient dy of the convolution's output (sisic_conv2d_wgrad)."""
    lib = _lib.load()
    B, c0, H, W = x.shape
    c1 = 0 if x2 is None else x2.shape[1]
    cout = dy.shape[1]
    dw = torch.empty((cout, c0 + c1, ksize, ksize), dtype=torch.float32, device=x.device)
    a = ConvArgs()
    a.in0 = _ptr(x, "x"); a.in1 = _ptr(x2, "x2"); a.c0 = c0; a.c1 = c1
    a.B = B; a.Hin = H; a.Win = W
    a.upsample = int(upsample); a.ksize = ksize; a.stride = stride; a.Cout = cout
    a.gn_scale = _ptr(gn_scale, "gn_scale"); a.gn_shift = _ptr(gn_shift, "gn_shift"); a.gn_silu = int(gn_silu)
    check(lib.sisic_conv2d_wgrad(context(x.device), C.byref(a), _ptr(dy, "dy"), dw.data_ptr(), _stream(x.device)))
    return dw


def attention_bwd(qkv: torch.Tensor, out: torch.Tensor, d_out: torch.Tensor, head_dim: int = 8) -> torch.Tensor:
    """gradient of ``attention`` w.r.t. qkv [B,3C,N]."""
    lib = _lib.load()
    B, C3, N = qkv.shape
    dqkv = torch.empty_like(qkv)
    check(lib.sisic_attention_bwd(context(qkv.device), _ptr(qkv, "qkv"), _ptr(out, "out"), _ptr(d_out, "d_out"),
                                  dqkv.data_ptr(), B, C3 // 3, N, head_dim, _stream(qkv.device)))
    return dqkv


def groupnorm_bwd(da: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float,
                  silu: bool):
    """(dx, dgamma, dbeta) of a = act(GroupNorm(x)), act = SiLU or identity."""
    lib = _lib.load()
    B, Cc = x.shape[0], x.shape[1]
    HW = x[0, 0].numel()
    dx = torch.zeros_like(x)
    dg = torch.empty(Cc, dtype=torch.float32, device=x.device)
    db = torch.empty_like(dg)
    check(lib.sisic_groupnorm_bwd(context(x.device), _ptr(da, "da"), _ptr(x, "x"), B, Cc, HW, groups, float(eps),
                                  _ptr(gamma, "gamma"), _ptr(beta, "beta"), int(silu), dx.data_ptr(), dg.data_ptr(),
                                  db.data_ptr(), _stream(x.device)))
    return dx, dg, db


DENORM_FORMS = {"image_generator": 0, "generate_test": 1, "diffusion_generator": 2}


def denorm_u8(x: torch.Tensor, form="image_generator") -> torch.Tensor:
    """[B,C,H,W] fp32 -> uint8 [B,H,W,C] in the fp32 operation order of one of the reference's three call sites:
    "image_generator" (image_generator.py:441-447), "generate_test" (generate_test.py:94-97, bit-equal to the first) or
    "diffusion_generator" (diffusion_generator.py:231-232, `(x+1)*127.5` -- rounds differently)."""
    lib = _lib.load()
    B, Cc, H, W = x.shape
    out = torch.empty((B, H, W, Cc), dtype=torch.uint8, device=x.device)
    f = DENORM_FORMS[form] if isinstance(form, str) else int(form)
    check(lib.sisic_denorm_u8_form(context(x.device), _ptr(x, "x"), out.data_ptr(), B, Cc, H, W, f, _stream(x.device)))
    return out


_KINDS = {"conv3x3": 0, "conv1x1": 1, "groupnorm": 2, "attention": 3, "ddpm_step": 4, "other": 5,
          "conv3x3_winograd_main": 6, "conv3x3_winograd_bf16x3": 7}


def profile_enable(device, on: bool) -> None:
    check(_lib.load().sisic_profile_enable(context(device), int(on)))


def profile_reset(device) -> None:
    check(_lib.load().sisic_profile_reset(context(device)))


def profile_read(device) -> dict:
    """{kind: {"ms", "launches", "bytes", "flops", "flops_executed"}} accumulated since the last reset."""
    lib = _lib.load()
    out = {}
    for name, kind in _KINDS.items():
        ms, by, fl, fx = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        n = C.c_int64()
        check(lib.sisic_profile_read(context(device), kind, C.byref(ms), C.byref(n), C.byref(by), C.byref(fl),
                                     C.byref(fx)))
        out[name] = {"ms": ms.value, "launches": n.value, "bytes": by.value, "flops": fl.value,
                     "flops_executed": fx.value}
    return out
