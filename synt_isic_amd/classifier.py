"""HipMelanomaClassifier -- drop-in for ``MelanomaClassifierAdaptive`` (xai/XAI.py:357-471), forward only.

The reference wraps ``torchvision.models.resnet18`` (fc -> num_classes) and feeds it diffusion latents in
[-1,1] through ``preprocess_for_classifier`` (clamp, bilinear resize to 224x224, ImageNet normalisation).
Here the whole forward -- pre-processing, the BN-folded convolutions on the f32 MFMA kernel, pooling, fc and
the softmax/log scores -- runs in libsisic_hip.so (``sisic_resnet_forward`` / ``sisic_class_scores``).
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, Iterator, Optional

import torch

from . import _lib, ops
from ._lib import check
from .arch import resnet18_param_spec

NUM_CLASSES = 7                                     # xai/XAI.py:196 ISIC classes
CLASS_NAMES = ("MEL", "NV", "BCC", "AKIEC", "BKL", "DF", "VASC")
CLASSIFIER_IMAGE_SIZE = 224


class HipMelanomaClassifier:
    def __init__(self, num_classes: int = NUM_CLASSES, architecture: str = "resnet18", pretrained: bool = False):
        if architecture not in ("resnet18", "auto"):
            raise NotImplementedError("only the built-in resnet18 architecture of the reference is implemented")
        if pretrained:
            raise NotImplementedError("IMAGENET1K_V1 weights are a network download (XAI.py:389); pass a "
                                      "state dict to load_state_dict instead")
        self.num_classes = num_classes
        self.architecture = "resnet18"
        self._spec = resnet18_param_spec(num_classes)
        self._params: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self._device = torch.device("cpu")
        self._handle: Optional[C.c_void_p] = None
        self._uploaded = False
        self.training = True

    # ---- module surface -------------------------------------------------------------------------
    @property
    def device(self) -> torch.device:
        return self._device

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("forward only: training the classifier is out of scope")
        return self.eval()

    def parameters(self) -> Iterator[torch.Tensor]:
        return (v for k, v in self._params.items() if "running_" not in k)

    def state_dict(self):
        return OrderedDict(self._params)

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        """strict=False keeps only keys whose name AND shape match, like load_classifier_with_fallback
        (XAI.py:518-527).  Integer buffers (num_batches_tracked) are ignored either way."""
        sd = {k: v for k, v in state_dict.items() if not k.endswith("num_batches_tracked")}
        cur = dict(self._params)
        new = OrderedDict()
        for name, shape in self._spec.items():
            t = sd.get(name)
            if t is not None and tuple(t.shape) == tuple(shape):
                new[name] = t.detach().to(device=self._device, dtype=torch.float32).contiguous().clone()
            elif strict:
                raise RuntimeError(f"Error(s) in loading state_dict for HipMelanomaClassifier: "
                                   f"{'missing key' if t is None else 'size mismatch for'} {name}")
            elif name in cur:
                new[name] = cur[name]
            else:
                raise RuntimeError(f"no value for {name} (non-strict load needs a previously loaded model)")
        if strict:
            extra = [k for k in sd if k not in self._spec]
            if extra:
                raise RuntimeError(f"unexpected keys in state_dict: {extra[:5]}")
        self._params = new
        self._uploaded = False
        if self._device.type == "cuda":
            self._upload()
        return self

    def to(self, device=None, *a, **k):
        if device is None:
            return self
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if device == self._device:
            return self
        self._release()
        self._params = OrderedDict((n, v.to(device)) for n, v in self._params.items())
        self._device = device
        if device.type == "cuda" and self._params:
            self._upload()
        return self

    # ---- library handle -------------------------------------------------------------------------
    def _upload(self) -> None:
        lib = _lib.load()
        if self._handle is None:
            h = C.c_void_p()
            check(lib.sisic_resnet_create(ops.context(self._device), self.num_classes, C.byref(h)))
            self._handle = h
            names = [lib.sisic_resnet_tensor_name(h, i).decode() for i in range(lib.sisic_resnet_num_tensors(h))]
            if sorted(names) != sorted(self._spec):
                raise RuntimeError("libsisic_hip.so and synt_isic_amd.arch disagree on the classifier keys")
        host = [(n, v.detach().to("cpu", torch.float32).contiguous()) for n, v in self._params.items()]
        n = len(host)
        names = (C.c_char_p * n)(*[k.encode() for k, _ in host])
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for _, t in host])
        numels = (C.c_int64 * n)(*[t.numel() for _, t in host])
        check(lib.sisic_resnet_load(self._handle, n, names, ptrs, numels))
        self._uploaded = True

    def _release(self) -> None:
        if self._handle is not None:
            _lib.load().sisic_resnet_destroy(self._handle)
            self._handle = None
            self._uploaded = False

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    @property
    def handle(self):
        if self._device.type != "cuda":
            raise RuntimeError("HipMelanomaClassifier runs on MI355X only: call .to('cuda') first (no CPU path)")
        if not self._params:
            raise RuntimeError("HipMelanomaClassifier has no weights: call load_state_dict() first")
        if not self._uploaded:
            self._upload()
        return self._handle

    # ---- forward --------------------------------------------------------------------------------
    # Frames per library call.  The callers in xai.py hand over whole trajectories (N frames = N inference steps, up to
    # 1000) and coalition sets; the library's activation workspace scales with the batch of a call (stem output:
    # 3.2 MB per image, the backward passes keep every block's activations), so the batch of one call is capped and
    # longer inputs are walked in chunks.  Rows do not depend on the chunking (no kernel choice depends on the batch).
    max_forward_batch = 512
    max_backward_batch = 128

    @staticmethod
    def _chunks(total: int, cap: int):
        lo = 0
        while lo < total:
            n = min(cap, total - lo)
            yield lo, n
            lo += n

    def workspace_bytes(self) -> int:
        """activation workspace the library keeps for this model (released when the input shape changes)"""
        return int(_lib.load().sisic_resnet_workspace_bytes(self._handle)) if self._handle is not None else 0

    @torch.no_grad()
    def forward(self, x: torch.Tensor, preprocessed: bool = False) -> torch.Tensor:
        """logits [B,num_classes].  x: [B,3,H,W] in [-1,1] (H,W <= 224); it is moved to the model's device like
        preprocess_for_classifier does (XAI.py:407-409)."""
        h = self.handle
        if x.device != self._device:
            x = x.to(self._device)
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"classifier input must be [B,3,H,W], got {tuple(x.shape)}")
        x = x.to(torch.float32).contiguous()
        B, _, H, W = x.shape
        out = torch.empty((B, self.num_classes), dtype=torch.float32, device=x.device)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        for lo, n in self._chunks(B, self.max_forward_batch):
            check(_lib.load().sisic_resnet_forward(h, x[lo:].data_ptr(), out[lo:].data_ptr(), n, H, W,
                                                   0 if preprocessed else 1, stream))
        return out

    __call__ = forward

    def _scores(self, x: torch.Tensor, target_class: int):
        logits = self.forward(x)
        B = logits.shape[0]
        prob = torch.empty(B, dtype=torch.float32, device=logits.device)
        logscore = torch.empty_like(prob)
        check(_lib.load().sisic_class_scores(ops.context(logits.device), logits.data_ptr(), B, self.num_classes,
                                             int(target_class), prob.data_ptr(), logscore.data_ptr(),
                                             C.c_void_p(torch.cuda.current_stream(logits.device).cuda_stream)))
        return prob, logscore

    def get_probabilities(self, x: torch.Tensor) -> torch.Tensor:
        """softmax over the 7 logits (XAI.py:438-441): [B,7]; assembled from the per-class score kernel."""
        logits = self.forward(x)
        cols = []
        for c in range(self.num_classes):
            p = torch.empty(logits.shape[0], dtype=torch.float32, device=logits.device)
            check(_lib.load().sisic_class_scores(ops.context(logits.device), logits.data_ptr(), logits.shape[0],
                                                 self.num_classes, c, p.data_ptr(), None,
                                                 C.c_void_p(torch.cuda.current_stream(logits.device).cuda_stream)))
            cols.append(p)
        return torch.stack(cols, dim=1)

    def get_per_class_score(self, x: torch.Tensor, target_class: int) -> torch.Tensor:
        """log(p(c|x) + 1e-8)   (XAI.py:443-459)."""
        return self._scores(x, target_class)[1]

    def get_confidence(self, x: torch.Tensor, target_class: int) -> torch.Tensor:
        """p(c|x)   (XAI.py:467-471)."""
        return self._scores(x, target_class)[0]

    @torch.no_grad()
    def stem_activation(self, x: torch.Tensor) -> torch.Tensor:
        """relu(bn1(conv1(preprocess(x)))) [B,64,112,112]: the tensor the stem's max-pool picks its arg-maxima from
        (sisic_resnet_stem; used by the parity tests of the backward pass)."""
        h = self.handle
        x = x.to(self._device).detach().to(torch.float32).contiguous()
        B, _, H, W = x.shape
        out = torch.empty((B, 64, 112, 112), dtype=torch.float32, device=x.device)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        for lo, n in self._chunks(B, self.max_forward_batch):
            check(_lib.load().sisic_resnet_stem(h, x[lo:].data_ptr(), out[lo:].data_ptr(), n, H, W, 1, stream))
        return out

    def input_gradient(self, x: torch.Tensor, target_class: int):
        """(d get_per_class_score / d x, logits): the gradient captum's IntegratedGradients and the plain-gradient
        fallback of XAI.py:1039-1109 take, through the pre-processing, with no autograd graph -- the transposed
        network runs on the same HIP convolution kernels (sisic_resnet_input_gradient)."""
        h = self.handle
        if x.device != self._device:
            x = x.to(self._device)
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"classifier input must be [B,3,H,W], got {tuple(x.shape)}")
        x = x.detach().to(torch.float32).contiguous()
        B, _, H, W = x.shape
        grad = torch.empty_like(x)
        logits = torch.empty((B, self.num_classes), dtype=torch.float32, device=x.device)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        for lo, n in self._chunks(B, self.max_backward_batch):
            check(_lib.load().sisic_resnet_input_gradient(h, x[lo:].data_ptr(), n, H, W, int(target_class),
                                                          grad[lo:].data_ptr(), logits[lo:].data_ptr(), stream))
        return grad, logits

    def grad_cam(self, x: torch.Tensor, target_class: int):
        """(cam [B,224,224] in [0,1], logits): pytorch_grad_cam's GradCAM on ``model.layer4[-1].conv2`` for the raw class
        logit, as xai/XAI.py:2945-3035 calls it on each trajectory frame (sisic_resnet_gradcam)."""
        h = self.handle
        if x.device != self._device:
            x = x.to(self._device)
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"classifier input must be [B,3,H,W], got {tuple(x.shape)}")
        x = x.detach().to(torch.float32).contiguous()
        B, _, H, W = x.shape
        cam = torch.empty((B, 224, 224), dtype=torch.float32, device=x.device)
        logits = torch.empty((B, self.num_classes), dtype=torch.float32, device=x.device)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        for lo, n in self._chunks(B, self.max_backward_batch):
            check(_lib.load().sisic_resnet_gradcam(h, x[lo:].data_ptr(), n, H, W, int(target_class),
                                                   cam[lo:].data_ptr(), logits[lo:].data_ptr(), stream))
        return cam, logits

    def predict(self, x: torch.Tensor) -> torch.Tensor:
        return torch.argmax(self.forward(x), dim=1)
