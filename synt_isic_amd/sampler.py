"""The T-step reverse-diffusion sampler: ``generate(seed, class_name, T)``.

Product-side equivalent of ``ImageGenerator.generate_single_image``
(core/generator/image_generator.py:308-500): seed policy (:586-592, :626-637),
initial noise and ``noise_hash`` (:369-389), the loop (:395-403) -- executed by
``sisic_sample`` in libsisic_hip.so without returning to Python between steps --
trajectory capture (:406-407) and de-normalisation to uint8 HWC (:441-447).

Noise contract (a defined extension): image b owns one CPU
``torch.Generator().manual_seed(seed_b)``; ``x_T[b]`` is drawn first, then ``z_t[b]``
for every step with t > 0 in loop order.  Results are independent of the batch an
image is sampled in and of how images are sharded over GPUs.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check
from .scheduler import HipDDPMScheduler
from .unet import HipUNet2DModel

ISIC_CLASSES = ("MEL", "NV", "BCC", "AKIEC", "BKL", "DF", "VASC")   # xai/XAI.py:196


def class_seed_offset(class_name: str) -> int:
    """image_generator.py:586-592 -- 31-bit md5 offset per class."""
    h = hashlib.md5(class_name.encode("utf-8")).hexdigest()
    return int(h[:8], 16) & 0x7FFFFFFF


def image_seed(base_seed: int, class_name: str, index: int) -> int:
    """image_generator.py:626-631."""
    return (int(base_seed) + class_seed_offset(class_name) + int(index)) & 0x7FFFFFFF


def noise_hash(x_T: torch.Tensor) -> str:
    """image_generator.py:383-389 -- sha256 of the fp32 bytes of x_T, first 16 hex digits."""
    return hashlib.sha256(x_T.detach().to("cpu").contiguous().numpy().tobytes()).hexdigest()[:16]


def draw_noise(seeds: Sequence[int], n_noise_steps: int, chw: Tuple[int, int, int],
               pin: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Host noise per the contract above: x_T [B,C,H,W] and z [n_noise_steps,B,C,H,W]."""
    B = len(seeds)
    x_T = torch.empty((B,) + tuple(chw), dtype=torch.float32)
    z = torch.empty((n_noise_steps, B) + tuple(chw), dtype=torch.float32, pin_memory=pin)
    for b, s in enumerate(seeds):
        g = torch.Generator(device="cpu")
        g.manual_seed(int(s))
        x_T[b] = torch.randn((1,) + tuple(chw), generator=g)[0]
        if n_noise_steps:
            # one draw of n*numel values == n consecutive draws of numel (numel is a multiple of 16)
            z[:, b] = torch.randn((n_noise_steps,) + tuple(chw), generator=g)
    return x_T, z


class NoiseStream:
    """The same noise as ``draw_noise``, produced while the GPU samples.

    ``draw_noise`` materialises every z_t first: 786 M normals (3.1 GB) for 64 images at 64x64, T=1000 -- seconds of
    single-threaded RNG in front of an 8 s sampling run.  Here the per-image generators are advanced on worker
    threads, one segment of steps at a time, into two pinned buffers that are uploaded on a side stream; the sampler
    consumes segment k while segment k+1 is drawn.  One big ``randn`` of n*numel values equals n consecutive draws of
    numel values (numel is a multiple of 16), and each generator is only ever advanced by one task at a time, so the
    stream is bit-identical to ``draw_noise`` whatever the segment length or worker count.
    """

    def __init__(self, seeds: Sequence[int], chw: Tuple[int, int, int], device: torch.device, segment_steps: int,
                 workers: Optional[int] = None, buffer_cache: Optional[dict] = None):
        self.chw = tuple(chw)
        self.B = len(seeds)
        self.device = device
        self.seg = max(1, int(segment_steps))
        # a caller that samples repeatedly (Sampler) keeps the worker pool, the copy stream and the staging buffers in
        # ``buffer_cache`` between calls: creating them costs milliseconds, a one-image T=50 run is ~110 ms
        self._own_pool = True
        if buffer_cache is not None and workers is None and "pool" in buffer_cache:
            self.pool = buffer_cache["pool"]
            self._own_pool = False
        else:
            if workers is None:
                try:
                    cpus = len(os.sched_getaffinity(0))
                except AttributeError:                      # pragma: no cover
                    cpus = os.cpu_count() or 1
                n_workers = max(1, min(cpus, 16))
            else:
                n_workers = max(1, int(workers))
            self.pool = ThreadPoolExecutor(max_workers=n_workers)
            if buffer_cache is not None and workers is None:
                buffer_cache["pool"] = self.pool
                self._own_pool = False
        self.gens = []
        for sd in seeds:
            g = torch.Generator(device="cpu")
            g.manual_seed(int(sd))
            self.gens.append(g)
        self.x_T = torch.empty((self.B,) + self.chw, dtype=torch.float32)
        list(self.pool.map(self._draw_x, range(self.B)))
        shape = (self.seg, self.B) + self.chw
        # pinning ~100 MB costs tens of milliseconds: a caller that samples repeatedly (Sampler) passes a dict in which
        # the two pinned and two device buffers of a shape are kept between calls
        key = (shape, str(device))
        if buffer_cache is not None and buffer_cache.get("shape") == key:
            self.host, self.dev = buffer_cache["buffers"]
        else:
            self.host = [torch.empty(shape, dtype=torch.float32, pin_memory=True) for _ in range(2)]
            self.dev = [torch.empty(shape, dtype=torch.float32, device=device) for _ in range(2)]
            if buffer_cache is not None:                # one shape at a time: bounded memory
                buffer_cache["shape"], buffer_cache["buffers"] = key, (self.host, self.dev)
        if buffer_cache is not None and buffer_cache.get("copy_stream_device") == str(device):
            self.copy_stream = buffer_cache["copy_stream"]
        else:
            self.copy_stream = torch.cuda.Stream(device)
            if buffer_cache is not None:
                buffer_cache["copy_stream"], buffer_cache["copy_stream_device"] = self.copy_stream, str(device)
        # uploads of less than 8 MB per segment (a few images) go on the sampling stream itself
        self.same_stream = (os.environ.get("SISIC_NOISE_SAME_STREAM", "1") != "0") and self.seg * self.B * int(np.prod(self.chw)) * 4 <= (8 << 20)
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]      # upload of slot i finished
        self.uploaded = [False, False]
        self.consumed = [None, None]                               # event after the sampler's last read of slot i
        self.pending = [None, None]

    def _draw_x(self, b: int) -> None:
        self.x_T[b] = torch.randn((1,) + self.chw, generator=self.gens[b])[0]

    def _draw_z(self, slot: int, n: int, b: int) -> None:
        self.host[slot][:n, b] = torch.randn((n,) + self.chw, generator=self.gens[b])

    def prefetch(self, slot: int, n: int) -> None:
        """start drawing the next n noise tensors of every image into pinned buffer ``slot``"""
        if n <= 0:
            self.pending[slot] = []
            return
        if self.uploaded[slot]:
            self.ready[slot].synchronize()           # the previous upload out of this pinned buffer has completed
        self.pending[slot] = [self.pool.submit(self._draw_z, slot, n, b) for b in range(self.B)]

    def acquire(self, slot: int, n: int) -> Optional[torch.Tensor]:
        """wait for the draws of ``slot``, upload them on the side stream and make the current stream wait"""
        if n <= 0:
            return None
        for f in self.pending[slot]:
            f.result()
        cur = torch.cuda.current_stream(self.device)
        if self.same_stream:
            # small runs (one image): the upload rides the sampling stream itself -- stream order replaces the two events
            self.dev[slot][:n].copy_(self.host[slot][:n], non_blocking=True)
            self.ready[slot].record(cur)
            self.uploaded[slot] = True
            return self.dev[slot][:n]
        with torch.cuda.stream(self.copy_stream):
            if self.consumed[slot] is not None:
                self.copy_stream.wait_event(self.consumed[slot])
            self.dev[slot][:n].copy_(self.host[slot][:n], non_blocking=True)
            self.ready[slot].record(self.copy_stream)
            self.uploaded[slot] = True
        cur.wait_event(self.ready[slot])
        return self.dev[slot][:n]

    def release(self, slot: int) -> None:
        """the sampler's reads of ``slot`` are enqueued on the current stream"""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.consumed[slot] = ev

    def close(self) -> None:
        if self._own_pool:
            self.pool.shutdown(wait=True)
        else:                                            # a shared pool: wait for this run's draws only
            for tasks in self.pending:
                for f in tasks or []:
                    f.result()
            for slot in range(2):                        # ... and for the uploads out of the shared pinned buffers
                if self.uploaded[slot]:
                    self.ready[slot].synchronize()


def trajectory_save_indices(timesteps: Sequence[int], save_every: int) -> List[int]:
    """Which steps of a run the XAI trajectory keeps (xai/XAI.py:751-777 `save_indices`, :815-822): every
    ``save_every``-th step by index and always the last one; when ``save_every`` is not smaller than the number of steps
    it is read as a stride in t instead: the steps nearest to t = 0, max(t) and the multiples of ``save_every`` up to 1000,
    and every step whose own t is such a multiple (or 0).  Sorted step indices."""
    ts = [int(float(t)) for t in timesteps]
    n = len(ts)
    every = int(save_every)
    if every <= 0:
        raise ValueError("save_every must be positive")
    keep = set(range(0, n, every))
    if n:
        keep.add(n - 1)
    if every >= n and n:
        want = {0, max(ts)} | set(range(0, 1001, every))
        for dt in want:
            keep.add(min(range(n), key=lambda i: abs(ts[i] - dt)))
        keep |= {i for i, t in enumerate(ts) if t % every == 0 or t == 0}
    return sorted(keep)


@dataclass
class SampleResult:
    images: torch.Tensor                       # uint8 [B,H,W,3] on the GPU
    latents: torch.Tensor                      # fp32 [B,3,H,W] final x_0 on the GPU
    trajectory: Optional[torch.Tensor] = None  # fp32 [n_kept,B,3,H,W] on the GPU: x after the steps of trajectory_steps
    trajectory_steps: List[int] = field(default_factory=list)   # step indices of the kept frames (all T without a stride)
    seeds: List[int] = field(default_factory=list)
    noise_hashes: List[str] = field(default_factory=list)
    timesteps: List[int] = field(default_factory=list)
    steps_done: int = 0
    cancelled: bool = False                    # the stop flag ended the loop early: images/latents are NOT a result


def _frame_rows(T: int, return_trajectory: bool, save_indices: Optional[Sequence[int]]):
    """(kept step indices, host int32 [T] row table or None) for sisic_sample_frames"""
    if not return_trajectory:
        return [], None
    if save_indices is None:
        return list(range(T)), None
    kept = sorted({int(i) for i in save_indices})
    if kept and (kept[0] < 0 or kept[-1] >= T):
        raise ValueError(f"save_indices outside 0..{T - 1}")
    rows = np.full((T,), -1, dtype=np.int32)
    rows[kept] = np.arange(len(kept), dtype=np.int32)
    return kept, rows


@torch.no_grad()
def run_sampling_loop(model: HipUNet2DModel, scheduler: HipDDPMScheduler, x_T: torch.Tensor,
                      noise, *, return_trajectory: bool = False, save_indices: Optional[Sequence[int]] = None,
                      cancel_flag: Optional[C.c_int] = None) -> SampleResult:
    """x_T: GPU fp32 [B,C,H,W]; noise: GPU fp32 [n_noise,B,C,H,W], None (no noise added), or a ``NoiseStream``
    (the loop then runs segment by segment while the stream draws and uploads the next segment's noise).
    return_trajectory keeps x after every step, or after the steps in ``save_indices`` only (``trajectory_save_indices``)."""
    if isinstance(noise, NoiseStream):
        return _run_streamed(model, scheduler, x_T, noise, return_trajectory, cancel_flag, save_indices)
    lib = _lib.load()
    dev = x_T.device
    if dev.type != "cuda":
        raise RuntimeError("the sampling loop runs on MI355X only")
    B, Cc, H, W = x_T.shape
    ts = scheduler.timesteps.to(torch.int64).contiguous()
    T = ts.numel()
    coef = scheduler.coefficient_table().contiguous()
    n_noise = int((coef[:, 4] != 0).sum())
    if noise is not None:
        if tuple(noise.shape) != (n_noise, B, Cc, H, W) or noise.device != dev or noise.dtype != torch.float32:
            raise ValueError(f"noise must be fp32 {(n_noise, B, Cc, H, W)} on {dev}, got {tuple(noise.shape)}")
        noise = noise.contiguous()
    x = x_T.to(torch.float32).contiguous().clone()
    kept, rows = _frame_rows(T, return_trajectory, save_indices)
    traj = torch.empty((len(kept), B, Cc, H, W), dtype=torch.float32, device=dev) if return_trajectory else None
    out_u8 = torch.empty((B, H, W, Cc), dtype=torch.uint8, device=dev)
    done = C.c_int(0)
    clip = scheduler.config.clip_sample_range if scheduler.config.clip_sample else 0.0
    rc = lib.sisic_sample_frames(model.handle, x.data_ptr(), B, H, W, T,
                                 C.cast(ts.data_ptr(), _lib.c_int64_p), C.cast(coef.data_ptr(), _lib.c_float_p),
                                 float(clip), noise.data_ptr() if noise is not None else None,
                                 traj.data_ptr() if traj is not None and len(kept) else None,
                                 rows.ctypes.data_as(C.POINTER(C.c_int)) if rows is not None and len(kept) else None,
                                 out_u8.data_ptr(),
                                 C.byref(cancel_flag) if cancel_flag is not None else None, C.byref(done),
                                 C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != _lib.SISIC_ECANCEL:
        check(rc)
    cancelled = rc == _lib.SISIC_ECANCEL
    if cancelled:
        out_u8.zero_()                         # never hand uninitialised pixels to a caller that ignores `cancelled`
    return SampleResult(images=out_u8, latents=x, trajectory=traj, trajectory_steps=kept, timesteps=[int(t) for t in ts],
                        steps_done=done.value, cancelled=cancelled)


def segment_bounds(T: int, seg: int, per_step: int) -> List[int]:
    """Step boundaries of the segments a streamed run is cut into (``NoiseStream``): ``seg`` steps each, except that a run
    whose segment is a noticeable amount of RNG (more than 1 M normals: one 128x128 image draws 3 M per 64 steps, 64 images
    at 64x64 draw 50 M) starts with 4, 8, 16, ... steps, so that the GPU starts after a millisecond or two instead of a
    whole segment's worth.  The draws do not depend on the segmentation.  (Round 2 kept one 128x128 image in ONE segment
    because every extra ``sisic_sample`` call cost ~20 ms: a segment of another length re-allocated the per-step tables,
    which rebuilt the captured graph, and every call ran its first step eagerly.  The tables now have a fixed size and a
    call that finds its graph replays from step 0, so an extra segment costs its launches only.)"""
    ramp_min = int(os.environ.get("SISIC_NOISE_RAMP_MIN", "1000000"))
    bounds, n = [0], (4 if per_step * seg > ramp_min else seg)
    while bounds[-1] < T:
        bounds.append(min(T, bounds[-1] + min(n, seg)))
        n *= 2
    return bounds


def _run_streamed(model: HipUNet2DModel, scheduler: HipDDPMScheduler, x_T: torch.Tensor, ns: NoiseStream,
                  return_trajectory: bool, cancel_flag: Optional[C.c_int],
                  save_indices: Optional[Sequence[int]] = None) -> SampleResult:
    lib = _lib.load()
    dev = x_T.device
    B, Cc, H, W = x_T.shape
    ts = scheduler.timesteps.to(torch.int64).contiguous()
    T = ts.numel()
    coef = scheduler.coefficient_table().contiguous()
    needs = (coef[:, 4] != 0).to(torch.int64)                   # 1 where the step adds noise
    bounds = segment_bounds(T, ns.seg, B * Cc * H * W)
    counts = [int(needs[a:b].sum()) for a, b in zip(bounds[:-1], bounds[1:])]
    x = x_T.to(torch.float32).contiguous().clone()
    kept, rows = _frame_rows(T, return_trajectory, save_indices)
    if return_trajectory and rows is None:
        rows = np.arange(T, dtype=np.int32)                     # a segment's steps go to their rows of the whole run
    traj = torch.empty((len(kept), B, Cc, H, W), dtype=torch.float32, device=dev) if return_trajectory else None
    out_u8 = torch.empty((B, H, W, Cc), dtype=torch.uint8, device=dev)
    clip = scheduler.config.clip_sample_range if scheduler.config.clip_sample else 0.0
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    done_total, rc = 0, 0
    ns.prefetch(0, counts[0])
    for k, (a, b) in enumerate(zip(bounds[:-1], bounds[1:])):
        slot = k & 1
        z = ns.acquire(slot, counts[k])
        if k + 1 < len(counts):
            ns.prefetch(slot ^ 1, counts[k + 1])                # drawn while the GPU runs this segment
        done = C.c_int(0)
        last = b == T
        seg_rows = np.ascontiguousarray(rows[a:b]) if traj is not None and len(kept) else None
        rc = lib.sisic_sample_frames(model.handle, x.data_ptr(), B, H, W, b - a,
                                     C.cast(ts[a:b].contiguous().data_ptr(), _lib.c_int64_p),
                                     C.cast(coef[a:b].contiguous().data_ptr(), _lib.c_float_p), float(clip),
                                     z.data_ptr() if z is not None else None,
                                     traj.data_ptr() if seg_rows is not None else None,
                                     seg_rows.ctypes.data_as(C.POINTER(C.c_int)) if seg_rows is not None else None,
                                     out_u8.data_ptr() if last else None,
                                     C.byref(cancel_flag) if cancel_flag is not None else None, C.byref(done), stream)
        ns.release(slot)
        done_total += done.value
        if rc != 0:
            break
    if rc != _lib.SISIC_ECANCEL:
        check(rc)
    cancelled = rc == _lib.SISIC_ECANCEL
    if cancelled:
        out_u8.zero_()
    return SampleResult(images=out_u8, latents=x, trajectory=traj, trajectory_steps=kept, timesteps=[int(t) for t in ts],
                        steps_done=done_total, cancelled=cancelled)


COLOR_BLEND = 0.35            # image_generator.py:532 "alpha"
COLOR_SCALE_CLIP = (0.6, 1.4)  # image_generator.py:527


def apply_color_statistics(images: np.ndarray, stats: Optional[dict]) -> np.ndarray:
    """Colour post-processing of the GUI path (``ImageGenerator._apply_color_postprocessing``,
    image_generator.py:502-545) for a batch of uint8 [B,H,W,3] images: each image's per-channel mean and standard
    deviation are pulled towards the class statistics of ``color_statistics.json`` -- scale clipped to [0.6, 1.4],
    35 % blend with the original, clip to [0,255], truncation to uint8.  Host-side numpy in the reference's float32
    operation order (bit-identical results); images are processed one by one because the reference's statistics are
    per image.  ``stats`` = the JSON entry of the class, or None / incomplete: images are returned unchanged."""
    if not stats or "rgb" not in stats or "mean" not in stats["rgb"]:
        return images
    t_mean = np.asarray(stats["rgb"].get("mean", [128, 128, 128]), dtype=np.float32)
    t_std = np.asarray(stats["rgb"].get("std", [50, 50, 50]), dtype=np.float32)
    out = np.empty_like(images)
    for b in range(images.shape[0]):
        img = images[b]
        mean = img.mean(axis=(0, 1)).astype(np.float32)             # float64 accumulation, then float32 like the reference
        std = img.std(axis=(0, 1)).astype(np.float32)
        scale = np.clip(t_std / np.maximum(std, 1e-6), *COLOR_SCALE_CLIP)
        f = img.astype(np.float32)
        moved = (f - mean) * scale + t_mean
        mixed = COLOR_BLEND * moved + (1.0 - COLOR_BLEND) * f
        out[b] = np.clip(mixed, 0, 255).astype(np.uint8)
    return out


class Sampler:
    """Holds one loaded UNet per class, like ``ModelManager.loaded_models`` (model_manager.py:19-171)."""

    def __init__(self, device="cuda", beta_schedule: str = "squaredcos_cap_v2", latency_mode: bool = False):
        """latency_mode=True: every model of this sampler uses the single-image kernel choices
        (HipUNet2DModel.set_latency_mode) -- for the GUI's one-image-at-a-time calls; throughput batches keep the default."""
        self.device = torch.device(device)
        self.beta_schedule = beta_schedule
        self.latency_mode = bool(latency_mode)
        self.models: Dict[str, HipUNet2DModel] = {}
        self.cancel = C.c_int(0)          # cooperative stop flag (image_generator.py:320,396)
        self.last_trajectory_steps: List[int] = []   # step indices of the frames the last generate(..., return_trajectory=True) kept
        self.noise_segment_steps = 64     # steps of noise drawn and uploaded per pipeline stage (NoiseStream)
        self.color_statistics: Dict[str, dict] = {}   # class -> color_statistics.json entry (image_generator.py:142-170)
        self._noise_buffers: dict = {}    # pinned/device staging buffers of the last noise shape (NoiseStream)

    def close(self) -> None:
        """Shut the noise producers' thread pool down and drop the pinned / device staging buffers (the models stay)."""
        pool = self._noise_buffers.pop("pool", None)
        if pool is not None:
            pool.shutdown(wait=True)
        self._noise_buffers.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:              # interpreter shutdown: the executor module may already be gone
            pass

    def load_color_statistics(self, path: str) -> int:
        """``checkpoints/color_statistics.json`` (image_generator.py:142-170); returns the number of classes read.
        A missing file leaves post-processing a no-op, as in the reference."""
        import json
        import os as _os
        if not _os.path.exists(path):
            return 0
        with open(path, "r") as f:
            self.color_statistics = json.load(f)
        return len(self.color_statistics)

    def add_model(self, class_name: str, state_dict: Dict[str, torch.Tensor], **unet_kwargs) -> HipUNet2DModel:
        m = HipUNet2DModel(**unet_kwargs)
        m.set_latency_mode(self.latency_mode)
        m.load_state_dict(state_dict)
        m = m.to(self.device)
        m.eval()
        self.models[class_name] = m
        return m

    def create_scheduler(self, T: int) -> HipDDPMScheduler:
        """model_manager.py:196-212."""
        s = HipDDPMScheduler(num_train_timesteps=1000, beta_schedule=self.beta_schedule)
        s.set_timesteps(max(1, min(1000, int(T))))
        return s

    def request_stop(self) -> None:
        """Cooperative stop (``stop_generation``, image_generator.py:784-786): the running loop ends at the next step."""
        self.cancel.value = 1

    def generate_images(self, class_name: str, seeds: Sequence[int], T: int, **kwargs) -> SampleResult:
        """Start of a generation run in the reference's sense (``generate_images`` clears ``stop_requested`` before
        its first image, image_generator.py:567): resets the stop flag, then samples ``seeds``.  A stop request that
        arrives later ends this run only."""
        self.cancel.value = 0
        return self.generate_seeds(class_name, seeds, T, **kwargs)

    def generate_seeds(self, class_name: str, seeds: Sequence[int], T: int, size: Tuple[int, int] = (128, 128),
                       return_trajectory: bool = False, save_every_n: Optional[int] = None) -> SampleResult:
        """save_every_n: keep only the trajectory frames the reference's XAI run keeps (``trajectory_save_indices``,
        xai/XAI.py:751-777) instead of all T -- 3.1 GB at 64 images x 64x64 x T = 1000 otherwise."""
        if class_name not in self.models:
            raise KeyError(f"no model loaded for class '{class_name}'")
        model = self.models[class_name]
        sched = self.create_scheduler(T)
        save_indices = None
        if return_trajectory and save_every_n is not None:
            save_indices = trajectory_save_indices([int(t) for t in sched.timesteps], save_every_n)
        n_noise = sum(1 for t in sched.timesteps if int(t) > 0)
        H, W = size
        # noise is drawn segment by segment on worker threads while the GPU samples (NoiseStream); the values are
        # those of draw_noise(seeds, n_noise, ...)
        ns = NoiseStream(seeds, (model.config.in_channels, H, W), self.device, self.noise_segment_steps,
                         buffer_cache=self._noise_buffers)
        try:
            hashes = [noise_hash(ns.x_T[b:b + 1]) for b in range(len(seeds))]
            res = run_sampling_loop(model, sched, ns.x_T.to(self.device), ns if n_noise else None,
                                    return_trajectory=return_trajectory, save_indices=save_indices, cancel_flag=self.cancel)
            torch.cuda.current_stream(self.device).synchronize()
        finally:
            ns.close()
        res.seeds = [int(s) for s in seeds]
        res.noise_hashes = hashes
        return res

    def generate(self, seed: int, class_name: str, T: int, *, count: int = 1, size: Tuple[int, int] = (128, 128),
                 return_trajectory: bool = False, seed_is_base: bool = False, postprocess: bool = False,
                 save_every_n: Optional[int] = None):
        """``generate(seed, class, T)``: returns (uint8 [count,H,W,3] numpy, trajectory list | None).

        save_every_n: with return_trajectory, the list holds only the frames of ``trajectory_save_indices`` (every n-th
        step and the last: xai/XAI.py:751-757), in step order; ``last_trajectory_steps`` then names their step indices.

        seed_is_base=False: image i uses ``manual_seed(seed + i)`` directly (the literal call);
        seed_is_base=True: ``seed`` is the GUI's base seed and image i uses
        ``(seed + md5_offset(class) + i) & 0x7fffffff`` (image_generator.py:626-631).
        postprocess=True applies the class colour statistics (``load_color_statistics``) to the uint8 images like
        ``generate_single_image(..., postprocess=True)`` does before saving (image_generator.py:449-452).
        Always returns a tuple (the reference's bare ``return False`` on early exit is a latent bug).
        """
        if seed_is_base:
            seeds = [image_seed(seed, class_name, i) for i in range(count)]
        else:
            seeds = [(int(seed) + i) & 0x7FFFFFFF for i in range(count)]
        res = self.generate_images(class_name, seeds, T, size=size, return_trajectory=return_trajectory,
                                   save_every_n=save_every_n)
        self.last_trajectory_steps = list(res.trajectory_steps)
        n_frames = sum(1 for i in res.trajectory_steps if i < res.steps_done)       # kept frames of the completed steps
        if res.cancelled:
            # the reference returns False for a stopped image (image_generator.py:396-398); the tuple shape is kept, with
            # no images, and the steps that did complete when a trajectory was asked for
            traj = [res.trajectory[i] for i in range(n_frames)] if return_trajectory else None
            return None, traj
        images = res.images.cpu().numpy()
        if postprocess:
            images = apply_color_statistics(images, self.color_statistics.get(class_name))
        traj = None
        if return_trajectory:
            # list of per-step (B,3,H,W) tensors, the shape xai_integration.py consumes
            traj = [res.trajectory[i] for i in range(n_frames)]
        return images, traj


_default_sampler: Optional[Sampler] = None


def generate(seed: int, class_name: str, T: int, **kwargs):
    """Module-level convenience over a process-wide ``Sampler`` whose models were registered with
    ``default_sampler().add_model(...)``."""
    return default_sampler().generate(seed, class_name, T, **kwargs)


def default_sampler() -> Sampler:
    global _default_sampler
    if _default_sampler is None:
        _default_sampler = Sampler()
    return _default_sampler
