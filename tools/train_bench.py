#!/usr/bin/env python3
"""Time of one training step (train_diffusion.py:215-233: add_noise, forward, MSE, backward, Adam) on one MI355X.

    python tools/train_bench.py [--batch 2 --size 128] [--steps 10]

The reference trains with batch 2 (train_diffusion.py:59) at IMAGE_SIZE (:58); a throughput-sized batch is timed beside it.
Prints one JSON line per configuration: ms per step, images/s, and the split forward / backward / optimizer measured with
HIP events on the stream.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from synt_isic_amd.scheduler import HipDDPMScheduler  # noqa: E402
from synt_isic_amd.train import HipAdam, HipGradScaler, mse_loss, train_step_fused  # noqa: E402
from synt_isic_amd.unet import HipUNet2DModel  # noqa: E402
from synt_isic_amd.weights import synthetic_unet_state_dict  # noqa: E402


def run(B, size, steps, latency=False):
    dev = torch.device("cuda")
    m = HipUNet2DModel()
    m.load_state_dict(synthetic_unet_state_dict())
    m = m.to(dev)
    if latency:
        m.set_latency_mode(True)        # small-batch tile choices (the reference trains with batch 2)
    sched = HipDDPMScheduler(num_train_timesteps=1000, beta_schedule="squaredcos_cap_v2")
    opt = HipAdam(m.parameters(), lr=1e-4)
    scaler = HipGradScaler()
    m.train()
    g = torch.Generator(device=dev).manual_seed(0)
    images = torch.rand(B, 3, size, size, generator=g, device=dev) * 2 - 1
    noise = torch.randn(B, 3, size, size, generator=g, device=dev)
    ts = torch.randint(0, 1000, (B,), generator=g, device=dev)
    for _ in range(2):
        train_step_fused(m, sched, images, noise, ts, opt, scaler)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = train_step_fused(m, sched, images, noise, ts, opt, scaler)
    torch.cuda.synchronize()
    fused_ms = (time.perf_counter() - t0) * 1e3 / steps
    # split, spelled-out loop
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    fw = bw = op = 0.0
    for _ in range(steps):
        ev[0].record()
        pred = m(sched.add_noise(images, noise, ts), ts).sample
        l = mse_loss(pred, noise)
        ev[1].record()
        scaler.scale(l).backward()
        ev[2].record()
        scaler.step(opt); scaler.update()
        ev[3].record()
        torch.cuda.synchronize()
        fw += ev[0].elapsed_time(ev[1]); bw += ev[1].elapsed_time(ev[2]); op += ev[2].elapsed_time(ev[3])
    print(json.dumps({"batch": B, "size": size, "steps": steps, "latency_mode": bool(latency), "ms_per_step_fused": fused_ms,
                      "images_per_sec": B / (fused_ms * 1e-3), "forward_ms": fw / steps, "backward_ms": bw / steps,
                      "optimizer_ms": op / steps, "last_loss": loss, "dtype": "f32"}), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--latency", action="store_true", help="latency-mode tile choices (set_latency_mode) for small batches")
    a = ap.parse_args()
    if a.batch:
        run(a.batch, a.size, a.steps, a.latency)
    else:
        run(2, 128, a.steps, a.latency)        # the reference's batch (train_diffusion.py:59)
        run(32, 64, a.steps, a.latency)
