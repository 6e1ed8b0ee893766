#!/usr/bin/env python3
"""Throughput of the BASELINE.json configurations bench.py does not print (they are parity-test cases there):

  config 4   batch 32, 3x128x128, T=1000 sampling        -> images/sec, per-kernel-class ms per step
  config 5   Time-SHAP: 16 coalitions x 32 images = 512 classifier forwards on [512,3,64,64] -> 224x224
             (seeded random ResNet18, fc -> 7)             -> forwards/sec, TFLOP/s; plus the as-coded N=50 form

One JSON line per configuration; run on one MI355X:  python tools/bench_configs.py [--steps 20] [--reps 5]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from synt_isic_amd import ops  # noqa: E402
from synt_isic_amd.classifier import HipMelanomaClassifier  # noqa: E402
from synt_isic_amd.sampler import Sampler, run_sampling_loop  # noqa: E402
from synt_isic_amd.scheduler import HipDDPMScheduler  # noqa: E402
from synt_isic_amd.weights import synthetic_resnet18_state_dict, synthetic_unet_state_dict  # noqa: E402
from synt_isic_amd.xai import compute_time_shap  # noqa: E402

UNET_GFLOP_128 = 75.277
RESNET_GFLOP_224 = 3.627


def sampling(batch, size, steps, warmup):
    dev = torch.device("cuda")
    s = Sampler()
    model = s.add_model("NV", synthetic_unet_state_dict())
    g = torch.Generator().manual_seed(0)
    x = torch.randn(batch, 3, size, size, generator=g).to(dev)
    z = torch.randn(max(steps, warmup, 3), batch, 3, size, size, generator=g).to(dev)

    def run(k):
        sched = HipDDPMScheduler(num_train_timesteps=1000, beta_schedule="squaredcos_cap_v2")
        sched.set_timesteps(1000)
        sched.timesteps = sched.timesteps[:k]                      # the first k steps of the T=1000 grid
        return run_sampling_loop(model, sched, x, z[:k])

    run(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    ops.profile_enable(dev, True)
    ops.profile_reset(dev)
    run(3)
    torch.cuda.synchronize()
    prof = ops.profile_read(dev)
    ops.profile_enable(dev, False)
    per = {k: round(v["ms"] / 3, 4) for k, v in prof.items() if v["launches"]}
    return ms, per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", type=int, default=0, help="4 or 5: just that configuration")
    a = ap.parse_args()
    dev = torch.device("cuda")

    if a.only in (0, 4):
        ms, per = sampling(32, 128, a.steps, 3)
        print(json.dumps({"config": "4: batch=32, 3x128x128, T=1000 DDPM sampling", "ms_per_step": round(ms, 3),
                          "images_per_sec": round(32 / (ms * 1000 / 1e3), 4), "unet_TFLOPs_algorithmic": round(
                              UNET_GFLOP_128 * 32 / ms, 1), "per_step_ms": per, "steps_timed": a.steps}), flush=True)
    if a.only == 4:
        return

    clf = HipMelanomaClassifier(num_classes=7)
    clf.load_state_dict(synthetic_resnet18_state_dict())
    clf = clf.to(dev).eval()
    x = torch.rand(512, 3, 64, 64, generator=torch.Generator().manual_seed(1)).mul_(2).sub_(1).to(dev)
    clf._scores(x, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        prob, score = clf._scores(x, 1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    ops.profile_enable(dev, True)
    ops.profile_reset(dev)
    clf._scores(x, 1)
    torch.cuda.synchronize()
    prof = ops.profile_read(dev)
    ops.profile_enable(dev, False)
    per = {k: round(v["ms"], 3) for k, v in prof.items() if v["launches"]}
    print(json.dumps({"config": "5: Time-SHAP coalitions, 512 classifier forwards [512,3,64,64] -> 224x224, ResNet18 fc->7",
                      "ms_per_pass": round(dt * 1e3, 3), "forwards_per_sec": round(512 / dt, 1),
                      "TFLOPs_algorithmic": round(512 * RESNET_GFLOP_224 / dt / 1e3, 1), "per_pass_ms": per}), flush=True)

    frames = torch.randn(50, 3, 64, 64, generator=torch.Generator().manual_seed(2)).clamp_(-1, 1).to(dev)
    ts = list(range(980, -1, -20))
    compute_time_shap(clf, frames, ts, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        imp, raw = compute_time_shap(clf, frames, ts, 1)
    dt = (time.perf_counter() - t0) / a.reps
    print(json.dumps({"config": "5b: Time-SHAP as coded (XAI.py:1179-1234), N=50 trajectory frames in one batch",
                      "ms_per_call": round(dt * 1e3, 3), "frames_per_sec": round(50 / dt, 1)}), flush=True)

    # Integrated Gradients (XAI.py:1039-1084): 50 Riemann points of one 3x64x64 image = one batched backward-to-input
    from synt_isic_amd.xai import compute_integrated_gradients
    img = frames[:1]
    base = torch.zeros_like(img)
    compute_integrated_gradients(clf, img, 1, baseline=base)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        compute_integrated_gradients(clf, img, 1, baseline=base)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    print(json.dumps({"config": "5c: Integrated Gradients, n_steps=50 (riemann_right), one 3x64x64 image: 50 forward+backward-to-input passes",
                      "ms_per_image": round(dt * 1e3, 3), "gradients_per_sec": round(50 / dt, 1)}), flush=True)


if __name__ == "__main__":
    main()
