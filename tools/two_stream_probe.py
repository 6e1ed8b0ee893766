#!/usr/bin/env python3
"""Does splitting a batch into independent lanes on separate HIP streams raise whole-job throughput?
Every image's reverse-diffusion chain is independent, so lanes are correct by construction; the question is only
whether the low-resolution levels (too few workgroups for 256 CUs) overlap with another lane's high-resolution work.

    python tools/two_stream_probe.py [--batch 64] [--steps 40] [--lanes 1,2,4]
"""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from synt_isic_amd.sampler import run_sampling_loop  # noqa: E402
from synt_isic_amd.scheduler import HipDDPMScheduler  # noqa: E402
from synt_isic_amd.unet import HipUNet2DModel  # noqa: E402
from synt_isic_amd.weights import synthetic_unet_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--lanes", default="1,2,4")
    a = ap.parse_args()
    dev = torch.device("cuda")
    sd = synthetic_unet_state_dict()
    for lanes in [int(v) for v in a.lanes.split(",")]:
        per = a.batch // lanes
        models, streams, xs = [], [], []
        for i in range(lanes):
            m = HipUNet2DModel()
            m.load_state_dict(sd)
            models.append(m.to(dev).eval())
            streams.append(torch.cuda.Stream(dev))
            xs.append(torch.randn(per, 3, a.size, a.size, device=dev))
        sched = HipDDPMScheduler(num_train_timesteps=1000, beta_schedule="squaredcos_cap_v2")

        def lane(i, steps):
            s = HipDDPMScheduler(num_train_timesteps=1000, beta_schedule="squaredcos_cap_v2")
            s.set_timesteps(steps)
            with torch.cuda.stream(streams[i]):
                run_sampling_loop(models[i], s, xs[i], None)

        def run(steps):
            th = [threading.Thread(target=lane, args=(i, steps)) for i in range(lanes)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()

        run(5)
        t0 = time.perf_counter()
        run(a.steps)
        dt = time.perf_counter() - t0
        print(f"lanes={lanes} x batch {per}: {dt / a.steps * 1e3:.3f} ms per step of {a.batch} images "
              f"-> {a.batch / (dt / a.steps * 1000):.3f} images/s at T=1000", flush=True)
        del models, xs


if __name__ == "__main__":
    main()
