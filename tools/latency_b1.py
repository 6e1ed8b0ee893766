#!/usr/bin/env python3
"""Single-image latency (the reference's own use: generate_single_image, B=1, image_generator.py:369-403): ms per denoising
step of the whole ``generate_seeds`` call (seeds in -> uint8 image on the host) against the loop alone (noise resident),
and where the difference goes.   python tools/latency_b1.py

The pieces are timed on the SECOND and later calls at a shape (the first sizes the workspace and captures the step)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from synt_isic_amd import _lib  # noqa: E402
from synt_isic_amd.sampler import NoiseStream, Sampler, draw_noise, run_sampling_loop  # noqa: E402
from synt_isic_amd.weights import synthetic_unet_state_dict  # noqa: E402


def timed(fn, reps=3):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best, r


def main():
    T = 50
    sd = synthetic_unet_state_dict()
    lib = _lib.load()
    for latency in ((True,) if os.environ.get("SISIC_LATENCY_ONLY128") else (True, False)):
        s = Sampler(latency_mode=latency)
        m = s.add_model("NV", sd)
        for size in ((128,) if os.environ.get("SISIC_LATENCY_ONLY128") else (128, 64)):
            for B in (1,):
                seeds = list(range(B))
                s.generate_seeds("NV", seeds, T, (size, size))               # first call at the shape: not timed
                builds0 = lib.sisic_unet_graph_builds(m.handle)
                whole, res = timed(lambda: s.generate_seeds("NV", seeds, T, (size, size)).images.cpu())
                builds1 = lib.sisic_unet_graph_builds(m.handle)
                # the loop alone: noise already on the device
                sched = s.create_scheduler(T)
                x_T, z = draw_noise(seeds, T - 1, (3, size, size))
                x_d, z_d = x_T.to(s.device), z.to(s.device)
                loop, _ = timed(lambda: run_sampling_loop(m, sched, x_d, z_d))
                # pieces of the host path
                t_sched, _ = timed(lambda: s.create_scheduler(T))
                t_ns, ns = timed(lambda: NoiseStream(seeds, (3, size, size), s.device, s.noise_segment_steps,
                                                     buffer_cache=s._noise_buffers), reps=1)
                ns.close()
                t_rng, _ = timed(lambda: torch.randn((T - 1, 3, size, size), generator=torch.Generator().manual_seed(0)))
                t_rng4, _ = timed(lambda: torch.randn((4, 3, size, size), generator=torch.Generator().manual_seed(0)))
                t_xup, _ = timed(lambda: x_T.to(s.device))
                t_dl, _ = timed(lambda: res.cpu() if hasattr(res, "cpu") else None)
                print(f"B={B} {size}x{size} latency_mode={int(latency)} T={T}: generate_seeds {whole / T * 1e3:.2f} ms per step "
                      f"({whole * 1e3:.1f} ms per call), loop alone {loop / T * 1e3:.2f} ms per step ({loop * 1e3:.1f} ms), "
                      f"ratio {whole / loop:.3f}; graphs built during the timed calls: {builds1 - builds0}")
                print(f"    pieces (ms): scheduler tables {t_sched * 1e3:.2f}, NoiseStream setup (generators, x_T, cached pool / "
                      f"buffers) {t_ns * 1e3:.2f}, RNG of all {T - 1} steps {t_rng * 1e3:.2f} (first segment of 4: "
                      f"{t_rng4 * 1e3:.2f}), x_T upload {t_xup * 1e3:.2f}", flush=True)


if __name__ == "__main__" and not (os.environ.get("SISIC_LATENCY_PROFILE") or os.environ.get("SISIC_LATENCY_FIXED")):
    main()


def profile_one(size=128, T=50):
    """cProfile of one generate_seeds call (second call at the shape): where the host time of the 128x128 case goes"""
    import cProfile
    import pstats
    s = Sampler(latency_mode=True)
    s.add_model("NV", synthetic_unet_state_dict())
    s.generate_seeds("NV", [0], T, (size, size))
    s.generate_seeds("NV", [0], T, (size, size))
    pr = cProfile.Profile()
    pr.enable()
    s.generate_seeds("NV", [0], T, (size, size)).images.cpu()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)


if __name__ == "__main__" and os.environ.get("SISIC_LATENCY_PROFILE"):
    profile_one()


def fixed_cost(size=128):
    """time of ONE sisic_sample call against its step count (noise resident): slope = ms per step, intercept = per-call cost"""
    s = Sampler(latency_mode=True)
    m = s.add_model("NV", synthetic_unet_state_dict())
    s.generate_seeds("NV", [0], 50, (size, size))
    for T in (4, 8, 16, 22, 50, 4, 8):
        sched = s.create_scheduler(50)
        sched.timesteps = sched.timesteps[:T]
        x_T, z = draw_noise([0], sum(1 for t in sched.timesteps if int(t) > 0), (3, size, size))
        x_d, z_d = x_T.to(s.device), z.to(s.device)
        dt, _ = timed(lambda: run_sampling_loop(m, sched, x_d, z_d, cancel_flag=s.cancel))
        dt2, _ = timed(lambda: run_sampling_loop(m, sched, x_d, z_d))
        print(f"one call of {T:2d} steps at {size}x{size}: {dt * 1e3:7.2f} ms with the cancel flag, {dt2 * 1e3:7.2f} ms without", flush=True)


if __name__ == "__main__" and os.environ.get("SISIC_LATENCY_FIXED"):
    fixed_cost()
