#!/usr/bin/env python3
"""Host-RNG ceiling of the sharded sampler.

Every rank of an N-GPU run draws its images' z_t on the CPU (one torch.Generator per image, the noise contract of
synt_isic_amd/sampler.py) while its GPU samples: at 64 images x 3x64x64 per rank and ~7 ms per step a rank consumes
64*12288 / 7.2e-3 = 109 M normals/s, eight ranks 0.87 G/s from ONE host.  This tool runs R producer processes (one per
would-be rank) with W worker threads each, exactly the work NoiseStream._draw_z does, for a fixed time, and prints the
aggregate rate next to that requirement.

    python tools/noise_producers.py --ranks 8 --workers 16 --seconds 5
"""
import argparse
import json
import multiprocessing as mp
import os
import time


def _producer(rank, workers, seconds, images, seg, q):
    import torch
    from concurrent.futures import ThreadPoolExecutor
    torch.set_num_threads(1)
    chw = (3, 64, 64)
    gens = [torch.Generator().manual_seed(1000 * rank + b) for b in range(images)]
    buf = torch.empty((seg, images) + chw, dtype=torch.float32)

    def draw(b):
        buf[:, b] = torch.randn((seg,) + chw, generator=gens[b])
        return seg * 3 * 64 * 64

    n = 0
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=workers) as pool:
        while time.perf_counter() - t0 < seconds:
            n += sum(pool.map(draw, range(images)))
    q.put((rank, n, time.perf_counter() - t0))


def measure(ranks=8, workers=16, seconds=3.0, images=64, seg=8):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_producer, args=(r, workers, seconds, images, seg, q)) for r in range(ranks)]
    t0 = time.perf_counter()
    for p in procs:
        p.start()
    res = [q.get() for _ in procs]
    for p in procs:
        p.join()
    wall = time.perf_counter() - t0
    total = sum(n for _, n, _ in res)
    rate = sum(n / dt for _, n, dt in res)
    need_per_rank = images * 3 * 64 * 64 / 7.2e-3
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count()
    return {"ranks": ranks, "workers_per_rank": workers, "images_per_rank": images, "normals": total,
            "aggregate_normals_per_sec": rate, "per_rank_normals_per_sec": rate / ranks,
            "needed_per_rank": need_per_rank, "needed_total": need_per_rank * ranks,
            "fraction_of_need": rate / (need_per_rank * ranks), "affinity_cpus": aff, "cpu_count": os.cpu_count(),
            "wall_s": wall}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--seconds", type=float, default=5.0)
    a = ap.parse_args()
    print(json.dumps(measure(a.ranks, a.workers, a.seconds)))
