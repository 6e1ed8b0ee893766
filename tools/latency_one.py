#!/usr/bin/env python3
"""One configuration of tools/latency_b1.py, also with the loop alone (noise resident in HBM), for profiling:
    python tools/latency_one.py <B> <size> <latency 0|1> [T]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from synt_isic_amd.sampler import Sampler, draw_noise, run_sampling_loop  # noqa: E402
from synt_isic_amd.weights import synthetic_unet_state_dict  # noqa: E402

B, size, lat = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
T = int(sys.argv[4]) if len(sys.argv) > 4 else 50
s = Sampler(latency_mode=bool(lat))
m = s.add_model("NV", synthetic_unet_state_dict())
s.generate_seeds("NV", list(range(B)), T, (size, size))
torch.cuda.synchronize()
t0 = time.perf_counter()
s.generate_seeds("NV", list(range(B)), T, (size, size)).images.cpu()
dt = time.perf_counter() - t0
print(f"B={B} {size}x{size} latency={lat} T={T}: generate_seeds {dt / T * 1e3:.2f} ms per step ({dt:.3f} s per image batch)")
sched = s.create_scheduler(T)
x_T, z = draw_noise(list(range(B)), T - 1, (3, size, size))
x_T, z = x_T.cuda(), z.cuda()
run_sampling_loop(m, sched, x_T, z)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    r = run_sampling_loop(m, sched, x_T, z)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"   loop alone (noise resident): {dt / T * 1e3:.2f} ms per step")
