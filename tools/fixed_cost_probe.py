"""Does a short sampling call cost more per step than a long one?  K = 20 / 40 / 100 steps of the T=1000 grid at batch 64 through
run_sampling_loop (the bench's timed region): the per-step time is the same (5.36 ms on the box measured), i.e. no fixed per-call
cost worth the name -- the driver's 20-step run reads what a full run reads.   python tools/fixed_cost_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synt_isic_amd.sampler import Sampler, run_sampling_loop
from synt_isic_amd.weights import synthetic_unet_state_dict
dev = torch.device("cuda", 0)
s = Sampler(dev); m = s.add_model("NV", synthetic_unet_state_dict())
def make(n):
    sched = s.create_scheduler(1000); sched.timesteps = sched.timesteps[:n]
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(64, 3, 64, 64, generator=g, device=dev)
    z = torch.randn(n, 64, 3, 64, 64, generator=g, device=dev)
    return sched, x, z
sched, x, z = make(5); run_sampling_loop(m, sched, x, z)
for n in (20, 20, 40, 100, 20):
    sched, x, z = make(n)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = run_sampling_loop(m, sched, x, z)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"K={n:4d}: {dt*1e3:8.2f} ms total, {dt*1e3/n:6.3f} ms/step", flush=True)
