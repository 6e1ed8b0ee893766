#!/usr/bin/env python3
"""bench.py -- images/sec of the DDPM reverse-diffusion sampler on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: batch=64 per GPU, 3x64x64, T=1000 DDPM sampling with the reference
architecture (model_manager.py:173-194), seeded synthetic weights and synthetic Gaussian inputs resident
in HBM.  One "step" is one pass of the loop body of image_generator.py:400-403 (UNet forward + scheduler
step) over the batch; K consecutive steps of the T=1000 grid are timed inside ONE sisic_sample call,
together with the uint8 epilogue and (N>1) the gather of finished samples.  value = images per second
of a full T=1000 run = N*64 / (ms_per_step * 1000 steps).  With the default K=1000 that is a measured
full run, for smaller K it is the per-step time extrapolated to 1000 steps (all steps launch identical
kernels).

Besides the driver's fields the JSON line carries
  roofline      conv3x3 (90 % of the FLOPs): algorithmic FLOP/s (and bytes/s) per launch, measured with HIP
                events on the launch stream over a few profiled steps, vs the fp32 MFMA peak (157.3 TFLOP/s)
  cpu_baseline  the CPU oracle (oracle/, torch fp32) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

BATCH_PER_GPU = 64
SIZE = 64
T_FULL = 1000
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense f32-in MFMA = vector fp32 peak
PEAK_HBM_GBPS = 8000.0            # HBM3E spec


def cpu_baseline(sd, seconds_budget: float = 25.0):
    """The oracle on the host cores: B=8, 3x64x64, consecutive steps from t=999 (2 warm-up, then timed)."""
    from oracle import ddpm as oddpm, unet as ounet
    # the box's CPU share, not the host's core count: oversubscribing OpenMP threads stalls the oracle
    try:
        n_aff = len(os.sched_getaffinity(0))
    except AttributeError:
        n_aff = os.cpu_count() or 1
    n_thr = max(1, min(n_aff, int(os.environ.get("SISIC_CPU_THREADS", "16"))))
    torch.set_num_threads(n_thr)
    B = 8
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 3, SIZE, SIZE, generator=g)
    sched = oddpm.DDPMSchedulerOracle()
    sched.set_timesteps(T_FULL)
    ts = [int(t) for t in sched.timesteps]
    done, t_used, i = 0, 0.0, 0
    t_start = time.perf_counter()
    with torch.no_grad():
        while True:
            t0 = time.perf_counter()
            eps = ounet.unet_forward(sd, x, ts[i])
            x = sched.step(eps, ts[i], x, noise=torch.randn(x.shape, generator=g))
            dt = time.perf_counter() - t0
            i += 1
            if i > 2:
                done += 1
                t_used += dt
                if t_used >= seconds_budget or done >= 20:
                    break
            if time.perf_counter() - t_start > 3 * seconds_budget and done >= 1:
                break
    s_per_step = t_used / done
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": B / (s_per_step * T_FULL),
        "unit": "images/sec",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"B={B}, 3x{SIZE}x{SIZE}, {done} consecutive steps after 2 warm-up steps, "
                  f"{s_per_step:.3f} s/step, extrapolated x{T_FULL} steps; cpu_count={os.cpu_count()}, "
                  f"cpu='{cpu_model}'",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="images per GPU (BASELINE config: 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=5)
    args = ap.parse_args()

    from synt_isic_amd import dist as sdist
    from synt_isic_amd import ops
    from synt_isic_amd.sampler import Sampler, run_sampling_loop
    from synt_isic_amd.weights import synthetic_unet_state_dict

    def log(msg):
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    rank, world, local = sdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    K, W, B = args.steps, args.warmup, args.batch
    if not (1 <= K <= T_FULL and 0 <= W <= T_FULL):
        raise SystemExit("--steps and --warmup must be within 1..1000")

    sd = synthetic_unet_state_dict()
    sampler = Sampler(dev)
    model = sampler.add_model("NV", sd)

    def make_run(n_steps):
        sched = sampler.create_scheduler(T_FULL)
        sched.timesteps = sched.timesteps[:n_steps]          # first n steps of the T=1000 grid
        n_noise = sum(1 for t in sched.timesteps if int(t) > 0)
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        x_T = torch.randn(B, 3, SIZE, SIZE, generator=g, device=dev)
        z = torch.randn(n_noise, B, 3, SIZE, SIZE, generator=g, device=dev)
        return sched, x_T, z

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    log(f"model loaded on {dev}; warm-up {W} steps")
    # warm-up: W untimed steps (also sizes the library workspace)
    if W > 0:
        sched, x_T, z = make_run(W)
        run_sampling_loop(model, sched, x_T, z)
    sched, x_T, z = make_run(K)
    barrier()
    t0 = time.perf_counter()
    res = run_sampling_loop(model, sched, x_T, z)            # K steps + uint8 epilogue
    gathered = sdist.gather_images(res.images, B * world, dst=0)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = sdist.max_over_ranks(elapsed, dev)
    assert res.steps_done == K
    if rank == 0:
        assert gathered.shape == (B * world, SIZE, SIZE, 3)

    ms_per_step = elapsed * 1e3 / K
    value = B * world / (ms_per_step * 1e-3 * T_FULL)
    log(f"timed {K} steps: {ms_per_step:.3f} ms/step -> {value:.3f} images/sec")

    roofline = None
    cpu = None
    if rank == 0:
        # ---- roofline leg: HIP events around every conv3x3 launch on the launch stream
        n_prof = max(1, min(args.profile_steps, K))
        sched_p, x_p, z_p = make_run(n_prof)
        ops.profile_reset(dev)
        ops.profile_enable(dev, True)
        run_sampling_loop(model, sched_p, x_p, z_p)
        torch.cuda.synchronize(dev)
        prof = ops.profile_read(dev)
        ops.profile_enable(dev, False)
        c3 = prof["conv3x3"]
        sec = c3["ms"] * 1e-3
        tflops = c3["flops"] / sec / 1e12                    # algorithmic: 2*MAC of the direct convolution
        tflops_exec = c3["flops_executed"] / sec / 1e12      # issued to the matrix pipe (Winograd launches need 16/36)
        gbps = c3["bytes"] / sec / 1e9
        traffic = None                                       # HBM bytes per launch from the committed PMC passes
        try:
            with open(os.path.join(ROOT, "profiles", "r01", "pmc_summary.json")) as f:
                traffic = json.load(f)["_conv3x3_all"]["hbm_MB_per_launch"] * 1e6
        except (OSError, KeyError, ValueError):
            pass
        roofline = {
            "kernel": "conv3x3 (52 launches per step: conv_winograd_kernel F(2x2,3x3) -- 9-position form for the"
                      " upsampled inputs, K-split form + splitk_reduce_kernel at 8x8 -- conv_mfma_kernel for stride 2,"
                      " conv3x3_smallcout_kernel), fp32 on v_mfma_f32_32x32x2_f32",
            "bound": "mfma",
            "achieved": tflops,
            "peak": PEAK_FP32_MFMA_TFLOPS,
            "unit": "TFLOP/s",
            "frac": tflops / PEAK_FP32_MFMA_TFLOPS,
            "traffic": traffic,
            "achieved_is": "algorithmic FLOPs (2*MAC of the direct form) / measured launch time",
            "executed_TFLOPs": tflops_exec,
            "executed_frac": tflops_exec / PEAK_FP32_MFMA_TFLOPS,
            "avg_launch_us": c3["ms"] * 1e3 / max(1, c3["launches"]),
            "launches": c3["launches"],
            "algorithmic_bytes_per_launch": c3["bytes"] / max(1, c3["launches"]),
            "algorithmic_GBps": gbps,
            "hbm_frac": gbps / PEAK_HBM_GBPS,
            "per_step_ms": {k: v["ms"] / n_prof for k, v in prof.items()},
        }
        log(f"conv3x3: {tflops:.1f} algorithmic / {tflops_exec:.1f} executed TFLOP/s over {c3['launches']} launches")
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(sd)
            log(f"cpu baseline: {cpu['value']:.5f} images/sec on {cpu['cores']} threads")

    if rank == 0:
        line = {
            "metric": "images/sec at 3x64x64 T=1000 (DDPM reverse-diffusion sampling)",
            "value": value,
            "unit": "images/sec",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"batch={B} per GPU, 3x{SIZE}x{SIZE}, T={T_FULL} DDPM sampling "
                                   f"(BASELINE configs[1]{'; configs[2] sharding' if world > 1 else ''})",
                       "batch_per_gpu": B, "global_batch": B * world, "T": T_FULL, "steps_timed": K,
                       "parallelism": f"independent seeds x{world}, one gather of uint8 images"},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
