"""BASELINE.json's configurations at their FULL sizes, under pytest on the GPU (VERDICT r01 "weak" item 3).

  config 2  batch=64, 3x64x64                -> tests/test_gpu_sampler.py::test_full_batch_properties
  config 3  512 seeds in 8 shards of 64      -> here, the eight shards sampled one after the other on this one GPU
  config 4  batch=32, 3x128x128              -> here
  config 5  16 coalitions x 32 images = 512 classifier forwards at 3x64x64 -> here

The T=1000 chain is shortened to a few steps (the oracle cannot follow a full-size run in test time); what is checked
at full size is what the domain offers independent of size: finiteness, run-to-run determinism, bit-equality of an
image with its own batch-1 run (the batch-1 run is what the golden fixtures pin), Shapley efficiency, and slices
against the CPU oracle.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
NV = 1                       # class id of "NV" (xai/XAI.py:196)


@pytest.fixture(scope="module")
def sampler(synthetic_sd):
    from synt_isic_amd.sampler import Sampler
    s = Sampler(DEV)
    s.add_model("NV", synthetic_sd)
    return s


@pytest.fixture(scope="module")
def clf():
    from synt_isic_amd.classifier import HipMelanomaClassifier
    from synt_isic_amd.weights import synthetic_resnet18_state_dict
    c = HipMelanomaClassifier(num_classes=7)
    c.load_state_dict(synthetic_resnet18_state_dict())
    return c.to(DEV).eval()


def test_config4_batch32_128(sampler, golden_dir):
    """[32,3,128,128]: attention over 1024 tokens (five blocks) and 256 tokens (mid block)."""
    from synt_isic_amd.sampler import draw_noise, run_sampling_loop
    model = sampler.models["NV"]
    g = np.load(os.path.join(golden_dir, "unet_forward_b1_128.npz"))
    # one forward of the full batch whose image 7 is the golden fixture's input: its row equals the golden output
    # (<= 2e-4) and is bit-equal to the batch-1 forward
    xg = torch.from_numpy(g["x"])
    x = torch.randn(32, 3, 128, 128, generator=torch.Generator().manual_seed(40))
    x[7] = xg[0]
    y = model(x.to(DEV), int(g["t"])).sample
    assert torch.isfinite(y).all()
    y1 = model(xg.to(DEV), int(g["t"])).sample
    assert torch.equal(y[7], y1[0])
    assert np.abs(y[7].cpu().numpy() - g["y"][0]).max() <= 2e-4
    # three steps of the T=1000 grid over the full batch: deterministic, finite, image k bit-equal to its own B=1 run
    seeds = list(range(100, 132))
    sched = sampler.create_scheduler(1000)
    sched.timesteps = sched.timesteps[:3]
    x_T, z = draw_noise(seeds, 3, (3, 128, 128))
    r1 = run_sampling_loop(model, sched, x_T.to(DEV), z.to(DEV))
    r2 = run_sampling_loop(model, sched, x_T.to(DEV), z.to(DEV))
    assert r1.steps_done == 3 and torch.equal(r1.latents, r2.latents) and torch.equal(r1.images, r2.images)
    assert torch.isfinite(r1.latents).all() and float(r1.latents.abs().max()) < 8.0
    for k in (0, 13, 31):
        rk = run_sampling_loop(model, sched, x_T[k:k + 1].to(DEV), z[:, k:k + 1].contiguous().to(DEV))
        assert torch.equal(rk.latents[0], r1.latents[k]) and torch.equal(rk.images[0], r1.images[k])


def test_config3_eight_shards_of_64_equal_one_run_of_512(sampler):
    """BASELINE config 3's partition (512 seeds, contiguous blocks of 64 per rank, dist.shard_seeds) with the eight ranks
    played one after the other on this GPU: the concatenation of the shard results is bit-identical to ONE batch-512 run
    -- what makes the 8-GPU gather equal to the single-GPU result.  (No 8-GPU run is available to the builder; the
    collective itself is exercised by test_nccl_world1_gather_and_max below and by the gloo test on the CPU.)"""
    from synt_isic_amd.dist import shard_seeds
    seeds = list(range(512))
    whole = sampler.generate_seeds("NV", seeds, T=2, size=(64, 64))
    assert whole.images.shape == (512, 64, 64, 3) and whole.steps_done == 2
    parts = [sampler.generate_seeds("NV", shard_seeds(seeds, 8, r), T=2, size=(64, 64)) for r in range(8)]
    assert all(p.images.shape[0] == 64 for p in parts)
    assert torch.equal(torch.cat([p.images for p in parts]), whole.images)
    assert torch.equal(torch.cat([p.latents for p in parts]), whole.latents)
    assert [h for p in parts for h in p.noise_hashes] == whole.noise_hashes


def test_config5_512_classifier_forwards(clf):
    """[512,3,64,64] -> 224x224 -> ResNet18 in ONE pass: rows bit-equal to the same images passed three at a time
    (no kernel choice depends on the batch), a slice against the CPU oracle (<= 2e-4 * max(1,|ref|))."""
    from oracle import resnet18 as ores
    from synt_isic_amd.weights import synthetic_resnet18_state_dict
    x = (torch.rand(512, 3, 64, 64, generator=torch.Generator().manual_seed(50)) * 2.4 - 1.2)
    xd = x.to(DEV)
    logits = clf(xd)
    assert logits.shape == (512, 7) and torch.isfinite(logits).all()
    assert torch.equal(logits, clf(xd))                                        # deterministic
    big = clf.workspace_bytes()                                                # pool of the 512-image pass
    for lo in (0, 255, 509):
        assert torch.equal(clf(xd[lo:lo + 3]), logits[lo:lo + 3])
    sd = synthetic_resnet18_state_dict()
    idx = [0, 1, 100, 511]
    ref = ores.classifier_forward(sd, x[idx])
    assert (logits[idx].cpu() - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
    # the workspace of the pass is released when the shape changes (the pool used to grow only)
    clf(xd[:4])
    assert 0 < clf.workspace_bytes() < big / 16


def test_config5_time_shap_16_coalitions_x_32_images(sampler, clf):
    """README.md:171-207 permutation Time-SHAP at the BASELINE size: T=15 steps -> the 16 nested coalitions of one
    permutation, 32 images at 3x64x64, scored in one [512,3,64,64] classifier batch.  Efficiency axiom (sum of phi =
    v(all) - v(none)), determinism, and independence from the classifier's chunking."""
    from synt_isic_amd import xai
    T, seeds = 15, list(range(32))
    gen = lambda: torch.Generator().manual_seed(3)
    r1 = xai.time_shap_permutation(sampler, clf, "NV", seeds, T, NV, n_permutations=1, size=(64, 64), generator=gen())
    assert r1["phi"].shape == (T,) and np.isfinite(r1["phi"]).all()
    assert abs(r1["phi"].sum() - (r1["v_full"] - r1["v_empty"])) < 1e-9
    r2 = xai.time_shap_permutation(sampler, clf, "NV", seeds, T, NV, n_permutations=1, size=(64, 64), generator=gen())
    assert np.array_equal(r1["phi"], r2["phi"])
    old = clf.max_forward_batch
    try:
        clf.max_forward_batch = 96                       # 512 forwards in chunks of 96: identical values
        r3 = xai.time_shap_permutation(sampler, clf, "NV", seeds, T, NV, n_permutations=1, size=(64, 64), generator=gen())
    finally:
        clf.max_forward_batch = old
    assert np.array_equal(r1["phi"], r3["phi"])


def test_nccl_world1_gather_and_max():
    """The N>1 code path of bench.py / examples/generate_sharded.py on the one GPU a test box has: torch.distributed with
    backend "nccl" (= RCCL), world_size 1 -- process-group creation on the GPU, dist.gather of uint8 images and the
    all_reduce(MAX) of the timing both go through RCCL."""
    import torch.distributed as dist
    from synt_isic_amd import dist as sdist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29517", rank=0, world_size=1)
    try:
        assert dist.get_backend() == "nccl"
        imgs = torch.randint(0, 256, (64, 64, 64, 3), dtype=torch.uint8, device=DEV)
        # with a group initialised, gather_images / max_over_ranks issue the collectives even for one rank
        out = sdist.gather_images(imgs, 64, dst=0)
        torch.cuda.synchronize()
        assert out is not imgs and torch.equal(out, imgs)
        assert sdist.max_over_ranks(1.25, torch.device(DEV)) == 1.25
        dist.barrier()
        with pytest.raises(ValueError):
            sdist.gather_images(imgs[:10], 64, dst=0)              # a shard of the wrong length is refused
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_the_gpu_rehearsal(tmp_path):
    """the N > 1 path beyond one rank, on the one GPU a test box has.  SISIC_SHARE_GPU=1
    (synt_isic_amd/dist.py) lets two ranks take the same device with gloo collectives on host copies: launcher, rendezvous on
    127.0.0.1, contiguous seed shards, two concurrent samplers, the gather and the max-over-ranks are the real code.
    (a) examples/generate_sharded.py with two ranks: the gathered images are bit-identical to the same seeds on one rank;
    (b) `python bench.py --gpus 2` starts its own two ranks, both pass the post-clock validation, one JSON line, n_gpus 2."""
    import json
    import subprocess
    import sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SISIC_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = {}
    for world in (1, 2):
        out = str(tmp_path / f"images_{world}.npy")
        if world == 1:
            cmd = [sys.executable, os.path.join(root, "examples", "generate_sharded.py")]
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                   "--master-port", "29531", os.path.join(root, "examples", "generate_sharded.py")]
        r = subprocess.run(cmd + ["--count", "6", "--T", "3", "--size", "64", "--batch", "4", "--out", out], env=env, cwd=root,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[world] = np.load(out)
    assert outs[1].shape == (6, 64, 64, 3) and np.array_equal(outs[1], outs[2])
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4",
                        "--no-cpu-baseline", "--no-e2e", "--profile-steps", "1"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["shared_gpu_rehearsal"] is True and d["backend"] == "gloo"
    assert d["validated"]["image0_bit_equal_to_its_B1_run"] and d["validated"]["all_latents_finite"]


BF3_LAUNCHES_PER_STEP = 33.0      # the stride-1 conv3x3 launches of a step with 64-channel x 16x16-pixel tiles (12 + 9 + 9) + 3 upsample


def test_bench_line_contract():
    """`python bench.py` (short run, no CPU leg) as the driver starts it: ONE JSON line on stdout carrying the metric of
    BASELINE.json, a roofline object for the dominant kernel (the bf16x3 Winograd form) and the
    host-inclusive leg; the HIP-event per-launch figures are self-consistent."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    assert d["metric"].startswith("images/sec at 3x64x64 T=1000") and base["metric"].startswith("images/sec at 3")
    assert d["unit"] == "images/sec"
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None and d["scaling"] == "weak"
    assert abs(d["value"] - 64.0 / (d["ms_per_step"] * 1e-3 * 1000)) < 1e-6 * d["value"]       # images/sec of a T=1000 run
    rf = d["roofline"]
    # the dominant kernel runs on the bf16 matrix pipe: frac = bf16 FLOPs issued / the dense bf16 peak, <= 1 by construction
    # (ADVICE r03); the fp32-equivalent rate against the f32 peak is a separately named field, not a utilisation
    assert rf["bound"] == "mfma" and rf["pipe"].startswith("bf16") and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0
    assert 0.05 < rf["frac"] <= 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert abs(rf["achieved"] - 6.0 * rf["executed_flops_per_launch"] / (rf["avg_launch_us"] * 1e-6) / 1e12) < 1e-6 * rf["achieved"]
    assert abs(rf["fp32_equivalent_TFLOPs"] * 6.0 - rf["achieved"]) < 1e-6 * rf["achieved"]
    assert abs(rf["fp32_equivalent_vs_f32_peak"] - rf["fp32_equivalent_TFLOPs"] / 157.3) < 1e-9
    assert 0.0 < rf["hbm_frac"] < 1.0 and (rf["traffic_over_algorithmic"] is None or rf["traffic_over_algorithmic"] > 0.9)
    v = d["validated"]
    assert v["all_latents_finite"] is True and v["image0_bit_equal_to_its_B1_run"] is True and v["steps"] == 6
    assert rf["launches_per_step"] == BF3_LAUNCHES_PER_STEP and "traffic" in rf and "traffic_source" in rf
    # (`traffic` and `rocprofv3_avg_launch_us` come from committed profiler files of another run and are labelled so in the
    #  line; comparing a live timing with them belongs to the measurement script, not to a correctness test -- ADVICE r02)
    # the host-inclusive leg has its own step count (>= 256) so that its fixed costs do not pose as the rate
    assert d["e2e"]["T"] >= 256 and d["e2e_images_per_sec"] > 0.5 * d["value"]

