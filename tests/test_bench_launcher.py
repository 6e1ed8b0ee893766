"""`python bench.py --gpus N` starts its own N ranks.  The launcher touches no GPU, so its plumbing is
tested here on the CPU with the `--dry-run` leg: rendezvous on 127.0.0.1 (gloo), shard_range, gather_images, max_over_ranks,
ONE JSON line of the bench schema relayed from rank 0, a non-zero exit when a rank dies."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")

# every key of the N=1 line of a real run (tests/test_gpu_configs.py::test_bench_line_contract checks their values on the GPU)
SCHEMA = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
          "dtype", "dtype_note", "data", "config", "roofline", "cpu_baseline", "e2e_images_per_sec", "e2e", "validated",
          "multi_gpu_note"}


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=e)


def test_launcher_two_ranks_dry_run_prints_one_line():
    r = _run(["--gpus", "2", "--dry-run", "--steps", "5", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["backend"] == "gloo"
    assert d["steps"] == 5 and d["warmup"] == 1 and d["scaling"] == "weak" and d["unit"] == "images/sec"
    assert d["config"]["global_batch"] == 128 and d["config"]["batch_per_gpu"] == 64
    assert SCHEMA <= set(d)
    assert "gather of 128 stand-in images" in r.stderr


def test_launcher_eight_ranks_dry_run():
    """the driver's largest case (`bench.py --gpus 8`): eight ranks rendezvous, shard 512 images, gather them, print ONE line"""
    r = _run(["--gpus", "8", "--dry-run", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 512 and d["config"]["batch_per_gpu"] == 64
    assert "gather of 512 stand-in images" in r.stderr


def test_single_rank_dry_run_has_the_same_schema():
    r = _run(["--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and set(d) == SCHEMA | {"dry_run", "backend"} and d["backend"] is None


def test_launcher_reports_a_dead_rank():
    r = _run(["--gpus", "2", "--dry-run"], env={"SISIC_BENCH_DRY_FAIL_RANK": "1"}, timeout=120)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")]


def test_torchrun_style_environment_still_works():
    """the driver's other form: the rank environment is already there (world 1 here), no launcher involved"""
    r = _run(["--gpus", "1", "--dry-run"], env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1",
                                               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29511"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    r = _run(["--gpus", "2", "--dry-run"], env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
