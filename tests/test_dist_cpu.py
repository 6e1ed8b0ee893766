"""The N>1 path on CPU: world_size-2 gloo processes exercise the sharding and the gather of finished samples
(synt_isic_amd/dist.py).  The images are stand-ins derived from the seeds -- the test covers the host logic
that makes the sharded result identical to the single-process one, not the sampler (that needs the GPU)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from synt_isic_amd import dist as sdist


def _fake_images(seeds):
    """uint8 [n,4,4,3] uniquely determined by each seed (stands in for the per-seed sampler output)."""
    out = torch.empty((len(seeds), 4, 4, 3), dtype=torch.uint8)
    for i, s in enumerate(seeds):
        g = torch.Generator().manual_seed(int(s))
        out[i] = torch.randint(0, 256, (4, 4, 3), generator=g, dtype=torch.uint8)
    return out


def _worker(rank, world, port, n_total, result_path):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    r, w, _ = sdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    seeds = list(range(100, 100 + n_total))
    mine = sdist.shard_seeds(seeds, w, r)
    local = _fake_images(mine)
    gathered = sdist.gather_images(local, n_total, dst=0)
    worst = sdist.max_over_ranks(float(rank + 1), "cpu")
    assert worst == float(world)
    if r == 0:
        assert torch.equal(gathered, _fake_images(seeds))        # same as the unsharded run, in seed order
        torch.save(gathered, result_path)
    else:
        assert gathered is None
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("n_total", [8, 7])       # even shards and ragged shards (4+3)
def test_two_rank_shard_and_gather(tmp_path, n_total):
    path = str(tmp_path / "gathered.pt")
    mp.spawn(_worker, args=(2, _free_port(), n_total, path), nprocs=2, join=True)
    got = torch.load(path)
    assert got.shape == (n_total, 4, 4, 3)
    assert torch.equal(got, _fake_images(list(range(100, 100 + n_total))))


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [sdist.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert sdist.shard_range(512, 8, 3) == (192, 256)            # BASELINE config 3: 64 contiguous seeds per GPU
    with pytest.raises(ValueError):
        sdist.shard_range(8, 2, 2)


def test_single_process_gather_is_identity():
    x = _fake_images([1, 2, 3])
    assert sdist.gather_images(x, 3) is x
    with pytest.raises(ValueError):
        sdist.gather_images(x, 4)
