"""The N>1 path on CPU: tile sharding + the padded all-gather with gloo, world_size 2 and 3."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffsplitting_amd import parallel
    ids = parallel.shard_ids(total, rank, world)
    assert len(ids) == parallel.shard_count(total, rank, world)
    # each rank "predicts" its tiles: tile i is filled with the value i (2 channels, 4x4)
    local = torch.stack([torch.full((2, 4, 4), float(i)) for i in ids]) if ids else torch.zeros((0, 2, 4, 4))
    full = parallel.all_gather_tiles(local, total)
    ok = full.shape == (total, 2, 4, 4) and all(float(full[i, 0, 0, 0]) == i for i in range(total))
    batch = parallel.all_gather_batch(torch.full((3, 2), float(rank)))
    ok = ok and batch.shape == (3 * world, 2) and float(batch[3 * (world - 1), 0]) == world - 1
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 45), (2, 490), (3, 7), (2, 1)])
def test_shard_and_all_gather(world, total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]


def test_shard_ids_partition():
    from diffsplitting_amd import parallel
    for total in (0, 1, 7, 490):
        for world in (1, 2, 8):
            ids = sorted(i for r in range(world) for i in parallel.shard_ids(total, r, world))
            assert ids == list(range(total))
            assert sum(parallel.shard_count(total, r, world) for r in range(world)) == total
