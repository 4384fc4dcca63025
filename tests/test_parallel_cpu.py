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


def _exchange_worker(rank, world, port, case, q):
    """The cropped exchange on CPU: the layout comes from the library (dsx_tileplan_pack_layout, host only), the
    collective is parallel.all_gather_flat on gloo, the byte moving (HIP kernels on the GPU) is the oracle's numpy."""
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffsplitting_amd import parallel
    from diffsplitting_amd.data.tiling import TilePlan
    from oracle import tiling
    data_shape, grid_shape, patch_shape, C = case
    plan = TilePlan(data_shape, grid_shape, patch_shape)
    oplan = tiling.TilePlan(data_shape, grid_shape, patch_shape)
    off, runs = plan.pack_layout(world)
    ooff, oruns = tiling.pack_layout(oplan, world)
    ok = np.array_equal(off, ooff) and np.array_equal(runs, oruns)
    stride = plan.rank_stride(world, C)
    rng = np.random.default_rng(11)                                   # every rank draws the same full set of tiles
    pred = rng.standard_normal((plan.total, C, patch_shape[1], patch_shape[2])).astype(np.float32)
    ids = parallel.shard_ids(plan.total, rank, world)
    flat = tiling.pack_rank(pred[ids], ids, oplan, off, stride)
    full = parallel.all_gather_flat(torch.from_numpy(flat))           # the one collective
    ok = ok and tuple(full.shape) == (world, stride)
    canvas = tiling.paste_packed(full.numpy(), oplan, off, C)
    ok = ok and np.array_equal(canvas, tiling.stitch(pred, oplan))    # bit-exact vs the one-rank stitch
    # the collective moves the canvas (every pixel exactly once: overlapping valid regions are clipped to what the
    # sequential paste leaves) plus the padding to equal runs
    valid = int(runs.sum()) * C * 4
    ok = ok and valid == int(np.prod(data_shape)) * C * 4
    gathered = world * stride * 4
    padding = int((runs.max() * world - runs.sum()) * C * 4)
    whole_tiles = world * ((plan.total + world - 1) // world) * C * patch_shape[1] * patch_shape[2] * 4
    ok = ok and gathered == valid + padding and gathered <= whole_tiles
    q.put((rank, bool(ok), gathered, whole_tiles))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [
    (2, ((5, 512, 512), (1, 128, 128), (1, 256, 256), 2)),            # 45 tiles (tests/test_tiling_setup.py)
    (3, ((2, 150, 210), (1, 16, 16), (1, 32, 32), 2)),                # ragged
    (3, ((7, 64, 64), (1, 32, 32), (1, 64, 64), 1)),                  # 7 tiles, one per frame
    (2, ((1, 64, 64), (1, 32, 32), (1, 64, 64), 2)),                  # 1 tile: rank 1 owns nothing
])
def test_cropped_exchange_gloo(world, case):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[:2] for r in res) == [(r, True) for r in range(world)], res


def test_cropped_exchange_bytes_hagen_490():
    """C3 (10 x 2048^2, grid 256, patch 512, 2 channels) at 8 ranks: the collective ships the canvas (335.5 MB) plus
    equal-run padding instead of 490 whole tiles (1.03 GB + padding to 62 per rank)."""
    from diffsplitting_amd.data.tiling import TilePlan
    plan = TilePlan((10, 2048, 2048), (1, 256, 256), (1, 512, 512))
    assert plan.total == 490
    off, runs = plan.pack_layout(8)
    assert int(runs.sum()) == 10 * 2048 * 2048
    gathered = 8 * plan.rank_stride(8, 2) * 4
    whole = 8 * 62 * 2 * 512 * 512 * 4
    assert 10 * 2048 * 2048 * 2 * 4 <= gathered < 1.12 * 10 * 2048 * 2048 * 2 * 4 and gathered < 0.36 * whole


def test_shard_ids_partition():
    from diffsplitting_amd import parallel
    for total in (0, 1, 7, 490):
        for world in (1, 2, 8):
            ids = sorted(i for r in range(world) for i in parallel.shard_ids(total, r, world))
            assert ids == list(range(total))
            assert sum(parallel.shard_count(total, r, world) for r in range(world)) == total


# ----------------------------------------------------------------------------- self-launching entry points
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launches_n_ranks_dry():
    """`python bench.py --gpus 2` without a launcher starts 2 fresh ranks itself (gloo, kernels stubbed by --dry):
    rank 0's JSON line reports n_gpus = 2, the MAX over ranks of the stubbed step time and the aggregate rate."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "1",
                        "--dry"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["dry"] is True and line["steps"] == 10
    assert abs(line["ms_per_step"] - 1.1) < 1e-9                      # MAX over ranks of the stub (rank 1: +10 %)
    assert abs(line["value"] - 2 * 16 / (1.1e-3 * 2000)) < 1e-6       # whole-job aggregate


def test_bench_refuses_mismatched_world():
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_self_launch_reports_a_failing_rank(tmp_path):
    import subprocess
    import sys
    script = tmp_path / "child.py"
    script.write_text("import os, sys\nprint('rank', os.environ['RANK'])\nsys.exit(3 if os.environ['RANK'] == '1' else 0)\n")
    code = ("import sys; sys.path.insert(0, %r)\nfrom diffsplitting_amd import parallel\n"
            "sys.exit(parallel.self_launch(2, [%r]))\n" % (ROOT, str(script)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "rank 0" in r.stdout and "ranks failed" in r.stderr


def test_self_launch_stops_the_others_when_one_rank_dies(tmp_path):
    """A rank that dies before the rendezvous must not leave rank 0 waiting: all children are watched together, the
    first failure (or the timeout) ends the rest."""
    import subprocess
    import sys
    import time
    script = tmp_path / "child.py"
    script.write_text("import os, sys, time\nprint('rank', os.environ['RANK'], flush=True)\n"
                      "sys.exit(5) if os.environ['RANK'] == '1' else time.sleep(600)\n")
    code = ("import sys; sys.path.insert(0, %r)\nfrom diffsplitting_amd import parallel\n"
            "sys.exit(parallel.self_launch(2, [%r]))\n" % (ROOT, str(script)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "ranks failed" in r.stderr and time.monotonic() - t0 < 60
    # the timeout form
    script.write_text("import time\ntime.sleep(600)\n")
    code = ("import sys; sys.path.insert(0, %r)\nfrom diffsplitting_amd import parallel\n"
            "sys.exit(parallel.self_launch(2, [%r], timeout=2))\n" % (ROOT, str(script)))
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "timed out" in r.stderr and time.monotonic() - t0 < 60
