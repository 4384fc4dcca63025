"""One process per GPU (torchrun): tile / batch sharding and the single
collective the sampling path needs — an all-gather of the finished tiles
(RCCL over xGMI with backend "nccl"; "gloo" on CPU for tests).

Tiles are independent, so the loop itself has no data-path collective: rank r
owns tiles ``r, r+W, r+2W, …`` (keeps every rank's share spread over frames),
runs its reverse loops, then all ranks exchange predictions once.
"""
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for 1 process)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or dist.is_initialized():
        return rank(), world_size()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = os.environ.get("DSX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        # one GPU per rank; with fewer GPUs than ranks (rehearsals of the N > 1 path on a one-GPU box, gloo transport)
        # the ranks share them -- RCCL itself refuses two ranks on one device
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    dist.init_process_group(backend=backend)
    return rank(), world_size()


def _gather_through_host(flat, group):
    """gloo has no all_gather_into_tensor for device tensors: stage through the host (rehearsal transport only)."""
    world = dist.get_world_size(group)
    src = flat.detach().to("cpu").contiguous().view(-1)
    out = torch.empty(world * src.numel(), dtype=src.dtype)
    dist.all_gather_into_tensor(out, src, group=group)
    return out.to(flat.device)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def needs_self_launch(n):
    """True when ``n`` > 1 ranks were asked for but this process was not started by a launcher."""
    return int(n) > 1 and "WORLD_SIZE" not in os.environ


def self_launch(n, argv, timeout=None):
    """Start ``n`` fresh children ``python argv...`` (one per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
    rendezvous on 127.0.0.1) and wait for them.  Must be called BEFORE this process touches the GPU: the children
    are new processes (never an exec of a process that has initialised HIP).  Rank 0's stdout is forwarded to ours,
    every rank's stderr to ours.  All children are watched together: the first non-zero exit (or the timeout) ends
    the others, so a rank that dies before the rendezvous does not leave its peers waiting for it.  Returns 0 only if
    every child exited 0."""
    import threading
    import time
    n = int(n)
    port = free_port()
    procs = []
    out0 = []
    try:
        for r in range(n):
            env = dict(os.environ)
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable] + list(argv), env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None))
        drain = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
        drain.start()
        deadline = None if timeout is None else time.monotonic() + float(timeout)
        bad, timed_out = [], False
        while True:
            rcs = [p.poll() for p in procs]
            bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
            if bad or all(rc is not None for rc in rcs):
                break
            if deadline is not None and time.monotonic() > deadline:
                timed_out = True
                break
            time.sleep(0.05)
        if bad or timed_out:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_end = time.monotonic() + 5.0
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, t_end - time.monotonic()))
                except subprocess.TimeoutExpired:
                    p.kill()
        drain.join(timeout=10)
        if out0 and out0[0]:
            sys.stdout.write(out0[0].decode(errors="replace"))
            sys.stdout.flush()
        if timed_out:
            print(f"[self_launch] timed out after {timeout} s: the ranks were stopped", file=sys.stderr)
            return 1
        if bad:
            print(f"[self_launch] ranks failed (rank, exit code): {bad}", file=sys.stderr)
            return 1
        return 0
    finally:
        for p in procs:                       # whatever happened above: no child outlives this call
            if p.poll() is None:
                p.kill()


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def shard_ids(total, rank_, world):
    """Tile ids owned by ``rank_``: i ≡ rank (mod world)."""
    return list(range(rank_, total, world))


def shard_count(total, rank_, world):
    return (total - rank_ + world - 1) // world if rank_ < total else 0


def all_gather_tiles(local, total, group=None):
    """``local``: this rank's predictions (n_local, C, h, w) for tile ids
    ``shard_ids(total, rank, world)``.  Returns (total, C, h, w) ordered by tile id on
    every rank.  One padded all-gather (equal counts per rank, as RCCL wants)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    r = dist.get_rank(group)
    per = (total + world - 1) // world
    n_local = shard_count(total, r, world)
    assert local.shape[0] == n_local, (local.shape, n_local)
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:n_local] = local
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    out = out.view((world, per) + tuple(local.shape[1:]))
    # rank q's k-th tile has id q + k*world  ->  interleave back to id order
    full = out.transpose(0, 1).reshape((per * world,) + tuple(local.shape[1:]))
    return full[:total].contiguous()


def all_gather_flat(flat, group=None):
    """The tiled path's one collective: every rank's packed run of CROPPED tiles (equal-sized 1-D tensors,
    ``TilePlan.rank_stride`` elements) -> (world, rank_stride) on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return flat.view(1, -1)
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        return _gather_through_host(flat, group).view(world, -1)
    out = torch.empty(world * flat.numel(), dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(out, flat.contiguous().view(-1), group=group)
    return out.view(world, -1)


def all_gather_batch(x, group=None):
    """Replica-parallel samplers (SR3 benchmark): concatenate every rank's finished batch."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return x
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x.contiguous(), group=group)
    return out
