#!/usr/bin/env python3
"""Headline benchmark: sampled images/s of the 2000-step 128x128 SR3
p_sample_loop (config sr_sr3_16_128: 16->128 SR3 UNet, batch 16 per GPU),
on N MI355X, one process per GPU.

    python bench.py --gpus 1 --steps 2000 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one reverse-diffusion step (UNet forward + posterior update) of the
whole per-GPU batch, replayed from a captured hipGraph with device (Philox)
noise; the default K=2000 is one complete sampling loop.  Every step costs the
same, so images/s = N*B / (2000 * ms_per_step).  Inputs (conditioning images,
initial noise, weights) are resident in HBM before the timed region.

Adds `roofline` (conv-MFMA kernel family, measured live with HIP events around
each launch of an eager replay) and `cpu_baseline` (the CPU oracle on the host
cores, a bounded sample) to the JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SAMPLE_STEPS = 2000
UNET = dict(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=(1, 2, 4, 8, 8),
            attn_res=(16,), res_blocks=2, image_size=128)             # config/sr_sr3_16_128.json
SCHEDULE = dict(schedule="linear", n_timestep=SAMPLE_STEPS, linear_start=1e-6, linear_end=1e-2)
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}                          # MI355X_MICROARCH.md, dense


def random_init_state_dict(names, shapes, seed=0):
    """Random-init weights of the architecture (no checkpoints exist offline)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for n, s in zip(names, shapes):
        if n.endswith("inv_freq"):
            continue
        if len(s) == 1:
            sd[n] = (1.0 + 0.2 * torch.randn(s, generator=g)) if n.endswith("weight") else 0.1 * torch.randn(s, generator=g)
        else:
            fan_in = int(np.prod(s[1:]))
            sd[n] = torch.randn(s, generator=g) * (1.0 / fan_in) ** 0.5
    return sd


def cpu_baseline(sd, batch=4, warm=2, timed=40):
    """The CPU oracle (oracle/, a checked restatement of the reference's p_sample)
    on this host's cores: a bounded sample (about 10-15 s) of `timed` reverse steps at batch 4,
    extrapolated to the 2000-step loop (per-step cost is step-independent)."""
    from oracle import samplers
    torch.set_grad_enabled(False)
    torch.set_num_threads(host_threads())
    sch = samplers.gaussian_schedule(SCHEDULE)
    osd = {"denoise_fn." + k: v for k, v in sd.items()}
    g = torch.Generator().manual_seed(1)
    cond = torch.randn((batch, 3, 128, 128), generator=g)
    img = torch.randn((batch, 3, 128, 128), generator=g)
    randn = lambda shape: torch.randn(shape, generator=g)
    ts = []
    for k in range(warm + timed):
        t0 = time.perf_counter()
        img = samplers.sr3_p_sample(osd, UNET, sch, img, SAMPLE_STEPS - 1 - k, cond, randn)
        ts.append(time.perf_counter() - t0)
    step = float(np.mean(ts[warm:]))
    return {"value": batch / (step * SAMPLE_STEPS), "unit": "images/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{timed} timed p_sample steps (after {warm} warm-up) of the torch-fp32 CPU oracle at "
                      f"batch {batch}, {step:.3f} s/step, extrapolated x{SAMPLE_STEPS}"}


def kernel_source_sha():
    """SHA-256 over the kernel and runtime sources the library is built from: what a counters file must name to be
    quoted next to a measurement of this build."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "diffsplitting_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".inc", ".h", ".cpp")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


def fp32_parity_line(sd, B, steps=20):
    """The parity build (fp32 MFMA operands, fp32 activations: the <= 1e-3 path) on the same workload: `steps` graph
    steps at batch B after 3 warm-up steps; the conv family's share is not separated here."""
    from diffsplitting_amd import engine
    cfg = engine.make_cfg("sr3", **{k: UNET[k] for k in ("in_channel", "out_channel", "inner_channel", "norm_groups",
                                                         "channel_mults", "attn_res", "res_blocks", "image_size")})
    eng = engine.UNetEngine(cfg, "sr3")
    eng.load_state_dict(sd)
    eng.finalize("f32")
    bufs, gam = engine.gaussian_buffers(SCHEDULE)
    full = engine.gaussian_step_table(bufs, gam, "sr3", clip_denoised=True)

    def sub(n):
        idx = np.arange(n) % SAMPLE_STEPS
        return engine.StepTableHost(full.tcond[idx], c1=full.c1[idx], c2=full.c2[idx], sigma=full.sigma[idx],
                                    a=full.a[idx], b=full.b[idx], predict_eps=True, clip=True)
    g = torch.Generator().manual_seed(5)
    cond = torch.randn((B, 3, 128, 128), generator=g).cuda()
    x = engine.randn((B, 3, 128, 128), seed=77)
    eng.sample_loop(sub(3), x, cond=cond, seed=1, use_graph=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.sample_loop(sub(steps), x, cond=cond, seed=2, use_graph=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    tf = eng.flops(128, 128) * B / (ms * 1e-3) / 1e12
    del eng
    return {"dtype": "f32", "batch_per_gpu": B, "steps": steps, "ms_per_step": ms, "sustained_tflops": tf,
            "peak": PEAK_TFLOPS["f32"], "frac": tf / PEAK_TFLOPS["f32"],
            "note": "whole step (UNet forward + update) of the fp32 parity build (v_mfma_f32_16x16x4_f32 in k_conv_ws, 32x32x2 in the other conv kernels); peak = dense fp32 MFMA"}


def names_of(k):
    return {0: "conv_mfma", 1: "conv_naive", 2: "gn_stats", 3: "gn_finalize", 4: "attn_gemm", 5: "softmax",
            6: "splitk_reduce"}[k]


def host_threads():
    """CPU threads this process may really use (cgroup quota / affinity), not the host's core count."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        pass
    return max(1, n)


def roofline(eng, ex, dtype, iters=3):
    from diffsplitting_amd._lib import check, lib
    n = lib.dsx_exec_num_ops(ex)
    ms = (C.c_float * n)()
    check(lib.dsx_exec_profile(ex, 1, ms, C.c_void_p(torch.cuda.current_stream().cuda_stream)))  # warm
    check(lib.dsx_exec_profile(ex, iters, ms, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    desc = C.create_string_buffer(256)
    kind, fl, by = C.c_int(), C.c_double(), C.c_double()
    rows = []
    for i in range(n):
        check(lib.dsx_exec_op_info(ex, i, desc, 256, C.byref(kind), C.byref(fl), C.byref(by)))
        rows.append((kind.value, desc.value.decode(), fl.value, by.value, ms[i]))
    conv = [r for r in rows if r[0] == 0]
    conv_ms_eager = sum(r[4] for r in conv)
    conv_fl = sum(r[2] for r in conv)
    total_ms = sum(r[4] for r in rows)
    # the dominant family's launch duration as it runs inside the captured step: all conv launches of one
    # forward replayed back-to-back as a hipGraph between two HIP events on a side stream (the eager per-launch
    # numbers above contain the launch gaps; rocprofv3 --kernel-trace of this command reports the same sum)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    gms, nl, gall, gother, n_all, n_other = C.c_float(), C.c_int(), C.c_float(), C.c_float(), C.c_int(), C.c_int()
    with torch.cuda.stream(side):
        sp = C.c_void_p(side.cuda_stream)
        check(lib.dsx_exec_time_kind(ex, 0, 20, C.byref(gms), C.byref(nl), sp))              # conv launches alone
        check(lib.dsx_exec_time_kind(ex, -1, 20, C.byref(gall), C.byref(n_all), sp))          # the whole forward
        check(lib.dsx_exec_time_kind(ex, -2, 20, C.byref(gother), C.byref(n_other), sp))      # everything but the convs
    torch.cuda.current_stream().wait_stream(side)
    assert nl.value == len(conv) and n_all.value - n_other.value == len(conv)
    conv_ms_alone = float(gms.value)
    # the conv family INSIDE the forward: the launches in front of a conv (k_gn_finalize) warm the L2s with its
    # weights, which a replay of the conv launches alone does not see; rocprofv3 --kernel-trace of the sampling
    # loop reports this in-context sum
    conv_ms = float(gall.value) - float(gother.value)
    achieved = conv_fl / (conv_ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[dtype]
    # HBM bytes per conv launch and MFMA-busy fraction come from rocprofv3 PMC passes of this same command
    # (tools/measure_r02.sh -> profiles/r02_counters.json): counters cannot be read live from inside the process
    # (tools/measure_r03.sh -> profiles/r03_counters.json, which records the SHA-256 of the kernel sources it was
    # taken on): counters cannot be read live from inside the process.  Quoted only when that hash is this build's;
    # otherwise null, and `counters_from` says why.
    traffic, mfma_busy, counters_from = None, None, None
    cpath = os.path.join(ROOT, "profiles", "r03_counters.json")
    if os.path.exists(cpath) and dtype == "bf16":
        try:
            cj = json.load(open(cpath))
            here = kernel_source_sha()
            if cj.get("kernel_source_sha") == here:
                traffic, mfma_busy = cj.get("hbm_bytes_per_conv_launch"), cj.get("mfma_busy_frac_conv")
                counters_from = f"profiles/r03_counters.json @ kernel sources {here[:12]} (rocprofv3 --pmc passes of this command)"
            else:
                counters_from = (f"none: profiles/r03_counters.json was taken on kernel sources "
                                 f"{str(cj.get('kernel_source_sha'))[:12]}, this build is {here[:12]}")
        except Exception:
            traffic, mfma_busy, counters_from = None, None, "none: profiles/r03_counters.json unreadable"
    by_kind = {}
    for r in rows:
        by_kind[r[0]] = by_kind.get(r[0], 0.0) + r[4]
    dump = os.environ.get("DSX_BENCH_OPS")
    if dump:
        with open(dump, "w") as f:
            json.dump([{"kind": names_of(r[0]), "desc": r[1], "gflop": r[2] / 1e9, "mbytes": r[3] / 1e6,
                        "ms": r[4]} for r in rows], f, indent=0)
    top = sorted(rows, key=lambda r: -r[4])[:8]
    print("[bench] eager per-launch profile: total %.3f ms/step over %d launches; conv-MFMA %.3f ms eager, %.3f ms "
          "inside the captured forward (%d launches; %.3f ms replayed alone)" % (total_ms, n, conv_ms_eager, conv_ms, len(conv), conv_ms_alone), file=sys.stderr)
    print("[bench] ms by kernel family: " + ", ".join(f"{names_of(k)} {v:.3f}" for k, v in sorted(by_kind.items())),
          file=sys.stderr)
    for r in top:
        tf = r[2] / (r[4] * 1e-3) / 1e12 if r[4] > 0 else 0.0
        print(f"[bench]   {r[4]:8.4f} ms  {tf:8.1f} TF/s  {r[1]}", file=sys.stderr)
    return {"bound": "mfma", "kernel": "k_conv_ws / k_conv_mfma / k_conv_img / k_conv_first (fused GN+Swish+conv implicit GEMM on MFMA; all conv launches of one UNet forward)",
            "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
            "mfma_busy": mfma_busy, "counters_from": counters_from,
            "launches": len(conv), "avg_launch_ms": conv_ms / max(1, len(conv)),
            "conv_ms_per_step": conv_ms, "conv_ms_per_step_replayed_alone": conv_ms_alone,
            "forward_ms_graph": float(gall.value), "conv_ms_per_step_eager_events": conv_ms_eager,
            "all_kernels_ms_per_step_eager_events": total_ms}


def dry_run(args, rank, world):
    """The distributed skeleton of main() on CPU with gloo: barrier, stubbed step time, MAX over ranks, the final
    all-gather of the images, rank 0 prints the JSON line (marked "dry": true, not a measurement)."""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
        dist.barrier()
    B, K = args.batch, args.steps
    elapsed = 1e-3 * K * (1.0 + 0.1 * rank)                        # stub: rank r is 10 r % slower
    x = torch.full((B, 3, 8, 8), float(rank))
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        out = torch.empty((world * B, 3, 8, 8))
        dist.all_gather_into_tensor(out, x)
        x = out
    assert x.shape[0] == world * B and float(x[-1, 0, 0, 0]) == world - 1
    ms = elapsed * 1e3 / K
    if rank == 0:
        print(json.dumps({"metric": "sampled images/sec (2000-step 128x128 SR3)", "dry": True,
                          "value": world * B / (ms * 1e-3 * SAMPLE_STEPS), "unit": "images/s", "n_gpus": world,
                          "steps": K, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "none (dry run)"}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=SAMPLE_STEPS)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU (BASELINE config: 16)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-fp32-parity", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--split", type=int, default=int(os.environ.get("DSX_BENCH_SPLIT", "1")),
                    help="run the per-GPU batch as this many independent sub-batches on separate HIP streams")
    ap.add_argument("--dry", action="store_true",
                    help="launcher / collective rehearsal on CPU (gloo): no kernels run, the timing is a stub and "
                         "the JSON line says so; used by the CPU tests of the N > 1 path")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start N fresh ranks ourselves, BEFORE anything touches the GPU
    from diffsplitting_amd import parallel
    if parallel.needs_self_launch(args.gpus):
        raise SystemExit(parallel.self_launch(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU "
                         f"(torchrun --nproc-per-node {args.gpus}, or plain `python bench.py --gpus {args.gpus}`)")
    if args.dry:
        return dry_run(args, rank, world)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    # one GPU per rank; DSX_DIST_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (the ranks
    # share them, collectives staged through the host; the JSON line then says so and is not a scaling measurement)
    backend = os.environ.get("DSX_DIST_BACKEND") or "nccl"
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"--gpus {world} but {ndev} GPU(s) visible: RCCL needs one device per rank")
    torch.cuda.set_device(local % ndev)
    dev = torch.device("cuda", local % ndev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    on_host = dist is not None and backend != "nccl"     # gloo: reductions / gathers of device tensors go through the host
    torch.set_grad_enabled(False)

    from diffsplitting_amd import engine
    cfg = engine.make_cfg("sr3", **{k: UNET[k] for k in ("in_channel", "out_channel", "inner_channel",
                                                         "norm_groups", "channel_mults", "attn_res",
                                                         "res_blocks", "image_size")})
    eng = engine.UNetEngine(cfg, "sr3")
    sd = random_init_state_dict(eng.param_names, eng.param_shapes, seed=0)
    eng.load_state_dict(sd)
    eng.finalize(args.dtype)

    B, K, W = args.batch, args.steps, args.warmup
    bufs, gam = engine.gaussian_buffers(SCHEDULE)
    full = engine.gaussian_step_table(bufs, gam, "sr3", clip_denoised=True)

    def sub(n):  # the first n reverse steps of the real T=2000 schedule (wraps for n > 2000)
        idx = np.arange(n) % SAMPLE_STEPS
        return engine.StepTableHost(full.tcond[idx], c1=full.c1[idx], c2=full.c2[idx], sigma=full.sigma[idx],
                                    a=full.a[idx], b=full.b[idx], predict_eps=True, clip=True)

    g = torch.Generator().manual_seed(100 + rank)
    cond = torch.randn((B, 3, 128, 128), generator=g).to(dev)      # stands in for the bicubic-upsampled LR
    x = engine.randn((B, 3, 128, 128), seed=1000 + rank)
    use_graph = not args.no_graph
    nsplit = max(1, args.split)
    assert B % nsplit == 0
    streams = [torch.cuda.Stream() for _ in range(nsplit)] if nsplit > 1 else [None]

    def run_loop(table, x):
        # batch items never interact (GroupNorm and attention are per image), so sub-batches are
        # independent sampling loops; on separate streams their kernels overlap on the GPU
        if nsplit == 1:
            return eng.sample_loop(table, x, cond=cond, seed=rank, use_graph=use_graph)[0]
        xs = [c.contiguous() for c in x.chunk(nsplit)]
        cs = [c.contiguous() for c in cond.chunk(nsplit)]
        outs = [eng.sample_loop(table, xs[i], cond=cs[i], seed=rank * 97 + i, use_graph=use_graph,
                                stream=streams[i], slot=i)[0] for i in range(nsplit)]
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
        return torch.cat(outs)

    if W > 0:
        run_loop(sub(W), x)
    x = engine.randn((B, 3, 128, 128), seed=2000 + rank)
    tab = sub(K)

    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x = run_loop(tab, x)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if on_host else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the sampler's only exchange: one all-gather of the finished images (outside the loop)
        src = x.contiguous().cpu() if on_host else x.contiguous()
        out = torch.empty((world * B, 3, 128, 128), dtype=torch.float32, device=src.device)
        dist.all_gather_into_tensor(out.view(-1), src.view(-1))
        x = out.to(dev)
    assert torch.isfinite(x).all(), "sampler produced non-finite pixels"
    assert eng.handoff_timeouts() == 0, "a bounded hand-off spin of k_conv_ws gave up: the timed pixels are wrong"

    ms_per_step = elapsed * 1e3 / K
    images_per_s = world * B / (ms_per_step * 1e-3 * SAMPLE_STEPS)
    line = {
        "metric": "sampled images/sec (2000-step 128x128 SR3)",
        "value": images_per_s, "unit": "images/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "sr_sr3_16_128: GaussianDiffusion.p_sample_loop, 16->128 SR3 UNet (97.8M params), "
                               "T=2000 linear schedule; one step = UNet forward + posterior update of the batch",
                   "batch_per_gpu": B, "global_batch": world * B, "image": "128x128x3", "sample_steps": SAMPLE_STEPS,
                   "parallelism": f"{world} independent replicas, final RCCL all-gather" if backend == "nccl" else
                                  f"{world} ranks sharing {ndev} GPU(s), {backend} transport through the host: a REHEARSAL of the N > 1 path, not a scaling measurement",
                   "hipgraph": use_graph, "streams": nsplit,
                   "noise": "device Philox4x32-10", "weights": "random-init, seed 0"},
    }
    if rank == 0:
        gf = eng.flops(128, 128)
        line["unet_gflop_per_image_step"] = gf / 1e9
        line["sustained_tflops_per_gpu"] = gf * B / (ms_per_step * 1e-3) / 1e12
        if not args.no_roofline:
            line["roofline"] = roofline(eng, eng.executor(B // nsplit, 128, 128, 3), args.dtype)
        if world == 1 and not args.no_fp32_parity and args.dtype == "bf16":
            line["fp32_parity"] = fp32_parity_line(sd, B)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd)                       # batch 4: the reference's CPU-runnable case
            b16 = cpu_baseline(sd, batch=16, warm=1, timed=8)             # the headline batch as well
            line["cpu_baseline"]["batch16"] = {"value": b16["value"], "unit": b16["unit"], "sample": b16["sample"]}
            line["speedup_vs_cpu_baseline"] = images_per_s / line["cpu_baseline"]["value"]
        print(json.dumps(line))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
