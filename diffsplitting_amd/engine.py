"""Python host of the HIP sampling engine: owns ``dsx_model`` / ``dsx_exec``
handles, maps reference state-dict keys onto the library's parameter table and
builds the per-step scalar tables (bit-exact with the reference's schedules).

torch is used for device memory, streams and host-side schedule arithmetic
only; all sampling compute happens inside ``libdsx.so``.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib
from ._lib import DsxError, check, lib


def _as_tuple(v):
    if v is None:
        return ()
    if isinstance(v, int):
        return (v,)
    return tuple(int(x) for x in v)


def make_cfg(flavour, in_channel, out_channel, inner_channel, norm_groups, channel_mults, attn_res,
             res_blocks, image_size, with_time_emb=True):
    """Keyword names of the reference ``UNet.__init__`` (sr3 unet.py:161-174)."""
    cfg = _lib.UnetCfg()
    cfg.flavour = _lib.FLAVOUR_SR3 if flavour == "sr3" else _lib.FLAVOUR_DDPM
    cfg.in_channel = int(in_channel)
    cfg.out_channel = int(out_channel if out_channel is not None else in_channel)
    cfg.inner_channel = int(inner_channel)
    cfg.norm_groups = int(norm_groups)
    mults = _as_tuple(channel_mults)
    attn = _as_tuple(attn_res)
    if len(mults) > 8 or len(attn) > 8:
        raise DsxError("at most 8 channel_mults / attn_res entries are supported")
    cfg.n_mults = len(mults)
    for i, m in enumerate(mults):
        cfg.channel_mults[i] = m
    cfg.n_attn_res = len(attn)
    for i, a in enumerate(attn):
        cfg.attn_res[i] = a
    cfg.res_blocks = int(res_blocks)
    cfg.image_size = int(image_size)
    cfg.with_time_emb = 1 if with_time_emb else 0
    return cfg


def dtype_code(dtype):
    """'f32' | 'bf16' | 'f16' (or the torch dtypes) -> DSX_DTYPE_*: the MFMA operand / activation storage type."""
    if dtype in ("bf16", torch.bfloat16):
        return _lib.DTYPE_BF16
    if dtype in ("f16", "fp16", torch.float16):
        return _lib.DTYPE_F16
    if dtype in ("f32", "fp32", torch.float32, None):
        return _lib.DTYPE_F32
    raise DsxError(f"unknown compute dtype {dtype!r} (f32, bf16, f16)")


def plan_dry_run(cfg, dtype, B, H, W, cond_channels=0):
    """Host-only planner check (no GPU): (sizing_bytes, planning_bytes, launches) of dsx_plan_dry_run."""
    a, b, n = C.c_size_t(), C.c_size_t(), C.c_int()
    check(lib.dsx_plan_dry_run(C.byref(cfg), dtype_code(dtype), int(B), int(H), int(W), int(cond_channels),
                               C.byref(a), C.byref(b), C.byref(n)))
    return int(a.value), int(b.value), int(n.value)


def _dptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class UNetEngine:
    """One UNet on the current GPU: parameter table + executors per (B,H,W,cond)."""

    def __init__(self, cfg, flavour):
        self.flavour = flavour
        self.cfg = cfg
        h = C.c_void_p()
        check(lib.dsx_model_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self._execs = {}
        self._finalized_dtype = None
        self.param_names, self.param_shapes = [], []
        name = C.create_string_buffer(256)
        nd = C.c_int()
        shp = (C.c_int64 * 4)()
        for i in range(lib.dsx_model_num_params(h)):
            check(lib.dsx_model_param_info(h, i, name, 256, C.byref(nd), shp))
            self.param_names.append(name.value.decode())
            self.param_shapes.append(tuple(int(shp[k]) for k in range(nd.value)))

    def __del__(self):
        try:
            for ex in self._execs.values():
                lib.dsx_exec_destroy(ex)
            if self._h:
                lib.dsx_model_destroy(self._h)
        except Exception:
            pass

    # ---- weights -----------------------------------------------------------
    def load_state_dict(self, sd, prefix=""):
        """``sd``: reference-keyed tensors (``prefix`` + UNet key).  Missing
        ``inv_freq`` is derived; anything else missing raises."""
        for i, (n, shp) in enumerate(zip(self.param_names, self.param_shapes)):
            key = prefix + n
            if key not in sd:
                if n.endswith("inv_freq"):
                    continue
                raise DsxError(f"state dict lacks {key}")
            t = sd[key].detach().to("cpu", torch.float32).contiguous()
            if tuple(t.shape) != shp:
                if "noise_func" in n and len(shp) >= 1 and t.shape[0] == 2 * shp[0] and tuple(t.shape[1:]) == tuple(shp[1:]):
                    raise DsxError(f"{key}: {tuple(t.shape)} is a FeatureWiseAffine with use_affine_level=True "
                                   f"((1 + gamma) x + beta, sr3 unet.py:34-50); the engine implements the additive form "
                                   f"every reference config uses (use_affine_level=False), expected {shp}")
                raise DsxError(f"{key}: shape {tuple(t.shape)} != {shp}")
            check(lib.dsx_model_set_param(self._h, i, C.c_void_p(t.data_ptr()), t.numel()))
        if self.flavour == "sr3" and self.cfg.with_time_emb:
            # PositionalEncoding's table, computed with the same torch ops as the
            # reference (sr3 unet.py:24-28) so the host constant is bit-identical
            count = self.cfg.inner_channel // 2
            step = torch.arange(count, dtype=torch.float32) / count
            freq = torch.exp(-math.log(1e4) * step).contiguous()
            check(lib.dsx_model_set_posenc_freq(self._h, C.c_void_p(freq.data_ptr()), count))
        self._drop_execs()
        self._finalized_dtype = None

    def _drop_execs(self):
        for ex in self._execs.values():
            lib.dsx_exec_destroy(ex)
        self._execs = {}

    def finalize(self, dtype="f32"):
        _lib.require_gpu()
        code = dtype_code(dtype)
        if self._finalized_dtype != code:
            self._drop_execs()
            check(lib.dsx_model_finalize(self._h, code))
            self._finalized_dtype = code

    # ---- packed-weight cache (N4: the repack next to `*_gen.pth`, model/model.py:153-166) -------------------
    _PACK_MAGIC = b"DSXPACK3"

    def _pack_header(self, dtype_code_, key):
        import hashlib
        cfg_bytes = bytes(memoryview(self.cfg))                      # the UNet configuration (ctypes struct)
        h = hashlib.sha256(cfg_bytes + bytes([dtype_code_, lib.dsx_abi_version()]) + str(key).encode()).digest()
        return self._PACK_MAGIC + h

    def save_packed(self, path, key):
        """Write the device image of the finalized model (fragment-ordered weights etc.) to ``path``."""
        if self._finalized_dtype is None:
            raise DsxError("finalize() before save_packed()")
        n = C.c_size_t()
        check(lib.dsx_model_packed_bytes(self._h, self._finalized_dtype, C.byref(n)))
        buf = np.empty(n.value, dtype=np.uint8)
        check(lib.dsx_model_export_packed(self._h, buf.ctypes.data_as(C.c_void_p), n.value))
        import hashlib
        tmp = f"{path}.tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(self._pack_header(self._finalized_dtype, key))
            f.write(hashlib.sha256(buf.data).digest())               # of the payload: a damaged file is refused
            buf.tofile(f)
        os.replace(tmp, path)

    def finalize_from_packed(self, path, dtype, key):
        """Finalize from a cached image if ``path`` holds one for this configuration, dtype, ABI and ``key`` (the
        checkpoint's hash); returns False (and does nothing) otherwise.  No parameter needs to be set."""
        _lib.require_gpu()
        code = dtype_code(dtype)
        if not os.path.exists(path):
            return False
        n = C.c_size_t()
        check(lib.dsx_model_packed_bytes(self._h, code, C.byref(n)))
        hdr = self._pack_header(code, key)
        import hashlib
        if os.path.getsize(path) != len(hdr) + 32 + n.value:
            return False
        with open(path, "rb") as f:
            if f.read(len(hdr)) != hdr:
                return False
            digest = f.read(32)
            buf = np.fromfile(f, dtype=np.uint8, count=n.value)
        if buf.size != n.value or hashlib.sha256(buf.data).digest() != digest:
            return False
        self._drop_execs()
        check(lib.dsx_model_finalize_packed(self._h, code, buf.ctypes.data_as(C.c_void_p), n.value))
        self._finalized_dtype = code
        return True

    def flops(self, H, W):
        return float(lib.dsx_model_flops(self._h, int(H), int(W)))

    # ---- execution ---------------------------------------------------------
    def executor(self, B, H, W, cond_channels=0, slot=0):
        """One executor (plan + workspace) per geometry and ``slot``; concurrent loops on
        different streams must use different slots (an executor's calls are not concurrent)."""
        if self._finalized_dtype is None:
            raise DsxError("finalize() must precede execution")
        key = (int(B), int(H), int(W), int(cond_channels), int(slot))
        ex = self._execs.get(key)
        if ex is None:
            ex = C.c_void_p()
            check(lib.dsx_exec_create(self._h, key[0], key[1], key[2], key[3], C.byref(ex)))
            self._execs[key] = ex
        return ex

    def workspace_bytes(self, B, H, W, cond_channels=0):
        return int(lib.dsx_exec_workspace_bytes(self.executor(B, H, W, cond_channels)))

    def handoff_timeouts(self):
        """Sum over this network's executors of dsx_exec_handoff_timeouts: bounded spins of the conv kernel's
        loader -> compute hand-off that gave up.  0 in every correct run (synchronises the device)."""
        total = 0
        for ex in self._execs.values():
            n = C.c_uint(0)
            check(lib.dsx_exec_handoff_timeouts(ex, C.byref(n)))
            total += int(n.value)
        return total

    def num_launches(self, B, H, W, cond_channels=0):
        return int(lib.dsx_exec_num_launches(self.executor(B, H, W, cond_channels)))

    def op_descriptions(self, B, H, W, cond_channels=0):
        """Descriptions of the plan's launches in order (dsx_exec_op_info), e.g. 'conv3x3 64->64 @128x128 tile128x64 ws'."""
        ex = self.executor(B, H, W, cond_channels)
        desc = C.create_string_buffer(256)
        kind, fl, by = C.c_int(), C.c_double(), C.c_double()
        out = []
        for i in range(int(lib.dsx_exec_num_ops(ex))):
            check(lib.dsx_exec_op_info(ex, i, desc, 256, C.byref(kind), C.byref(fl), C.byref(by)))
            out.append(desc.value.decode())
        return out

    def forward(self, x, time=None, cond_channels=0):
        """denoise_fn(x, t): x (B,Cin,H,W) fp32 cuda NCHW -> (B,Cout,H,W)."""
        if not x.is_cuda or x.dtype != torch.float32:
            raise DsxError("x must be a float32 CUDA tensor")
        x = x.contiguous()
        B, _, H, W = x.shape
        ex = self.executor(B, H, W, cond_channels)
        y = torch.empty((B, self.cfg.out_channel, H, W), dtype=torch.float32, device=x.device)
        if self.cfg.with_time_emb:
            t = time.to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
            if t.numel() not in (1, B):
                raise DsxError(f"time must have 1 or B={B} elements, got {t.numel()}")
            check(lib.dsx_unet_forward(ex, _dptr(x), _dptr(t), t.numel(), _dptr(y), _stream_ptr()))
        else:
            check(lib.dsx_unet_forward(ex, _dptr(x), C.c_void_p(0), 0, _dptr(y), _stream_ptr()))
        return y

    def sample_loop(self, table, x_init, cond=None, noise=None, seed=0, snapshot_steps=(),
                    use_graph=True, stream=None, slot=0):
        """Runs ``table`` (a ``StepTableHost``) from ``x_init`` (B,C,H,W) in place.
        Returns (final_state, snapshots) with snapshots (n_snap,B,C,H,W) or None.
        Asynchronous on the current (or given) stream."""
        x = x_init.contiguous()
        B, Cx, H, W = x.shape
        cc = 0 if cond is None else cond.shape[1]
        ex = self.executor(B, H, W, cc, slot)
        if cond is not None:
            cond = cond.to(dtype=torch.float32).contiguous()
        if noise is not None:
            noise = noise.to(device=x.device, dtype=torch.float32).contiguous()
            if tuple(noise.shape) != (table.n_steps, B, Cx, H, W):
                raise DsxError(f"noise must be {(table.n_steps, B, Cx, H, W)}, got {tuple(noise.shape)}")
        snaps = None
        steps = np.asarray(sorted(int(s) for s in snapshot_steps), dtype=np.int32)
        if len(steps):
            snaps = torch.empty((len(steps), B, Cx, H, W), dtype=torch.float32, device=x.device)
        if stream is not None:
            # the loop runs asynchronously on `stream`: tell torch's caching allocator, or a
            # tensor dropped by the caller is recycled while the kernels still read it
            stream.wait_stream(torch.cuda.current_stream())
            for t in (x, cond, noise, snaps):
                if t is not None:
                    t.record_stream(stream)
        sp = C.c_void_p(stream.cuda_stream) if stream is not None else _stream_ptr()
        check(lib.dsx_sample_loop(ex, C.byref(table.c_struct()), _dptr(cond), _dptr(x), _dptr(noise),
                                  C.c_uint64(int(seed) & (2 ** 64 - 1)),
                                  steps.ctypes.data_as(C.POINTER(C.c_int32)) if len(steps) else None,
                                  len(steps), _dptr(snaps), 1 if use_graph else 0, sp))
        return x, snaps


class StepTableHost:
    """Host-side per-step scalars handed to ``dsx_sample_loop`` (include/dsx.h)."""

    def __init__(self, tcond, c1, c2, sigma, a=None, b=None, predict_eps=False, clip=False, per_sample=0):
        """Columns of shape (n_steps,), or (n_steps, B) with ``per_sample=B`` (one schedule per batch element)."""
        f = lambda v: None if v is None else np.ascontiguousarray(np.asarray(v, dtype=np.float32))
        self.tcond, self.a, self.b, self.c1, self.c2, self.sigma = f(tcond), f(a), f(b), f(c1), f(c2), f(sigma)
        self.n_steps = int(self.tcond.shape[0])
        self.per_sample = int(per_sample)
        if self.per_sample and tuple(self.tcond.shape) != (self.n_steps, self.per_sample):
            raise DsxError("per-sample step tables hold (n_steps, B) values per column")
        self.predict_eps, self.clip = bool(predict_eps), bool(clip)

    def c_struct(self):
        p = lambda v: v.ctypes.data_as(C.POINTER(C.c_float)) if v is not None else None
        return _lib.StepTable(self.n_steps, int(self.predict_eps), int(self.clip), p(self.tcond), p(self.a),
                              p(self.b), p(self.c1), p(self.c2), p(self.sigma), self.per_sample)


# ---------------------------------------------------------------------------
# schedules (host arithmetic, bit-exact with the reference)
# ---------------------------------------------------------------------------
def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """model/sr3_modules/diffusion.py:19-49 (float64 numpy)."""
    if schedule == "quad":
        betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=np.float64) ** 2
    elif schedule == "linear":
        betas = np.linspace(linear_start, linear_end, n_timestep, dtype=np.float64)
    elif schedule in ("warmup10", "warmup50"):
        frac = 0.1 if schedule == "warmup10" else 0.5
        betas = linear_end * np.ones(n_timestep, dtype=np.float64)
        warm = int(n_timestep * frac)
        betas[:warm] = np.linspace(linear_start, linear_end, warm, dtype=np.float64)
    elif schedule == "const":
        betas = linear_end * np.ones(n_timestep, dtype=np.float64)
    elif schedule == "jsd":
        betas = 1.0 / np.linspace(n_timestep, 1, n_timestep, dtype=np.float64)
    elif schedule == "cosine":
        ts = torch.arange(n_timestep + 1, dtype=torch.float64) / n_timestep + cosine_s
        alphas = torch.cos(ts / (1 + cosine_s) * math.pi / 2).pow(2)
        alphas = alphas / alphas[0]
        betas = (1 - alphas[1:] / alphas[:-1]).clamp(max=0.999).numpy()
    else:
        raise NotImplementedError(schedule)
    return betas


def gaussian_buffers(schedule_opt):
    """The buffers ``set_new_noise_schedule`` registers (sr3 diffusion.py:92-139):
    dict of fp32 torch tensors + the float64 ``sqrt_alphas_cumprod_prev`` table."""
    betas = make_beta_schedule(schedule_opt["schedule"], schedule_opt["n_timestep"],
                               schedule_opt["linear_start"], schedule_opt["linear_end"])
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    t32 = lambda v: torch.tensor(v, dtype=torch.float32)
    pv = betas * (1.0 - ac_prev) / (1.0 - ac)
    bufs = {
        "betas": t32(betas),
        "alphas_cumprod": t32(ac),
        "alphas_cumprod_prev": t32(ac_prev),
        "sqrt_alphas_cumprod": t32(np.sqrt(ac)),
        "sqrt_one_minus_alphas_cumprod": t32(np.sqrt(1.0 - ac)),
        "log_one_minus_alphas_cumprod": t32(np.log(1.0 - ac)),
        "sqrt_recip_alphas_cumprod": t32(np.sqrt(1.0 / ac)),
        "sqrt_recipm1_alphas_cumprod": t32(np.sqrt(1.0 / ac - 1)),
        "posterior_variance": t32(pv),
        "posterior_log_variance_clipped": t32(np.log(np.maximum(pv, 1e-20))),
        "posterior_mean_coef1": t32(betas * np.sqrt(ac_prev) / (1.0 - ac)),
        "posterior_mean_coef2": t32((1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac)),
    }
    return bufs, np.sqrt(np.append(1.0, ac))


def gaussian_step_table(bufs, gamma_table_f64, kind, clip_denoised=True):
    """Rows in execution order i = T-1 .. 0 (sr3 diffusion.py:196-199).

    ``kind`` "sr3": tcond = fp32(sqrt_alphas_cumprod_prev[i+1]) (:153-154);
    "ddpm": tcond = float(i) (ddpm diffusion.py:216-217).  sigma = exp(0.5*logvar)
    evaluated with torch fp32 like the reference (:175); sigma = 0 at i == 0
    (no noise at t == 0: :174 / ddpm :199-203)."""
    T = bufs["betas"].shape[0]
    order = np.arange(T - 1, -1, -1)
    if kind == "sr3":
        tcond = torch.tensor(gamma_table_f64[order + 1], dtype=torch.float64).to(torch.float32).numpy()
    else:
        tcond = order.astype(np.float32)
    sigma = (0.5 * bufs["posterior_log_variance_clipped"]).exp().numpy().copy()
    sigma[0] = 0.0
    g = lambda k: bufs[k].numpy()[order]
    return StepTableHost(tcond, c1=g("posterior_mean_coef1"), c2=g("posterior_mean_coef2"),
                         sigma=sigma[order], a=g("sqrt_recip_alphas_cumprod"),
                         b=g("sqrt_recipm1_alphas_cumprod"), predict_eps=True, clip=clip_denoised)


def indi_step_table(num_timesteps, t_float_start, e=0.01):
    """The scalars of InDI.inference (indi.py:62-69,83-88): float64 ``cur_t -= delta``
    accumulation on the host, each op rounded to fp32 as torch does for
    python-scalar (op) fp32-tensor.  The drift assert of indi.py:64 is dropped (R3)."""
    delta = t_float_start / num_timesteps
    cur_t = t_float_start
    ts, c1, c2, sg = [], [], [], []
    for _ in range(num_timesteps):
        # the very torch expressions of indi.py:65-68 on a (1,) fp32 tensor, so scalar
        # promotion and rounding (e.g. python_float / tensor == reciprocal * scalar) match
        t_cur = torch.Tensor([cur_t])
        r = delta / t_cur
        ts.append(t_cur.item())
        c1.append(r.item())
        c2.append((1 - r).item())
        sg.append((e * (t_cur - delta)).item())
        cur_t -= delta
    return StepTableHost(ts, c1=c1, c2=c2, sigma=sg, predict_eps=False, clip=False)


def indi_step_table_per_sample(num_timesteps, t_starts, e=0.01):
    """One InDI schedule per batch element (``t_starts``: B floats): the columns of ``indi_step_table`` for every
    sample, stacked to (n_steps, B).  This is what the reference's refinement driver gets by looping over the batch
    one sample at a time (core/psnr_based_t_refinement.py:22-36)."""
    tabs = [indi_step_table(num_timesteps, float(t), e) for t in t_starts]
    st = lambda k: np.stack([getattr(t, k) for t in tabs], axis=1)
    return StepTableHost(st("tcond"), c1=st("c1"), c2=st("c2"), sigma=st("sigma"), predict_eps=False, clip=False,
                         per_sample=len(tabs))


def indi_snapshot_steps(num_timesteps):
    """indi.py:77,89-90: idx % (1|(n//20)) == 0 or idx == n-1."""
    inter = 1 | (num_timesteps // 20)
    return [i for i in range(num_timesteps) if i % inter == 0 or i == num_timesteps - 1]


def gaussian_snapshot_steps(T):
    """sr3 diffusion.py:180,198: i % (1|(T//10)) == 0 on the descending index i;
    returned as 0-based step ordinals (step s handles i = T-1-s)."""
    inter = 1 | (T // 10)
    return [s for s in range(T) if (T - 1 - s) % inter == 0]


def randn(shape, seed, subsequence=0, device="cuda"):
    """N(0,1) from the engine's Philox stream (perf-mode initial states)."""
    _lib.require_gpu()
    out = torch.empty(shape, dtype=torch.float32, device=device)
    check(lib.dsx_randn(_dptr(out), out.numel(), C.c_uint64(int(seed)), C.c_uint64(int(subsequence)),
                        _stream_ptr()))
    return out
