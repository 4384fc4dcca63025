"""Oracle: the reverse-sampling loops (SR3, DDPM, InDI, JointIndi).

Test infrastructure only (see ``oracle/__init__.py``).  Noise is drawn through
a ``randn(shape)`` callable in exactly the reference's draw order, so that with
``torch.manual_seed(s)`` + ``torch.randn`` the outputs equal the reference's
bit for bit, and with a recorded list of draws the same tensors can be injected
into the HIP engine.
"""
import numpy as np
import torch

from .unet import unet_forward


# --------------------------------------------------------------------------
# schedules
# --------------------------------------------------------------------------
def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """sr3_modules/diffusion.py:19-49 (float64)."""
    if schedule == "quad":
        return np.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=np.float64) ** 2
    if schedule == "linear":
        return np.linspace(linear_start, linear_end, n_timestep, dtype=np.float64)
    if schedule in ("warmup10", "warmup50"):
        frac = 0.1 if schedule == "warmup10" else 0.5
        betas = linear_end * np.ones(n_timestep, dtype=np.float64)
        w = int(n_timestep * frac)
        betas[:w] = np.linspace(linear_start, linear_end, w, dtype=np.float64)
        return betas
    if schedule == "const":
        return linear_end * np.ones(n_timestep, dtype=np.float64)
    if schedule == "jsd":
        return 1.0 / np.linspace(n_timestep, 1, n_timestep, dtype=np.float64)
    if schedule == "cosine":
        import math
        ts = torch.arange(n_timestep + 1, dtype=torch.float64) / n_timestep + cosine_s
        alphas = torch.cos(ts / (1 + cosine_s) * math.pi / 2).pow(2)
        alphas = alphas / alphas[0]
        betas = 1 - alphas[1:] / alphas[:-1]
        return betas.clamp(max=0.999).numpy()
    raise NotImplementedError(schedule)


def gaussian_schedule(schedule_opt):
    """sr3_modules/diffusion.py:92-139: the fp32 buffers + float64 gamma table."""
    betas = make_beta_schedule(schedule_opt["schedule"], schedule_opt["n_timestep"],
                               schedule_opt["linear_start"], schedule_opt["linear_end"])
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    f32 = lambda a: torch.tensor(a, dtype=torch.float32)
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    return {
        "num_timesteps": int(betas.shape[0]),
        "sqrt_alphas_cumprod_prev": np.sqrt(np.append(1.0, ac)),  # float64, T+1
        "betas": f32(betas),
        "alphas_cumprod": f32(ac),
        "alphas_cumprod_prev": f32(ac_prev),
        "sqrt_alphas_cumprod": f32(np.sqrt(ac)),
        "sqrt_one_minus_alphas_cumprod": f32(np.sqrt(1.0 - ac)),
        "log_one_minus_alphas_cumprod": f32(np.log(1.0 - ac)),
        "sqrt_recip_alphas_cumprod": f32(np.sqrt(1.0 / ac)),
        "sqrt_recipm1_alphas_cumprod": f32(np.sqrt(1.0 / ac - 1)),
        "posterior_variance": f32(post_var),
        "posterior_log_variance_clipped": f32(np.log(np.maximum(post_var, 1e-20))),
        "posterior_mean_coef1": f32(betas * np.sqrt(ac_prev) / (1.0 - ac)),
        "posterior_mean_coef2": f32((1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac)),
    }


# --------------------------------------------------------------------------
# SR3  (sr3_modules/diffusion.py:141-213)
# --------------------------------------------------------------------------
def sr3_p_sample(sd, cfg, sch, img, i, cond, randn, clip_denoised=True, prefix="denoise_fn."):
    b = img.shape[0]
    gamma = torch.FloatTensor([sch["sqrt_alphas_cumprod_prev"][i + 1]]).repeat(b, 1)  # :153-154
    x_in = torch.cat([cond, img], dim=1) if cond is not None else img
    eps = unet_forward(sd, cfg, "sr3", x_in, gamma, prefix)
    x0 = sch["sqrt_recip_alphas_cumprod"][i] * img - sch["sqrt_recipm1_alphas_cumprod"][i] * eps
    if clip_denoised:
        x0.clamp_(-1.0, 1.0)
    mean = sch["posterior_mean_coef1"][i] * x0 + sch["posterior_mean_coef2"][i] * img
    logvar = sch["posterior_log_variance_clipped"][i]
    noise = randn(img.shape) if i > 0 else torch.zeros_like(img)  # :174
    return mean + noise * (0.5 * logvar).exp()


def sr3_p_sample_loop(sd, cfg, sch, x_in, randn=torch.randn, clip_denoised=True, continous=False,
                      channels=3, conditional=True, return_full=False, prefix="denoise_fn."):
    """:177-203.  ``return_full`` additionally returns the final full batch."""
    T = sch["num_timesteps"]
    sample_inter = 1 | (T // 10)
    if not conditional:
        shape = tuple(x_in)
        img = randn(shape)
        ret = img
        cond = None
    else:
        cond = x_in
        shape = list(cond.shape)
        shape[1] = channels
        img = randn(tuple(shape))
        ret = cond.repeat((1, channels // cond.shape[1], 1, 1))
    for i in reversed(range(T)):
        img = sr3_p_sample(sd, cfg, sch, img, i, cond, randn, clip_denoised, prefix)
        if i % sample_inter == 0:
            ret = torch.cat([ret, img], dim=0)
    out = ret if continous else ret[-1]
    return (out, img) if return_full else out


# --------------------------------------------------------------------------
# DDPM (ddpm_modules/diffusion.py:64-75,:194-237)
# --------------------------------------------------------------------------
def ddpm_p_sample_loop(sd, cfg, sch, x_in, randn=torch.randn, clip_denoised=True, continous=False,
                       channels=3, conditional=True, return_full=False, prefix="denoise_fn."):
    T = sch["num_timesteps"]
    sample_inter = 1 | (T // 10)
    if not conditional:
        shape = tuple(x_in)
        cond = None
        img = randn(shape)
        ret = img
    else:
        cond = x_in
        shape = list(cond.shape)
        shape[1] = channels
        img = randn(tuple(shape))
        ret = cond.repeat((1, channels // cond.shape[1], 1, 1))
    b = img.shape[0]
    for i in reversed(range(T)):
        t = torch.full((b,), i, dtype=torch.long)
        x_cat = torch.cat([cond, img], dim=1) if cond is not None else img
        eps = unet_forward(sd, cfg, "ddpm", x_cat, t, prefix)
        ex = lambda a: a.gather(-1, t).reshape(b, 1, 1, 1)
        x0 = ex(sch["sqrt_recip_alphas_cumprod"]) * img - ex(sch["sqrt_recipm1_alphas_cumprod"]) * eps
        if clip_denoised:
            x0.clamp_(-1.0, 1.0)
        mean = ex(sch["posterior_mean_coef1"]) * x0 + ex(sch["posterior_mean_coef2"]) * img
        logvar = ex(sch["posterior_log_variance_clipped"])
        noise = randn(img.shape)  # drawn even at t == 0, then masked (:199-203)
        mask = (1 - (t == 0).float()).reshape(b, 1, 1, 1)
        img = mean + mask * (0.5 * logvar).exp() * noise
        if i % sample_inter == 0:
            ret = torch.cat([ret, img], dim=0)
    if not conditional:
        out = img  # :222 returns img for the unconditional branch
    else:
        out = ret if continous else ret[-1]
    return (out, img) if return_full else out


# --------------------------------------------------------------------------
# InDI (ddpm_modules/indi.py:62-110)
# --------------------------------------------------------------------------
def indi_inference(sd, cfg, x_in, num_timesteps, out_channel, randn=torch.randn, continuous=False,
                   t_float_start=1.0, e=0.01, return_full=False, prefix="denoise_fn.",
                   check_assert=False):
    """indi.py:71-95.  ``check_assert`` re-enables the drift assert of :64 (R3)."""
    sample_inter = 1 | (num_timesteps // 20)
    x_in = torch.cat([x_in] * out_channel, dim=1)
    x_t = x_in + randn(x_in.shape) * (e * torch.Tensor([t_float_start]))
    delta = t_float_start / num_timesteps
    cur_t = t_float_start
    ret = x_t
    for idx in range(num_timesteps):
        if check_assert:
            assert delta <= cur_t
        t_cur = torch.Tensor([cur_t])
        x0 = unet_forward(sd, cfg, "ddpm", x_t, t_cur, prefix)
        noise = randn(x_t.shape) * (e * (t_cur - delta))
        x_t = delta / t_cur * x0 + (1 - delta / t_cur) * x_t + noise
        cur_t -= delta
        if idx % sample_inter == 0 or idx == num_timesteps - 1:
            ret = torch.cat([ret, x_t], dim=0)
    out = ret if continuous else ret[-1:]
    return (out, x_t) if return_full else out


def joint_indi_inference(sd, cfg, x_in, num_timesteps, out_channel, randn=torch.randn,
                         continuous=False, t_float_start=0.5, e=0.01, return_full=False):
    """joint_indi.py:131-135: all of indi1's steps, then all of indi2's; cat on channels."""
    r1 = indi_inference(sd, cfg, x_in, num_timesteps, out_channel, randn, continuous,
                        t_float_start, e, return_full, prefix="indi1.denoise_fn.")
    r2 = indi_inference(sd, cfg, x_in, num_timesteps, out_channel, randn, continuous,
                        1 - t_float_start, e, return_full, prefix="indi2.denoise_fn.")
    if return_full:
        return torch.cat([r1[0], r2[0]], dim=1), torch.cat([r1[1], r2[1]], dim=1)
    return torch.cat([r1, r2], dim=1)


def indi_schedule(num_timesteps, t_float_start):
    """The per-step fp32 scalars InDI.inference effectively uses (indi.py:62-69,83-88).

    Returns float32 arrays ``t`` (UNet conditioning), ``c_x0 = δ/t``,
    ``c_xt = 1-δ/t``, ``c_noise = e·(t-δ)`` without the factor e, following
    torch's scalar/tensor promotion (python float -> fp32 before the op).
    """
    delta = t_float_start / num_timesteps
    cur_t = t_float_start
    ts, c0, c1, cn = [], [], [], []
    for _ in range(num_timesteps):
        t_cur = torch.Tensor([cur_t])
        ts.append(t_cur.item())
        r = delta / t_cur
        c0.append(r.item())
        c1.append((1 - r).item())
        cn.append((t_cur - delta).item())
        cur_t -= delta
    f = lambda a: np.asarray(a, dtype=np.float32)
    return f(ts), f(c0), f(c1), f(cn)
