"""The reverse-sampling loops behind the reference's sampler classes.

Each class keeps the reference's constructor / method signatures and return
shapes and drives ``UNetEngine.sample_loop`` (a captured
hipGraph per step).  Training entry points raise: the engine is inference-only.

Noise: by default the per-step noise is drawn on the device (Philox, seeded
from torch's generator); set ``noise_source`` to a ``randn(shape)`` callable to
inject host draws in the reference's draw order (parity mode, Q4).
"""
import torch
from torch import nn

from .. import engine
from .._lib import DsxError


class _SamplerBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.noise_source = None      # callable(shape) -> CPU/GPU tensor; None = device RNG
        self.use_graph = True
        self.last_full_batch = None   # final state of the whole batch (Q1 keeps only one element)

    def _draw(self, shape, device):
        if self.noise_source is not None:
            return self.noise_source(tuple(shape)).to(device=device, dtype=torch.float32)
        return torch.randn(tuple(shape), device=device)

    def _seed(self):
        if self.noise_source is not None:
            return 0   # injected noise: leave torch's generator untouched (the draws must stay in order)
        return int(torch.randint(0, 2 ** 31 - 1, (1,)).item())

    def forward(self, x, *args, **kwargs):
        raise NotImplementedError("training (p_losses) is out of scope of the MI355X sampling engine")

    def get_current_log(self):
        return {}

    @property
    def prediction_channels(self):
        """Channels of ``last_full_batch`` (what tiled prediction stitches)."""
        return int(getattr(self, "out_channel", None) or self.channels)


class GaussianSampler(_SamplerBase):
    """SR3 / DDPM ancestral sampling (sr3 diffusion.py:141-213, ddpm diffusion.py:194-247)."""

    kind = "sr3"

    def __init__(self, denoise_fn, image_size, channels=3, loss_type="l1", conditional=True,
                 schedule_opt=None, **unused):
        # `unused` swallows out_channel / lr_reduction / val_schedule_opt that define_G always
        # passes (networks.py:159-170, rot R1)
        super().__init__()
        self.channels = channels
        self.image_size = image_size
        self.denoise_fn = denoise_fn
        self.loss_type = loss_type
        self.conditional = conditional
        self.num_timesteps = None
        self._table = {}

    def set_loss(self, device):
        self._device = device

    def set_new_noise_schedule(self, schedule_opt, device):
        bufs, gamma = engine.gaussian_buffers(schedule_opt)
        self.sqrt_alphas_cumprod_prev = gamma                    # float64 numpy, as in the reference
        self.num_timesteps = int(bufs["betas"].shape[0])
        for k, v in bufs.items():                                # same buffer names as diffusion.py:104-139
            if hasattr(self, k):
                delattr(self, k)
            self.register_buffer(k, v.to(device))
        self._bufs_cpu, self._table = bufs, {}

    def _step_table(self, clip):
        if clip not in self._table:
            self._table[clip] = engine.gaussian_step_table(self._bufs_cpu, self.sqrt_alphas_cumprod_prev,
                                                           self.kind, clip)
        return self._table[clip]

    @torch.no_grad()
    def p_sample_loop(self, x_in, clip_denoised=True, continous=False):
        if self.num_timesteps is None:
            raise DsxError("set_new_noise_schedule() first")
        dev = self.betas.device
        T = self.num_timesteps
        if not self.conditional:
            shape, cond = tuple(x_in), None
        else:
            cond = x_in.to(dev).float()
            shape = (cond.shape[0], self.channels) + tuple(cond.shape[2:])
        img = self._draw(shape, dev)
        # the loop updates `img` in place: keep a copy of the initial noise for ret_img[0] (sr3 diffusion.py:183-185)
        first = (img.clone() if continous else None) if cond is None else cond.repeat((1, self.channels // cond.shape[1], 1, 1))
        noise = None
        if self.noise_source is not None:  # reference draw order: one per step, none at t == 0 (sr3)
            n_draw = T - 1 if self.kind == "sr3" else T
            noise = torch.zeros((T,) + shape, device=dev)
            for s in range(n_draw):
                noise[s] = self._draw(shape, dev)
        snaps = engine.gaussian_snapshot_steps(T) if continous else []
        x, sn = self.denoise_fn.engine().sample_loop(self._step_table(bool(clip_denoised)), img, cond=cond,
                                                     noise=noise, seed=self._seed(), snapshot_steps=snaps,
                                                     use_graph=self.use_graph)
        self.last_full_batch = x
        if self.kind == "ddpm" and not self.conditional:
            return x                                                 # ddpm diffusion.py:222
        if continous:
            return torch.cat([first] + [s for s in sn], dim=0)
        return x[-1]                                                 # diffusion.py:200-203: ret_img[-1]

    @torch.no_grad()
    def sample(self, batch_size=1, continous=False):
        return self.p_sample_loop((batch_size, self.channels, self.image_size, self.image_size),
                                  continous=continous)

    @torch.no_grad()
    def super_resolution(self, x_in, clip_denoised=True, continous=False):
        return self.p_sample_loop(x_in, clip_denoised=clip_denoised, continous=continous)

    predict = super_resolution                                       # ddpm diffusion.py:245-247


class GaussianSamplerDdpm(GaussianSampler):
    kind = "ddpm"

    def __init__(self, denoise_fn, image_size, channels=3, loss_type="l1", lr_reduction=None,
                 conditional=True, schedule_opt=None, **unused):
        super().__init__(denoise_fn, image_size, channels, loss_type, conditional, schedule_opt)
        self.lr_reduction = lr_reduction or "sum"


class InDISampler(_SamplerBase):
    """InDI.inference (ddpm_modules/indi.py:62-110)."""

    def __init__(self, denoise_fn, image_size, channels=3, loss_type="l1", out_channel=2, lr_reduction=None,
                 conditional=True, schedule_opt=None, val_schedule_opt=None, e=0.01, **unused):
        super().__init__()
        self.denoise_fn = denoise_fn
        self.image_size, self.channels, self.loss_type = image_size, channels, loss_type
        self.out_channel = out_channel
        self.conditional = conditional
        self.lr_reduction = lr_reduction or "sum"
        self.e = e
        self.num_timesteps = None
        self.val_num_timesteps = val_schedule_opt["n_timestep"] if val_schedule_opt else None

    def set_loss(self, device):
        self._device = device

    def set_new_noise_schedule(self, schedule_opt, device):
        self.num_timesteps = schedule_opt["n_timestep"]              # indi.py:46-47

    def _start(self, x_in, t_float_start):
        dev = x_in.device
        x_in = torch.cat([x_in.float()] * self.out_channel, dim=1)   # indi.py:80
        scale = (self.e * torch.Tensor([t_float_start])).to(dev)     # get_t_times_e, indi.py:106-110
        return x_in + self._draw(x_in.shape, dev) * scale            # indi.py:82

    @staticmethod
    def _per_sample_t(t_float_start, batch):
        """None for the reference's scalar start time; else the B per-sample start times as python floats."""
        if torch.is_tensor(t_float_start):
            t = t_float_start.detach().reshape(-1).to("cpu", torch.float64).tolist()
        elif isinstance(t_float_start, (list, tuple)) or (hasattr(t_float_start, "shape") and getattr(t_float_start, "ndim", 0) > 0):
            t = [float(v) for v in t_float_start]
        else:
            return None
        if len(t) == 1:
            t = t * batch
        if len(t) != batch:
            raise DsxError(f"t_float_start has {len(t)} entries for a batch of {batch}")
        return t

    @torch.no_grad()
    def _inference_per_sample(self, x_in, t_list, continuous, num_timesteps, stream):
        """One start time per batch element, one batched loop (per-sample step tables).  Equivalent to calling
        ``inference`` on every sample alone, as core/psnr_based_t_refinement.py:26-34 does; with a ``noise_source``
        the draws are taken sample by sample in that order (start draw, then one draw per step)."""
        dev = x_in.device
        B = x_in.shape[0]
        xr = torch.cat([x_in.float()] * self.out_channel, dim=1)
        noise = None
        if self.noise_source is not None:
            starts, steps = [], []
            for b in range(B):
                starts.append(self._draw((1,) + tuple(xr.shape[1:]), dev))
                steps.append(torch.cat([self._draw((1,) + tuple(xr.shape[1:]), dev) for _ in range(num_timesteps)]))
            d0 = torch.cat(starts)
            noise = torch.stack(steps, dim=1).contiguous()            # (n, B, C, H, W)
        else:
            d0 = self._draw(xr.shape, dev)
        scale = torch.stack([(self.e * torch.Tensor([t])) for t in t_list]).to(dev).view(B, 1, 1, 1)
        x_t = xr + d0 * scale
        table = engine.indi_step_table_per_sample(num_timesteps, t_list, self.e)
        snaps = engine.indi_snapshot_steps(num_timesteps) if continuous else []
        first = x_t.clone() if continuous else None
        x, sn = self.denoise_fn.engine().sample_loop(table, x_t, noise=noise, seed=self._seed(), snapshot_steps=snaps,
                                                     use_graph=self.use_graph, stream=stream)
        self.last_full_batch = x
        if continuous:
            if stream is not None:
                stream.synchronize()
            return torch.cat([first] + [s for s in sn], dim=0)
        return x[-1:]

    def _noise(self, shape, n, dev):
        if self.noise_source is None:
            return None
        return torch.stack([self._draw(shape, dev) for _ in range(n)])  # drawn on every step (Q4)

    @torch.no_grad()
    def inference(self, x_in, continuous=False, num_timesteps=None, t_float_start=1.0, eps=1e-8,
                  stream=None):
        if num_timesteps is None:
            num_timesteps = self.num_timesteps
        assert self.conditional is False
        if not x_in.is_cuda:
            raise DsxError("inference runs on the MI355X only; pass a CUDA tensor (no CPU fallback)")
        t_list = self._per_sample_t(t_float_start, x_in.shape[0])
        if t_list is not None:
            return self._inference_per_sample(x_in, t_list, continuous, num_timesteps, stream)
        x_t = self._start(x_in, t_float_start)
        noise = self._noise(x_t.shape, num_timesteps, x_t.device)
        table = engine.indi_step_table(num_timesteps, t_float_start, self.e)   # no drift assert (R3)
        snaps = engine.indi_snapshot_steps(num_timesteps) if continuous else []
        first = x_t.clone() if continuous else None
        x, sn = self.denoise_fn.engine().sample_loop(table, x_t, noise=noise, seed=self._seed(),
                                                     snapshot_steps=snaps, use_graph=self.use_graph,
                                                     stream=stream)
        self.last_full_batch = x
        if continuous:
            if stream is not None:
                stream.synchronize()
            return torch.cat([first] + [s for s in sn], dim=0)
        return x[-1:]                                                # indi.py:92-95: ret_img[-1:]


class JointIndiSampler(_SamplerBase):
    """JointIndi (ddpm_modules/joint_indi.py:40-149): indi1 at t0, indi2 at 1-t0; the two
    independent loops run concurrently on two HIP streams instead of back to back."""

    def __init__(self, denoise_fn, image_size, channels=3, loss_type="l1", out_channel=2, lr_reduction=None,
                 denoise_fn_ch1=None, denoise_fn_ch2=None, conditional=True, schedule_opt=None,
                 val_schedule_opt=None, w_input_loss=0.0, e=0.01, allow_full_translation=False):
        super().__init__()
        assert denoise_fn_ch1 is not None and denoise_fn_ch2 is not None and denoise_fn is None
        kw = dict(channels=channels, loss_type=loss_type, out_channel=out_channel, lr_reduction=lr_reduction,
                  conditional=conditional, schedule_opt=schedule_opt, val_schedule_opt=val_schedule_opt, e=e)
        self.indi1 = InDISampler(denoise_fn_ch1, image_size, **kw)
        self.indi2 = InDISampler(denoise_fn_ch2, image_size, **kw)
        self.val_num_timesteps = self.indi1.val_num_timesteps
        self.alpha_param = nn.Parameter(torch.tensor(0.0))           # kept for *_gen.pth compatibility
        self.offset_param = nn.Parameter(torch.tensor(0.0))
        self.scale_param = nn.Parameter(torch.tensor(1.0))
        self.w_input_loss = w_input_loss
        self._streams = None
        self.concurrent = True        # the two loops on two HIP streams; False: back to back on the current stream

    @property
    def prediction_channels(self):
        return self.indi1.prediction_channels + self.indi2.prediction_channels   # joint_indi.py:135 (channel cat)

    def set_loss(self, device):
        self.indi1.set_loss(device)
        self.indi2.set_loss(device)

    def set_new_noise_schedule(self, schedule_opt, device):
        self.indi1.set_new_noise_schedule(schedule_opt, device)
        self.indi2.set_new_noise_schedule(schedule_opt, device)

    @torch.no_grad()
    def inference(self, x_in, continuous=False, num_timesteps=None, t_float_start=0.5, eps=1e-8):
        for s in (self.indi1, self.indi2):
            s.noise_source, s.use_graph = self.noise_source, self.use_graph   # indi1's draws first (Q4)
        if self.noise_source is not None or not self.concurrent:
            # host draws are order-dependent: finish indi1's before indi2's start
            ch1 = self.indi1.inference(x_in, continuous, num_timesteps, t_float_start, eps)
            ch2 = self.indi2.inference(x_in, continuous, num_timesteps, 1 - t_float_start, eps)
        else:
            if self._streams is None:
                self._streams = (torch.cuda.Stream(), torch.cuda.Stream())
            s1, s2 = self._streams
            ch1 = self.indi1.inference(x_in, continuous, num_timesteps, t_float_start, eps, stream=s1)
            ch2 = self.indi2.inference(x_in, continuous, num_timesteps, 1 - t_float_start, eps, stream=s2)
            torch.cuda.current_stream().wait_stream(s1)
            torch.cuda.current_stream().wait_stream(s2)
        self.last_full_batch = torch.cat([self.indi1.last_full_batch, self.indi2.last_full_batch], dim=1)
        return torch.cat([ch1, ch2], dim=1)
