"""``DDPM`` — the model wrapper of model/model.py (feed_data / test /
get_current_visuals / load_network), running on the HIP engine.

``test()`` is a superset of the reference's (rot R2): it routes to
``inference`` (InDI family), ``super_resolution`` (sr3) or ``predict`` (ddpm)."""
import logging
from collections import OrderedDict

import torch

from . import networks
from .base_model import BaseModel

logger = logging.getLogger("base")


class DDPM(BaseModel):
    def __init__(self, opt):
        super().__init__(opt)
        self.netG = self.set_device(networks.define_G(opt))
        self.schedule_phase = None
        self.set_loss()
        self.set_new_noise_schedule(opt["model"]["beta_schedule"]["train"], schedule_phase="train")
        self.log_dict = OrderedDict()
        self.load_network()

    def feed_data(self, data):
        self.data = self.set_device(data)

    def optimize_parameters(self):
        raise NotImplementedError("training is out of scope of the MI355X sampling engine")

    def test(self, continuous=False, clip_denoised=True):
        self.netG.eval()
        x = self.data["input"]
        with torch.no_grad():
            if hasattr(self.netG, "inference"):
                self.prediction = self.netG.inference(x, continuous=continuous)   # model.py:63-76
            elif hasattr(self.netG, "super_resolution"):
                self.prediction = self.netG.super_resolution(x, clip_denoised=clip_denoised, continous=continuous)
            else:
                self.prediction = self.netG.predict(x, clip_denoised=clip_denoised, continous=continuous)

    def sample(self, batch_size=1, continous=False):
        self.netG.eval()
        with torch.no_grad():
            self.prediction = self.netG.sample(batch_size, continous)

    def set_loss(self):
        self.netG.set_loss(self.device)

    def set_new_noise_schedule(self, schedule_opt, schedule_phase="train"):
        if self.schedule_phase is None or self.schedule_phase != schedule_phase:   # model.py:93-100
            self.schedule_phase = schedule_phase
            self.netG.set_new_noise_schedule(schedule_opt, self.device)

    def get_current_log(self):
        return self.log_dict

    def get_current_visuals(self, need_LR=True, sample=False):
        out = OrderedDict()
        if sample:
            out["SAM"] = self.prediction.detach().float().cpu()
        else:
            out["prediction"] = self.prediction.detach().float().cpu()
            out["input"] = self.data["input"].detach().float().cpu()
            if "target" in self.data and self.data["target"] is not None:
                out["target"] = self.data["target"].detach().float().cpu()
        return out

    def save_network(self, epoch, iter_step):
        import os
        gen_path = os.path.join(self.opt["path"]["checkpoint"], "I{}_E{}_gen.pth".format(iter_step, epoch))
        torch.save({k: v.cpu() for k, v in self.netG.state_dict().items()}, gen_path)   # model.py:131-142
        logger.info("Saved model in [{:s}] ...".format(gen_path))

    def load_network(self):
        load_path = self.opt["path"]["resume_state"] if self.opt["path"] else None
        if load_path is not None:
            logger.info("Loading pretrained model for G [{:s}] ...".format(load_path))
            gen_path = "{}_gen.pth".format(load_path)
            sd = torch.load(gen_path, map_location="cpu", weights_only=True)
            self.netG.load_state_dict(sd, strict=(not self.opt["model"]["finetune_norm"]))
            # packed-weight cache next to the checkpoint, keyed by the checkpoint's hash: the MFMA-fragment repack of
            # every conv is done once per (checkpoint, dtype), later loads upload the cached image
            import hashlib
            h = hashlib.sha256()
            with open(gen_path, "rb") as f:
                for blk in iter(lambda: f.read(1 << 24), b""):
                    h.update(blk)
            key = h.hexdigest()
            k = 0
            for m in self.netG.modules():
                if hasattr(m, "attach_pack_cache"):
                    m.attach_pack_cache("{}_gen.unet{}".format(load_path, k), key)
                    k += 1
