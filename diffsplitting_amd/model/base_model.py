"""model/base_model.py of the reference (device placement helpers)."""
import torch


class BaseModel:
    def __init__(self, opt):
        self.opt = opt
        self.device = torch.device("cuda" if opt["gpu_ids"] is not None else "cpu")   # base_model.py:9-10 (Q10)
        self.begin_step = 0
        self.begin_epoch = 0

    def set_device(self, x):
        if isinstance(x, dict):
            for key, item in x.items():
                if item is not None:
                    x[key] = item.to(self.device)
        elif isinstance(x, list):
            x = [item.to(self.device) if item is not None else None for item in x]
        else:
            x = x.to(self.device)
        return x

    def get_network_description(self, network):
        return str(network), sum(p.numel() for p in network.parameters())
