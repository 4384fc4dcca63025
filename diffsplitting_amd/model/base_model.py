"""model/base_model.py of the reference (device placement helpers)."""
import torch


class BaseModel:
    def __init__(self, opt):
        self.opt = opt
        # base_model.py:9-10 (Q10): cuda iff gpu_ids is given.  The index is the current device, which the entry
        # point set from gpu_ids[LOCAL_RANK] (split.py); without an entry point, the first id of the list.
        if opt["gpu_ids"] is not None:
            idx = 0
            if torch.cuda.is_available():
                idx = torch.cuda.current_device()
                ids = [int(i) for i in opt["gpu_ids"]] if opt["gpu_ids"] else []
                if ids and idx not in ids and ids[0] < torch.cuda.device_count():
                    idx = ids[0]
                    torch.cuda.set_device(idx)
            self.device = torch.device("cuda", idx)
        else:
            self.device = torch.device("cpu")
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
