"""Config system of the reference, consumed unchanged (core/logger.py):
JSON with ``//`` comments stripped line-wise (:20-27,:35-40) and ``NoneDict``
(missing key -> None, :107-122)."""
import json
import os
from collections import OrderedDict


def load_json(path):
    s = ""
    with open(path, "r") as f:
        for line in f:
            s += line.split("//")[0] + "\n"
    return json.loads(s, object_pairs_hook=OrderedDict)


class NoneDict(dict):
    def __missing__(self, key):
        return None


def dict_to_nonedict(opt):
    if isinstance(opt, dict):
        return NoneDict(**{k: dict_to_nonedict(v) for k, v in opt.items()})
    if isinstance(opt, list):
        return [dict_to_nonedict(v) for v in opt]
    return opt


def parse(args):
    """core/logger.py:29-104, the parts the sampling path reads: phase, gpu_ids,
    distributed (string-length quirk Q10 kept), debug overrides."""
    opt = load_json(args.config)
    opt["phase"] = getattr(args, "phase", "val")
    gpu_ids = getattr(args, "gpu_ids", None)
    if gpu_ids is not None:
        opt["gpu_ids"] = [int(i) for i in str(gpu_ids).split(",")]
        gpu_list = str(gpu_ids)
    else:
        gpu_list = ",".join(str(x) for x in (opt.get("gpu_ids") or []))
    opt["distributed"] = len(gpu_list) > 1
    opt["enable_wandb"] = bool(getattr(args, "enable_wandb", False))
    rootdir = getattr(args, "rootdir", None) or "."
    opt.setdefault("path", OrderedDict())
    for key in ("log", "results", "checkpoint"):
        opt["path"].setdefault(key, os.path.join(rootdir, key))
    if getattr(args, "debug", False) or "debug" in str(opt.get("name", "")):
        opt["model"]["beta_schedule"]["train"]["n_timestep"] = 10
        opt["model"]["beta_schedule"]["val"]["n_timestep"] = 10
    return dict_to_nonedict(opt)
