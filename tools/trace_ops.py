import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from diffsplitting_amd import engine
from diffsplitting_amd._lib import lib, check
torch.set_grad_enabled(False)
cfg = engine.make_cfg("sr3", **{k: bench.UNET[k] for k in ("in_channel", "out_channel", "inner_channel", "norm_groups", "channel_mults", "attn_res", "res_blocks", "image_size")})
eng = engine.UNetEngine(cfg, "sr3")
eng.load_state_dict(bench.random_init_state_dict(eng.param_names, eng.param_shapes)); eng.finalize("bf16")
ex = eng.executor(16, 128, 128, 3)
n = lib.dsx_exec_num_ops(ex); ms = (C.c_float * n)()
check(lib.dsx_exec_profile(ex, 1, ms, None))
torch.cuda.synchronize()
print("all ops ran")
