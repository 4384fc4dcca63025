"""Raw in-kernel stamps of one conv launch: DSX_STAMP_OP=<conv ordinal>[,<block>] DSX_LIB_PATH=<-DDSX_STAMPS build> python tools/stamps_raw.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from diffsplitting_amd import engine
from diffsplitting_amd._lib import lib, check
torch.set_grad_enabled(False)
cfg = engine.make_cfg("sr3", **{k: bench.UNET[k] for k in ("in_channel", "out_channel", "inner_channel", "norm_groups", "channel_mults", "attn_res", "res_blocks", "image_size")})
eng = engine.UNetEngine(cfg, "sr3")
eng.load_state_dict(bench.random_init_state_dict(eng.param_names, eng.param_shapes)); eng.finalize(os.environ.get("DT", "bf16"))
ex = eng.executor(16, 128, 128, 3)
n = lib.dsx_exec_num_ops(ex); ms = (C.c_float * n)()
x = torch.randn(16, 6, 128, 128, device="cuda"); t = torch.rand(16, 1, device="cuda")
eng.forward(x, t, cond_channels=3)
check(lib.dsx_exec_profile(ex, 2, ms, None))
buf = (C.c_uint64 * 128)(); check(lib.dsx_exec_read_stamps(ex, buf))
st = np.array(buf[:], dtype=np.int64)
desc = C.create_string_buffer(256); kind = C.c_int(); fl = C.c_double(); by = C.c_double()
want = int(os.environ["DSX_STAMP_OP"].split(",")[0]); k = -1
for i in range(n):
    lib.dsx_exec_op_info(ex, i, desc, 256, C.byref(kind), C.byref(fl), C.byref(by))
    if kind.value == 0:
        k += 1
        if k == want: print("op:", desc.value.decode(), " measured %.1f us" % (ms[i] * 1e3))
nz = [(i, v) for i, v in enumerate(st) if v > 0]
t0 = min(v for _, v in nz) if nz else 0
prev = t0
for i, v in sorted(nz, key=lambda iv: iv[1]):
    print("stamp %3d: +%7d  (delta %6d)" % (i, v - t0, v - prev)); prev = v
