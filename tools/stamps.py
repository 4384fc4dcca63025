"""Print in-kernel phase stamps of one conv launch: DSX_STAMP_OP=<conv ordinal>[,<block>] python tools/stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from diffsplitting_amd import engine
from diffsplitting_amd._lib import lib, check
torch.set_grad_enabled(False)
cfg = engine.make_cfg("sr3", **{k: bench.UNET[k] for k in ("in_channel", "out_channel", "inner_channel", "norm_groups", "channel_mults", "attn_res", "res_blocks", "image_size")})
eng = engine.UNetEngine(cfg, "sr3")
eng.load_state_dict(bench.random_init_state_dict(eng.param_names, eng.param_shapes)); eng.finalize("bf16")
ex = eng.executor(16, 128, 128, 3)
n = lib.dsx_exec_num_ops(ex); ms = (C.c_float * n)()
x = torch.randn(16, 6, 128, 128, device="cuda"); t = torch.rand(16, 1, device="cuda")
eng.forward(x, t, cond_channels=3); eng.forward(x, t, cond_channels=3)
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
print("prologue: load-issue %d  store %d  barrier %d" % (st[1]-st[0], st[2]-st[1], st[3]-st[2]))
g = 0
while 11 + 4*g < 120 and st[8 + 4*g] > 0:
    b = st[8+4*g:12+4*g]; prev = st[3] if g == 0 else st[11 + 4*(g-1)]
    print("group %2d: load-issue %5d  mfma-loop %6d  stage_store %5d  barrier %5d" % (g, b[0]-prev, b[1]-b[0], b[2]-b[1], b[3]-b[2]))
    g += 1
print("epilogue %d  stats %d  total %d ticks" % (st[5]-st[4], st[6]-st[5], st[6]-st[0]))
