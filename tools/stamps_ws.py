"""Stamps of one warp-specialised conv launch: DSX_STAMP_OP=<conv ordinal>[,<block>] python tools/stamps_ws.py"""
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
t0 = st[0]
print("compute waves (cycles): item: mfma-loop | barrier-wait | epilogue")
prev = st[0]
for v in range(20):
    a, b, e = st[1 + 3*v], st[2 + 3*v], st[3 + 3*v]
    if a == 0: break
    line = "  item %2d: mfma %6d  barrier %6d" % (v, a - prev, b - a)
    prev = b
    if e > 0:
        line += "  epilogue %6d" % (e - b); prev = e
    print(line)
print("loader waves: item: issue | wait-DMA | consume(math) | barrier-wait     (relative start %d)" % (st[64] - t0))
prev = st[64]
for v in range(15):
    a, b, c, d = st[65 + 4*v], st[66 + 4*v], st[67 + 4*v], st[68 + 4*v]
    if a == 0: break
    print("  item %2d: issue %5d  wait %6d  consume %6d  barrier %6d" % (v, a - prev, (b - a) if b else 0, (c - b) if b else (c - a), d - c))
    prev = d
# second tile's epilogue breakdown (stamps 61: after load_aff, 62: after the store loop)
eps = [v for v in range(20) if st[3 + 3*v] > 0 and st[1 + 3*v] > 0]
if len(eps) >= 2 and st[61] > 0:
    v = eps[1]; b, e = st[2 + 3*v], st[3 + 3*v]
    print("epilogue of tile 1: load_aff %d | loads+adds+stores %d | stats %d" % (st[61] - b, st[62] - st[61], e - st[62]))

if st[120] > 0:
    print("convert breakdown (one item): lds-wait %d | math %d | writes+drain %d" % (st[121]-st[120], st[122]-st[121], st[123]-st[122]))
