"""Per-kernel vector-ALU / vector-memory / LDS instruction activity of the conv launches (one rocprofv3 --pmc pass):

    python tools/pmc_valu.py <pmc dir> <ops.json>

VALU busy = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x kernel cycles)   (rocprof's VALUBusy; the counter ticks in quad-cycles),
kernel cycles = GRBM_GUI_ACTIVE / 8 (summed over the 8 XCDs).  VMEM and LDS likewise.  MFMA issue is part of VALU activity."""
import collections, csv, glob, json, os, sys

d, opsf = sys.argv[1], sys.argv[2]
f = max(glob.glob(f"{d}/*/*_counter_collection.csv"), key=os.path.getmtime)
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    e = rows.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "dur": float(r["End_Timestamp"]) - float(r["Start_Timestamp"])})
    e[r["Counter_Name"]] = float(r["Counter_Value"])
conv = [e for e in rows.values() if "k_conv" in e["name"] and "naive" not in e["name"]]
ops = [o for o in json.load(open(opsf)) if o["kind"] == "conv_mfma"]
assert len(conv) % len(ops) == 0, (len(conv), len(ops))
agg = collections.OrderedDict()
for i, e in enumerate(conv):
    desc = ops[i % len(ops)]["desc"].split(" tile")[0].split(" img")[0].split(" first")[0]
    g = agg.setdefault(desc, collections.Counter())
    g["n"] += 1; g["dur"] += e["dur"]; g["cyc"] += e.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY"):
        g[k] += e.get(k, 0.0)
print(f"{'layer':34s} {'n':>3s} {'us':>7s} {'VALU%':>7s} {'VMEM%':>7s} {'LDS%':>7s} {'ANY%':>7s}")
for k, g in sorted(agg.items(), key=lambda kv: -kv[1]["dur"]):
    den = 1024.0 * g["cyc"] / 4.0
    print(f"{k:34s} {g['n']:3d} {g['dur'] / g['n'] / 1e3:7.1f} " + " ".join(
        f"{100.0 * g[c] / den:7.1f}" for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY")))
