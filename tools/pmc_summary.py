"""Per-kernel summary of the three rocprofv3 --pmc passes of tools/measure_r03.sh (eager bench.py steps; the same
program gives the same dispatch order in every pass, so dispatches are joined by ordinal among the conv kernels).

    python tools/pmc_summary.py <pmc_mfma dir> <pmc_fetch dir> <pmc_write dir> <ops.json> <out counters.json>

MFMA busy  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3
             reports the sum over the 8 XCDs: MI355X_MICROARCH.md, DVFS give-back)
HBM bytes  = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950
             (wide coalesced reads are tallied at half their size); separate passes (TCC counter budget)
"""
import collections, csv, glob, json, os, sys


def load(d):
    f = max(glob.glob(f"{d}/*/*_counter_collection.csv"), key=os.path.getmtime)
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        e = rows.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]),
                                                    "wg": int(r["Workgroup_Size"]), "lds": int(r["LDS_Block_Size"]),
                                                    "dur": float(r["End_Timestamp"]) - float(r["Start_Timestamp"])})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
    return list(rows.values())


def is_conv(e):
    return "k_conv" in e["name"] and "naive" not in e["name"]


mf, fe, wr = (load(sys.argv[i]) for i in (1, 2, 3))
ops = json.load(open(sys.argv[4])) if os.path.exists(sys.argv[4]) else None
cm, cf, cw = ([e for e in x if is_conv(e)] for x in (mf, fe, wr))
assert len(cm) == len(cf) == len(cw) and cm, (len(cm), len(cf), len(cw))
conv_ops = [o for o in ops if o["kind"] == "conv_mfma"] if ops else None
per_step = len(conv_ops) if conv_ops else None
agg = collections.OrderedDict()
tot = collections.Counter()
for i, (a, b, c) in enumerate(zip(cm, cf, cw)):
    assert a["name"] == b["name"] == c["name"] and a["grid"] == b["grid"]
    short = a["name"].split("(")[0].replace("void dsx::", "").replace("dsx::", "")
    desc = conv_ops[i % per_step]["desc"] if per_step and len(cm) % per_step == 0 else ""
    fl = conv_ops[i % per_step]["gflop"] if desc else 0.0
    ab = conv_ops[i % per_step]["mbytes"] if desc else 0.0
    k = (short[:58], a["grid"] // a["wg"], a["lds"], desc.split(" tile")[0].split(" img")[0].split(" first")[0].split(" ws")[0])
    g = agg.setdefault(k, collections.Counter())
    cyc = a.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    hbm = (2.0 * b.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0
    for key, v in (("n", 1), ("dur", a["dur"]), ("mfma", a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)), ("cyc", cyc), ("hbm", hbm),
                   ("gflop", fl), ("algo_mb", ab)):
        g[key] += v
        tot[key] += v
print(f"{'kernel':58s} {'WGs':>5s} {'LDS':>7s} {'n':>3s} {'us':>7s} {'MFMAbusy%':>9s} {'HBM GB/s':>9s} {'MB/launch':>9s} {'algoMB':>7s} {'TF/s':>6s}  layer")
for k, g in sorted(agg.items(), key=lambda kv: -kv[1]["dur"]):
    n = g["n"]; dur = g["dur"] / n
    busy = 100.0 * g["mfma"] / (1024.0 * g["cyc"]) if g["cyc"] else 0.0
    print(f"{k[0]:58s} {k[1]:5d} {k[2]:7d} {n:3d} {dur / 1e3:7.1f} {busy:9.1f} {g['hbm'] / g['dur']:9.0f} {g['hbm'] / n / 1e6:9.1f} "
          f"{g['algo_mb'] / n:7.1f} {g['gflop'] / g['dur'] * 1e3 if g['dur'] else 0:6.0f}  {k[3]}")
n = tot["n"]
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench as _bench
out = {"kernel_source_sha": _bench.kernel_source_sha(), "conv_launches_profiled": n, "hbm_bytes_per_conv_launch": tot["hbm"] / n,
       "mfma_busy_frac_conv": tot["mfma"] / (1024.0 * tot["cyc"]) if tot["cyc"] else None,
       "hbm_gbps_conv": tot["hbm"] / tot["dur"], "avg_conv_launch_us_profiled": tot["dur"] / n / 1e3,
       "algorithmic_bytes_per_conv_launch": tot["algo_mb"] * 1e6 / n if tot["algo_mb"] else None,
       "_note": "rocprofv3 --pmc passes over eager bench.py steps (tools/measure_r03.sh): SQ_VALU_MFMA_BUSY_CYCLES / (1024 x "
                "GRBM_GUI_ACTIVE / 8); HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 correction of "
                "MI355X_MICROARCH.md); separate passes"}
print(json.dumps(out, indent=1))
json.dump(out, open(sys.argv[5], "w"), indent=1)
