"""HBM bytes per conv launch from the two rocprofv3 PMC passes of tools/final_measure.sh.

    python tools/traffic.py gpurun_out/final_pmc_fetch gpurun_out/final_pmc_write [out.json]

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 summed over the conv kernels' dispatches, divided by their
count.  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (16-byte-per-lane streaming reads
are tallied at half their size); the two counters need separate passes (TCC counter budget).
"""
import csv, glob, json, os, sys

def total(d, counter):
    f = max(glob.glob(f"{d}/*/*_counter_collection.csv"), key=os.path.getmtime)   # newest run
    s, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter or "k_conv" not in r["Kernel_Name"] or "naive" in r["Kernel_Name"]:
            continue
        s += float(r["Counter_Value"]); n += 1
    return s, n

fetch, nf = total(sys.argv[1], "FETCH_SIZE")
write, nw = total(sys.argv[2], "WRITE_SIZE")
assert nf == nw and nf > 0, (nf, nw)
per = (2.0 * fetch + write) * 1024.0 / nf
out = {"bf16": per, "launches": nf,
       "_note": "HBM bytes per conv launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 / launches; FETCH_SIZE doubled per "
                "MI355X_MICROARCH.md (gfx950 tallies wide coalesced reads at half size); separate --pmc passes; "
                "eager steps of bench.py (tools/final_measure.sh)"}
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
