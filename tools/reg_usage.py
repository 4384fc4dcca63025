"""Registers / spills / scratch per kernel: hipcc -Rpass-analysis=kernel-resource-usage of one source file.
    python tools/reg_usage.py diffsplitting_amd/csrc/dsx_conv.hip [filter]"""
import re, subprocess, sys
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
                    "-c", src, "-o", "/dev/null"] + sys.argv[3:], capture_output=True, text=True)
cur = None; rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|SGPRs|Occupancy \[waves/SIMD\]): (\S+)", line)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name": cur = v; rows[cur] = {}
    elif cur: rows[cur][k] = v
for name, d in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in dem: continue
    print(f"V {str(d.get('VGPRs')):>4} A {str(d.get('AGPRs')):>3} S {str(d.get('SGPRs')):>4} sspill {str(d.get('SGPRs Spill')):>3} vspill {str(d.get('VGPRs Spill')):>3} scratch {str(d.get('ScratchSize [bytes/lane]')):>4} occ {d.get('Occupancy [waves/SIMD]')}  {dem[:110]}")
