#!/bin/bash
# A/B of library variants inside ONE gpurun call (boxes differ by a few percent): usage tools/ab_bench.sh name=ENV1=v,ENV2=v[,LIB=path] ...
# every variant: bench.py (2000 graph steps) with the per-launch table dumped to gpurun_out/ab_<name>_ops.json
set -o pipefail
mkdir -p gpurun_out
for spec in "$@"; do
  name=${spec%%=*}; envs=${spec#*=}
  (
    IFS=',' read -ra kv <<< "$envs"
    for e in "${kv[@]}"; do
      [ -z "$e" ] && continue
      if [[ $e == LIB=* ]]; then export DSX_LIB_PATH=$PWD/${e#LIB=}; else export "$e"; fi
    done
    DSX_BENCH_OPS=gpurun_out/ab_${name}_ops.json timeout -k 10 240 python bench.py --no-cpu-baseline --no-fp32-parity > gpurun_out/ab_${name}.json 2> gpurun_out/ab_${name}.err || { echo "$name FAILED"; tail -3 gpurun_out/ab_${name}.err; exit 1; }
  ) || exit 1
  python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
d = json.load(open(f"gpurun_out/ab_{n}.json"))
r = d["roofline"]
print(f"{n:12s} ms/step {d['ms_per_step']:.4f}  conv {r['conv_ms_per_step']:.4f}  fwd graph {r['forward_ms_graph']:.4f}  launches(conv) {r['launches']}  eager all {r['all_kernels_ms_per_step_eager_events']:.3f}")
PY
done
