#!/bin/bash
# usage: tools/tune.sh "<label>|ENV=.. ENV=.." ...   (each arg = one bench run)
mkdir -p gpurun_out
for spec in "$@"; do
  label="${spec%%|*}"; envs="${spec#*|}"
  env $envs DSX_BENCH_OPS=gpurun_out/ops_$label.json timeout -k 10 200 python bench.py --steps 50 --warmup 3 --no-cpu-baseline > gpurun_out/tune_$label.json 2> gpurun_out/tune_$label.err
  python - "$label" <<'PY'
import json, sys
l = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/tune_{l}.json"))
    r = d["roofline"]
    print(f"{l:24s} ms/step {d['ms_per_step']:.3f}  img/s {d['value']:.3f}  conv {r['conv_ms_per_step']:.3f} ms @ {r['achieved']:.0f} TF/s  all-kernels(eager) {r['all_kernels_ms_per_step_eager_events']:.3f}")
except Exception as e:
    print(l, "FAILED", e); print(open(f"gpurun_out/tune_{l}.err").read()[-800:])
PY
done
