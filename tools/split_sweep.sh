#!/bin/bash
# bench.py with the batch of 16 run as 1 / 2 / 4 independent sub-batch loops on separate HIP streams
mkdir -p gpurun_out
for n in 1 2 4; do
  timeout -k 10 200 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-roofline --split $n > gpurun_out/split_$n.json 2> gpurun_out/split_$n.err || { echo "split $n failed"; tail -5 gpurun_out/split_$n.err; exit 1; }
  python -c "import json; d=json.load(open('gpurun_out/split_$n.json')); print('split', $n, 'ms/step %.3f  img/s %.3f' % (d['ms_per_step'], d['value']))"
done
