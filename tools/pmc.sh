#!/bin/bash
# usage: tools/pmc.sh <label> "<COUNTERS...>"   — one rocprofv3 --pmc pass over 3 eager steps of bench.py
R=$PWD; L=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc $* --output-format csv -d $R/gpurun_out/pmc_$L -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-roofline --no-cpu-baseline > $R/gpurun_out/pmc_$L.log 2>&1
ls $R/gpurun_out/pmc_$L/*/ | head -5
