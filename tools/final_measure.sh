#!/bin/bash
# Round-end measurement on the GPU box: full GPU tests, default bench, rocprofv3 kernel stats, HBM traffic (PMC).
set -o pipefail
R=$PWD; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/final_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/final_tests.log
tail -3 gpurun_out/final_tests.log
timeout -k 10 400 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"
tail -12 gpurun_out/final_bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_prof -- python $R/bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/final_prof.log 2>&1; echo "rocprof rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/final_pmc_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > $R/gpurun_out/final_pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/final_pmc_write -- python $R/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > $R/gpurun_out/final_pmc_write.log 2>&1; echo "pmc write rc=$?"
cd $R; ls gpurun_out/final_prof/*/ gpurun_out/final_pmc_fetch/*/ | head
