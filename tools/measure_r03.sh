#!/bin/bash
# Round-3 measurement on the GPU box (one gpurun call):
#   rocprofv3 --kernel-trace --stats of the same command, three separate --pmc passes
#   (MFMA busy / FETCH_SIZE / WRITE_SIZE: TCC counters do not fit one pass), per-kernel summary.
# rocprofv3 gets the program itself after `--` (python ...): no env / bash -c hop (the profiler initialises the GPU).
set -o pipefail
R=$PWD; mkdir -p gpurun_out
DSX_BENCH_OPS=gpurun_out/r03_ops.json timeout -k 10 300 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-fp32-parity > gpurun_out/r03_bench200.json 2> gpurun_out/r03_bench200.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_prof -- python $R/bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-roofline --no-fp32-parity > $R/gpurun_out/r03_prof.log 2>&1 || { echo "rocprof stats failed"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r03_pmc_mfma -- python $R/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-fp32-parity > $R/gpurun_out/r03_pmc_mfma.log 2>&1 || { echo "pmc mfma failed"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r03_pmc_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-fp32-parity > $R/gpurun_out/r03_pmc_fetch.log 2>&1 || { echo "pmc fetch failed"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r03_pmc_write -- python $R/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-fp32-parity > $R/gpurun_out/r03_pmc_write.log 2>&1 || { echo "pmc write failed"; exit 1; }
cd $R
python tools/pmc_summary.py gpurun_out/r03_pmc_mfma gpurun_out/r03_pmc_fetch gpurun_out/r03_pmc_write gpurun_out/r03_ops.json gpurun_out/r03_counters.json > gpurun_out/r03_pmc_summary.txt 2>&1; tail -40 gpurun_out/r03_pmc_summary.txt
find gpurun_out/r03_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r03_kernel_stats.csv
# the bench line last, with the counters of THIS build in place (bench.py quotes profiles/r03_counters.json only when its
# kernel-source hash is the running build's); the box's copy of profiles/ is scratch, the file travels back in gpurun_out/
cp gpurun_out/r03_counters.json profiles/r03_counters.json
timeout -k 10 500 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err || { echo "bench failed"; tail -5 gpurun_out/r03_bench.err; exit 1; }
tail -14 gpurun_out/r03_bench.err; cat gpurun_out/r03_bench.json
