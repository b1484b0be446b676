#!/bin/bash
# Run on the GPU box through gpurun:  gpurun --timeout 900 -- 'bash tools/profile_gpu.sh [workload]'
# Three separate passes (kernel trace + stats; PMC FETCH_SIZE; PMC WRITE_SIZE) -- gpurun refuses
# PMC combined with sys/hip traces.  Results under gpurun_out/prof_*; summarise with
# tools/pmc_summary.py and copy what should be judged into profiles/.
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${1:-c3c5}
cd /tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $WL > $R/gpurun_out/prof_kt.json 2> $R/gpurun_out/prof_kt.err && \
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $WL > $R/gpurun_out/prof_fetch.json 2> $R/gpurun_out/prof_fetch.err && \
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $WL > $R/gpurun_out/prof_write.json 2> $R/gpurun_out/prof_write.err
echo profile_exit=$?
cd $R && python3 tools/pmc_summary.py gpurun_out/prof_fetch gpurun_out/prof_write > gpurun_out/pmc_summary.txt; cat gpurun_out/prof_kt/*/*_kernel_stats.csv | head -12
