#!/bin/bash
# Run on the GPU box through gpurun:  gpurun --timeout 900 -- 'bash tools/profile_gpu.sh [workload] [tag]'
# Three separate passes (kernel trace + stats; PMC FETCH_SIZE; PMC WRITE_SIZE) -- gpurun refuses
# PMC combined with sys/hip traces.  Results under gpurun_out/prof_<tag>/ : kernel_stats.csv,
# pmc_traffic.json (stamped with the source hash), the bench lines of the three runs.  Copy what
# should be judged into profiles/ (tracked).
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${1:-c3c5}
TAG=${2:-r02}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $WL > $O/bench_under_rocprof.json 2> $O/kt.err && \
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $WL > $O/bench_fetch.json 2> $O/fetch.err && \
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $WL > $O/bench_write.json 2> $O/write.err
echo profile_exit=$?
cd $R
cp $(ls $O/kt/*/*_kernel_stats.csv | head -1) $O/kernel_stats.csv
python3 tools/pmc_to_json.py $WL $O/fetch $O/write 3 $O/bench_fetch.json $O/pmc_traffic.json
python3 tools/pmc_summary.py $O/fetch $O/write > $O/pmc_fetch_write_summary.txt
rm -rf $O/kt $O/fetch $O/write      # raw traces are large; the summaries above are what is kept
head -14 $O/kernel_stats.csv
