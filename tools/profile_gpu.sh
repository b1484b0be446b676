#!/bin/bash
# Run on the GPU box through gpurun:  gpurun --timeout 1100 -- 'bash tools/profile_gpu.sh <tag> <workload> [workload ...]'
# Per workload three separate passes (kernel trace + stats; PMC FETCH_SIZE; PMC WRITE_SIZE) -- gpurun refuses
# PMC combined with sys/hip traces.  Results under gpurun_out/prof_<tag>/<workload>/ : kernel_stats.csv,
# pmc_fetch_write_summary.txt, the bench lines of the three runs; gpurun_out/prof_<tag>/pmc_traffic.json gets one
# entry per workload (stamped with the source hash).  Copy what should be judged into profiles/ (tracked).
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04}
shift
WLS=${@:-c3c5x12}
B="--no-cpu-baseline --no-latency-mode"
# entries of earlier calls (gpurun_out/ does not travel to the box; profiles/ does): keep them if the sources still match
mkdir -p $R/gpurun_out/prof_$TAG
[ -f $R/gpurun_out/prof_$TAG/pmc_traffic.json ] || cp $R/profiles/pmc_traffic.json $R/gpurun_out/prof_$TAG/pmc_traffic.json 2>/dev/null
for WL in $WLS; do
    O=$R/gpurun_out/prof_$TAG/$WL
    mkdir -p $O
    cd /tmp
    timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 3 --warmup 1 $B --workload $WL > $O/bench_under_rocprof.json 2> $O/kt.err && \
    timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 2 --warmup 1 $B --workload $WL > $O/bench_fetch.json 2> $O/fetch.err && \
    timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 2 --warmup 1 $B --workload $WL > $O/bench_write.json 2> $O/write.err
    rc=$?
    echo "$WL profile_exit=$rc"
    cd $R
    if [ $rc -ne 0 ]; then tail -5 $O/*.err; exit $rc; fi
    cp $(ls $O/kt/*/*_kernel_stats.csv | head -1) $O/kernel_stats.csv
    python3 tools/pmc_to_json.py $WL $O/fetch $O/write $O/bench_fetch.json $R/gpurun_out/prof_$TAG/pmc_traffic.json
    python3 tools/pmc_summary.py $O/fetch $O/write > $O/pmc_fetch_write_summary.txt
    rm -rf $O/kt $O/fetch $O/write $O/*.err     # raw traces are large; the summaries above are what is kept
    head -6 $O/kernel_stats.csv | cut -c1-160
done
