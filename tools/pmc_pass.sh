#!/bin/bash
# One rocprofv3 --pmc pass per argument (a quoted, space-separated counter list each), on
# `bench.py --workload $WL`; per-kernel averages to gpurun_out/pmc_<tag>_<n>.txt.
#   gpurun -- 'WL=c3c5 TAG=x bash tools/pmc_pass.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum"'
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${WL:-c3c5}
TAG=${TAG:-x}
EXTRA=${EXTRA:-}
n=0
for counters in "$@"; do
    n=$((n + 1))
    O=$R/gpurun_out/pmc_${TAG}_$n
    rm -rf $O && mkdir -p $O
    (cd /tmp && timeout -k 10 280 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $WL $EXTRA > $O.bench.json 2> $O.err) || exit 1
    python3 $R/tools/pmc_summary.py $O > $O.txt
    rm -rf $O
    grep -i "prepass\|k_sweep\|k_box\|k_wta\|k_pix" $O.txt
done
