#!/bin/bash
# Everything profiles/<tag>/ holds besides the rocprof passes (tools/profile_gpu.sh), from one build:
#   gpurun --timeout 1100 -- 'bash tools/round_artifacts.sh r03'
# -> gpurun_out/art_<tag>/ : bench_<workload>.json + _stages.txt (the default workload with cpu_baseline and latency_mode),
#    bench_default_verify.json, host_rate.txt, rehearse_n2_*.json
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04}
O=$R/gpurun_out/art_$TAG
mkdir -p $O
cd $R
python bench.py --stages > $O/bench_default.json 2> $O/bench_default_stages.txt || echo "default bench failed"
for w in c3c5 c1 c1x8 c1t c2 c3 c4 c5 nb c3c5x2 c3c5x3 c3c5x6 c3c5x8 c3c5x12 c3c5x14 c3c5x16 c3c5x18 c3c5x24 c5x12 c5x17 c4t; do
    python bench.py --workload $w --stages --no-cpu-baseline --no-latency-mode > $O/bench_$w.json 2> $O/bench_${w}_stages.txt || echo "bench $w failed"
done
for f in $O/bench_*_stages.txt; do grep -v amdgpu.ids $f > $O/t; mv $O/t $f; done
for w in default c3c5 c1 c1x8 c1t c2 c3 c4 c5 nb c3c5x2 c3c5x3 c3c5x6 c3c5x8 c3c5x12 c3c5x14 c3c5x16 c3c5x18 c3c5x24 c5x12 c5x17 c4t; do
    python3 -c "import json; d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); print('$w', round(d['ms_per_step'],3), 'ms/step', round(d['ms_per_pair'],3), 'ms/pair', round(d['pairs_per_s'],1), 'pairs/s', round(d['value']), 'Mdisp/s')"
done
python bench.py --verify --no-cpu-baseline > $O/bench_default_verify.json 2> /dev/null; python3 -c "import json; d=json.loads(open('$O/bench_default_verify.json').read().strip().splitlines()[-1]); print('verify', d.get('verify'))"
python tools/host_rate.py > $O/host_rate.txt 2>&1; cat $O/host_rate.txt
BENCH_REHEARSE=1 python bench.py --gpus 2 --workload c4 --steps 2 --warmup 1 --no-cpu-baseline > $O/rehearse_n2_c4.json 2> $O/rehearse_n2_c4.err; tail -c 600 $O/rehearse_n2_c4.json
BENCH_REHEARSE=1 python bench.py --gpus 2 --workload c3c5x6 --steps 2 --warmup 1 --no-cpu-baseline > $O/rehearse_n2_c3c5x6.json 2> $O/rehearse_n2_c3c5x6.err; tail -c 600 $O/rehearse_n2_c3c5x6.json
