#!/bin/bash
# Everything profiles/<tag>/ holds besides the rocprof passes (tools/profile_gpu.sh), from one build:
#   gpurun --timeout 1100 -- 'bash tools/round_artifacts.sh r02'
# -> gpurun_out/art_<tag>/ : bench_<workload>.json + _stages.txt (with cpu_baseline), bench_c3c5_verify.json,
#    pytest_gpu.log, host_rate.txt, rehearse_n2_*.json
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
O=$R/gpurun_out/art_$TAG
mkdir -p $O
cd $R
python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
for w in c3c5 c1 c2 c3 c4 c5 nb c3c5x2; do
    python bench.py --workload $w --stages > $O/bench_$w.json 2> $O/bench_${w}_stages.txt || echo "bench $w failed"
    grep -v amdgpu.ids $O/bench_${w}_stages.txt > $O/t && mv $O/t $O/bench_${w}_stages.txt
    python3 -c "import json; d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); print('$w', round(d['ms_per_step'],3), 'ms/step', round(d['value']), d['unit'])"
done
python bench.py --verify --no-cpu-baseline > $O/bench_c3c5_verify.json 2> /dev/null; python3 -c "import json; d=json.loads(open('$O/bench_c3c5_verify.json').read().strip().splitlines()[-1]); print('verify', d.get('verify'))"
python tools/host_rate.py > $O/host_rate.txt 2>&1; cat $O/host_rate.txt
BENCH_REHEARSE=1 python bench.py --gpus 2 --workload c4 --steps 2 --warmup 1 --no-cpu-baseline > $O/rehearse_n2_c4.json 2> $O/rehearse_n2_c4.err; tail -c 400 $O/rehearse_n2_c4.json
BENCH_REHEARSE=1 python bench.py --gpus 2 --workload c5 --ingest rank0 --steps 2 --warmup 1 --no-cpu-baseline > $O/rehearse_n2_c5_ingest_rank0.json 2> $O/rehearse_n2_c5.err; tail -c 400 $O/rehearse_n2_c5_ingest_rank0.json
