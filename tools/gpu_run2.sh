cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-latency-mode --steps 8 --warmup 3 "$@" 2>/dev/null | python -c "
import sys,json,os
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(os.environ.get('GPU_MAX_HW_QUEUES','default'), '$*', '-> ms/pair %.3f' % d['ms_per_pair'], {k: round(v,2) for k,v in d['stage_ms'].items() if v > 0.5})
"
}
run --workload c3c5x17
export GPU_MAX_HW_QUEUES=8
run --workload c3c5x17
export GPU_MAX_HW_QUEUES=16
run --workload c3c5x17
export GPU_MAX_HW_QUEUES=2
run --workload c3c5x17
