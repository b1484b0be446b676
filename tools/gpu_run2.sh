cd $GRAFT_REPO_ROOT
export SGM_ALLOW_WRONG_RESULTS=1
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-latency-mode --steps 20 --warmup 3 --workload c3c5 "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$*', '-> ms/pair %.3f' % d['ms_per_pair'], {k: round(v,2) for k,v in d['stage_ms'].items() if k.startswith('cost')})
"
}
run
run --debug 262144
run --debug 524288
run --debug 1048576
run --debug 131072
