cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/session_pytest.log 2>&1; rc=$?
tail -5 gpurun_out/session_pytest.log
[ $rc -eq 0 ] || exit $rc
run() {
  timeout -k 10 300 python bench.py --workload $1 --stages --no-cpu-baseline --no-latency-mode --steps ${2:-10} --warmup 3 2>&1 | python -c "
import sys,json
o=['$1']
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); o.append('ms/pair %.3f' % d['ms_per_pair'])
    elif l.startswith('  ') and not l.startswith('  sum'): o.append(' '.join(l.split()[:2]))
print(' | '.join(o))
"
}
run c3c5x17 8 && run c3c5x24 6 && run c3c5x12 8 && run c3c5 20 && run c2 20 && run c4t 10 && run c1t 10 && run c5x17 8
