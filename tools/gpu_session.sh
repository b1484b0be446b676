cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_chain.py -x -q 2>&1 | tail -3
run() {
  echo "== $*"
  timeout -k 10 300 python bench.py --workload $1 $2 $3 $4 $5 $6 $7 $8 --stages --no-cpu-baseline --no-latency-mode 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms/step',round(d['ms_per_step'],3),'ms/pair',round(d['ms_per_pair'],3),'pairs/s',round(d['pairs_per_s'],2))
    elif 'chain' in l or 'sum' in l: print(l.rstrip())
"
}
run c1t
run c4t
run c4t64
run c3c5x12
