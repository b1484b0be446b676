cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_paths.py -x -q 2>&1 | tail -3
run() {
  echo "== $*"
  timeout -k 10 300 python bench.py --workload $1 $2 $3 $4 $5 --stages --no-cpu-baseline --no-latency-mode 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms/step',round(d['ms_per_step'],3),'ms/pair',round(d['ms_per_pair'],3),'pairs/s',round(d['pairs_per_s'],2))
    elif 'chain' in l or 'sum' in l: print(l.rstrip())
"
}
run c3c5x12
run c3c5x6
run c4t
run c5x12
run c3c5 --schedule 2
