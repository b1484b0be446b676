set -e
cd $GRAFT_REPO_ROOT
run() {
  echo "== $*"
  timeout -k 10 300 python bench.py --workload $1 --schedule $2 --chain-wgs $3 --steps 6 --warmup 2 --no-cpu-baseline --stages 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms/step',round(d['ms_per_step'],3),'pairs/s',round(d['pairs_per_s'],2))
    elif 'chain' in l or 'sum' in l or 'sweep' in l or 'prepass' in l: print(l.rstrip())
"
}
export GPU_MAX_HW_QUEUES=8
run c3c5x4 2 0
run c3c5x6 2 0
run c3c5x6 2 40
run c3c5x6 1 0
export GPU_MAX_HW_QUEUES=16
run c3c5x6 2 0
