set -e
cd $GRAFT_REPO_ROOT
run() {
  echo "== $*"
  timeout -k 10 300 python bench.py --workload $1 --schedule $2 --chain-wgs $3 $4 $5 $6 --steps 4 --warmup 1 --no-cpu-baseline --stages 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms/step',round(d['ms_per_step'],3),'pairs/s',round(d['pairs_per_s'],2))
    elif 'chain' in l or 'sum' in l or 'sweep' in l or 'prepass' in l or 'cost' in l or 'wta' in l: print(l.rstrip())
"
}
run c3c5x8 2 0 --batch
run c3c5x12 2 0 --batch
run c3c5x12 2 0 --batch --debug 2
run c3c5x6 2 0 --batch --debug 2
run c3c5x12 2 0 --batch --sweep-rows 12
