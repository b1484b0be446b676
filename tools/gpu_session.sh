cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
run() {
  echo "== $*"
  timeout -k 10 300 python bench.py --workload $1 $2 $3 $4 $5 --stages --no-cpu-baseline --no-latency-mode 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms/step',round(d['ms_per_step'],3),'ms/pair',round(d['ms_per_pair'],3),'pairs/s',round(d['pairs_per_s'],2))
    elif 'amdgpu' not in l: print(l.rstrip())
"
}
run c1
run nb
run c1x8
run c4
run c2
run c4t
