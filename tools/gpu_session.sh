cd $GRAFT_REPO_ROOT
run() {
  echo "== $*"
  timeout -k 10 300 python bench.py --workload $1 $2 $3 $4 $5 $6 $7 $8 --stages --no-cpu-baseline --no-latency-mode 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms/step',round(d['ms_per_step'],3),'ms/pair',round(d['ms_per_pair'],3),'pairs/s',round(d['pairs_per_s'],2))
    elif 'amdgpu' not in l: print(l.rstrip())
"
}
run c1x16 --schedule 2 --batch 1 --sweep-rows 12
run c1x32 --schedule 2 --batch 1 --sweep-rows 12
run c1x32 --schedule 2 --batch 1 --sweep-rows 6
run c1x16
run c1x8 --debug 4096
