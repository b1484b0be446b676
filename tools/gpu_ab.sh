cd $GRAFT_REPO_ROOT
run() {
  echo "== $* (LDS5=$SGM_EXP_LDS5)"
  timeout -k 10 300 python bench.py --workload $1 --stages --no-cpu-baseline --no-latency-mode --steps 40 --warmup 5 2>&1 | python -c "
import sys,json
o=[]
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); o.append('ms/pair %.3f' % d['ms_per_pair'])
    elif l.startswith('  ') and any(k in l for k in ('path','lines','prepass','sweep','wta')): o.append(' '.join(l.split()[:2]))
print(' | '.join(o))
"
}
for w in nb c1 c1x8; do
for v in 0 20000 27000 40000 54000 81000; do
export SGM_EXP_LDS5=$v
run $w || exit 1
done; done
