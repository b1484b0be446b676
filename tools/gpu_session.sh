cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/clk
rm -rf $O; mkdir -p $O
cd /tmp
for w in nb c3c5; do
timeout -k 10 280 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/$w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency-mode --workload $w > $O/$w.json 2> $O/$w.err
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for w in ('nb','c3c5'):
    dur = collections.defaultdict(list)
    for f in glob.glob(f'gpurun_out/clk/{w}/**/*_kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r['Kernel_Name'].split('(')[0]].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f'gpurun_out/clk/{w}/**/*_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            cnt[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
    print('==', w)
    for k, c in cnt.items():
        if k not in dur or not dur[k]: continue
        d = sum(dur[k])/len(dur[k])
        g = sum(c['GRBM_GUI_ACTIVE'])/len(c['GRBM_GUI_ACTIVE']) if c.get('GRBM_GUI_ACTIVE') else 0
        row = {n: round(sum(v)/len(v)) for n, v in c.items()}
        print(f'{k[:60]:60s} {d/1e3:9.1f} us  clock ~ {g/8/d:5.2f} GHz', {n: row[n] for n in row if n != 'GRBM_GUI_ACTIVE'})
PY
