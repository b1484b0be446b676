#!/bin/bash
# SQ / GRBM counters per kernel of one workload (one --pmc pass; serialises the kernels):
#   gpurun -- 'bash tools/sq_counters.sh c3c5x12 > gpurun_out/sq_c3c5x12.txt'
# clock ~ GRBM_GUI_ACTIVE / 8 / duration; issue share = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES (both in quad-cycles);
# cycles per vector instruction on a SIMD that holds n waves = 4 * SQ_ACTIVE_INST_ANY / SQ_INSTS_VALU (per wave).
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${1:-c3c5x12}
O=$R/gpurun_out/sq_$WL
rm -rf $O; mkdir -p $O
(cd /tmp && timeout -k 10 280 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency-mode --workload $WL > $O.json 2> $O.err) || { tail -5 $O.err; exit 1; }
cd $R
python3 - "$O" "$WL" <<'PY'
import collections, csv, glob, sys
o, wl = sys.argv[1], sys.argv[2]
dur = collections.defaultdict(list)
for f in glob.glob(o + '/**/*_kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Kernel_Name'].split('(')[0]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(o + '/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
print(f'# {wl}: per kernel, averages over its launches (one PMC pass: kernels run one at a time)')
print('kernel | launches | us | clock GHz | wave quad-cycles | issuing | parked (wait_any) | stalled at issue | VALU wave-instructions | VALU wave-instr per us per SIMD')
for k, c in sorted(cnt.items(), key=lambda kv: -sum(dur.get(kv[0], [0]))):
    if not dur.get(k):
        continue
    d = sum(dur[k]) / len(dur[k])
    a = {n: sum(v) / len(v) for n, v in c.items()}
    wc = a.get('SQ_WAVE_CYCLES', 0) or 1
    print(f"{k[:70]} | {len(dur[k])} | {d / 1e3:.1f} | {a.get('GRBM_GUI_ACTIVE', 0) / 8 / d:.2f} | {wc:.3g} | {a.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} | "
          f"{a.get('SQ_WAIT_ANY', 0) / wc:.2f} | {a.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} | {a.get('SQ_INSTS_VALU', 0):.4g} | {a.get('SQ_INSTS_VALU', 0) / (d / 1e3) / 1024:.1f}")
PY
rm -rf $O
