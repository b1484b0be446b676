cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu_r04.log 2>&1; rc=$?
tail -5 gpurun_out/pytest_gpu_r04.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --stages > gpurun_out/bench_default_r04.json 2> gpurun_out/bench_default_r04_stages.txt; echo bench rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default_r04.json').read().strip().splitlines()[-1])
print(d['ms_per_pair'], d['pairs_per_s'], d['roofline']['frac'], d['cpu_baseline']['all_threads']['value'], d['speedup_vs_cpu_all_threads'], d['latency_mode']['ms_per_pair'])
PY
python -c "import __graft_entry__ as g; g.smoke()"
