cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r03b.log 2>&1; tail -5 gpurun_out/pytest_gpu_r03b.log
timeout -k 10 300 python bench.py --stages > gpurun_out/bench_default_r03b.json 2> gpurun_out/bench_default_r03b.err; tail -20 gpurun_out/bench_default_r03b.err; cat gpurun_out/bench_default_r03b.json | cut -c1-3000
