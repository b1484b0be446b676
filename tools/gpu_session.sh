cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout -k 10 900 python tools/soak_medium.py 40 > gpurun_out/soak_medium_r03.log 2>&1; tail -3 gpurun_out/soak_medium_r03.log
