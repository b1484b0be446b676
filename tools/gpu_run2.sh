cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_batch_entries.py tests/test_gpu_configs.py::test_c4_batch_1080p_d128_both_batch_entries -x -q -m gpu > gpurun_out/session_new.log 2>&1; rc=$?
tail -5 gpurun_out/session_new.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/host_rate.py > gpurun_out/host_rate_r04b.txt 2>&1; tail -12 gpurun_out/host_rate_r04b.txt
