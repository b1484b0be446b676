#!/bin/bash
# Host side of the engine (argument validation, geometry, error paths, option handling) under AddressSanitizer + UBSan:
# hipcc accepts -fsanitize=address,undefined with -fno-gpu-sanitize on this image (device code stays uninstrumented).
# Runs the C-ABI tests that need no GPU against that build (in this container there is no GPU: everything behind
# sgm_create's device check is out of reach; on the GPU box sanitizer runs are not offered).
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
make -s -C $R/stereo_reconstruction_cv_amd/csrc build/libsgm_hip_asan.so || exit 1
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd $R
SGM_HIP_LIB=$R/stereo_reconstruction_cv_amd/csrc/build/libsgm_hip_asan.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
    UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 python -m pytest -q -p no:cacheprovider tests/test_abi.py \
    -k "header or exports or parameter_validation or no_silent or never_imports" "$@" && echo "sanitize_engine_host: no report"
