#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer pass over the CPU side (CPU only: GPU ASan / xnack are not available
# on this pool).  Builds oracle/liboracle_sgbm_asan.so (-fsanitize=address,undefined) and runs the oracle's own tests
# (known answers, brute-force cross-check, golden vectors, rectification, real pairs) against it, with libasan
# preloaded into the interpreter, then the oracle on the reference's full-resolution d3 pair at the notebook's setting
# (the input of tests/test_oracle_vs_notebook_figure.py; that test itself imports matplotlib / scipy, whose extension
# modules do not survive a preloaded ASan run-time).  Any report makes the run fail (halt_on_error, -fno-sanitize-recover).
#   bash tools/sanitize_oracle.sh            -> profiles/<tag>/sanitize_oracle.log by hand
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
make -s -C $R/oracle liboracle_sgbm_asan.so || exit 1
ASAN=$(gcc -print-file-name=libasan.so)
cd $R
ORACLE_SANITIZE=1 LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
    python -m pytest -q -x -p no:cacheprovider tests/test_oracle_known_answers.py tests/test_oracle_vs_bruteforce.py tests/test_golden.py \
    tests/test_oracle_rectify.py tests/test_real_pairs.py "$@" || exit 1
if [ -f /root/reference/dataset/d3/img1.jpg ]; then
ORACLE_SANITIZE=1 LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 python - <<'PY' || exit 1
import sys
import numpy as np
from PIL import Image
sys.path.insert(0, '.')
from oracle import oracle as O
g = lambda f: np.asarray(Image.open(f).convert('L'), dtype=np.uint8)
l, r = g('/root/reference/dataset/d3/img1.jpg'), g('/root/reference/dataset/d3/img2.jpg')
for mode in (0, 1):
    d, t = O.sgbm_compute(l, r, taps='light', minDisparity=0, numDisparities=16, blockSize=11, P1=2904, P2=11616, disp12MaxDiff=1,
                          preFilterCap=63, uniquenessRatio=10, speckleWindowSize=100, speckleRange=32, mode=mode)
    x = O.reproject(O.disp_to_float(d), np.array([[1, 0, 0, -1909.9754], [0, 1, 0, -1057.74529], [0, 0, 0, 2045.48384], [0, 0, -1.0, 0]]))
    print('d3 full resolution, mode', mode, 'under ASan + UBSan: ok, valid fraction', float((d >= 0).mean()))
PY
fi
echo "sanitize_oracle: no report"
