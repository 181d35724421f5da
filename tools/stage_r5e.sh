#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5e; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_rts.py tests/test_gpu_random.py tests/test_gpu_asymmetric.py tests/test_gpu_wideobs.py -x -q -p no:cacheprovider 2>&1 | tail -15 ) | tee $O/tests.log
( timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -p no:cacheprovider -k "config4" 2>&1 | tail -15 ) | tee $O/tests_full.log
timeout 200 python tools/config4_time.py 2>&1 | tail -n 2 | tee $O/config4.txt
