#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5b; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_rts.py -x -q -p no:cacheprovider 2>&1 | tail -15 ) | tee $O/tests_rts.log
timeout 200 python tools/config4_time.py 2>&1 | tail -n 2 | tee $O/config4.txt
