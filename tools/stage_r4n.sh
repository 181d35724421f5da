#!/bin/bash
# The seeded random models (sin terms: their F holds cos of the same arguments) on the device after the sin / cos lowering; the 24-state
# one now ships its fused run (no scratch any more): its asymmetric-covariance contract as well.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4n; mkdir -p $O
timeout 55 python -m pytest tests/test_gpu_random.py tests/test_gpu_asymmetric.py -q -m gpu > $O/random.log 2>&1; echo "pytest rc $?" >> $O/random.log
grep -E "FAILED|Error" $O/random.log | head -8; tail -n 3 $O/random.log
