#!/bin/bash
# Last GPU call of round 4: the lowered sin / cos pairs (rn::sincos_fast) on the device -- live parity against oracle and golden vectors,
# then one config-4 chunk (forward / backward) and the live stream of bench.py for the timing.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4k; mkdir -p $O
timeout 170 python -m pytest tests/test_gpu_live.py -x -q -m gpu > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log
tail -n 4 $O/tests.log
timeout 70 python tools/config4_time.py > $O/c4.log 2>&1; tail -n 2 $O/c4.log
timeout 60 python bench.py --model live --steps 420 --warmup 42 --no-extras --no-cpu-baseline > $O/live.json 2> $O/live.err; cut -c1-600 $O/live.json
