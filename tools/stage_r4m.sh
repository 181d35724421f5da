#!/bin/bash
# Repeat of the driver-style short run and the other run_model workloads after the warm-up reorder.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4m; mkdir -p $O
for i in 1 2; do timeout 30 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=o['roofline']; print('short run', o['value'], r['launch_us'], r['frac'], r['launch_us_median'])"; done | tee $O/short.log
for m in kinematic9 kinematic; do timeout 30 python bench.py --model $m --steps 300 --warmup 30 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=o['roofline']; print('$m', o['value'], r['launch_us'], r['frac'])"; done | tee $O/models.log
