#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4j; mkdir -p $O
for i in 1 2; do RN_BENCH_MARK_EVERY=1 timeout 100 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=o['roofline']; print('short run', r['launch_us'], r['frac'], r.get('group_us'))"; done | tee $O/short2.log
