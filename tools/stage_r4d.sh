#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4d; mkdir -p $O
timeout 300 python tools/gap_probe4.py 2>&1 | grep -v amdgpu.ids > $O/gap4.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_short.json 2>> $O/bench_default.err
cat $O/gap4.log; tail -3 $O/bench_default.err
