#!/bin/bash
# bench.py after the warm-up reorder (steady-state groups first, initial state restored, then the --warmup steps): the live stream and
# the driver-style short headline run.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4l; mkdir -p $O
timeout 50 python bench.py --model live --steps 420 --warmup 42 --no-extras --no-cpu-baseline > $O/live.json 2> $O/live.err; tail -n 2 $O/live.err; cut -c1-300 $O/live.json
timeout 40 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/short.json 2> $O/short.err; tail -n 2 $O/short.err; cut -c1-200 $O/short.json
