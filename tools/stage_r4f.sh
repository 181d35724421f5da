#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4f; mkdir -p $O
RN_ALLOW_SPILLS=1 RN_TUNE=wide_timeline=1 RN_GEN_DIR=gen_tl timeout 300 python tools/timeline.py rts 2>&1 | grep -v amdgpu.ids > $O/timeline_rts.txt
RN_ALLOW_SPILLS=1 RN_TUNE=wide_timeline=1 RN_GEN_DIR=gen_tl timeout 300 python tools/timeline.py run 8192 trace 2>&1 | grep -v amdgpu.ids > $O/timeline_run.txt
timeout 300 python bench.py --no-extras --no-cpu-baseline > $O/bench_noextras.json 2> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_short.json 2>> $O/bench.err
cat $O/timeline_rts.txt $O/timeline_run.txt; cat $O/bench_noextras.json $O/bench_short.json | cut -c1-900
