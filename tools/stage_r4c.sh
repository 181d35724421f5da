#!/bin/bash
# Round 4, third GPU stage: parity of what changed since stage b, the smoother's early request of the filtered records (A/B), the gap
# probe's third stage, and a default bench run.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_asymmetric.py tests/test_gpu_random.py tests/test_gpu_rts.py tests/test_gpu_run.py tests/test_gpu_live.py tests/test_gpu_parity.py tests/test_gpu_mid.py -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log
{
echo "== config 4 chunk: default (filtered records requested early in phase J)"
timeout 300 python tools/config4_time.py
echo "== config 4 chunk: rts3_early_pf=0"
RN_GEN_DIR=gen_ab/rl RN_TUNE=rts3_early_pf=0 timeout 300 python tools/config4_time.py
echo "== config 4 chunk: default again"
timeout 300 python tools/config4_time.py
} 2>&1 | grep -v amdgpu.ids > $O/config4.log
bash tools/gap_probe3.sh > $O/gap3.log 2>&1
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_short.json 2>> $O/bench_default.err
tail -3 $O/tests.log; cat $O/config4.log $O/gap3.log
