#!/bin/bash
# Round 4, second GPU stage: (1) the whole -m gpu suite on the rebuilt libraries (general S factorisation in the step kernels,
# symmetric-part / lower-triangle contracts of batch_run / batch_rts, blocked traced run, coefficient batching), (2) A/B of the two
# knobs still open -- the unrolled row pass of the live step kernels, the transposed trace store of the lane-group fused run --,
# (3) which part of `import torch` costs the Python-hosted launches their 6-30 %.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4b; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log
S=tools/ab_step
{
echo "== live step kernels (16384): default (unroll 2) | unroll 22 | default"
for k in 4 12 10; do
  timeout 60 $S live $k 23 22 3 16384 20 200 0.01 gen_ab/in/live$k.bin generated/liblive.so gen_ab/lu/liblive.so generated/liblive.so
done
echo "== live dt = 0 launches"
timeout 60 $S live 10 23 22 3 16384 20 200 0.0 gen_ab/in/live10.bin generated/liblive.so gen_ab/lu/liblive.so
echo "== headline kernel: default | kernel arguments preloaded into SGPRs (-mllvm -amdgpu-kernarg-preload-count=16) | default"
timeout 60 $S kinematic6 1 6 6 3 65536 100 1000 0.01 - generated/libkinematic6.so gen_ab/kp/libkinematic6.so generated/libkinematic6.so
echo "== live gyro step: default | kernarg preload"
timeout 60 $S live 4 23 22 3 16384 20 200 0.01 gen_ab/in/live4.bin generated/liblive.so gen_ab/kp/liblive.so
echo "== fused runs (new defaults)"
timeout 60 tools/ab_run kinematic6 6 6 3 65536 500 3 0 generated/libkinematic6.so
timeout 60 tools/ab_run kinematic6 6 6 3 8192 200 3 1 generated/libkinematic6.so
timeout 60 tools/ab_run kinematic 2 2 1 65536 2000 3 0 generated/libkinematic.so
} > $O/ab.log 2>&1
{
echo "== config 4 chunk: default trace path"
timeout 300 python tools/config4_time.py
echo "== config 4 chunk: transposed trace store (run_trace_t=1)"
RN_GEN_DIR=gen_ab/tt RN_TUNE=run_trace_t=1 timeout 300 python tools/config4_time.py
echo "== config 4 chunk: default again"
timeout 300 python tools/config4_time.py
} > $O/config4.log 2>&1
bash tools/gap_probe2.sh generated > $O/gap2.log 2>&1
tail -4 $O/tests.log
