#!/bin/bash
# Round 4, GPU stage h: reciprocal-square-root quaternion normalisation and rsqrt / rcp powers in the lowered scalars (the one lane per
# filter that evaluates them) against the IEEE sqrt + division builds, same call; then the parity suites that touch them.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4h; mkdir -p $O
{
echo "== live gyro / accelerometer / position launches (16384, dt = 0.01): IEEE build | fast build | IEEE build"
for k in 4 10 12; do
  timeout 60 tools/ab_step live $k 23 22 3 16384 20 200 0.01 gen_ab/in/live$k.bin gen_ab/ieee/liblive.so generated/liblive.so gen_ab/ieee/liblive.so
done
echo "== config 4 chunk: fast scalars"
timeout 300 python tools/config4_time.py
echo "== config 4 chunk: IEEE scalars"
RN_GEN_DIR=gen_ab/ieee RN_NO_GEN=1 timeout 300 python tools/config4_time.py
echo "== config 4 chunk: fast scalars again"
timeout 300 python tools/config4_time.py
} 2>&1 | grep -v amdgpu.ids > $O/ab.log
timeout 1200 python -m pytest tests/test_gpu_live.py tests/test_gpu_parity.py tests/test_gpu_attitude.py tests/test_gpu_rts.py tests/test_gpu_asymmetric.py tests/test_gpu_run.py tests/test_gpu_random.py -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log
cat $O/ab.log; tail -5 $O/tests.log
