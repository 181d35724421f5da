#!/bin/bash
# Round 4, GPU stage e: the smoother with its read-back through the image and the diagonal process noise in registers against the
# previous build (same call), its parity tests, the bisection switch of the gap probe, a default and a short bench run.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rts.py tests/test_gpu_asymmetric.py tests/test_gpu_random.py tests/test_gpu_msckf.py tests/test_gpu_timelines.py tests/test_gpu_cpp.py "tests/test_gpu_fullsize.py::test_config4_resynchronised_backward_steps" tests/test_gpu_mid.py tests/test_gpu_attitude.py -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log
{
echo "== config 4 chunk: new smoother"
timeout 300 python tools/config4_time.py
echo "== config 4 chunk: previous build (strided read-back, Q rows from memory)"
RN_GEN_DIR=gen_ab/rts_old RN_NO_GEN=1 timeout 300 python tools/config4_time.py
echo "== config 4 chunk: new smoother again"
timeout 300 python tools/config4_time.py
} 2>&1 | grep -v amdgpu.ids > $O/config4.log
{
RN_PROBE_MODE=tensors timeout 200 python tools/notorch_probe.py
RN_PROBE_MODE=tensors RN_PROBE_BEKF=1 timeout 200 python tools/notorch_probe.py
RN_PROBE_MODE=tensors RN_PROBE_BEKF=2 timeout 200 python tools/notorch_probe.py
} 2>&1 | grep -v amdgpu.ids > $O/gap5.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_short.json 2>> $O/bench_default.err
tail -3 $O/tests.log; cat $O/config4.log $O/gap5.log; tail -3 $O/bench_default.err
