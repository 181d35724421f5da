#!/bin/bash
# Round 4, GPU stage i: the deferred, woven covariance-trace copy of the lane-group fused run against the previous structure (same
# call), and the parity suites of everything rebuilt (guarded reciprocals in the lowered scalars, woven run).
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4i; mkdir -p $O
{
echo "== config 4 chunk: woven trace copy"
timeout 300 python tools/config4_time.py
echo "== config 4 chunk: previous structure (gen_ab/noweave)"
RN_GEN_DIR=gen_ab/noweave RN_NO_GEN=1 timeout 300 python tools/config4_time.py
echo "== config 4 chunk: woven again"
timeout 300 python tools/config4_time.py
} 2>&1 | grep -v amdgpu.ids > $O/config4.log
timeout 1200 python -m pytest tests/test_gpu_msckf.py tests/test_gpu_run.py tests/test_gpu_live.py tests/test_gpu_asymmetric.py tests/test_gpu_mid.py tests/test_gpu_attitude.py "tests/test_gpu_random.py::test_fused_run_and_step_path_vs_oracle" "tests/test_gpu_random.py::test_trace_vs_step_path_many_shapes" tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log
cat $O/config4.log; tail -6 $O/tests.log
