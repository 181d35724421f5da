#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4g; mkdir -p $O
{
echo "== config 4 chunk: xk_k committed before the image copy is issued"
timeout 300 python tools/config4_time.py
echo "== config 4 chunk: round-3 structure (gen_ab/rts_old)"
RN_GEN_DIR=gen_ab/rts_old RN_NO_GEN=1 timeout 300 python tools/config4_time.py
echo "== config 4 chunk: new again"
timeout 300 python tools/config4_time.py
} 2>&1 | grep -v amdgpu.ids > $O/config4.log
timeout 600 python -m pytest tests/test_gpu_rts.py tests/test_gpu_asymmetric.py "tests/test_gpu_random.py::test_smoother_many_shapes" "tests/test_gpu_fullsize.py::test_config4_resynchronised_backward_steps" -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log
cat $O/config4.log; tail -3 $O/tests.log
