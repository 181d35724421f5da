#!/bin/bash
# Stage 1c: the lane-group fused run with the filtered trace (config 4 forward, one chunk of 8 192 filters, 210 steps): HEAD build
# against the build whose next observation lands before a step's stores are issued.
cd "$(dirname "$0")/.." || exit 1
export AB_KINDS=4,10,4,10,12
echo "== live_maha traced run, 8192 x 210"
timeout 100 tools/ab_run live_maha 23 22 3 8192 210 3 1 gen_ab/old/liblive_maha.so gen_ab/z1/liblive_maha.so gen_ab/old/liblive_maha.so
echo "== live_maha untraced run, 8192 x 210"
timeout 100 tools/ab_run live_maha 23 22 3 8192 210 3 0 gen_ab/old/liblive_maha.so gen_ab/z1/liblive_maha.so
echo "== ragged: 1003 x 37 traced"
timeout 100 tools/ab_run live_maha 23 22 3 1003 37 1 1 gen_ab/old/liblive_maha.so gen_ab/z1/liblive_maha.so
