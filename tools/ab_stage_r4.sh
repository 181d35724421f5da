#!/bin/bash
# GPU side (after tools/ab_prepare_r4.sh): every prepared knob against the default build, results and times in one log.
cd "$(dirname "$0")/.." || exit 1
S=tools/ab_step; R=tools/ab_run; D=gen_ab/default
echo "== headline step kernel (kinematic6, 65536): default | small_sym"
timeout 60 $S kinematic6 1 6 6 3 65536 100 1000 0.01 - $D/libkinematic6.so gen_ab/sym/libkinematic6.so $D/libkinematic6.so
echo "== kinematic6 step, 1 M filters"
timeout 60 $S kinematic6 1 6 6 3 1048576 10 100 0.01 - $D/libkinematic6.so gen_ab/sym/libkinematic6.so
echo "== fused runs, untraced (65536 filters): default | small_sym"
timeout 60 $R kinematic6 6 6 3 65536 500 3 0 $D/libkinematic6.so gen_ab/sym/libkinematic6.so
timeout 60 $R kinematic 2 2 1 65536 2000 3 0 $D/libkinematic.so gen_ab/sym/libkinematic.so
echo "== fused runs WITH trace (8192 x 200): k_run | k_run_blk_tr | k_run_blk_tr + small_sym"
timeout 60 $R kinematic6 6 6 3 8192 200 3 1 $D/libkinematic6.so gen_ab/bt/libkinematic6.so gen_ab/symbt/libkinematic6.so
timeout 60 $R kinematic 2 2 1 65536 200 3 1 $D/libkinematic.so gen_ab/bt/libkinematic.so
echo "== live step kernels (16384): default | coefficient batching | + G from the row + unrolled row pass"
for k in 4 12; do
  timeout 60 $S live $k 23 22 3 16384 20 200 0.01 gen_ab/in/live$k.bin $D/liblive.so gen_ab/lc/liblive.so gen_ab/ls/liblive.so $D/liblive.so
done
timeout 60 $S live 10 23 22 3 16384 20 200 0.0 gen_ab/in/live10.bin $D/liblive.so gen_ab/lc/liblive.so gen_ab/ls/liblive.so
