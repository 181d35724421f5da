#!/bin/bash
# Stage 1b: step kernels with Q / shared R requested behind the tile loads (lane-per-filter) and asynchronous x / z loads in the
# scalar phase (lane groups), against the HEAD builds in gen_ab/old; fused runs at the chosen block sizes.
cd "$(dirname "$0")/.." || exit 1
S=tools/ab_step
echo "== kinematic6 step, 65536 filters (headline kernel)"
timeout 60 $S kinematic6 1 6 6 3 65536 100 1000 0.01 - gen_ab/old/libkinematic6.so generated/libkinematic6.so gen_ab/q2/libkinematic6.so gen_ab/old/libkinematic6.so
echo "== kinematic6 step, 1048576 filters"
timeout 60 $S kinematic6 1 6 6 3 1048576 10 100 0.01 - gen_ab/old/libkinematic6.so generated/libkinematic6.so gen_ab/q2/libkinematic6.so
echo "== kinematic step, 65536 filters"
timeout 60 $S kinematic 1 2 2 1 65536 100 1000 0.01 - gen_ab/old/libkinematic.so generated/libkinematic.so
echo "== live gyro step dt > 0, 16384 filters"
timeout 60 $S live 4 23 22 3 16384 20 200 0.01 gen_ab/in/live4.bin gen_ab/old/liblive.so generated/liblive.so gen_ab/old/liblive.so
echo "== live gyro step dt = 0"
timeout 60 $S live 4 23 22 3 16384 20 200 0.0 gen_ab/in/live4.bin gen_ab/old/liblive.so generated/liblive.so
echo "== live ECEF position step dt > 0"
timeout 60 $S live 12 23 22 3 16384 20 200 0.01 gen_ab/in/live12.bin gen_ab/old/liblive.so generated/liblive.so
echo "== fused runs at the chosen block sizes"
timeout 60 tools/ab_run kinematic 2 2 1 65536 2000 3 0 gen_ab/old/libkinematic.so generated/libkinematic.so
timeout 60 tools/ab_run kinematic6 6 6 3 65536 500 3 0 gen_ab/old/libkinematic6.so generated/libkinematic6.so
timeout 60 tools/ab_run kinematic6 6 6 3 1000 67 1 0 gen_ab/old/libkinematic6.so generated/libkinematic6.so
