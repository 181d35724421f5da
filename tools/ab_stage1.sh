#!/bin/bash
# Stage 1 of an A/B decision on the GPU box (no Python): the untraced fused run of the lane-per-filter models, HEAD build
# (gen_ab/old) against the blocked kernel at several block sizes.  Prints one line per build; first line of a group is the reference.
cd "$(dirname "$0")/.." || exit 1
R=tools/ab_run
echo "== kinematic parity, ragged n, T not a multiple of any block"
timeout 60 $R kinematic 2 2 1 1000 203 1 0 gen_ab/old/libkinematic.so generated/libkinematic.so gen_ab/k8/libkinematic.so gen_ab/k16/libkinematic.so gen_ab/k64/libkinematic.so
echo "== kinematic6 parity"
timeout 60 $R kinematic6 6 6 3 777 67 1 0 gen_ab/old/libkinematic6.so generated/libkinematic6.so gen_ab/k2/libkinematic6.so gen_ab/k8/libkinematic6.so
echo "== kinematic 65536 x 2000 (bench extra kinematic_fused)"
timeout 120 $R kinematic 2 2 1 65536 2000 3 0 gen_ab/old/libkinematic.so generated/libkinematic.so gen_ab/k8/libkinematic.so gen_ab/k16/libkinematic.so gen_ab/k64/libkinematic.so
echo "== kinematic6 65536 x 500 (bench extra fused_run)"
timeout 120 $R kinematic6 6 6 3 65536 500 3 0 gen_ab/old/libkinematic6.so generated/libkinematic6.so gen_ab/k2/libkinematic6.so gen_ab/k8/libkinematic6.so
echo "== kinematic 1048576 x 96"
timeout 120 $R kinematic 2 2 1 1048576 96 2 0 gen_ab/old/libkinematic.so generated/libkinematic.so gen_ab/k16/libkinematic.so
