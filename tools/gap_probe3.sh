#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
run() { PP_Z=$1 PP_CALL=$2 PP_EV=$3 PP_STATE=$4 timeout 200 python tools/gap_probe3.py 2>&1 | grep -v amdgpu.ids; }
run h2d raw hip random
run h2d raw hip init
run randn raw hip random
run h2d batched hip random
run h2d raw torch random
run randn batched torch init
