#!/bin/bash
# CPU side of the first A/B of the next round: builds the variant libraries of the knobs prepared at the end of round 3 (all verified
# against the oracle in the host emulation, none timed yet) into gen_ab/<variant>/, next to a copy of the default builds.  Then
# `gpurun -- tools/ab_stage_r4.sh` (well under a GPU-minute).
cd "$(dirname "$0")/.." || exit 1
set -e
/opt/rocm/bin/hipcc -O2 -std=c++17 tools/ab_run.cpp -o tools/ab_run -ldl
/opt/rocm/bin/hipcc -O2 -std=c++17 tools/ab_step.cpp -o tools/ab_step -ldl
gen() { RN_TUNE="$1" RN_GEN_DIR="gen_ab/$2" python -c "from examples import ensure_generated; ensure_generated($3)"; }
python -c "from examples import ensure_generated; ensure_generated(['kinematic', 'kinematic6', 'live'])"
mkdir -p gen_ab/default gen_ab/in
cp generated/libkinematic.so generated/libkinematic6.so generated/liblive.so gen_ab/default/
gen small_sym=1 sym "['kinematic', 'kinematic6']"
gen run_block_trace=1 bt "['kinematic', 'kinematic6']"
gen small_sym=1,run_block_trace=1 symbt "['kinematic6']"
gen wide_lean_coef=1 lc "['live']"
gen wide_lean_coef=1,wide_lean_sym=1,wide_lean_unroll=22 ls "['live']"
for k in 4 10 12; do python tools/ab_inputs.py $k gen_ab/in/live$k.bin; done
grep -H "k_step_1<true>\|k_run\|k_step_4<true>" gen_ab/*/kinematic6.kernels.txt gen_ab/*/live.kernels.txt
