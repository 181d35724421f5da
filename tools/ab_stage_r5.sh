#!/bin/bash
# GPU side: one config-4 chunk (8 192 filters x 2 100 steps: forward with trace, backward) per variant built by tools/ab_prepare_r5.sh,
# default build first and last (gen_ab/ travels with the push).   ~20 s per variant.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5ab; mkdir -p $O
for d in "" gen_ab/run_jb_6 gen_ab/run_jb_8 gen_ab/rts3_np_6 gen_ab/rts3_np_8 ""; do
  if [ -z "$d" ]; then timeout 90 python tools/config4_time.py 2>/dev/null | tail -n 1
  else RN_NO_GEN=1 RN_GEN_DIR=$PWD/$d RN_TUNE=$(basename $d | sed 's/_\([0-9]*\)$/=\1/') timeout 90 python tools/config4_time.py 2>/dev/null | tail -n 1; fi
done | tee $O/config4_ab.txt
