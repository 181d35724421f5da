#!/bin/bash
# SQ counters of a development build of k_rts4 (tools/rts4_time.py on one directory), two passes of <= 8 counters
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5d; mkdir -p $O
D=${1:-gen_ab/rts4_v2}
SQA="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
SQB="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"
i=0
for set in "$SQA" "$SQB"; do
  i=$((i + 1))
  rm -rf /tmp/p_sq$i
  RTS4_T=100 rocprofv3 --pmc $set --kernel-trace -d /tmp/p_sq$i -o r -- python tools/rts4_time.py $D > /tmp/p_sq$i.log 2>&1
  { echo "# rocprofv3 --pmc $set --kernel-trace -- RTS4_T=100 python tools/rts4_time.py $D"; python profiles/summarize_rocpd.py pmc "$(find /tmp/p_sq$i -name '*.db' | head -1)" k_rts4; } | tee $O/sq_set$i.txt
done
