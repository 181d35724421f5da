#!/bin/bash
# SQ / instruction-cache counters of the fused multi-step run (k_run, live, 8 192 filters x 252 steps), one --pmc pass each.
#   usage: profiles/collect_run_counters.sh [round-tag]   -> gpurun_out/prof/<tag>_sq_counters_fused_run.txt
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
tag=${1:-r2}
out=gpurun_out/prof; mkdir -p $out
db() { find "$1" -name "*.db" | head -1; }
for c in SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_MISSES; do
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pr_$c -o r -- python tools/run_time.py > /tmp/pr_$c.log 2>&1
  python profiles/summarize_rocpd.py pmc "$(db /tmp/pr_$c)" k_run
done > $out/${tag}_sq_counters_fused_run.txt 2>&1
tail -3 /tmp/pr_SQ_WAVES.log >> $out/${tag}_sq_counters_fused_run.txt
cat $out/${tag}_sq_counters_fused_run.txt
