#!/bin/bash
# Regenerates the profile summaries committed under profiles/ (run on the MI355X box through gpurun; raw rocpd databases stay
# in /tmp, only text / JSON summaries are written to gpurun_out/prof/).  PMC counters are collected in passes of their own
# (--pmc with --kernel-trace only), as MI355X_MICROARCH.md prescribes; FETCH_SIZE and WRITE_SIZE in separate passes.
# Every pass but the first runs profiles/pmc_workload.py: every kernel configuration bench.py times, in labelled sections,
# config 4 as the 8 192 x 2 100 chunk the bench launches -- so each timed kernel gets the same counter set.
#   usage: profiles/collect.sh [round-tag] [quick]   (default r3) -> gpurun_out/prof/<tag>_*.txt, <tag>_pmc_traffic.json
#          quick: skip the SQ passes (HBM traffic + times only)
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
tag=${1:-r3}
quick=${2:-}
out=gpurun_out/prof; mkdir -p $out
db() { find "$1" -name "*.db" | head -1; }
W="python profiles/pmc_workload.py"

# 1. the bench's own default command under the kernel trace (the average k_step_1<true> duration the bench line must agree with)
B6="python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d /tmp/p_k6 -o r -- $B6 > /tmp/p_k6.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $B6"; python profiles/summarize_rocpd.py stats "$(db /tmp/p_k6)"; grep '^{' /tmp/p_k6.log | tail -1; } > $out/${tag}_kernel_trace_stats.txt

# 2. every timed configuration: durations (no counters)
rocprofv3 --kernel-trace --stats -d /tmp/p_w -o r -- $W > /tmp/p_w.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $W    (per section: the launches bench.py times; durations without counters)";
  python profiles/summarize_sections.py /tmp/p_w.log "$(db /tmp/p_w)"; } > $out/${tag}_sections_kernel_trace.txt 2>&1
python profiles/summarize_sections.py /tmp/p_w.log "$(db /tmp/p_w)" json > $out/${tag}_sections_kernel_trace.json 2>/dev/null

# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in passes of their own (KiB per dispatch), calibration copy in the same pass
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d /tmp/p_$c -o r -- $W > /tmp/p_$c.log 2>&1
  { echo "# rocprofv3 --pmc $c --kernel-trace -- $W"; python profiles/summarize_sections.py /tmp/p_$c.log "$(db /tmp/p_$c)"; } > $out/${tag}_pmc_$c.txt 2>&1
  python profiles/summarize_sections.py /tmp/p_$c.log "$(db /tmp/p_$c)" json > $out/${tag}_pmc_$c.json 2>/dev/null
done
python profiles/make_traffic_json.py $out/${tag}_pmc_FETCH_SIZE.json $out/${tag}_pmc_WRITE_SIZE.json $tag > $out/${tag}_pmc_traffic.json
# bench.py reads profiles/pmc_traffic.json (and prints a figure only for the library digests recorded there): a bench run that follows in
# the same call sees this measurement; the copy under gpurun_out/ is what gets committed
[ -s $out/${tag}_pmc_traffic.json ] && cp $out/${tag}_pmc_traffic.json profiles/pmc_traffic.json

# 4. one uniform SQ set for every kernel, two passes of <= 8 SQ counters
if [ -z "$quick" ]; then
  SQA="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
  SQB="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM"
  i=0
  for set in "$SQA" "$SQB"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --kernel-trace -d /tmp/p_sq$i -o r -- $W > /tmp/p_sq$i.log 2>&1
    { echo "# rocprofv3 --pmc $set --kernel-trace -- $W"; python profiles/summarize_sections.py /tmp/p_sq$i.log "$(db /tmp/p_sq$i)"; } > $out/${tag}_sq_counters_set$i.txt 2>&1
  done
fi
tail -2 /tmp/p_w.log /tmp/p_FETCH_SIZE.log /tmp/p_WRITE_SIZE.log
ls -la $out
