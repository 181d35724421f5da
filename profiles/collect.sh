#!/bin/bash
# Regenerates the profile summaries committed under profiles/ (run on the MI355X box through gpurun; raw rocpd databases stay
# in /tmp, only the text summaries are written to gpurun_out/prof/).  PMC counters are collected in passes of their own
# (--pmc with --kernel-trace only), as MI355X_MICROARCH.md prescribes.
#   usage: profiles/collect.sh [round-tag]        (default r2) -> gpurun_out/prof/<tag>_*.txt, <tag>_pmc_traffic.json
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
tag=${1:-r2}
out=gpurun_out/prof; mkdir -p $out
sum() { python profiles/summarize_rocpd.py "$@"; }
db() { find "$1" -name "*.db" | head -1; }

B6="python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-extras"
BL="python bench.py --model live --steps 420 --warmup 42 --no-cpu-baseline --no-extras"
BX="python bench.py --steps 100 --warmup 10 --no-cpu-baseline"

rocprofv3 --kernel-trace --stats -d /tmp/p_k6 -o r -- $B6 > /tmp/p_k6.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $B6"; sum stats "$(db /tmp/p_k6)"; } > $out/${tag}_kernel_trace_stats.txt
rocprofv3 --kernel-trace --stats -d /tmp/p_live -o r -- $BL > /tmp/p_live.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $BL"; sum stats "$(db /tmp/p_live)"; } > $out/${tag}_live_kernel_trace_stats.txt
rocprofv3 --kernel-trace --stats -d /tmp/p_x -o r -- $BX > /tmp/p_x.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $BX   (extras: live, live dt > 0, kinematic, kinematic6 1M, kinematic9, fused runs, MSCKF, config 4 = gate + trace + RTS at 16 384 x 2 100)"; sum stats "$(db /tmp/p_x)"; } > $out/${tag}_extras_kernel_trace_stats.txt
grep '^{' /tmp/p_x.log | tail -1 > $out/${tag}_bench_under_rocprof.json

P6="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras"
PL="python bench.py --model live --steps 105 --warmup 21 --no-cpu-baseline --no-extras"
{
  echo "# HBM traffic: separate passes, KiB per dispatch"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace -d /tmp/p_$c -o r -- $P6 > /tmp/p_$c.log 2>&1
    echo "#   rocprofv3 --pmc $c --kernel-trace -- $P6"
    sum pmc "$(db /tmp/p_$c)" k_step
    rocprofv3 --pmc $c --kernel-trace -d /tmp/pl_$c -o r -- $PL > /tmp/pl_$c.log 2>&1
    echo "#   rocprofv3 --pmc $c --kernel-trace -- $PL"
    sum pmc "$(db /tmp/pl_$c)" k_step
  done
  echo "# calibration in passes of the same kind: profiles/pmc_calibrate.py copies a 1 GiB fp64 tensor 5x (2^30 B read + 2^30 B written per copy)"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace -d /tmp/pc_$c -o r -- python profiles/pmc_calibrate.py > /tmp/pc_$c.log 2>&1
    sum pmc "$(db /tmp/pc_$c)" copyBuffer
  done
} > $out/${tag}_pmc_hbm_traffic.txt
python profiles/make_traffic_json.py $out/${tag}_pmc_hbm_traffic.txt $tag > $out/${tag}_pmc_traffic.json

# where the cycles of the live step kernel and of the smoother go
for c in SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64; do
  rocprofv3 --pmc $c --kernel-trace -d /tmp/ps_$c -o r -- python tools/rts_time.py > /tmp/ps_$c.log 2>&1
  sum pmc "$(db /tmp/ps_$c)" k_rts
done > $out/${tag}_sq_counters_smoother.txt 2>&1
# phase timeline of the fused run (needs the library built with the timeline knob: RN_TUNE=wide_timeline=1 RN_GEN_DIR=generated_tl)
if [ -f generated_tl/liblive.so ]; then
  { RN_TUNE=wide_timeline=1 RN_GEN_DIR=$PWD/generated_tl python tools/timeline.py run 8192;
    RN_TUNE=wide_timeline=1 RN_GEN_DIR=$PWD/generated_tl python tools/timeline.py run 8192 trace; } 2>&1 | grep -v amdgpu.ids > $out/${tag}_fused_run_timeline.txt
fi
ls -la $out
