#!/bin/bash
# Regenerates the profile summaries committed under profiles/ (run on the MI355X box through gpurun; raw rocpd databases stay
# in /tmp, only the text summaries are written to gpurun_out/prof/).  PMC counters are collected in passes of their own
# (--pmc with --kernel-trace only), as MI355X_MICROARCH.md prescribes.
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out/prof; mkdir -p $out
sum() { python profiles/summarize_rocpd.py "$@"; }
db() { find "$1" -name "*.db" | head -1; }

B6="python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-extras"
BL="python bench.py --model live --steps 420 --warmup 42 --no-cpu-baseline --no-extras"
BX="python bench.py --steps 100 --warmup 10 --no-cpu-baseline"

rocprofv3 --kernel-trace --stats -d /tmp/p_k6 -o r -- $B6 > /tmp/p_k6.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $B6"; sum stats "$(db /tmp/p_k6)"; } > $out/kernel_trace_stats.txt
rocprofv3 --kernel-trace --stats -d /tmp/p_live -o r -- $BL > /tmp/p_live.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $BL"; sum stats "$(db /tmp/p_live)"; } > $out/live_kernel_trace_stats.txt
rocprofv3 --kernel-trace --stats -d /tmp/p_x -o r -- $BX > /tmp/p_x.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $BX   (extras: live, kinematic, kinematic6 1M, kinematic9, fused run, MSCKF, gate + trace + RTS)"; sum stats "$(db /tmp/p_x)"; } > $out/extras_kernel_trace_stats.txt

P6="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras"
{
  echo "# HBM traffic of k_step_1<true> (kinematic6, batch 65536): separate passes, KiB per dispatch"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace -d /tmp/p_$c -o r -- $P6 > /tmp/p_$c.log 2>&1
    echo "#   rocprofv3 --pmc $c --kernel-trace -- $P6"
    sum pmc "$(db /tmp/p_$c)" k_step
  done
  echo "# calibration in passes of the same kind: profiles/pmc_calibrate.py copies a 1 GiB fp64 tensor 5x (2^30 B read + 2^30 B written per copy)"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace -d /tmp/pc_$c -o r -- python profiles/pmc_calibrate.py > /tmp/pc_$c.log 2>&1
    sum pmc "$(db /tmp/pc_$c)" copyBuffer
  done
} > $out/pmc_hbm_traffic.txt
ls -la $out
