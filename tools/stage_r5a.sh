#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5a; mkdir -p $O
timeout 120 tools/dpp_probe > $O/dpp_probe.txt 2>&1
cat $O/dpp_probe.txt
timeout 500 bash tools/ab_stage_r5.sh
cp gpurun_out/r5ab/config4_ab.txt $O/ 2>/dev/null
