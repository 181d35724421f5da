#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5c; mkdir -p $O
timeout 300 python tools/rts4_time.py $(ls -d gen_ab/rts4_* | sort) 2>&1 | grep -v amdgpu.ids | tee $O/rts4_time.txt
