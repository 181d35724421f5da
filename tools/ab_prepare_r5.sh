#!/bin/bash
# CPU side of the first A/B of the next round: variant builds of live_maha for the two knobs prepared at the end of round 4
# (more independent fp64 chains for a lone wavefront: profiles/r4_issue_probe.txt).  ~8 minutes, four hipcc runs in parallel.
cd "$(dirname "$0")/.." || exit 1
for v in run_jb=6 run_jb=8 rts3_np=6 rts3_np=8; do
  d=gen_ab/$(echo $v | tr '=' '_')
  RN_GEN_DIR=$PWD/$d RN_TUNE=$v python -c "
from examples import ensure_generated
import os
ensure_generated(['live_maha'], folder=os.environ['RN_GEN_DIR'])" > /tmp/ab_$(echo $v | tr '=' '_').log 2>&1 &
done
wait
for d in gen_ab/run_jb_* gen_ab/rts3_np_*; do echo "== $d"; grep -E "^k_run |^k_rts3" $d/live_maha.kernels.txt; done
