#!/bin/bash
# First GPU call of the next round: the whole -m gpu suite on the build round 4 ended with (its last changes -- sin / cos lowering,
# the 24-state model's fused run, bench.py's warm-up order -- ran only in parts there: profiles/r4_late_measurements.txt), then the profile
# passes (kernel trace, FETCH_SIZE / WRITE_SIZE: refreshes profiles/pmc_traffic.json for the live libraries, whose records are carried) and
# the default and driver-style bench lines.    /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/stage_r5_first.sh'
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r5a; mkdir -p $O
( time timeout 780 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log
( time timeout 400 profiles/collect.sh r5 quick ) > $O/collect.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null
( time timeout 300 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
( time timeout 100 python bench.py --steps 20 --warmup 5 --no-extras ) > $O/bench_short.json 2> $O/bench_short.err
[ -x tools/issue_probe ] || hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/issue_probe.hip -o tools/issue_probe > /dev/null 2>&1
timeout 60 tools/issue_probe > $O/issue_probe.txt 2>&1
tail -5 $O/tests.log; tail -3 $O/collect.log; cut -c1-400 $O/bench_short.json
