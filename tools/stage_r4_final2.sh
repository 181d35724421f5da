#!/bin/bash
# Round 4, last GPU stage: profile summaries of the SHIPPED build (kernel trace of the bench command, per-section durations, FETCH_SIZE /
# WRITE_SIZE passes; the SQ counter sets were taken one build earlier, gpurun_out/prof of stage r4final), default and driver-style bench
# lines, then as much of the -m gpu suite as the remaining budget allows.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4final2; mkdir -p $O
( time timeout 600 profiles/collect.sh r4 quick ) > $O/collect.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
( time timeout 200 python bench.py --steps 20 --warmup 5 --no-extras ) > $O/bench_short.json 2> $O/bench_short.err
( time timeout 720 python -m pytest tests -m gpu -q -p no:cacheprovider -x ) > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log
tail -5 $O/tests.log; tail -3 $O/collect.log
