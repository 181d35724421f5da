#!/bin/bash
# Round 4, final GPU stage: every profile summary (profiles/collect.sh r4: kernel trace of the bench command, per-section durations,
# FETCH_SIZE / WRITE_SIZE passes, two SQ counter sets; installs profiles/pmc_traffic.json), the default and the driver-style short bench
# lines, then the whole -m gpu suite.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r4final; mkdir -p $O
( time timeout 900 profiles/collect.sh r4 ) > $O/collect.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json 2>/dev/null
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
( time timeout 300 python bench.py --steps 20 --warmup 5 ) > $O/bench_short.json 2> $O/bench_short.err
( time timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 ) > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log
tail -5 $O/tests.log; tail -3 $O/collect.log; ls gpurun_out/prof | head -30
