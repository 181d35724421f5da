#!/bin/bash
# One gpurun call at the end of a round: a few targeted GPU tests, every profile summary (profiles/collect.sh, which also installs
# the fresh traffic json) and the default bench line, all into gpurun_out/<dir>/.  ~6.5 GPU-minutes.
#   usage: profiles/collect_and_bench.sh [out-dir-name] [round-tag]
cd "$(dirname "$0")/.." || exit 1
d=gpurun_out/${1:-final}
tag=${2:-r3}
mkdir -p "$d"
( time timeout 200 python -m pytest tests/test_gpu_run_blk.py tests/test_gpu_run.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider ) > "$d/pytest.log" 2>&1
tail -5 "$d/pytest.log"
( time timeout 420 profiles/collect.sh "$tag" ) > "$d/collect.log" 2>&1
tail -3 "$d/collect.log"
( time timeout 240 python bench.py ) > "$d/bench.json" 2> "$d/bench.err"
tail -c 600 "$d/bench.err"
# the driver-style short run (what `--steps 20 --warmup 5` reports: frac from the HIP events, frac_wall from the host's clock)
( time timeout 120 python bench.py --steps 20 --warmup 5 --no-extras ) > "$d/bench_short.json" 2> "$d/bench_short.err"
python - "$d/bench.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(d['value'], d['roofline']['frac'], d['roofline']['traffic'])
for k, v in d['extra'].items():
  r = v.get('roofline') or {}
  print(k, v.get('value'), r.get('bound'), r.get('frac'), r.get('traffic'), v.get('forward_steps_per_s'), v.get('backward_steps_per_s'))
PY
