cd /root/repo
mkdir -p gpurun_out/r3i
( time timeout 200 python -m pytest tests/test_gpu_run_blk.py tests/test_gpu_run.py tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider ) > gpurun_out/r3i/pytest.log 2>&1
tail -5 gpurun_out/r3i/pytest.log
( time timeout 330 profiles/collect.sh r3 ) > gpurun_out/r3i/collect.log 2>&1
tail -3 gpurun_out/r3i/collect.log
( time timeout 170 python bench.py ) > gpurun_out/r3i/bench.json 2> gpurun_out/r3i/bench.err
tail -c 600 gpurun_out/r3i/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3i/bench.json') if l.startswith('{')][-1])
print(d['value'], d['roofline']['frac'], d['roofline']['traffic'])
for k,v in d['extra'].items():
    r=v.get('roofline') or {}
    print(k, v.get('value'), r.get('bound'), r.get('frac'), r.get('traffic'), v.get('forward_steps_per_s'), v.get('backward_steps_per_s'))
PY
