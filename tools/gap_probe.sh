#!/bin/bash
# Python-vs-harness gap (profiles/tuning_notes.md, "unresolved"): the same library call under (A) the C++ harness on the system ROCm
# runtime, (B) the C++ harness on the HIP / HSA runtime bundled with the torch wheel, (C) Python + torch as bench.py runs it,
# (D) Python + torch with the SYSTEM runtime preloaded, (E) Python without torch on the system runtime.
cd "$(dirname "$0")/.." || exit 1
TL=$(python -c "import os, importlib.util as u; print(os.path.join(os.path.dirname(u.find_spec('torch').origin), 'lib'))")
R=tools/ab_run; D=${1:-generated}
echo "== A: harness, system runtime"
timeout 120 $R kinematic 2 2 1 65536 2000 3 0 $D/libkinematic.so
echo "== B: harness, torch-bundled libamdhip64 + libhsa-runtime64 (LD_PRELOAD=$TL/libamdhip64.so)"
LD_PRELOAD=$TL/libamdhip64.so timeout 120 $R kinematic 2 2 1 65536 2000 3 0 $D/libkinematic.so
echo "== C: python + torch (bundled runtime)"
timeout 300 python tools/preload_probe.py
echo "== D: python + torch, system libhsa-runtime64 + libamdhip64 preloaded"
LD_PRELOAD="/opt/rocm/lib/libhsa-runtime64.so.1 /opt/rocm/lib/libamdhip64.so.7" timeout 300 python tools/preload_probe.py
echo "== D2: python + torch, system libhsa-runtime64 only preloaded"
LD_PRELOAD="/opt/rocm/lib/libhsa-runtime64.so.1" timeout 300 python tools/preload_probe.py
echo "== E: python without torch, system runtime"
timeout 300 python tools/notorch_probe.py
echo "== E2: python without torch, torch-bundled runtime"
RN_HIP_LIB=$TL/libamdhip64.so timeout 300 python tools/notorch_probe.py
