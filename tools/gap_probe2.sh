#!/bin/bash
# Second stage of the Python-vs-harness gap: round 4's first stage (gap_probe.sh) showed that neither the HIP nor the HSA runtime
# version matters -- a torch-free Python process is as fast as the C++ harness on either runtime -- so it is something `import torch`
# or its HIP context / allocator does.  Which step?
cd "$(dirname "$0")/.." || exit 1
for m in none import init tensors; do
  echo "== RN_PROBE_MODE=$m"
  RN_PROBE_MODE=$m RN_GEN=${1:-generated} timeout 300 python tools/notorch_probe.py 2>&1 | grep -v amdgpu.ids
done
echo "== tensors, PYTORCH_NO_HIP_MEMORY_CACHING=1"
PYTORCH_NO_HIP_MEMORY_CACHING=1 RN_PROBE_MODE=tensors RN_GEN=${1:-generated} timeout 300 python tools/notorch_probe.py 2>&1 | grep -v amdgpu.ids
echo "== env of a torch process that might matter"
python - <<'PY'
import os
before = dict(os.environ)
import torch
torch.zeros(1, device="cuda:0")
for k, v in sorted(os.environ.items()):
  if before.get(k) != v:
    print("  set by torch:", k, "=", v)
print("  HSA/HIP/ROC env:", {k: v for k, v in os.environ.items() if k.startswith(("HSA_", "HIP_", "ROC", "AMD_", "GPU_"))})
PY
