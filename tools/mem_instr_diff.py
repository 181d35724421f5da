"""Do two builds of a filter library issue the same vector-memory instructions?  Per kernel, the multiset of global_* / buffer_* / flat_* /
scratch_* mnemonics of both device code objects (llvm-objdump).  Used before a record of profiles/pmc_traffic.json is carried over to a
later build whose change was arithmetic only (`carried_to` / `carried_note`, read by bench.measured_traffic).
  python tools/mem_instr_diff.py <old lib.so> <new lib.so>"""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def mix(lib):
  with tempfile.TemporaryDirectory() as d:
    fb, co = os.path.join(d, "a.hipfb"), os.path.join(d, "a.co")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fb], check=True, capture_output=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--output={co}"], check=True, capture_output=True)
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
  out, cur = {}, None
  for line in dis.split("\n"):
    m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
    if m:
      cur = m.group(1)
      out[cur] = collections.Counter()
      continue
    m = re.match(r"^\s+((?:global|buffer|flat|scratch)_[a-z_0-9]+)", line)
    if cur and m:
      out[cur][m.group(1)] += 1
  return out


if __name__ == "__main__":
  a, b = mix(sys.argv[1]), mix(sys.argv[2])
  diff = [k for k in sorted(set(a) | set(b)) if a.get(k) != b.get(k)]
  print(f"{len(a)} / {len(b)} kernels, {len(diff)} with a different vector-memory instruction mix")
  for k in diff:
    print("  ", k, dict(a.get(k, {})), "->", dict(b.get(k, {})))
  sys.exit(1 if diff else 0)
