#!/usr/bin/env python3
"""profiles/<tag>_pmc_hbm_traffic.txt (profiles/collect.sh) -> the JSON bench.py reads for `roofline.traffic`.

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB per dispatch -> bytes): FETCH_SIZE under-reports wide coalesced
reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section; the calibration copy in the same file confirms it per run),
WRITE_SIZE is taken as is.  Each entry carries the digest of the library it was measured on ({generated}/{name}.digest):
bench.py prints a traffic figure only for the library with that digest."""
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(txt, tag):
  rows = {}
  model = None
  with open(txt, encoding="utf-8") as f:
    for line in f:
      m = re.search(r"rocprofv3 --pmc (\w+) .*bench\.py(?: --model (\w+))?", line)
      if m:
        model = m.group(2) or "kinematic6"
        continue
      m = re.match(r"(k_step_\S+)\(.*?\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([\d.]+)", line)
      if m and model:
        rows.setdefault((model, m.group(1)), {})[m.group(2)] = (int(m.group(3)), float(m.group(4)))
  calib = {}
  with open(txt, encoding="utf-8") as f:
    for line in f:
      m = re.match(r"__amd_rocclr_copyBuffer\s+(FETCH_SIZE|WRITE_SIZE)\s+\d+\s+([\d.]+)", line)
      if m:
        calib[m.group(1)] = float(m.group(2)) / (2**30 / 1024)
  out = {}
  for model, n in (("kinematic6", 65536), ("live", 16384)):
    ks = {k: v for (mdl, k), v in rows.items() if mdl == model and "FETCH_SIZE" in v and "WRITE_SIZE" in v and "<true>" in k}
    if not ks:
      continue
    calls = sum(v["FETCH_SIZE"][0] for v in ks.values())
    fetch = sum(v["FETCH_SIZE"][0] * v["FETCH_SIZE"][1] for v in ks.values()) / calls
    write = sum(v["WRITE_SIZE"][0] * v["WRITE_SIZE"][1] for v in ks.values()) / sum(v["WRITE_SIZE"][0] for v in ks.values())
    dg = os.path.join(REPO, "generated", f"{model}.digest")
    digest = open(dg, encoding="utf-8").read().strip() if os.path.exists(dg) else None
    out[f"{model}_b{n}"] = {"hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0, "fetch_size_kib_raw": fetch, "fetch_correction": 2.0,
                           "write_size_kib_raw": write, "write_correction": 1.0, "kernels": sorted(ks), "calibration_copy_ratio": calib,
                           "source": f"profiles/{tag}_pmc_hbm_traffic.txt", "lib_digest": digest}
  print(json.dumps(out, indent=1))


if __name__ == "__main__":
  main(sys.argv[1], sys.argv[2])
