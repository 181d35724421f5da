#!/usr/bin/env python3
"""<tag>_pmc_FETCH_SIZE.json + <tag>_pmc_WRITE_SIZE.json (profiles/collect.sh, one section per timed kernel configuration)
-> the JSON bench.py reads for `roofline.traffic`, keyed by section label.

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB per dispatch -> bytes): FETCH_SIZE under-reports wide coalesced
reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section; the 1 GiB calibration copy of the same pass is recorded next to
every entry), WRITE_SIZE is taken as is.  Each entry carries the digest of the library it was measured on
({generated}/{name}.digest): bench.py prints a traffic figure only for the library with that digest."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(fetch_json, write_json, tag):
  with open(fetch_json, encoding="utf-8") as f:
    fe = json.load(f)
  with open(write_json, encoding="utf-8") as f:
    wr = json.load(f)
  gen = os.environ.get("RN_GEN_DIR") or os.path.join(REPO, "generated")
  calib = {}
  for name, rec in (("FETCH_SIZE", fe), ("WRITE_SIZE", wr)):
    c = rec.get("calibration") or {}
    if c.get("counters", {}).get(name):
      calib[name] = c["counters"][name] * 1024.0 / c["bytes_read"]        # counter bytes / true bytes (expected 0.5 and 1.0)
  wsec = {s["label"]: s for s in wr["sections"]}
  out = {}
  for s in fe["sections"]:
    w = wsec.get(s["label"])
    if w is None or "FETCH_SIZE" not in s["counters"] or "WRITE_SIZE" not in w["counters"]:
      continue
    fetch, write = s["counters"]["FETCH_SIZE"], w["counters"]["WRITE_SIZE"]
    dg = os.path.join(gen, f"{s['lib']}.digest")
    digest = open(dg, encoding="utf-8").read().strip() if os.path.exists(dg) else None
    hbm = (2.0 * fetch + write) * 1024.0
    out[s["label"]] = {"hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": s["algorithmic_bytes_per_dispatch"],
                       "traffic_ratio": hbm / s["algorithmic_bytes_per_dispatch"], "fetch_size_kib_raw": fetch, "fetch_correction": 2.0,
                       "write_size_kib_raw": write, "write_correction": 1.0, "kernels": sorted(s["kernels"]), "dispatches": s["dispatches"],
                       "calibration_copy_ratio": calib, "source": f"profiles/{tag}_pmc_FETCH_SIZE.txt, profiles/{tag}_pmc_WRITE_SIZE.txt",
                       "lib": s["lib"], "lib_digest": digest}
  print(json.dumps(out, indent=1))


if __name__ == "__main__":
  main(*sys.argv[1:4])
