#!/usr/bin/env python3
"""Join the manifest of profiles/pmc_workload.py (its stdout) with the rocpd database of the same rocprofv3 pass.

  python profiles/summarize_sections.py <workload.log> <results.db>  -> one line per (section, kernel, counter | duration)

Product kernels (k_step_* / k_predict / k_run / k_rts*) are assigned to sections by dispatch order: every SECTION line of
the manifest says how many warm-up product dispatches the section issued before its n measured ones.
Counter values are means per dispatch; durations come from the same pass (under --pmc they are inflated: use the
kernel-trace pass for time)."""
import json
import re
import sqlite3
import sys

PRODUCT = re.compile(r"\bk_(step|predict|run|rts|maha)")


def load(log, db):
  sections = []
  calib = None
  with open(log, encoding="utf-8") as f:
    for line in f:
      p = line.split()
      if p[:1] == ["SECTION"]:
        sections.append(dict(label=p[1], lib=p[2], n=int(p[3]), bytes=float(p[4]), steps=int(p[5]), warmup=int(p[6])))
      if p[:1] == ["CALIBRATION"]:
        calib = dict(n=int(p[2]), bytes=int(p[3]))
  cur = sqlite3.connect(db).cursor()
  disp = {}
  has_pmc = bool(list(cur.execute("select count(*) from rocpd_pmc_event"))[0][0])
  for did, name, dur, grid, vg, ag, lds in cur.execute("select dispatch_id, name, duration, grid_x, vgpr_count, accum_vgpr_count, lds_size from kernels order by dispatch_id"):
    disp[did] = dict(name=name, dur=dur, grid=grid, vgpr=vg, agpr=ag, lds=lds, ctr={})
  if has_pmc:
    for did, cname, val in cur.execute("select dispatch_id, counter_name, value from counters_collection"):
      if did in disp:
        disp[did]["ctr"][cname] = val
  return sections, calib, disp


def main(log, db, as_json=False):
  sections, calib, disp = load(log, db)
  prod = [d for _, d in sorted(disp.items()) if PRODUCT.search(d["name"])]
  pos, rows = 0, []
  for s in sections:
    tot = s["warmup"] + s["n"]
    mine = prod[pos + s["warmup"]: pos + tot]
    pos += tot
    by_kernel = {}
    for d in mine:
      by_kernel.setdefault(re.sub(r"\(anonymous namespace\)::|^void ", "", d["name"]).split("(")[0], []).append(d)
    rec = dict(label=s["label"], lib=s["lib"], dispatches=len(mine), algorithmic_bytes_per_dispatch=s["bytes"], filter_steps_per_dispatch=s["steps"],
               avg_us=sum(d["dur"] for d in mine) / max(1, len(mine)) / 1e3, kernels={})
    names = set()
    for d in mine:
      names.update(d["ctr"])
    rec["counters"] = {c: sum(d["ctr"].get(c, 0.0) for d in mine) / max(1, len(mine)) for c in sorted(names)}
    for k, ds in by_kernel.items():
      rec["kernels"][k] = dict(calls=len(ds), avg_us=sum(d["dur"] for d in ds) / len(ds) / 1e3, grid=ds[0]["grid"], vgpr=ds[0]["vgpr"], agpr=ds[0]["agpr"],
                               lds=ds[0]["lds"], counters={c: sum(d["ctr"].get(c, 0.0) for d in ds) / len(ds) for c in sorted(names)})
    rows.append(rec)
  assert pos == len(prod), f"manifest accounts for {pos} product dispatches, the database holds {len(prod)}"
  cal = None
  if calib:
    copies = [d for _, d in sorted(disp.items()) if not PRODUCT.search(d["name"]) and d["grid"] >= 1024 and d["dur"] > 100e3][-calib["n"]:]
    names = set()
    for d in copies:
      names.update(d["ctr"])
    cal = dict(kernel=copies[0]["name"][:60] if copies else None, bytes_read=calib["bytes"], bytes_written=calib["bytes"], avg_us=sum(d["dur"] for d in copies) / max(1, len(copies)) / 1e3,
               counters={c: sum(d["ctr"].get(c, 0.0) for d in copies) / max(1, len(copies)) for c in sorted(names)})
  if as_json:
    print(json.dumps(dict(sections=rows, calibration=cal), indent=1))
    return
  for r in rows:
    print(f"== {r['label']}  lib{r['lib']}.so  {r['dispatches']} dispatches  avg {r['avg_us']:.3f} us  algorithmic {r['algorithmic_bytes_per_dispatch'] / 1e6:.3f} MB/dispatch")
    for k, kr in r["kernels"].items():
      print(f"   {k:40s} calls {kr['calls']:3d} avg_us {kr['avg_us']:12.3f} grid {kr['grid']:7d} vgpr {kr['vgpr']:3d}+{kr['agpr']:3d} lds {kr['lds']:6d}  " +
            "  ".join(f"{c}={v:.6g}" for c, v in kr["counters"].items()))
  if cal:
    print(f"== calibration: {cal['kernel']} x{calib['n']}  2^30 B read + 2^30 B written per copy  avg {cal['avg_us']:.1f} us  " +
          "  ".join(f"{c}={v:.6g}" for c, v in cal["counters"].items()))


if __name__ == "__main__":
  main(sys.argv[1], sys.argv[2], as_json=len(sys.argv) > 3 and sys.argv[3] == "json")
