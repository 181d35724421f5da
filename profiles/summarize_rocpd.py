#!/usr/bin/env python3
"""Turn rocprofv3 rocpd SQLite output (gpurun_out/...) into the compact text summaries committed under profiles/.

  python profiles/summarize_rocpd.py stats <results.db>            -> per-kernel calls / total / avg / min / max (us)
  python profiles/summarize_rocpd.py pmc   <results.db> [substr]   -> per-kernel mean counter value
"""
import re
import sqlite3
import sys


def short(name, n=90):
  name = re.sub(r"\(anonymous namespace\)::", "", name)
  name = re.sub(r"^void ", "", name)
  return name if len(name) <= n else name[:n - 3] + "..."


def stats(db):
  cur = sqlite3.connect(db).cursor()
  rows = list(cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc"))
  tot = sum(r[2] for r in rows)
  print(f"{'kernel':92s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
  for name, calls, total, avg, mn, mx in rows[:15]:
    print(f"{short(name):92s} {calls:6d} {total / 1e3:10.1f} {avg / 1e3:9.3f} {mn / 1e3:9.3f} {mx / 1e3:9.3f} {100 * total / tot:6.2f}")


def pmc(db, sub=""):
  cur = sqlite3.connect(db).cursor()
  rows = list(cur.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) from counters_collection "
                          "group by kernel_name, counter_name order by sum(duration) desc"))
  print(f"{'kernel':92s} {'counter':>12s} {'calls':>6s} {'mean':>14s} {'min':>14s} {'max':>14s} {'avg_us':>9s}")
  for name, ctr, calls, avg, mn, mx, dur in rows:
    if sub and sub not in name:
      continue
    print(f"{short(name):92s} {ctr:>12s} {calls:6d} {avg:14.2f} {mn:14.2f} {mx:14.2f} {dur / 1e3:9.3f}")


if __name__ == "__main__":
  {"stats": stats, "pmc": pmc}[sys.argv[1]](*sys.argv[2:])
