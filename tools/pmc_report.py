#!/usr/bin/env python
# coding=utf-8
"""Per-kernel PMC summary from rocprofv3 rocpd databases (one database per
`--pmc` pass; FETCH_SIZE and WRITE_SIZE need separate passes on gfx950).

  pmc_report.py KERNEL_SUBSTR out.json db1 [db2 ...]

FETCH_SIZE / WRITE_SIZE are in KiB.  MI355X_MICROARCH.md (HBM section): on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming
read, so the read side is quoted raw AND doubled (upper bound)."""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiverse_amd import buildinfo  # noqa: E402


def main(kernel, out, dbs):
  agg = {}
  for db in dbs:
    c = sqlite3.connect(db)
    rows = c.execute(
        "select counter_name, sum(value), count(*), sum(duration) from "
        "counters_collection where kernel_name like ? group by counter_name",
        ("%" + kernel + "%",)).fetchall()
    for name, total, n, dur in rows:
      agg[name] = {"total": total, "launches": n, "per_launch": total / n,
                   "avg_launch_us": dur / n / 1e3}
  # the sources the profiled library was built from: bench.py quotes roofline.traffic from
  # this file only while they are the sources of the tree it runs in
  res = {"kernel": kernel, "counters": agg,
         "kernel_source_sha16": buildinfo.kernel_source_hash()}
  if "GRBM_GUI_ACTIVE" in agg:
    gui = agg["GRBM_GUI_ACTIVE"]
    cyc = gui["total"] / 8.0                     # summed over the 8 XCDs
    # GRBM_GUI_ACTIVE also counts the front-end's launch / drain cycles around a dispatch:
    # for launches shorter than ~50 us the quotient exceeds the 2.4 GHz maximum clock, so
    # the field is only reported where it means something
    if gui["avg_launch_us"] >= 50.0:
      res["effective_clock_GHz"] = cyc / (gui["avg_launch_us"] * 1e3 * gui["launches"])
    else:
      res["effective_clock_note"] = "launches < 50 us: GRBM_GUI_ACTIVE / duration is not a clock"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in agg:
      res["mfma_busy_frac"] = agg["SQ_VALU_MFMA_BUSY_CYCLES"]["total"] / (cyc * 1024)
    if "SQ_WAVE_CYCLES" in agg:
      wc = agg["SQ_WAVE_CYCLES"]["total"]
      res["waves_per_simd_avg"] = wc * 4 / cyc / 1024
      for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if k in agg:
          res[k + "_per_wave_cycle"] = agg[k]["total"] / wc
  if "FETCH_SIZE" in agg and "WRITE_SIZE" in agg:
    f = agg["FETCH_SIZE"]["per_launch"] * 1024.0
    w = agg["WRITE_SIZE"]["per_launch"] * 1024.0
    res["hbm_bytes_per_launch"] = {"fetch_raw": f, "fetch_x2_gfx950_correction": 2 * f,
                                   "write": w, "total_raw": f + w,
                                   "total_corrected": 2 * f + w}
  with open(out, "w") as fo:
    json.dump(res, fo, indent=1, sort_keys=True)
  print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
  main(sys.argv[1], sys.argv[2], sys.argv[3:])
