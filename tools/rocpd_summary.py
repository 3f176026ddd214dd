#!/usr/bin/env python
# coding=utf-8
"""Turn a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME`
writes DIR/NAME_results.db on ROCm 7.2) into the per-kernel stats table that is
committed under profiles/.  Usage: rocpd_summary.py results.db > profiles/xxx.md"""
import sqlite3
import sys


def main(path):
  c = sqlite3.connect(path)
  rows = c.execute(
      "select name, count(*), sum(duration), avg(duration), min(duration), "
      "max(duration), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), "
      "max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name "
      "order by sum(duration) desc").fetchall()
  total = sum(r[2] for r in rows) or 1
  print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | sgpr | lds B | max grid | wg |")
  print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
  for r in rows:
    name = r[0].split("(")[0]
    print("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.2f | %s | %s | %s | %s | %s | %s |" % (
        name, r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3,
        100.0 * r[2] / total, r[6], r[7], r[8], r[9], r[10], r[11]))


if __name__ == "__main__":
  main(sys.argv[1])
