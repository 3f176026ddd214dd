#!/usr/bin/env python
"""Markdown table of an ablation session: ablate_table.py <dir> <variant...>"""
import json
import os
import sys

d, names = sys.argv[1], sys.argv[2:]
print("| variant | traj/s | launch ms (hipEvent) | executed MFMA TF/s | MFMA busy | clock GHz "
      "| busy x clock | waves/SIMD | wait_any | wait_inst | active |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for v in names:
  b, p = {}, {}
  try:
    b = json.loads(open(os.path.join(d, v + ".bench.json")).read().strip().split("\n")[-1])
  except Exception:  # pylint: disable=broad-except
    pass
  try:
    p = json.load(open(os.path.join(d, v + ".pmc.json")))
  except Exception:  # pylint: disable=broad-except
    pass
  r = b.get("roofline", {})
  f = lambda x, n=3: ("%.*f" % (n, x)) if isinstance(x, (int, float)) else "-"
  busy, clk = p.get("mfma_busy_frac"), p.get("effective_clock_GHz")
  print("| %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (
      v, f(b.get("value"), 1), f(r.get("avg_launch_ms"), 4), f(r.get("executed_mfma_TFLOPs"), 1),
      f(busy), f(clk), f(busy * clk if busy and clk else None), f(p.get("waves_per_simd_avg"), 2),
      f(p.get("SQ_WAIT_ANY_per_wave_cycle")), f(p.get("SQ_WAIT_INST_ANY_per_wave_cycle")),
      f(p.get("SQ_ACTIVE_INST_ANY_per_wave_cycle"))))
