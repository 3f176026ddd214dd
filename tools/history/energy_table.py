#!/usr/bin/env python
"""Markdown table of tools/energy_attribution.sh: energy_table.py <dir> <name:c:a ...>.

Per row: traj/s, ms per grouped gate launch (hipEvents), sclk / package power while running
(median of the upper half of the rocm-smi samples: the ramp in and out is dropped), energy per
launch = (power - idle power) x launch time (dynamic) and power x launch time (package), and the
difference to the row above."""
import json
import os
import re
import sys


def smi(path):
  try:
    txt = open(path).read()
  except OSError:
    return None, None, 0
  sclk = [int(x) for x in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)]
  pw = [float(x) for x in re.findall(r"Power \(W\): ([0-9.]+)", txt)]
  med_hi = lambda a: (sorted(a)[len(a) // 2:][len(a[len(a) // 2:]) // 2]) if a else None
  return med_hi(sclk), med_hi(pw), len(pw)


def main():
  d, rows = sys.argv[1], sys.argv[2:]
  idle_txt = os.path.join(d, "smi_idle.log")
  pw_idle = None
  if os.path.exists(idle_txt):
    pw = [float(x) for x in re.findall(r"Power \(W\): ([0-9.]+)", open(idle_txt).read())]
    pw_idle = sorted(pw)[len(pw) // 2] if pw else None
  print("idle package power: %s W\n" % pw_idle)
  print("| row | ABLC build | MV_WINO_ABL | traj/s | ms / launch | sclk MHz | package W | "
        "package mJ / launch | dynamic mJ / launch | delta dynamic mJ | samples |")
  print("|---|---|---|---|---|---|---|---|---|---|---|")
  prev = None
  for row in rows:
    name, c, a = row.split(":")[:3]
    if name == "idle":
      continue
    try:
      b = json.loads(open(os.path.join(d, name + ".json")).read().strip().split("\n")[-1])
    except Exception:  # pylint: disable=broad-except
      print("| %s | %s | %s | run failed |" % (name, c, a))
      continue
    ms = b["roofline"].get("avg_launch_ms")
    sclk, pw, n = smi(os.path.join(d, "smi_%s.log" % name))
    e_pkg = pw * ms if pw and ms else None
    e_dyn = (pw - pw_idle) * ms if pw and ms and pw_idle else None
    delta = (e_dyn - prev) if (e_dyn is not None and prev is not None) else None
    f = lambda x, k=1: ("%.*f" % (k, x)) if isinstance(x, (int, float)) else "-"
    print("| %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %d |" % (
        name, c, a, f(b.get("value")), f(ms, 4), sclk, f(pw, 0), f(e_pkg, 0), f(e_dyn, 0),
        f(delta, 0), n))
    prev = e_dyn


if __name__ == "__main__":
  main()
