# coding=utf-8
"""Identity of the kernel sources a measurement was taken on.

`kernel_source_hash()` = sha256 (first 16 hex digits) over multiverse_amd/csrc/* and
include/multiverse_hip.h, in name order.  tools/pmc_report.py stores it in every PMC
summary under profiles/; bench.py quotes `roofline.traffic` from such a summary only when
the hash equals the one of the tree it runs from, so a counter figure can never outlive
the kernel it was collected on."""

from __future__ import annotations

import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)


def kernel_source_files():
  csrc = os.path.join(_HERE, "csrc")
  files = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc))
           if f.endswith((".h", ".hip", ".cpp"))]
  files.append(os.path.join(_ROOT, "include", "multiverse_hip.h"))
  return files


def kernel_source_hash():
  h = hashlib.sha256()
  for path in kernel_source_files():
    h.update(os.path.basename(path).encode() + b"\0")
    with open(path, "rb") as f:
      h.update(f.read())
  return h.hexdigest()[:16]
