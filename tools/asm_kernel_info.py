#!/usr/bin/env python
"""Resource usage and instruction mix of one kernel in a `hipcc --cuda-device-only -S`
listing:  asm_kernel_info.py file.s kernel_substr"""
import re, sys, collections
path, sub = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = None
for i, l in enumerate(lines):
  if re.match(r"^_Z\w*%s\w*:" % re.escape(sub), l):
    start = i; name = l[:-1]; break
if start is None: sys.exit("kernel not found")
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
ops = collections.Counter()
for l in body:
  m = re.match(r"^\t([a-z_0-9]+)", l)
  if m: ops[m.group(1)] += 1
print(name, "instructions:", sum(ops.values()))
for k in ("v_mfma_f32_32x32x16_f16","v_mfma_f32_32x32x16_bf16","v_mfma_f32_32x32x2_f32","ds_read_b128","global_load_dwordx4","global_load_lds_dwordx4","s_barrier","s_waitcnt","scratch_load_dword","scratch_store_dword","v_accvgpr_read_b32","v_accvgpr_write_b32"):
  if ops.get(k): print("  %-28s %d" % (k, ops[k]))
txt = "\n".join(lines)
m = re.search(r"\.amdhsa_kernel %s(.*?)\.end_amdhsa_kernel" % re.escape(name), txt, re.S)
if m:
  for key in ("next_free_vgpr","next_free_sgpr","accum_offset","group_segment_fixed_size","private_segment_fixed_size"):
    mm = re.search(r"\.amdhsa_%s (\S+)" % key, m.group(1))
    if mm: print("  %s = %s" % (key, mm.group(1)))
