#!/usr/bin/env python3
"""blocks.py <file.s> <kernel-substring> : basic blocks of one kernel in a hipcc -S listing -- per block the VALU / SALU /
memory / branch instruction counts and the transcendentals, so that the hot loop of two builds can be compared."""
import re, sys
fn, key = sys.argv[1], sys.argv[2]
lines = open(fn).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(("E:",)) or (l.startswith("_Z") and key in l and ": " in l and l.split(":")[0].find(key) >= 0))
blocks, cur = [], ["entry", []]
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith(".Lfunc_end"): break
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        blocks.append(cur); cur = [m.group(1), []]
        continue
    if not t or t.startswith(";") or t.startswith("."): continue
    cur[1].append(t.split(";")[0].strip())
blocks.append(cur)
tot = 0
for name, ins in blocks:
    v = sum(1 for i in ins if i.startswith("v_"))
    s = sum(1 for i in ins if i.startswith("s_") and not i.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_nop")))
    br = sum(1 for i in ins if i.startswith(("s_cbranch", "s_branch")))
    mem = sum(1 for i in ins if i.startswith(("ds_", "global_", "scratch_", "buffer_", "flat_", "s_load")))
    tr = sum(1 for i in ins if re.match(r"v_(rsq|sqrt|rcp|log|exp|sin|cos)_", i))
    tgt = [i.split()[-1] for i in ins if i.startswith(("s_cbranch", "s_branch"))]
    tot += len(ins)
    if len(sys.argv) > 3 and sys.argv[3] == "all" or v >= 20 or tr:
        print("%-12s n=%4d valu=%4d salu=%3d br=%d mem=%2d trans=%d -> %s" % (name, len(ins), v, s, br, mem, tr, ",".join(tgt)))
print("total instructions", tot)
