#!/usr/bin/env python3
"""Put ROCr's fault message and the recorder's ring together: resolve.py <loop dir>
For every run output w*_r*.txt that holds "Memory access fault by GPU node-N (Agent handle: H) on address A. Reason: R" find the recorder
file fault_<pid>.txt of the same process (the one whose [heap] mapping holds the agent handle: ASLR gives every process its own), and
print: which mapping A lies in, every recorded page lock / registration / allocation / copy whose range holds A (with what happened to it
afterwards), and the last calls before the abort."""
import glob, os, re, sys

d = sys.argv[1]
tail_n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
faults = []
for fn in sorted(glob.glob(os.path.join(d, "w*_r*.txt"))):
    txt = open(fn, errors="replace").read()
    m = re.search(r"Memory access fault by GPU node-(\d+) \(Agent handle: (0x[0-9a-f]+)\) on address (0x[0-9a-f]+)\. Reason: ([^\n]*)", txt)
    if m:
        ntests = len(re.findall(r"[.sF]", txt.split("Memory access fault")[0].splitlines()[-1])) if "Memory access fault" in txt else 0
        faults.append((fn, int(m.group(2), 16), int(m.group(3), 16), m.group(4).strip(), ntests))
recs = {}
for fn in glob.glob(os.path.join(d, "fault_*.txt")):
    lines = open(fn, errors="replace").read().splitlines()
    if "== maps" not in lines:
        continue
    k = lines.index("== maps")
    maps = []
    for ln in lines[k + 1:]:
        m = re.match(r"([0-9a-f]+)-([0-9a-f]+) (\S+) \S+ \S+ \S+\s*(.*)", ln)
        if m:
            maps.append((int(m.group(1), 16), int(m.group(2), 16), m.group(3), m.group(4)))
    recs[fn] = (lines[:k], maps)
print("%d faulting runs, %d recorder files" % (len(faults), len(recs)))
for fn, agent, va, reason, ntests in faults:
    print("=" * 120)
    print("%s: fault on address 0x%x (%s) in test #%d; agent handle 0x%x" % (os.path.basename(fn), va, reason, ntests + 1, agent))
    match = [r for r, (_, maps) in recs.items() if any(lo <= agent < hi for lo, hi, _, _ in maps)]
    if not match:
        print("  no recorder file of this process (run without the recorder?)")
        continue
    ring, maps = recs[match[0]]
    print("  recorder file %s (%s)" % (os.path.basename(match[0]), ring[0]))
    inmap = [(lo, hi, pr, nm) for lo, hi, pr, nm in maps if lo <= va < hi]
    print("  the address lies in: %s" % (", ".join("%x-%x %s %s" % t for t in inmap) if inmap else "NO mapping of the process at abort time"))
    near = sorted(maps, key=lambda t: min(abs(t[0] - va), abs(t[1] - va)))[:3]
    for lo, hi, pr, nm in near:
        print("    nearby mapping %x-%x %s %s" % (lo, hi, pr, nm))
    ents = []
    for ln in ring[1:]:
        f = ln.split()
        if len(f) < 5:
            continue
        ents.append((int(f[0]), f[1], int(f[2], 16), int(f[3], 16), int(f[4], 16), f[5] if len(f) > 5 else ""))
    t_end = ents[-1][0] if ents else 0
    print("  calls whose range holds the address (time before the abort, call, arguments):")
    def holds(base, size):
        return base and size and base <= va < base + size
    hit = []
    for t, what, a, b, c, ra in ents:
        if what in ("hsa_memory_lock", "hsa_memory_lock_to_pool", "hipHostRegister", "hipMalloc", "hipHostMalloc", "hsa_pool_allocate", "hipExtMallocWithFlags") and holds(a, b):
            hit.append((t, what, "range 0x%x + 0x%x%s" % (a, b, (" -> agent ptr 0x%x" % c) if what.startswith("hsa_memory_lock") else ""), a))
        if what.startswith("hipMemcpy") and what.endswith(">"):
            for base, role in ((a, "dst"), (b, "src")):
                if holds(base, c):
                    hit.append((t, what, "%s 0x%x + 0x%x (dst 0x%x src 0x%x)" % (role, base, c, a, b), base))
    bases = set(h[3] for h in hit)
    for t, what, a, b, c, ra in ents:
        if what in ("hsa_memory_unlock>", "hsa_memory_unlock<", "hipHostUnregister", "hipFree>", "hipHostFree>", "hsa_pool_free>") and a in bases:
            hit.append((t, what, "0x%x (status 0x%x)" % (a, c), a))
    for t, what, txt, _ in sorted(hit)[-40:]:
        print("    -%12.6f s  %-24s %s" % ((t_end - t) / 1e9, what, txt))
    print("  last %d calls before the abort:" % tail_n)
    for t, what, a, b, c, ra in ents[-tail_n:]:
        print("    -%12.6f s  %-24s 0x%x 0x%x 0x%x %s" % ((t_end - t) / 1e9, what, a, b, c, ra))
