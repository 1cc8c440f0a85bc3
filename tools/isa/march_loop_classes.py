#!/usr/bin/env python3
"""march_loop_classes.py [-o profiles/r06_march_loop_classes.txt] -- the achievable-issue roof of the headline kernel, from its ISA.

Compiles csrc/rmdf_render.hip to gfx950 assembly with the product's flags (csrc/Makefile CXXFLAGS; hipcc --cuda-device-only -S: the same
code generator input as the shipped library, ~15 s, no GPU), takes k_render<2, true, OUT_RGBA8> (BASELINE config 3), finds the loops the
code generator annotated, and prices every instruction in ISSUE SLOTS of the vector port (1 slot = one wave64 instruction of a pure
v_mul_f32 stream = 2 cycles at the spec rate) with the costs MEASURED on MI355X by tools/ubench/valu_rates (profiles/r04_form_costs.txt,
eight waves per SIMD, the instruction among seven multiplies):
    plain VALU (fma / mul / add / min3 / cmp / cndmask / mov / cvt / shifts, literals and SGPR operands, DPP)      1.0
    transcendental (v_rsq / v_rcp / v_sqrt / v_log / v_exp / v_sin / v_cos)                                          6.6 in a mix (3.5 back to back; 2 by the quarter-rate spec)
    v_readlane / v_writelane                                                                                          1.9 / 1.0
    SALU, s_cbranch, s_waitcnt, s_nop (another port: measured 0.2 of a slot beside seven multiplies)                  0.2
    LDS / global / scalar memory instructions (issue only; their latency is what the other waves cover)               1.0
Output: per loop the instruction classes (the "named list" behind "1.48 issued lane-slots per as-written operation"), cycles per Mandelbulb
iteration pass at 100 % lane utilisation against the measured ones, and the achievable-issue roof next to the as-written one."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "ray-marching-distance-fields_amd", "csrc")
KERNEL = "_ZN4rmdf8k_renderILi2ELb1ELi0EEEvNS_11FrameParamsE"

SLOT = {"valu": 1.0, "trans": 6.6, "readlane": 1.9, "scalar": 0.2, "mem": 1.0}
TRANS = re.compile(r"^v_(rsq|sqrt|rcp|log|exp|sin|cos)_")

# as-written IEEE operations of the reference's GLSL (bench.py: mb8_flops) and the headline frame's counters from the
# instrumented oracle (bench.py HEADLINE_COUNTERS): F = 79 I + 11 E + 9 S + 150 H + 30 P
OPS_PER = {"I": 79, "E": 11, "S": 9, "H": 150, "P": 30}
COUNTERS = {"I": 128870630, "E": 39739969, "S": 32356849, "H": 1230520, "P": 2073600}
SIMDS, CLOCK, SPEC_SLOT_RATE = 1024, 2.4e9, 1.2e9      # 256 CUs x 4; one wave64 instruction per SIMD every 2 cycles
SUSTAINED_SLOT_RATE = 1.0e9                             # what a pure v_mul_f32 stream sustains on this chip (valu_rates: 0.98-1.03 G/s/SIMD)


def product_flags():
    mk = open(os.path.join(CSRC, "Makefile")).read().replace("\\\n", " ")
    flags = re.search(r"^CXXFLAGS\s*=\s*(.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").split()
    return [f for f in flags if f != "-fPIC"]


def kernel_asm(path=None):
    if path is None:
        tmp = tempfile.mkdtemp()
        path = os.path.join(tmp, "render.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + product_flags() + ["-x", "hip", "--cuda-device-only", "-S", os.path.join(CSRC, "rmdf_render.hip"), "-o", path],
                              stderr=subprocess.DEVNULL)
    lines = open(path).read().split("\n")
    a = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    b = next(i for i in range(a, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[a + 1:b]


def classify(op):
    if op.startswith(("v_readlane", "v_readfirstlane")):
        return "readlane"
    if op.startswith("v_"):
        if TRANS.match(op):
            return "trans"
        return "valu"
    if op.startswith(("ds_", "global_", "buffer_", "flat_", "scratch_", "s_load", "s_buffer_load")):
        return "mem"
    return "scalar"


def detail(op):
    """finer names for the table"""
    if TRANS.match(op): return "transcendental (%s)" % op.split("_e")[0]
    if op.startswith(("v_fma", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_mac", "v_fmac", "v_mad_f32", "v_subrev_f32", "v_mul_legacy")): return "float mul / add / fma"
    if op.startswith(("v_min", "v_max", "v_med3")): return "min / max / med3"
    if op.startswith("v_cmp"): return "compare (guards, bailout, selects)"
    if op.startswith("v_cndmask"): return "v_cndmask (selects)"
    if op.startswith(("v_mov", "v_accvgpr", "v_swap")): return "register copy"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")): return "lane <-> scalar"
    if op.startswith("v_cvt"): return "conversion"
    if op.startswith(("v_ldexp", "v_frexp", "v_rndne", "v_floor", "v_fract", "v_trunc")): return "ldexp / frexp / round"
    if op.startswith("v_"): return "integer / bit vector op"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_setprio")): return "s_waitcnt / s_nop / s_barrier"
    if op.startswith(("ds_",)): return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vector memory"
    if op.startswith(("s_load", "s_buffer_load")): return "scalar memory"
    return "SALU (masks, counters, compares)"


def parse(lines):
    """[(label or None, op, args, branch target or None)] in layout order + loop headers {label: depth}"""
    out, headers, parents, cur = [], {}, {}, None
    for l in lines:
        m = re.match(r"^(\.LBB\d+_\d+):(.*)$", l)
        if m:
            cur = m.group(1)
            out.append((cur, None, None, None))
            d = re.search(r"Loop Header: Depth=(\d+)", m.group(2))
            if d: headers[cur] = int(d.group(1))
            q = re.search(r"Parent Loop (BB\d+_\d+)", m.group(2))
            if q: parents.setdefault(cur, ".L" + q.group(1))          # (the innermost parent is printed last; one level here)
            continue
        t = l.split(";")[0].strip()
        if not t or t.startswith("."):
            d = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", l)
            if d and cur: headers[cur] = int(d.group(1))
            q = re.search(r"Parent Loop (BB\d+_\d+)", l)
            if q and cur: parents[cur] = ".L" + q.group(1)
            continue
        op, _, args = t.partition(" ")
        op = op.strip(); args = args.strip()
        tgt = args.split()[-1] if op.startswith(("s_cbranch", "s_branch")) else None
        out.append((None, op, args, tgt))
    return out, headers, parents


def loops(seq, headers, parents):
    """{header: sorted list of instruction indexes of the loop's blocks}.  The code generator rotates loops (the header sits in the middle
    of the layout), so extents will not do: blocks = the strongly connected component of the header once the headers of its enclosing
    loops are taken out of the control-flow graph."""
    # basic blocks: cut at labels and behind branches
    starts = [0]
    for i, (lab, op, a, t) in enumerate(seq):
        if lab is not None and i != 0: starts.append(i)
        if op and op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")) and i + 1 < len(seq): starts.append(i + 1)
    starts = sorted(set(starts))
    bid = {}
    for k, st in enumerate(starts):
        for i in range(st, starts[k + 1] if k + 1 < len(starts) else len(seq)): bid[i] = k
    label_block = {lab: bid[i] for i, (lab, op, a, t) in enumerate(seq) if lab}
    succ = {k: set() for k in range(len(starts))}
    for k, st in enumerate(starts):
        en = (starts[k + 1] if k + 1 < len(starts) else len(seq)) - 1
        last = next(((op, t) for lab, op, a, t in reversed(seq[st:en + 1]) if op), (None, None))
        if last[0] and last[0].startswith(("s_cbranch", "s_branch")) and last[1] in label_block: succ[k].add(label_block[last[1]])
        if not (last[0] and last[0].startswith(("s_branch", "s_endpgm", "s_setpc"))) and k + 1 < len(starts): succ[k].add(k + 1)
    pred = {k: set() for k in succ}
    for k, ss in succ.items():
        for x in ss: pred[x].add(k)

    def reach(start, edges, banned):
        seen, todo = {start}, [start]
        while todo:
            n = todo.pop()
            for x in edges[n]:
                if x not in seen and x not in banned: seen.add(x); todo.append(x)
        return seen

    def block_range(k):
        return range(starts[k], starts[k + 1] if k + 1 < len(starts) else len(seq))
    weight = {k: sum(1 for i in block_range(k) if seq[i][1]) for k in succ}

    def trans_of(k):
        return sum(1 for i in block_range(k) if seq[i][1] and TRANS.match(seq[i][1]))

    def cond_taken(n, x):
        """the edge n -> x is the TAKEN side of n's conditional branch (the code generator lays the likely side out as the fall-through)"""
        en = block_range(n)[-1]
        last = next(((op, t) for lab, op, a, t in reversed([seq[i] for i in block_range(n)]) if op), (None, None))
        return bool(last[0] and last[0].startswith("s_cbranch") and label_block.get(last[1]) == x and x != n + 1)

    def hot_cycle(hb, comp):
        """Of the simple cycles through hb inside the component: the one that takes the fewest conditional branches (fall-through = the side the
        code generator expects), then the fewest transcendentals, then the most instructions.  For the iteration pass that is the pass itself:
        the roots' slow path (guard tripped) and the last-pass exit both hang off taken branches.  (Components of more than 40 blocks: all of it.)"""
        if len(comp) > 40: return set(comp)
        best, stack = None, [(hb, (hb,), 0)]
        while stack:
            n, path, taken = stack.pop()
            for x in succ[n]:
                tk = taken + (1 if cond_taken(n, x) else 0)
                if x == hb:
                    key = (tk, sum(trans_of(k) for k in path), -sum(weight[k] for k in path))
                    if best is None or key < best[0]: best = (key, set(path))
                elif x in comp and x not in path:
                    stack.append((x, path + (x,), tk))
        return best[1] if best else {hb}

    out, hot = {}, {}
    for h in headers:
        banned, q = set(), parents.get(h)
        while q:
            banned.add(label_block[q]); q = parents.get(q)
        hb = label_block[h]
        comp = reach(hb, succ, banned) & reach(hb, pred, banned)
        out[h] = sorted(i for i in range(len(seq)) if bid[i] in comp)
        cyc = hot_cycle(hb, comp)
        hot[h] = sorted(i for i in range(len(seq)) if bid[i] in cyc)
    return out, hot


def tally(seq, idx, skip=()):
    cls, det = {}, {}
    skip = set(skip)
    for i in idx:
        if i in skip: continue
        lab, op, args, t = seq[i]
        if op is None: continue
        c = classify(op)
        cls[c] = cls.get(c, 0) + 1
        d = detail(op)
        det[d] = det.get(d, 0) + 1
    return cls, det


def slots(cls):
    return sum(SLOT[c] * n for c, n in cls.items())


def table(P, seq, idx, skip=()):
    cls, det = tally(seq, idx, skip)
    P("%-44s %6s %10s %8s" % ("class", "count", "slots each", "slots"))
    tot = 0.0
    skipset = set(skip)
    for name, n in sorted(det.items(), key=lambda kv: -kv[1]):
        anyop = next(seq[i][1] for i in idx if i not in skipset and seq[i][1] and detail(seq[i][1]) == name)
        sl = SLOT[classify(anyop)]
        tot += sl * n
        P("%-44s %6d %10.1f %8.1f" % (name, n, sl, sl * n))
    P("%-44s %6d %10s %8.1f" % ("total", sum(cls.values()), "", tot))
    vec = SLOT["valu"] * cls.get("valu", 0) + SLOT["trans"] * cls.get("trans", 0) + SLOT["readlane"] * cls.get("readlane", 0)
    return cls, tot, vec


def report(asm_path=None):
    seq, headers, parents = parse(kernel_asm(asm_path))
    body, hot = loops(seq, headers, parents)
    rows = []
    P = lambda s="": rows.append(s)
    total_cls, _ = tally(seq, range(len(seq)))
    nvec = lambda c: c.get("valu", 0) + c.get("trans", 0) + c.get("readlane", 0)
    P("# k_render<2, true, OUT_RGBA8> (BASELINE config 3) compiled with the product's flags: %d instructions, %d vector (%d transcendental)" %
      (sum(total_cls.values()), nvec(total_cls), total_cls.get("trans", 0)))
    P("# slot prices: profiles/r04_form_costs.txt (MI355X, tools/ubench/valu_rates forms): plain vector 1.0, transcendental 6.6 in a mix, v_readlane 1.9, scalar-port 0.2, memory issue 1.0")
    # the iteration passes: innermost loops with exactly two transcendentals (the reciprocal square roots that seed the pass's two correctly
    # rounded roots) and >= 70 vector instructions
    passes = []
    for h in headers:
        inner = [g for g in headers if g != h and set(body[g]) < set(body[h])]
        cls, _ = tally(seq, hot[h])
        if not inner and cls.get("trans", 0) == 2 and nvec(cls) >= 70:
            passes.append(h)
    passes.sort(key=lambda h: body[h][0])
    P("# %d inlined copies of the Mandelbulb iteration pass (the march's plain and pooled arms, the normal's and the AO's estimates); vector instructions per copy: %s" %
      (len(passes), " ".join(str(nvec(tally(seq, hot[h])[0])) for h in passes)))
    nested = [h for h in passes if headers[h] >= 2]
    h = (nested or passes)[0]
    P()
    P("## one iteration pass of the march (loop %s, depth %d): the hot cycle through the loop header" % (h, headers[h]))
    cls, tot, vslots = table(P, seq, hot[h])
    scls, _ = tally(seq, body[h], hot[h])
    P("(beside it in the loop: the roots' slow path -- %d instructions, %d transcendental -- taken when the shared root guard trips: ~10^3 estimates per headline frame)" % (sum(scls.values()), scls.get("trans", 0)))
    nv = nvec(cls)
    P("-> %d vector instructions for %d as-written operations (fragment.shd:74-158).  The difference by name: the two correctly rounded roots are 12 instructions" % (nv, OPS_PER["I"]))
    P("   instead of 2 (+10), the power-of-two folds give 5 back, the guards -- bailout compare, root guard's two compares, the fold bound's two v_min3 -- add 5, the")
    P("   per-lane iteration counter 1 (the v_mov); no v_cndmask, no loop-control vector instruction.  Scalar side: %d instructions (EXEC masks of the bailout," % cls.get("scalar", 0))
    P("   the guard's OR + test, the pass counter, three branches) on the scalar port: %.1f slots if none of it overlapped, measured 0.2 each beside vector work." % (tot - vslots))
    P("-> %.1f vector-port slots per pass = %.2f per as-written operation; the two v_rsq_f32 are %.1f of them." % (vslots, vslots / OPS_PER["I"], 2 * SLOT["trans"]))
    P("-> cycles per wave-pass with every slot issued back to back at 100 %% lane utilisation: %.0f at the 2-cycle spec rate = %.2f cycles per Mandelbulb iteration per SIMD lane-group" % (2 * vslots, 2 * vslots / 64))
    P("   hoisting the guard's scalar OR + test + branch to once per estimate (a sticky flag) would save 3 scalar-port instructions = %.1f slots of %.1f: %.1f %%." % (3 * SLOT["scalar"], tot, 100 * 3 * SLOT["scalar"] / tot))
    P()
    # the step loop around it, without any pass loop
    outer = [g for g in headers if set(body[h]) < set(body[g])]
    est_vec_slots = None
    if outer:
        g = min(outer, key=lambda x: len(body[x]))
        skip = set()
        for q in passes:
            if set(body[q]) < set(body[g]): skip |= set(body[q])
        P("## the step loop around it (loop %s, depth %d) without its pass loops: one distance estimate's fixed part -- first pass peeled from the loop, pinned log," % (g, headers[g]))
        P("## final quotient, root of the escaped radius, ray step, bailout tests -- plus the pooling's bookkeeping (LDS mailboxes, ballots, barriers).  STATIC count: both")
        P("## arms (plain march, pooled march with the hand-over code) and the written fall-back of the folded passes are in here; one estimate runs one arm")
        ocls, otot, ovec = table(P, seq, body[g], skip)
        est_vec_slots = ovec
        P()
    # ---- roofs
    ops = sum(OPS_PER[k] * COUNTERS[k] for k in OPS_PER)
    P("## roofs for the headline frame (1920x1080, 256 steps): %.3f G as-written lane-operations = 79 x %d passes + 11 x %d estimates + 9 x %d steps + 150 x %d hit" %
      (ops / 1e9, COUNTERS["I"], COUNTERS["E"], COUNTERS["S"], COUNTERS["H"]))
    P("## pixels + 30 x %d pixels (instrumented oracle; bench.py)" % COUNTERS["P"])
    as_written_peak = SIMDS * 32 * CLOCK                  # 78.6 T lane-ops/s: every slot an as-written operation on 64 live lanes
    t_aw = ops / as_written_peak
    P("(A) as-written roof (SURVEY 8d): every issue slot an as-written operation on 64 live lanes: %.1f T lane-ops/s -> the frame in %.4f ms" % (as_written_peak / 1e12, t_aw * 1e3))
    # (B) the instruction stream the kernel must issue at 100 % lane utilisation.  Passes: the table above x passes / 64.  Everything else: the
    # measured dynamic count (PMC SQ_INSTS_VALU 351.2 M wave-instructions of which 8.8 M transcendental, SQ lane utilisation 0.735:
    # profiles/r04_pmc_sq.csv -- the march loop of that build is instruction-identical to this one, profiles/r05_kernel_isa_vs_r04.txt)
    # minus the passes' share at the march's measured lane utilisation (0.741, profiles/r04_sched_regroup.txt), scaled to full lanes.
    M_INSTR, M_TRANS, M_UTIL = 351.2e6, 8.8e6, 0.735
    wave_pass_full = COUNTERS["I"] / 64.0
    pass_instr_meas = wave_pass_full / 0.741 * nv                                   # what the passes cost in the measured run
    rest_instr_meas = M_INSTR - pass_instr_meas
    rest_trans_meas = M_TRANS - wave_pass_full / 0.741 * 2
    rest_slots_full = (rest_instr_meas + (SLOT["trans"] - 1.0) * rest_trans_meas) * M_UTIL       # the rest at its measured utilisation -> full lanes
    pass_slots_full = wave_pass_full * vslots
    full = pass_slots_full + rest_slots_full
    t_spec, t_sust = full / SIMDS / SPEC_SLOT_RATE, full / SIMDS / SUSTAINED_SLOT_RATE
    P("(B) achievable-issue roof: the instructions this kernel has to issue, priced in slots, on 64 live lanes: passes %.1f M wave-slots (%d x %.1f / 64) + everything else" %
      (pass_slots_full / 1e6, COUNTERS["I"], vslots))
    P("    %.1f M (measured 351.2 M vector instructions, 8.8 M transcendental, minus the passes' share, at lane utilisation 0.735 -> 1.0) = %.1f M wave-slots" % (rest_slots_full / 1e6, full / 1e6))
    P("    -> %.4f ms at the 2-cycle spec issue rate (1.2 G slots/s/SIMD) = %.1f T as-written lane-ops/s = %.3f of (A)" % (t_spec * 1e3, ops / t_spec / 1e12, t_aw / t_spec))
    P("    -> %.4f ms at the rate a pure v_mul_f32 stream sustains on this chip (1.0 G slots/s/SIMD: clock / power management) = %.1f T = %.3f of (A)" % (t_sust * 1e3, ops / t_sust / 1e12, t_aw / t_sust))
    M_MS = 0.3795
    P("(C) measured (profiles/r04_kernel_stats_s1.csv, 1242 launches): %.4f ms = %.1f T as-written lane-ops/s = %.3f of (A), %.3f of (B) at the spec rate, %.3f of (B) at the sustained rate." %
      (M_MS, ops / (M_MS * 1e-3) / 1e12, t_aw / (M_MS * 1e-3), t_spec / (M_MS * 1e-3), t_sust / (M_MS * 1e-3)))
    r = t_sust / (M_MS * 1e-3)
    P("    (C) / (B-sustained) = %.3f = lane utilisation 0.735 x %.3f: with the lanes it has live, the kernel issues at %.0f %% of what a pure v_mul_f32 stream sustains --" % (r, r / 0.735, 100 * r / 0.735))
    P("    it IS at the issue limit.  What is left to take is (1) lane utilisation: 0.265 of the lane-slots belong to rays that have ended while their packet marches")
    P("    on (profiles/r04_sched_regroup.txt: perfect regrouping inside a workgroup -4.4 %, hand-over across workgroups -7 % idealised), and (2) instructions:")
    P("    per pass 8 above the 79 as-written (exact roots +10, folds -5, guards +5, counter +1: all needed for bit-exact escape counts), the two v_rsq_f32 at")
    P("    6.6 slots each (13 % of the pass), scalar-port work 2.6 % (hoisting the guard's OR / test / branch: 0.6 %).  Below 3 % is left in the march loop's issue.")
    P("    What separates (B) from (A) is the instruction stream: %.2f slots per as-written operation in the passes, more in the estimate's fixed part (pinned log)." % (vslots / OPS_PER["I"]))
    return "\n".join(rows), dict(vslots=vslots, pass_instr=nv, full_slots=full, t_spec_ms=t_spec * 1e3, t_sust_ms=t_sust * 1e3, t_as_written_ms=t_aw * 1e3, ops=ops)


if __name__ == "__main__":
    asm = None
    out = None
    a = sys.argv[1:]
    while a:
        if a[0] == "-o": out = a[1]; a = a[2:]
        elif a[0] == "--asm": asm = a[1]; a = a[2:]
        else: raise SystemExit(__doc__)
    text, d = report(asm)
    print(text)
    if out:
        import json
        open(out, "w").write(text + "\n")
        d["source"] = "tools/isa/march_loop_classes.py: ISA of k_render<2,true,0> priced with profiles/r04_form_costs.txt; dynamic counts outside the passes from profiles/r04_pmc_sq.csv (march loop instruction-identical)"
        json.dump(d, open(os.path.splitext(out)[0] + ".json", "w"), indent=1, sort_keys=True)
