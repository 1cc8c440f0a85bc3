#!/usr/bin/env python3
"""kernel_diff.py A.so B.so -- per-kernel comparison of the gfx950 code two builds of the library carry.  Every offload bundle of each
file is unbundled (clang-offload-bundler), disassembled (llvm-objdump -d) and cut at the kernel symbols; per kernel: instruction count
and whether the instruction streams are IDENTICAL (mnemonics and every operand, addresses and encodings stripped; branch targets kept as
offsets from the kernel's start).  Kernels present on one side only are listed.  Used for the claim "the product's kernels are the ones
the GPU tier last ran" (profiles/r06_kernel_isa_vs_gpu_tested.txt).
kernel_diff.py --hashes A.so prints {kernel: sha256 of its instruction stream} (tests/golden/gpu_tested_kernels.json is made of these)."""
import hashlib, os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(so, tmp):
    out = []
    blob = open(so, "rb").read()
    # bundles: "__CLANG_OFFLOAD_BUNDLE__" magic; let the bundler list and extract the gfx950 targets of each .hip_fatbin piece
    sec = os.path.join(tmp, os.path.basename(so) + ".fatbin")
    subprocess.check_call([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, sec])
    data = open(sec, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), data)]
    for i, st in enumerate(starts):
        piece = os.path.join(tmp, "%s.%d.bundle" % (os.path.basename(so), i))
        open(piece, "wb").write(data[st:starts[i + 1] if i + 1 < len(starts) else len(data)])
        targets = subprocess.check_output([LLVM + "/clang-offload-bundler", "--list", "--type=o", "--input=" + piece]).decode().split()
        for t in targets:
            if "gfx950" not in t:
                continue
            co = piece + ".co"
            subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=" + t, "--input=" + piece, "--output=" + co])
            out.append(co)
    return out


def kernels(so, tmp):
    ks = {}
    for co in code_objects(so, tmp):
        txt = subprocess.check_output([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", "-C", co]).decode()
        name, start = None, 0
        for ln in txt.split("\n"):
            m = re.match(r"^([0-9a-f]+) <(.*)>:$", ln)
            if m:
                name, start = m.group(2), int(m.group(1), 16)
                ks[name] = []
                continue
            if name is None or not ln.startswith(" ") and not ln.startswith("\t"):
                continue
            ins = ln.split("//")[0].strip()
            if not ins:
                continue
            ks[name].append(ins)
    return ks


def kernel_hashes(so):
    """{demangled kernel / device function: sha256 of its instruction stream} of every gfx950 code object in `so`"""
    with tempfile.TemporaryDirectory() as tmp:
        return {k: hashlib.sha256("\n".join(v).encode()).hexdigest() for k, v in kernels(so, tmp).items()}


def main():
    if sys.argv[1] == "--hashes":
        import json
        print(json.dumps(kernel_hashes(sys.argv[2]), indent=1, sort_keys=True))
        return
    a, b = sys.argv[1], sys.argv[2]
    with tempfile.TemporaryDirectory() as tmp:
        ka, kb = kernels(a, tmp), kernels(b, tmp)
    print("# A = %s  sha256 %s" % (a, hashlib.sha256(open(a, "rb").read()).hexdigest()))
    print("# B = %s  sha256 %s" % (b, hashlib.sha256(open(b, "rb").read()).hexdigest()))
    same = diff = 0
    for k in sorted(set(ka) | set(kb)):
        if k not in ka:
            print("only in B   %6d instructions  %s" % (len(kb[k]), k)); continue
        if k not in kb:
            print("only in A   %6d instructions  %s" % (len(ka[k]), k)); continue
        if ka[k] == kb[k]:
            same += 1
            print("IDENTICAL   %6d instructions  %s" % (len(ka[k]), k))
        else:
            diff += 1
            n = sum(1 for x, y in zip(ka[k], kb[k]) if x != y) + abs(len(ka[k]) - len(kb[k]))
            print("DIFFERENT   %6d vs %6d instructions, %d positions differ  %s" % (len(ka[k]), len(kb[k]), n, k))
            if "-v" in sys.argv:
                for i, (x, y) in enumerate(zip(ka[k], kb[k])):
                    if x != y:
                        print("      [%d]  A: %s\n            B: %s" % (i, x, y))
    print("# %d kernels identical, %d different" % (same, diff))


if __name__ == "__main__":
    main()
