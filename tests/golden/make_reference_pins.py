#!/usr/bin/env python3
"""Pin the constant tables of the reference by machine (build container only: reads /root/reference at run time and writes
tests/golden/reference_pins.json -- values and digests, never source text).

    python tests/golden/make_reference_pins.py

* cornell: the 64 `V3` literals of CornellBox.hs:48-129 (terms like `548.8 - 0.1` evaluated in Float, as GHC does for a
  `V3 Float` literal), turned into the 96 triangle vertices exactly as mkCornellBoxVerticesTex does (CornellBox.hs:25-38:
  quads (q0,q1,q2,q3) -> triangles (q0,q1,q3),(q3,q1,q2); (v / toUnit - 1) ^* scale in Float) -> sha256 of the 96x3 float32
  table, plus the raw quads (data).
* shader_constants: the numeric literals of the fragment.shd lines the render path's constants are stated on (named here,
  located by line number and checked to be where SURVEY.md says they are): bailout, iterations, MAX_STEPS, MIN_DIST, the bounding
  sphere radii, both distance-AO tap sets and the fudge factors, the Fresnel / shading constants, the finite-difference epsilon,
  the step back, the camera distance, the field of view, gamma.

tests/test_reference_pins.py (CPU tier) compares the oracle's AND the product's tables with this file, so that a typo cannot hide
behind "oracle == kernel".
"""
import hashlib
import json
import os
import re

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_pins.json")
f32 = np.float32


def cornell():
    lines = open(os.path.join(REF, "CornellBox.hs")).read().split("\n")
    body = lines[47:129]                                        # CornellBox.hs:48-129
    quads = []
    term = r"\(?([0-9.]+(?:\s*[-+]\s*[0-9.]+)?)\)?"                  # a literal, or a parenthesised `a - b`
    for ln in body:
        m = re.search(r"V3\s+" + term + r"\s+" + term + r"\s+" + term + r"\s*$", ln.split("--")[0].rstrip())
        if not m:
            continue
        v = []
        for t in m.groups():
            parts = re.split(r"\s*([-+])\s*", t)
            acc = f32(float(parts[0]))
            for op, rhs in zip(parts[1::2], parts[2::2]):
                acc = f32(acc - f32(float(rhs))) if op == "-" else f32(acc + f32(float(rhs)))
            v.append(acc)
        quads.append(v)
    quads = np.array(quads, f32)
    assert quads.shape == (64, 3), quads.shape
    # mkCornellBoxVerticesTex (CornellBox.hs:25-38), Float arithmetic
    to_unit = f32(f32(559.2) / f32(2))
    scale = f32(f32(f32(1) / f32(np.sqrt(f32(2 * 2 + 2 * 2 + 2 * 2)) / f32(2))) * f32(0.99))
    order = (0, 1, 3, 3, 1, 2)
    tri = np.empty((96, 3), f32)
    for q in range(16):
        for k in range(6):
            tri[q * 6 + k] = (quads[q * 4 + order[k]] / to_unit - f32(1)) * scale
    return {"source": "CornellBox.hs:25-38,48-129", "quads_64x3": [[float(x) for x in row] for row in quads],
            "quads_sha256": hashlib.sha256(quads.tobytes()).hexdigest(),
            "triangle_vertices_96x3_sha256": hashlib.sha256(tri.tobytes()).hexdigest(),
            "to_unit": float(to_unit), "scale": float(scale)}


def shader_constants():
    src = open(os.path.join(REF, "fragment.shd")).read().split("\n")
    num = r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?"

    def at(line, pattern, group=1):
        m = re.search(pattern, src[line - 1])
        assert m, "fragment.shd:%d does not match %r: %r" % (line, pattern, src[line - 1])
        return float(m.group(group))
    c = {}
    c["mb_bailout"] = at(121, r"bailout\s*=\s*(%s)" % num)
    c["mb_iterations"] = at(122, r"iterations\s*=\s*(%s)" % num)
    c["march_max_steps_default"] = at(634, r"MAX_STEPS\s*=\s*(%s)" % num)
    c["march_min_dist"] = at(635, r"MIN_DIST\s*=\s*(%s)" % num)
    c["bsphere_r_power8"] = at(643, r"(%s);" % num)
    c["bsphere_r_general"] = at(645, r"(%s);" % num)
    c["bsphere_r_other"] = at(648, r"(%s);" % num)
    for i, (lw, ld) in enumerate(((548, 549), (552, 553))):
        c["ao_w%d" % i] = at(lw, r"weight\s*=\s*(%s)" % num)
        c["ao_d%d" % i] = at(ld, r"delta\s*=\s*(%s)" % num)
    c["ao_bias"] = at(558, r"occl_sum\s*-=\s*(%s)" % num)
    c["ao_gain"] = at(559, r"occl_sum\s*\*=\s*(%s)" % num)
    for i, (lw, ld) in enumerate(((571, 572), (575, 576), (579, 580), (583, 584))):
        c["cornell_ao_w%d" % i] = at(lw, r"weight\s*=\s*(%s)" % num)
        c["cornell_ao_d%d" % i] = at(ld, r"delta\s*=\s*(%s)" % num)
    c["normal_eps"] = at(466, r"eps\s*=\s*(%s)" % num)
    c["isec_step_back"] = at(751, r"dir\s*\*\s*(%s)" % num)
    c["fresnel_eta"] = at(799, r"isec_n\),\s*(%s),\s*(%s)\)" % (num, num), 1)
    c["fresnel_k"] = at(799, r"isec_n\),\s*(%s),\s*(%s)\)" % (num, num), 2)
    c["diff_weight"] = at(801, r"diff_weight\s*=\s*(%s)" % num)
    m = re.search(r"vec3\((%s),\s*(%s),\s*(%s)\)" % (num, num, num), src[801])
    c["diff_r"], c["diff_g"], c["diff_b"] = (float(x) for x in m.groups())
    m = re.search(r"vec3\((%s),\s*(%s),\s*(%s)\)" % (num, num, num), src[802])
    c["spec_r"], c["spec_g"], c["spec_b"] = (float(x) for x in m.groups())
    c["spec_weight_one_minus"] = at(804, r"spec_weight\s*=\s*(%s)\s*-\s*diff_weight" % num)
    c["phong_lobe_n"] = at(808, r"normalize_phong_lobe\((%s)\)" % num)
    c["refl_weight"] = at(809, r"fresnel\s*\*\s*(%s)" % num)
    c["exposure"] = at(810, r"\)\s*\*\s*(%s)\s*\*\s*ao" % num)
    c["phong_lobe_plus"] = at(723, r"power\s*\+\s*(%s)\)\s*/\s*(%s)" % (num, num), 1)
    c["phong_lobe_div"] = at(723, r"power\s*\+\s*(%s)\)\s*/\s*(%s)" % (num, num), 2)
    c["camera_distance"] = at(897, r"\*\s*(%s);" % num)
    c["camera_cornell_radius"] = at(888, r"\*\s*(%s);" % num)
    c["camera_cornell_z"] = at(889, r"=\s*(%s);" % num)
    c["hfov_deg_a"] = at(910, r"(%s)\s*\*\s*(%s)," % (num, num), 1)
    c["hfov_deg_b"] = at(910, r"(%s)\s*\*\s*(%s)," % (num, num), 2)
    c["gamma"] = at(959, r"vec3\(1\.0\s*/\s*(%s)\)" % num)
    return {"source": "fragment.shd:121-122,466,548-584,634-648,723,751,799-810,888-897,910,959", "values": c}


def main():
    pins = {"note": "written by tests/golden/make_reference_pins.py from /root/reference (values and digests only)",
            "cornell": cornell(), "shader_constants": shader_constants()}
    with open(OUT, "w") as f:
        json.dump(pins, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", OUT, "(%d shader constants, Cornell table %s...)" % (
        len(pins["shader_constants"]["values"]), pins["cornell"]["triangle_vertices_96x3_sha256"][:12]))


if __name__ == "__main__":
    main()
