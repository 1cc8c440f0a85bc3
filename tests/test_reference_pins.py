"""CPU tier: the constant tables of the oracle AND of the product against values a script extracted from the reference itself
(tests/golden/make_reference_pins.py -> tests/golden/reference_pins.json).  Everything else in the parity suite compares the kernels
with the oracle; a constant mistyped the same way on both sides would pass there.  Here each side is compared with the reference:

* the Cornell box: the 96 triangle vertices (CornellBox.hs:21-46 applied to the 64 literals of :48-129) -- sha256 of the float32 table;
* the shader constants (bailout, iterations, MIN_DIST, bounding-sphere radii, distance-AO taps and fudge factors, Fresnel and
  shading weights, finite-difference epsilon, camera distance, field of view, gamma): name by name, as float32.

Host-only entry points on both sides (no GPU, no compute)."""
import hashlib
import json
import os

import numpy as np
import pytest

PINS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_pins.json")))


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, np.float32).tobytes()).hexdigest()


def test_cornell_table_oracle_equals_reference(orc):
    assert _sha(orc.cornell_vertices()) == PINS["cornell"]["triangle_vertices_96x3_sha256"]


def test_cornell_table_product_equals_reference(rmdf):
    v = rmdf.cornell_vertices()
    assert v.shape == (96, 3)
    assert _sha(v) == PINS["cornell"]["triangle_vertices_96x3_sha256"]


def test_cornell_pin_is_self_consistent():
    """the committed raw quads reproduce the committed digest with mkCornellBoxVerticesTex's arithmetic (so the JSON cannot be edited on
    one side only), and the one computed literal (`548.8 - 0.1`, the light: CornellBox.hs:81-84) is in it"""
    f32 = np.float32
    q = np.array(PINS["cornell"]["quads_64x3"], f32)
    assert _sha(q) == PINS["cornell"]["quads_sha256"]
    to_unit = f32(f32(559.2) / f32(2))
    scale = f32(f32(f32(1) / f32(np.sqrt(f32(12)) / f32(2))) * f32(0.99))
    tri = np.empty((96, 3), f32)
    for quad in range(16):
        for k, o in enumerate((0, 1, 3, 3, 1, 2)):
            tri[quad * 6 + k] = (q[quad * 4 + o] / to_unit - f32(1)) * scale
    assert _sha(tri) == PINS["cornell"]["triangle_vertices_96x3_sha256"]
    assert (q[:, 1] == f32(f32(548.8) - f32(0.1))).sum() == 4


@pytest.mark.parametrize("side", ["oracle", "product"])
def test_shader_constants_equal_reference(side, orc, rmdf):
    have = orc.shader_constants() if side == "oracle" else rmdf.shader_constants()
    want = PINS["shader_constants"]["values"]
    assert set(have) == set(want), (sorted(set(have) ^ set(want)))
    for name, v in want.items():
        assert np.float32(have[name]) == np.float32(v), (side, name, have[name], v)


def test_the_viewer_patch_applies_to_the_reference(tmp_path):
    """INTEGRATION.md section 3 is prose about three of the reference's files; hs/apply_viewer_patch.py is the same change as anchored edits.
    Where the reference checkout is present (this container, not the GPU box) the script must apply cleanly -- every anchor found exactly as
    often as stated -- and leave exactly the described differences.  (Compiling the result needs GHC, which the image does not have.)"""
    import subprocess
    import sys
    ref = "/root/reference"
    if not os.path.exists(os.path.join(ref, "App.hs")):
        pytest.skip("no reference checkout here")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "patched")
    r = subprocess.run([sys.executable, os.path.join(root, "ray-marching-distance-fields_amd", "hs", "apply_viewer_patch.py"), ref, out],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    import difflib
    delta = {}
    for name in ("App.hs", "Main.hs", "rmdf.cabal"):
        a = open(os.path.join(ref, name)).read().splitlines()
        b = open(os.path.join(out, name)).read().splitlines()
        d = [l for l in difflib.unified_diff(a, b, lineterm="", n=0) if l[:1] in "+-" and l[:3] not in ("+++", "---")]
        delta[name] = (sum(1 for l in d if l[0] == "-"), sum(1 for l in d if l[0] == "+"), d)
    assert delta["App.hs"][:2] == (4, 7) and delta["Main.hs"][:2] == (1, 2) and delta["rmdf.cabal"][:2] == (1, 3), {k: v[:2] for k, v in delta.items()}
    added = "\n".join(l[1:] for l in delta["App.hs"][2] if l[0] == "+")
    assert added.count("hipShader FS") == 4 and "import RmdfFFI" in added and "_aeHR" in added and "drawHipTile _aeHR" in added
    assert "withHipRenderer reflMapFn $ \\_aeHR -> do" in "\n".join(delta["Main.hs"][2])
    assert os.path.exists(os.path.join(out, "RmdfFFI.hs"))
    # the script refuses to write the reference's source into this repository
    r = subprocess.run([sys.executable, os.path.join(root, "ray-marching-distance-fields_amd", "hs", "apply_viewer_patch.py"), ref, os.path.join(root, "gpurun_out", "x")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "outside this repository" in r.stderr
