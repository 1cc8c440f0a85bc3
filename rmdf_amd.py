"""Import shim: `import rmdf_amd` == the package in ./ray-marching-distance-fields_amd/ (its
directory name has hyphens, which the `import` statement cannot spell)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
sys.modules[__name__] = importlib.import_module("ray-marching-distance-fields_amd")
