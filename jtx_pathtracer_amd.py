"""Import shim: the package directory is `jtx-pathtracer_amd/` (hyphenated, as prescribed); this module
makes it importable under a valid Python identifier."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("jtx-pathtracer_amd")
sys.modules[__name__] = _pkg
