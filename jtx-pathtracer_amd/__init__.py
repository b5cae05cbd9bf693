"""jtx-pathtracer_amd -- MI355X-native path-tracing core for JTX-PathTracer (hot path only).

The directory name has a hyphen (it is the name the task prescribes); import it as
`import jtx_pathtracer_amd` (shim module at the repo root) or
`importlib.import_module("jtx-pathtracer_amd")`.
"""
from . import _capi, scenes, api, distributed, gltf     # noqa: F401
from .api import Scene, MultiScene, StaticCamera, DynamicCamera, JtxMiError  # noqa: F401
from .build import build_all, build_test_hooks, lib_is_built      # noqa: F401
