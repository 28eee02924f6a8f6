"""Import shim: `import vadx` loads the package directory `voice-activity-detection-vad-onnx_amd/`
(the directory name mandated for this repo is not a valid Python identifier)."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "voice-activity-detection-vad-onnx_amd")
_spec = _u.spec_from_file_location("vadx", _os.path.join(_dir, "__init__.py"),
                                   submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["vadx"] = _mod
_spec.loader.exec_module(_mod)
