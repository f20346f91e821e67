"""Import shim: the package directory is named `lightly-train_amd/` (repo contract), which is not a
valid Python identifier.  `import lightly_train_amd` loads that directory as the package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lightly-train_amd")
_spec = importlib.util.spec_from_file_location(
    "lightly_train_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["lightly_train_amd"] = _mod
_spec.loader.exec_module(_mod)
