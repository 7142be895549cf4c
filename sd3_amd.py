"""Import alias for the package directory `stable-diffusion-3-from-scratch_amd/` (its name is not a
valid Python identifier).  `import sd3_amd` loads that directory as the package `sd3_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stable-diffusion-3-from-scratch_amd")
_spec = importlib.util.spec_from_file_location("sd3_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sd3_amd"] = _mod
_spec.loader.exec_module(_mod)
