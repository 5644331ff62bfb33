"""Import shim: the package directory is `gf-nerf_amd/` (not an importable name), so
`import gfnerf_amd` loads that directory as the package `gfnerf_amd`."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "gf-nerf_amd")
_spec = _ilu.spec_from_file_location("gfnerf_amd", _os.path.join(_pkg_dir, "__init__.py"),
                                     submodule_search_locations=[_pkg_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["gfnerf_amd"] = _mod
_spec.loader.exec_module(_mod)
