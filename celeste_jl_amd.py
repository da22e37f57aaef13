"""Import alias for the package directory ``celeste.jl_amd/``.

The directory name contains a dot, which Python's import statement cannot
spell, so ``import celeste_jl_amd`` loads ``celeste.jl_amd/__init__.py`` as the
package ``celeste_jl_amd`` (submodules resolve inside that directory).
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "celeste.jl_amd")
_spec = _ilu.spec_from_file_location(
    "celeste_jl_amd", _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["celeste_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
