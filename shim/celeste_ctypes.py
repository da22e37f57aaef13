"""The tested counterpart of CelesteMI355X.jl: the ctypes binding of
include/celeste_mi355x.h lives in the package as `celeste_jl_amd.cabi` (structs field for field, every exported
symbol, `Problem` marshalling in the same call sequence as the Julia shim); this module re-exports it under the
name the survey uses."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celeste_jl_amd.cabi import *  # noqa: F401,F403,E402
from celeste_jl_amd.cabi import Problem, load_library, check, EXPORTED_SYMBOLS  # noqa: F401,E402
