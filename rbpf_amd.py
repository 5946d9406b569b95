"""Alias: `import rbpf_amd` == the package in ./rao-blackwellized-slam-smoothing_amd/ (hyphenated name)."""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_pkg = importlib.import_module("rao-blackwellized-slam-smoothing_amd")
sys.modules[__name__] = _pkg
