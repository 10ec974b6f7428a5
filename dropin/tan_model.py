"""Bare-name shim: `import tan_model` (as the reference's drivers do, train/main.py:16,21) -> temporalalignnet_amd.tan_model."""
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from temporalalignnet_amd.tan_model import *  # noqa: E402,F401,F403
from temporalalignnet_amd import tan_model as _impl  # noqa: E402

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
