"""Importable alias of the package directory `gr-lora_amd/` (a hyphen is not a Python identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("gr-lora_amd")
sys.modules[__name__] = _pkg
