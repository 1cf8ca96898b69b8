"""Importable alias for the hyphen-named package directory
`representation-disentanglement_amd/` (a hyphen cannot appear in an `import`
statement).  `import mrdis` yields that package object itself."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module('representation-disentanglement_amd')
sys.modules[__name__] = _pkg
