"""Importable alias of the package directory
``multi-modal-multi-label-facial-action-unit-detection-with-transformer_amd`` (its name is not a
Python identifier, so ``import avformer_amd`` is the convenient spelling)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)
PACKAGE = "multi-modal-multi-label-facial-action-unit-detection-with-transformer_amd"
_pkg = importlib.import_module(PACKAGE)
sys.modules[__name__] = _pkg
