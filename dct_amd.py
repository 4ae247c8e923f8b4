"""Import shim: ``import dct_amd`` loads the package that lives in the directory
``deep-co-training-for-semi-supervised-image-segmentation_amd/`` (not a valid Python
identifier, hence this loader)."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "deep-co-training-for-semi-supervised-image-segmentation_amd")
_spec = importlib.util.spec_from_file_location(
    "dct_amd", os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dct_amd"] = _mod
_spec.loader.exec_module(_mod)
