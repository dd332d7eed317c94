"""Import alias: makes `efficient-visual-document-retrieval_amd/` (not a valid Python identifier) importable
as the package `evdr_amd`.  `import evdr_amd` from the repo root (or with the repo root on sys.path)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "efficient-visual-document-retrieval_amd")
_spec = importlib.util.spec_from_file_location("evdr_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["evdr_amd"] = _mod
_spec.loader.exec_module(_mod)
