"""The Python shown in README.md and INTEGRATION.md is at least valid Python, and every `evdr_amd...` name the snippets import
exists (the ctypes stub of INTEGRATION.md §2 is also EXECUTED on the GPU: tests/test_gpu_cabi.py)."""
import importlib
import os
import re

import pytest

import evdr_amd  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOCKS = [(doc, i, b) for doc in ("README.md", "INTEGRATION.md")
          for i, b in enumerate(re.findall(r"```python\n(.*?)```", open(os.path.join(ROOT, doc)).read(), flags=re.S))]


@pytest.mark.parametrize("doc,i,block", BLOCKS, ids=[f"{d}#{i}" for d, i, _ in BLOCKS])
def test_python_blocks_compile_and_their_imports_resolve(doc, i, block):
    import ast
    tree = ast.parse(block, f"{doc}#{i}")
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.module and node.module.startswith("evdr_amd"):
            m = importlib.import_module(node.module)
            for alias in node.names:
                assert hasattr(m, alias.name), f"{doc}: {node.module} has no {alias.name}"


def test_files_the_documents_point_to_exist():
    for doc in ("README.md", "DESIGN.md", "INTEGRATION.md", "profiles/README.md"):
        text = open(os.path.join(ROOT, doc)).read()
        for path in set(re.findall(r"`((?:profiles|scratch|tests|oracle|include)/[\w./-]+\.(?:py|json|csv|txt|sh|hip|h|log|npz|cpp|c))`", text)):
            assert os.path.exists(os.path.join(ROOT, path)), f"{doc} mentions {path}, which does not exist"
