#!/usr/bin/env python3
"""Dump the call signatures of every public function / method of the four reference modules this repo mirrors
(evaluator/retrieval.py, criterion.py, utils/preprocess_data.py, utils/utils.py) to tests/golden/signatures.json.

Runs only in the build container (needs /root/reference).  The fixture is DATA (names, parameter names, kinds and the
repr of defaults) -- no reference source text.  tests/test_signatures.py checks the drop-in modules against it.
"""
import inspect
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402  (registers the mteb / tensorboard stubs, puts the reference on sys.path)


def describe(fn):
    out = []
    for p in inspect.signature(fn).parameters.values():
        out.append({"name": p.name, "kind": p.kind.name,
                    "default": None if p.default is inspect.Parameter.empty else repr(p.default)})
    return out


def public_api(mod):
    api = {}
    for name, obj in vars(mod).items():
        if name.startswith("_") or getattr(obj, "__module__", None) != mod.__name__:
            continue
        if inspect.isfunction(obj):
            api[name] = describe(obj)
        elif inspect.isclass(obj):
            for mname, m in vars(obj).items():
                f = m.__func__ if isinstance(m, (staticmethod, classmethod)) else m
                if inspect.isfunction(f) and (not mname.startswith("_") or mname == "__init__"):
                    api[f"{name}.{mname}"] = {"static": isinstance(m, staticmethod), "params": describe(f)}
    return api


def main():
    import_reference()
    import criterion
    import evaluator.retrieval
    import utils.preprocess_data
    import utils.utils
    mods = {"evaluator.retrieval": evaluator.retrieval, "criterion": criterion,
            "utils.preprocess_data": utils.preprocess_data, "utils.utils": utils.utils}
    out = {k: public_api(m) for k, m in mods.items()}
    path = os.path.join(HERE, "signatures.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(path, {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
