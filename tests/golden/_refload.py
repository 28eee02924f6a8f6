"""Generation-time helper: import pieces of the read-only reference at /root/reference.

Used ONLY by tests/golden/make_golden.py in the build container (the reference does not exist on
the GPU box and nothing under tests/ reads it at test time).  No reference text is copied into
the repo: classes/functions are taken from the reference files in place (importlib for the
side-effect-free modules, AST node selection + exec for the Export_*/Inference_* scripts whose
module level has side effects) and only their numeric inputs/outputs are saved as fixtures.
"""
from __future__ import annotations

import ast
import importlib.util
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

REF = "/root/reference"


def install_stubs(melscale_fbanks):
    """sys.modules stand-ins for the third-party packages the reference imports at module top
    (onnxruntime, onnxslim, torchaudio, funasr.register, pydub)."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mod("onnxruntime")
    mod("onnxslim", slim=lambda *a, **k: None)
    fn = mod("torchaudio.functional", melscale_fbanks=melscale_fbanks)
    mod("torchaudio", functional=fn)

    class _Tables:
        @staticmethod
        def register(*_a, **_k):
            return lambda cls: cls
    mod("funasr")
    mod("funasr.register", tables=_Tables)
    mod("pydub", AudioSegment=object)
    mod("kaldiio")


def load_module(relpath, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def select_nodes(relpath, names, namespace, consts=()):
    """exec the ClassDef/FunctionDef nodes in `names` and the simple constant assignments in
    `consts` from a reference script into `namespace` (module-level side effects are skipped)."""
    path = os.path.join(REF, relpath)
    with open(path, "r", encoding="utf-8") as fh:
        tree = ast.parse(fh.read(), filename=path)
    keep = []
    for node in tree.body:
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)) and node.name in names:
            keep.append(node)
        elif isinstance(node, ast.Assign) and len(node.targets) == 1 and \
                isinstance(node.targets[0], ast.Name) and node.targets[0].id in consts:
            keep.append(node)
    code = compile(ast.Module(body=keep, type_ignores=[]), path, "exec")
    exec(code, namespace)
    return namespace


def select_lines(relpath, first, last, namespace):
    """exec an inclusive 1-based line range of a reference script (used for the module-level
    host loops that are not wrapped in a function)."""
    path = os.path.join(REF, relpath)
    with open(path, "r", encoding="utf-8") as fh:
        lines = fh.readlines()
    src = "\n" * (first - 1) + "".join(lines[first - 1:last])
    exec(compile(src, path, "exec"), namespace)
    return namespace
