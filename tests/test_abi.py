"""CPU-only checks of the drop-in boundary: the header, the ctypes table and the built
library agree symbol for symbol, and the product refuses to run without a GPU."""
import ctypes
import os
import re

import pytest

from faster_rcnn_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = ""
    inc = os.path.join(ROOT, "include")
    for f in sorted(os.listdir(inc)):
        if f.endswith(".h"):
            txt += open(os.path.join(inc, f)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(frcnn_[a-z0-9_]+)\s*\(", txt)))


def test_header_matches_ctypes_table():
    assert header_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_symbol():
    from faster_rcnn_amd.build import build_library
    build_library(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), name
    loaded = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "frcnn_hip.h")).read()
    assert loaded.frcnn_version() == _lib.ABI_VERSION == int(re.search(r"#define FRCNN_ABI_VERSION (\d+)", hdr).group(1))


def test_header_cites_reference_lines():
    txt = open(os.path.join(ROOT, "include", "frcnn_hip.h")).read()
    for needle in ("rpn_util.py:276", "util.py:146", "det_util.py:209", "custom_layers.py:35", "rpn_util.py:54"):
        assert needle in txt


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from faster_rcnn_amd import ops
    with pytest.raises(_lib.FrcnnError):
        ops.anchors_image(2, 2, [[16, 16]], 16)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "faster_rcnn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
