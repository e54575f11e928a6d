"""CPU test: libqocx.so loads and exports every symbol include/qocx.h declares (no GPU call)."""

import os
import re

from qoc_amd import engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "qocx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qocx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = engine.load_library()
    names = declared_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(engine.SIGNATURES) == names
    assert lib.qocx_version() >= 100


def test_missing_library_fails_loudly(tmp_path):
    import pytest
    with pytest.raises(ImportError):
        engine.load_library(str(tmp_path / "absent.so"))
