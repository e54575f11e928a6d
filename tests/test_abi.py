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


# ---- struct layouts: include/qocx.h == qoc_amd/engine.py == the stub in INTEGRATION.md ------------

_C_TO_CTYPES = {"int32_t": "c_int32", "double": "c_double", "const double*": "_dp",
                "const int32_t*": "_ip", "const qocx_cost_desc*": "POINTER(CostDesc)"}


def header_struct_fields(name):
    """[(field, ctypes-spelling)] of `typedef struct <name> {...}` in include/qocx.h."""
    text = open(os.path.join(ROOT, "include", "qocx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"(.*?)\s*([a-z0-9_]+)$", decl)
        ctype = m.group(1).replace(" *", "*")
        fields.append((m.group(2), _C_TO_CTYPES[ctype]))
    return fields


def python_fields(source, class_name):
    """[(field, ctypes-spelling)] of `class <class_name>(ctypes.Structure)` in Python source text."""
    block = re.search(r"class %s\(ctypes\.Structure\):.*?_fields_ = \[(.*?)\]\n" % class_name,
                      source, flags=re.S).group(1)
    block = re.sub(r"#.*", "", block)
    out = []
    for field, spelling in re.findall(r'\("([a-z0-9_]+)",\s*([^()]+(?:\([^()]*\))?)\)', block):
        spelling = spelling.strip().replace("ctypes.", "")
        spelling = {"_c_double_p": "_dp", "_c_int_p": "_ip",
                    "POINTER(_CostDesc)": "POINTER(CostDesc)"}.get(spelling, spelling)
        out.append((field, spelling))
    return out


def integration_stub():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return "\n".join(re.findall(r"```python\n(.*?)```", text, flags=re.S))


def test_struct_layouts_agree_between_header_binding_and_integration_doc():
    engine_src = open(os.path.join(ROOT, "qoc_amd", "engine.py")).read()
    stub = integration_stub()
    for c_name, engine_cls, stub_cls in (
            ("qocx_cost_desc", "_CostDesc", "CostDesc"),
            ("qocx_schroedinger_problem", "_SchroedingerProblem", "Problem"),
            ("qocx_lindblad_problem", "_LindbladProblem", "LindbladProblem")):
        want = header_struct_fields(c_name)
        assert len(want) >= 5
        assert python_fields(engine_src, engine_cls) == want, c_name
        assert python_fields(stub, stub_cls) == want, c_name
        # and the live ctypes class really has that layout
        live = getattr(engine, engine_cls)
        assert [f[0] for f in live._fields_] == [f[0] for f in want]


def test_integration_stub_structs_have_the_size_the_library_expects():
    """Execute the struct definitions of the INTEGRATION.md stub (no library call) and compare
    their sizes with the binding's: a stale stub would be rejected by struct_size, not over-read."""
    import ctypes
    stub = integration_stub()
    ns = {"ctypes": ctypes, "_dp": ctypes.POINTER(ctypes.c_double),
          "_ip": ctypes.POINTER(ctypes.c_int32)}
    for cls in ("CostDesc", "Problem", "LindbladProblem"):
        src = re.search(r"(class %s\(ctypes\.Structure\):.*?\]\n)" % cls, stub, flags=re.S).group(1)
        exec(src, ns)
    assert ctypes.sizeof(ns["CostDesc"]) == ctypes.sizeof(engine._CostDesc)
    assert ctypes.sizeof(ns["Problem"]) == ctypes.sizeof(engine._SchroedingerProblem)
    assert ctypes.sizeof(ns["LindbladProblem"]) == ctypes.sizeof(engine._LindbladProblem)
