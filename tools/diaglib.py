"""
diaglib.py - BUILD-CONTAINER / GPU-BOX TOOLING: make the scripts under tools/ run on the
measurement build of the engine (qoc_amd/libqocx_diag.so, `make -C qoc_amd/csrc diag`), which
accepts the diagnostic knobs and environment switches the product library leaves out
(qoc_amd/csrc/qocx_diag.h). Call load() before the first Engine is created.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def load(build=True):
    from qoc_amd import engine
    if build and not os.path.exists(engine.DIAG_LIBRARY_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "qoc_amd", "csrc"), "-j8", "diag"])
    lib = engine.load_library(engine.DIAG_LIBRARY_PATH)
    assert lib.qocx_build_is_diag() == 1
    return lib
