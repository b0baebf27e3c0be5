"""CPU: libgprhip.so builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports every
function include/gpr_hip.h declares.  No compute is called here."""
import ctypes
import os
import re
import subprocess
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
HEADER = os.path.join(ROOT, "include", "gpr_hip.h")
LIB = os.path.join(ROOT, "gpr_calculator_amd", "libgprhip.so")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b([a-z_0-9]+)\s*\([^;{]*\)\s*;", text)
    return sorted(set(n for n in names if n.startswith(("gpr_", "rbf_", "dot_"))))


@pytest.fixture(scope="module")
def built_lib():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "gpr_calculator_amd", "csrc"), "-j4"])
    import torch  # noqa: F401  -- like the package: torch's HIP runtime is resolved first, the library binds to the same one
    return ctypes.CDLL(LIB)


def test_header_declares_the_13_reference_natives():
    names = declared_functions()
    natives = [n for n in names if n.startswith(("rbf_", "dot_"))]
    assert len(natives) == 13, natives


def test_every_declared_symbol_is_exported(built_lib):
    missing = [n for n in declared_functions() if not hasattr(built_lib, n)]
    assert not missing, missing


def test_build_info_and_error_channel(built_lib):
    built_lib.gpr_build_info.restype = ctypes.c_char_p
    assert b"gfx950" in built_lib.gpr_build_info()
    built_lib.gpr_last_error.restype = ctypes.c_char_p
    assert built_lib.gpr_last_error() is not None


def test_bad_arguments_are_reported_not_executed(built_lib):
    """argument validation happens on the host before any HIP call"""
    built_lib.gpr_last_error.restype = ctypes.c_char_p
    out = ctypes.c_void_p()
    counts = (ctypes.c_int * 1)(3)
    rc = built_lib.gpr_set_create(ctypes.byref(out), None, 0, 129, 3, 0, 3, 1, counts, None, None, None, 0)
    assert rc == -3 and b"unsupported" in built_lib.gpr_last_error()      # d = 129 > 128 (three plane widths: 32, 64 and 128 columns)
    rc = built_lib.gpr_set_create(ctypes.byref(out), None, 0, 30, 3, 0, 5, 1, counts, None, None, None, 0)
    assert rc == -2 and b"null device pointer" in built_lib.gpr_last_error()
