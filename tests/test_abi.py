"""CPU: the C-ABI library loads (against torch's HIP runtime, no GPU needed) and exports every
symbol include/drakegpt_hip.h declares; the ctypes signatures agree with the header."""
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "drakegpt_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|int64_t|const char\*)\s+(dg_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        out[m.group(1)] = n
    return out


def test_library_loads_and_exports_every_declared_symbol():
    from drakegpt_amd import _lib
    decl = _declared()
    assert len(decl) >= 20
    for name in decl:
        assert hasattr(_lib.lib, name), f"{name} declared in the header but not exported"
    assert _lib.lib.dg_version() == _lib.ABI_VERSION
    assert b"invalid argument" in _lib.lib.dg_error_string(-1)


def test_ctypes_signatures_match_header():
    from drakegpt_amd import _lib
    decl = _declared()
    for name, argtypes in _lib.SIGNATURES.items():
        assert name in decl, f"{name} bound in python but not declared in the header"
        assert len(argtypes) == decl[name], (name, len(argtypes), decl[name])
    missing = set(decl) - set(_lib.SIGNATURES) - {"dg_error_string"}
    assert not missing, missing


def test_library_binds_to_torch_hip_runtime():
    """no rpath to /opt/rocm: the kernels must share the HIP runtime torch loaded."""
    import subprocess
    from drakegpt_amd import _lib
    out = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "libamdhip64.so" in out
    assert "RPATH" not in out and "RUNPATH" not in out
    maps = open("/proc/self/maps").read()
    hips = {line.split()[-1] for line in maps.splitlines() if "libamdhip64" in line}
    assert len(hips) == 1, hips
