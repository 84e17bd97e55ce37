"""Build libdrakegpt_hip.so (gfx950) in-tree with hipcc -- no torch headers, no hipify.

    python -m drakegpt_amd.build [--force] [--save-temps]

The library links against libamdhip64.so.7 by SONAME only (no rpath): at run time it binds to
the HIP runtime that `import torch` has already loaded, so device pointers and streams are
shared with PyTorch-ROCm ("two HIP runtimes in one process").
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
# objects live outside the tree: only the linked .so has to travel to the GPU box
OBJDIR = os.environ.get("DRAKEGPT_OBJDIR", os.path.join(os.environ.get("TMPDIR", "/tmp"), f"drakegpt_amd_build_{os.getuid()}"))
LIB = os.path.join(LIBDIR, "libdrakegpt_hip.so")
ARCH = "gfx950"

SOURCES = [
    "elementwise.hip",
    "layernorm.hip",
    "gemm.hip",
    "gemm_fp8.hip",
    "fp8.hip",
    "attention.hip",
    "attention_simple.hip",
    "attention_mfma.hip",
    "cross_entropy.hip",
    "chain.hip",
    "chain_bwd.hip",
]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force: bool = False, save_temps: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_nt_ws.h"), os.path.join(PKG, "..", "include", "drakegpt_hip.h")]
    hdr_digest = _digest(headers)
    flags = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17",
             "-Wno-unused-result", "-I", os.path.join(PKG, "..", "include")]
    if save_temps:
        flags += ["-save-temps=obj"]

    def compile_one(src: str):
        sp = os.path.join(CSRC, src)
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        stamp = obj + ".sha"
        dig = _digest([sp]) + hdr_digest + " ".join(flags)
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
            return obj, False
        cmd = [hipcc, *flags, "-c", sp, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip() and verbose:
            sys.stderr.write(r.stderr)
        with open(stamp, "w") as f:
            f.write(dig)
        return obj, True

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        results = list(ex.map(compile_one, SOURCES))
    objs = [o for o, _ in results]
    rebuilt = any(c for _, c in results)
    if rebuilt or force or not os.path.exists(LIB):
        # link with the host C++ driver: hipcc's own link step hard-wires -rpath /opt/rocm/lib, which
        # could pull a second HIP runtime into a process where torch already loaded its own
        rocm_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib")
        cmd = ["g++", "-shared", "-fPIC", "-o", LIB, *objs, "-L" + rocm_lib, "-lamdhip64", "-Wl,--no-undefined"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[drakegpt_amd.build] linked {LIB}")
    elif verbose:
        print(f"[drakegpt_amd.build] up to date: {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv)
