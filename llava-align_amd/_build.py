"""Builds libvdd_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).  One object per source, compiled in
parallel and only when stale (objects under build/, which is git-ignored), then one link."""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(ROOT, "build", "obj")
LIB = os.path.join(PKG, "lib", "libvdd_hip.so")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
          "-ffp-contract=off"]   # no FMA contraction: every torch op rounds separately
FLAGS = CFLAGS + ["-shared"]     # (kept for callers that print the full command line)


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm >= 7.0)")
    return exe


def _headers():
    deps = [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    return deps


def _obj_of(src):
    return os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def is_stale() -> bool:
    return _stale(LIB, sources() + _headers())


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    os.makedirs(OBJ, exist_ok=True)
    cc, hdrs = hipcc(), _headers()
    todo = [s for s in sources() if force or _stale(_obj_of(s), [s] + hdrs)]

    def compile_one(src):
        cmd = [cc, *CFLAGS, "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", src, "-o", _obj_of(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=max(1, min(len(todo), os.cpu_count() or 1))) as ex:
        list(ex.map(compile_one, todo))
    # objects of sources that no longer exist must not be linked
    keep = {_obj_of(s) for s in sources()}
    for f in os.listdir(OBJ):
        if os.path.join(OBJ, f) not in keep:
            os.remove(os.path.join(OBJ, f))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", *sorted(keep), "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
