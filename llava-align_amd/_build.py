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


# the model kernels are compiled once per storage type (csrc/vdd_elem.h): -DVDD_ELEM = the vdd_dtype value
PER_DTYPE = ("vdd_llm_kernels.hip", "vdd_prefill_kernels.hip", "vdd_gemm.hip")
ELEMS = (("bf16", 2), ("f16", 1))


def units():
    """(source, object, extra flags) of every compilation unit."""
    out = []
    for src in sources():
        base = os.path.basename(src)
        if base in PER_DTYPE:
            out += [(src, os.path.join(OBJ, f"{base[:-4]}.{tag}.o"), [f"-DVDD_ELEM={val}"]) for tag, val in ELEMS]
        else:
            out.append((src, os.path.join(OBJ, base[:-4] + ".o"), []))
    return out


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
    todo = [u for u in units() if force or _stale(u[1], [u[0]] + hdrs)]

    def compile_one(unit):
        src, obj, extra = unit
        cmd = [cc, *CFLAGS, *extra, "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=max(1, min(len(todo), os.cpu_count() or 1))) as ex:
        list(ex.map(compile_one, todo))
    # objects of sources that no longer exist must not be linked
    keep = {u[1] for u in units()}
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
