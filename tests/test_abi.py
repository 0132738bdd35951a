"""The C-ABI shared library loads, exports every symbol include/*.h declares, and the
ctypes mirror of `vdd_sample_params` has the C compiler's layout.  No GPU needed."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from importlib import import_module
    import_module("llava_align_amd._build").build_lib()
    import llava_align_amd as L
    return L.load_lib()


def declared_functions():
    names = []
    for fn in os.listdir(os.path.join(ROOT, "include")):
        src = open(os.path.join(ROOT, "include", fn)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"^\s*(?:const\s+)?(?:int64_t|int|void|char\s*\*|const char\s*\*)\s*\*?\s*(vdd_\w+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_exports_every_declared_symbol(lib):
    names = declared_functions()
    assert "vdd_contrast_sample" in names and "vdd_add_diffusion_noise" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ but not exported"


def test_exports_nothing_but_the_declared_symbols(lib):
    """The per-dtype instantiations (`*_bf16` / `*_f16`, csrc/vdd_elem.h) are hidden: the dynamic symbol table holds the header's
    entries and nothing else."""
    from llava_align_amd._lib import lib_path
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path()], check=True, capture_output=True, text=True).stdout
    exported = sorted(l.split()[2] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] == "T" and l.split()[2].startswith("vdd_"))
    assert exported == declared_functions()


def _declared_params():
    """name -> list of parameter declarations of every function in include/vdd_hip.h."""
    src = open(os.path.join(ROOT, "include", "vdd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"^\s*(?:int64_t|int)\s+(vdd_\w+)\s*\((.*?)\)\s*;", src, flags=re.M | re.S):
        out[m.group(1)] = [a.strip() for a in m.group(2).split(",")]
    return out


def test_ctypes_signatures_match_the_header():
    """Every binding in ops._SIGS has the header's arity and pointer / integer / float kinds, and every model entry takes the storage
    type right in front of the stream (ABI 3)."""
    from llava_align_amd import ops
    decl = _declared_params()
    kind = {ops._P: "p", ops._I: "i", ops._L: "l", ops._F: "f", C.c_uint32: "i"}
    for name, sig in ops._SIGS.items():
        params = decl[name]
        assert len(params) == len(sig), (name, len(params), len(sig))
        for prm, ct in zip(params, sig):
            want = "p" if "*" in prm else ("f" if prm.startswith("float") else ("l" if prm.startswith("int64_t") else "i"))
            assert kind[ct] == want, (name, prm, ct)
        if name not in ("vdd_stop_words_match", "vdd_repetition_penalty"):
            assert params[-2] == "int dtype" and params[-1].startswith("void*"), (name, params[-2:])


def test_model_entries_refuse_other_storage_types(lib):
    """dtype is checked before anything is launched: fp32 (or garbage) -> VDD_ERR_INVALID_ARG, no GPU needed."""
    from llava_align_amd import ops
    ops._lib_ready()
    for bad in (0, 3, -1):
        assert lib.vdd_silu_mul(None, None, 4, 8, bad, None) == -1
        assert lib.vdd_add(None, None, None, 8, bad, None) == -1
    assert lib.vdd_silu_mul(None, None, 0, 8, 2, None) == 0 and lib.vdd_silu_mul(None, None, 0, 8, 1, None) == 0     # empty: nothing to do


def test_abi_version(lib):
    from llava_align_amd._lib import ABI_VERSION
    assert lib.vdd_abi_version() == ABI_VERSION == 3
    assert lib.vdd_lds_row_capacity(1) >= 32000 and lib.vdd_lds_row_capacity(0) >= 32000
    assert lib.vdd_kernel_name(2, 32000).decode() == "vdd_contrast_sample_kernel"


def test_struct_layout_matches_c(tmp_path):
    from llava_align_amd._lib import VddSampleParams
    fields = [f[0] for f in VddSampleParams._fields_]
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "vdd_hip.h"', "int main(void){",
            'printf("%zu\\n", sizeof(vdd_sample_params));']
    prog += [f'printf("%zu\\n", offsetof(vdd_sample_params, {f}));' for f in fields]
    prog += ["return 0;}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(prog))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) == C.sizeof(VddSampleParams)
    for f, off in zip(fields, out[1:]):
        assert getattr(VddSampleParams, f).offset == int(off), f


def test_invalid_arguments_return_status_not_crash(lib):
    from llava_align_amd._lib import VddSampleParams
    assert lib.vdd_contrast_sample(None, None) == -1
    p = VddSampleParams()
    p.abi_version = 99
    assert lib.vdd_contrast_sample(C.byref(p), None) == -1
    assert b"abi_version" in lib.vdd_last_error()
    from llava_align_amd._lib import ABI_VERSION
    p.abi_version = ABI_VERSION
    p.B, p.V = 1, 0
    assert lib.vdd_contrast_sample(C.byref(p), None) == -1
    p.B, p.V = 0, 10
    assert lib.vdd_contrast_sample(C.byref(p), None) == 0      # empty batch: nothing to do
