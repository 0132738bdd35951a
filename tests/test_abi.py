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
        names += re.findall(r"^\s*(?:const\s+)?(?:int|void|char\s*\*|const char\s*\*)\s*\*?\s*(vdd_\w+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_exports_every_declared_symbol(lib):
    names = declared_functions()
    assert "vdd_contrast_sample" in names and "vdd_add_diffusion_noise" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ but not exported"


def test_abi_version(lib):
    from llava_align_amd._lib import ABI_VERSION
    assert lib.vdd_abi_version() == ABI_VERSION == 2
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
