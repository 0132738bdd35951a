"""Import shim: the real package lives in `llava-align_amd/` (the directory name the
project layout prescribes, which is not a valid Python identifier).  `import
llava_align_amd` executes that directory's __init__ with its sub-modules resolvable."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "llava-align_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
