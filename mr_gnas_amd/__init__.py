"""Import shim: the package sources live in the directory ``mr-gnas_amd/`` (a
name Python cannot import directly); this module makes them importable as
``mr_gnas_amd`` by pointing the package path there and running its __init__."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "mr-gnas_amd")
__path__[:] = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
