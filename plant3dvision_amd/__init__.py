"""Importable alias of the package directory ``plant-3d-vision_amd/`` (its name, fixed by
the repo layout, is not a valid Python identifier).  Submodules resolve there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "plant-3d-vision_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
