"""Import alias: the package directory name contains hyphens, so ``import uvad_amd`` (and
``from uvad_amd.<submodule> import ...``) resolve to ``universal-voice-activity-detection_amd``.
Every submodule is registered under both names so there is exactly one module object each."""
import importlib
import os
import pkgutil
import sys

_REAL = "universal-voice-activity-detection_amd"
_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module(_REAL)
for _m in pkgutil.iter_modules(_pkg.__path__):
    if not os.path.exists(os.path.join(_pkg.__path__[0], _m.name + ".py")):
        continue  # libuvad.so is a C-ABI library, not a Python extension module
    sys.modules[f"{__name__}.{_m.name}"] = importlib.import_module(f"{_REAL}.{_m.name}")
sys.modules[__name__] = _pkg
