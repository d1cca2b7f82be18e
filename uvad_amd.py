"""Import alias: the package directory name contains hyphens, so ``import uvad_amd`` resolves it."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
sys.modules[__name__] = importlib.import_module("universal-voice-activity-detection_amd")
