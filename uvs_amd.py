"""Import alias: ``import uvs_amd`` returns the package in ``uncalibrated-visual-servoing_amd/`` (a directory name that
is not a Python identifier).  Use attribute access (``uvs_amd.experiment``) or ``from uvs_amd import experiment``."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
sys.modules[__name__] = importlib.import_module('uncalibrated-visual-servoing_amd')
