"""Importable alias of the `3d_poseestimation_amd` package (a directory name that starts
with a digit cannot follow `import`)."""
import importlib
import sys

_pkg = importlib.import_module("3d_poseestimation_amd")
sys.modules[__name__] = _pkg
