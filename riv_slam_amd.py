"""Import alias: the package directory is `riv-slam_amd/` (hyphen), which `import` cannot spell."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("riv-slam_amd")
