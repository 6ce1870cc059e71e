"""`snowvision` on MI355X: the package name SnowMocap's main.py imports (`from snowvision import *`, main.py:6),
served by snowmocap_amd.  Put this directory on PYTHONPATH *before* the reference checkout:

    PYTHONPATH=/path/to/this/repo/compat:/path/to/this/repo python main.py

Same names, signatures and result schema as snowvision/__init__.py:1-4 re-exports (camera, util, triangulation,
blender); the triangulation path runs through libsnowtri.so (include/snowtri.h), see INTEGRATION.md.
"""
from snowmocap_amd import *          # noqa: F401,F403
from snowmocap_amd import __all__    # noqa: F401
from snowmocap_amd import camera, triangulation, blender, util   # noqa: F401  (snowvision.camera, ... submodule names)
