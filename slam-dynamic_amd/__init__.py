"""MI355X-native per-frame front end of li-guihai/slam-dynamic (ORB extract, Hamming match,
dynamic-point cull).  The product is lib/libsd_frontend.so (C ABI: include/sd_frontend.h);
this package holds the ctypes plumbing (`frontend`, `yolo`), the synthetic-input generator (`synth`) and the
reference drivers' file formats / per-frame call sequence (`harness`).

The directory name contains a hyphen, so load it through `__graft_entry__.load_package()`,
which registers it as module `slam_dynamic_amd`.
"""
from . import synth  # noqa: F401
from . import frontend  # noqa: F401
from . import yolo  # noqa: F401
from . import harness  # noqa: F401
