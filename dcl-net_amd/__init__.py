"""dcl-net_amd: MI355X-native implementation of DCL-Net's per-crop RGB-D -> 6-DoF pose forward.

Package layout (only what the hot path needs):
  csrc/            hand-written HIP kernels for gfx950 + the C ABI (include/dclnet_hip.h) -> libdclnet_hip.so
  _native.py       ctypes loader (fails loudly when the library is missing; there is no CPU fallback)
  ops.py           torch-tensor front end of the C ABI
  spconv/ libs/    mirrors of the reference's extension-module Python APIs (spconv, pointnet_sp,
                   pointnet_lib, pointgroup_ops) on top of ops.py
  models/          DCL_Net.Network, Modules, refiner.Refiner: drop-ins for the reference's models/*.py
  synth.py         procedural YCB-V-shaped crops + seeded weights (no datasets/checkpoints offline)
  sharding.py      frame sharding across ranks + exact ADD-S metric reduction
  crops.py         device-side crop builder (the loader step in front of forward): image -> the `data` dict in HBM

The directory name contains a hyphen (it is the project's name); import it with
    import importlib; dcl = importlib.import_module("dcl-net_amd")
"""
import os as _os

# The HIP runtime deals streams -- and the streams it makes for the parallel branches of a hipGraph -- onto GPU_MAX_HW_QUEUES
# hardware queues (default 4).  A whole-forward graph whose second branch lands on another queue than the launch stream replays
# 1.3x (32 crops) to 4x (one crop) slower for as long as it lives: every fork / join edge becomes a cross-queue dependency
# (tools/recapture_probe.py: with 4 queues one capture in two to four is fast, with 2 every one, with 16 none; launch by launch the
# stress step runs at the same speed with 1, 2 or 4).  The variable is read when the runtime initialises, i.e. it takes effect when
# this package is imported before the process first touches the GPU; a value the host set itself is respected.  Where it comes
# too late, Network._select_capture still keeps slow captures out.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

from . import _native, ops  # noqa: F401,E402
from . import spconv  # noqa: F401,E402
from . import synth, sharding, crops, autograd  # noqa: F401,E402
from .models import DCL_Net, Modules, refiner  # noqa: F401,E402

build = _native.build
__all__ = ["ops", "spconv", "DCL_Net", "Modules", "refiner", "synth", "sharding", "crops", "build"]
