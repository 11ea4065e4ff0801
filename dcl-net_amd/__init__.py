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
from . import _native, ops  # noqa: F401
from . import spconv  # noqa: F401
from . import synth, sharding, crops, autograd  # noqa: F401
from .models import DCL_Net, Modules, refiner  # noqa: F401

build = _native.build
__all__ = ["ops", "spconv", "DCL_Net", "Modules", "refiner", "synth", "sharding", "crops", "build"]
