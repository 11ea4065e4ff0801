"""Synthetic stand-in for one YCB-V test frame (colour, 16-bit depth, label image, PoseCNN rois, CAD clouds): the generator
lives in the package (dcl-net_amd/synth.py::make_frame; bench.py's eval-stream legs use it too) -- re-exported here under the
name the crop tests and the golden generators use."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_synth = importlib.import_module("dcl-net_amd").synth
H, W = _synth.FRAME_H, _synth.FRAME_W
make_scene = _synth.make_frame
