"""Mirror of libs/pointgroup_ops/functions/pointgroup_ops.py:11-75 (the two ops DCL-Net uses)."""
import torch

from .... import ops as _ops
from ....autograd import VoxelizationFn


def voxelization_idx(coords, batchsize, mode=4):
    """Voxelization_Idx.apply: coords long (N, 3|4) on the HOST ->
    (output_coords long (M, ncol), input_map int (N), output_map int (M, maxActive+1))."""
    assert coords.is_contiguous()
    return _ops.voxelize_idx(coords, batchsize, mode)


def voxelization(feats, map_rule, mode=4):
    """Voxelization.apply: feats cuda float (N,C), map_rule cuda int (M, maxActive+1) -> cuda float (M,C)."""
    assert map_rule.is_contiguous()
    assert feats.is_contiguous()
    return VoxelizationFn.apply(feats, map_rule, mode)
