"""Mirror of libs/pointnet_sp/pointnet2_utils.py:9-86: flat (batch-column) 3-NN + interpolation."""
import torch

from ... import ops as _ops
from ...autograd import ThreeInterpolateFn


def three_nn(unknown, known, known_seg=None):
    """unknown (N,4) [b,x,y,z], known (M,4) -> (dist (N,3) = sqrt(dist2), idx (N,3) int32).
    known_seg (optional, i32[nbatch+1]): row ranges of `known` per batch id (speeds the scan up)."""
    assert unknown.is_contiguous()
    assert known.is_contiguous()
    dist2, idx = _ops.three_nn_sp(unknown, known, known_seg)
    return torch.sqrt(dist2), idx


def three_interpolate(features, idx, weight):
    """features (M,C), idx (n,3), weight (n,3) -> (n,C)."""
    assert features.is_contiguous()
    assert idx.is_contiguous()
    assert weight.is_contiguous()
    return ThreeInterpolateFn.apply(features, idx, weight)
