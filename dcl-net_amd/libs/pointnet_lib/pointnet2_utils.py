"""Mirror of libs/pointnet_lib/pointnet2_utils.py:10-271 (forward passes of the PointNet++ primitives)."""
import torch

from ... import ops as _ops


def furthest_point_sample(xyz, npoint):
    """xyz (B,N,3) -> (B,npoint) int32; starts at index 0 (sampling_gpu.cu:111-113)."""
    return _ops.furthest_point_sampling(xyz.contiguous(), npoint)


def gather_operation(features, idx):
    """features (B,C,N), idx (B,npoint) -> (B,C,npoint)."""
    return _ops.gather_points(features.contiguous(), idx.contiguous())


def knn(k, unknown, known):
    """(B,N,3),(B,M,3) -> (dist (B,N,k), idx (B,N,k) int32), k <= 200."""
    dist2, idx = _ops.knn(k, unknown.contiguous(), known.contiguous())
    return torch.sqrt(dist2), idx


def three_nn(unknown, known):
    dist2, idx = _ops.three_nn(unknown.contiguous(), known.contiguous())
    return torch.sqrt(dist2), idx


def three_interpolate(features, idx, weight):
    """features (B,C,M), idx (B,n,3), weight (B,n,3) -> (B,C,n)."""
    return _ops.three_interpolate(features.contiguous(), idx.contiguous(), weight.contiguous())


def grouping_operation(features, idx):
    """features (B,C,N), idx (B,npoint,nsample) -> (B,C,npoint,nsample)."""
    return _ops.group_points(features.contiguous(), idx.contiguous().int())


def ball_query(radius, nsample, xyz, new_xyz):
    """xyz (B,N,3), new_xyz (B,npoint,3) -> idx (B,npoint,nsample) int32."""
    return _ops.ball_query(radius, nsample, xyz.contiguous(), new_xyz.contiguous())
