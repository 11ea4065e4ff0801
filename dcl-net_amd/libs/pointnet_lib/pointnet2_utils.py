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


# ---- grouping modules (libs/pointnet_lib/pointnet2_utils.py:274-386); forward passes only, like the ops above
class QueryAndGroup(torch.nn.Module):
    """Ball query around `new_xyz`, then the neighbours' centre-relative coordinates and/or features:
    (B, 3 + C, npoint, nsample) with use_xyz, features first then xyz -- the reference's channel order (:292-307)."""

    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
        if features is not None and not self.use_xyz:
            return grouping_operation(features, idx)
        # both blocks are gathered straight into the result (no torch.cat copy of the (B,C+3,npoint,nsample) tensor)
        B, npoint, ns = idx.shape
        c = 0 if features is None else features.shape[1]
        out = torch.empty((B, c + 3, npoint, ns), dtype=torch.float32, device=xyz.device)
        if c:
            _ops.group_points(features.contiguous(), idx, out=out[:, :c])
        gx = out[:, c:]
        _ops.group_points(xyz.transpose(1, 2).contiguous(), idx, out=gx)
        gx -= new_xyz.transpose(1, 2).unsqueeze(-1)
        return out


class GroupAll(torch.nn.Module):
    """One group holding every point: (B, 3 + C, 1, N), xyz first (:310-333)."""

    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped_features = features.unsqueeze(2)
        return torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features


class KNNAndGroup(torch.nn.Module):
    """k-nearest-neighbour grouping (:335-386): xyz first, then features.  The reference builds `idx` with
    `knn(xyz, new_xyz, radius, nsample)`, which does not match its own `knn(k, unknown, known)`; here a missing `idx` is
    the `nsample` nearest points of `xyz` around every `new_xyz` centre."""

    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz=None, idx=None, features=None):
        if new_xyz is None:
            new_xyz = xyz
        if idx is None:
            _, idx = knn(self.nsample, new_xyz, xyz)
        idx = idx.detach()
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz -= new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            return grouped_xyz
        grouped_features = grouping_operation(features, idx)
        return torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
