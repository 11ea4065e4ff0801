"""Host-side mirror of the reference's models/Modules.py (same class names, constructor arguments and
state_dict keys) on top of the HIP kernels.  These per-module forwards are the COMPATIBILITY path
(`Network(fused=False)`): they compose exactly like the reference graph.  The default fused path in
DCL_Net.py reads the same parameters but runs a restructured pipeline.
"""
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from .. import spconv
from ..libs.pointnet_sp import pointnet2_utils as pointnet2_utils_sp
from ..spconv import SparseAvgPool3d

_ACTS = {"relu": nn.ReLU, "sigmoid": nn.Sigmoid, "tanh": nn.Tanh}


def _act_layer(act):
    if act == "none":
        return None
    if act not in _ACTS:
        raise NotImplementedError
    return _ACTS[act]()


class BasicBlock_SPCONV(nn.Module):
    """sparse conv (SparseConv3d | SubMConv3d) -> BatchNorm1d -> act  (reference Modules.py:12-57)."""

    def __init__(self, subm, dim_in, dim_out, bias, size, stride, padding, norm, act, drop, indice_key):
        super().__init__()
        self.subm, self.dim_in, self.dim_out = subm, dim_in, dim_out
        if subm:
            conv = spconv.SubMConv3d(dim_in, dim_out, size, padding=padding, bias=bias, indice_key=indice_key)
        else:
            conv = spconv.SparseConv3d(dim_in, dim_out, size, (stride,) * 3, padding=padding, bias=bias,
                                       indice_key=indice_key)
        layers = [conv]
        if norm:
            layers.append(nn.BatchNorm1d(dim_out))
        a = _act_layer(act)
        if a is not None:
            layers.append(a)
        if drop > 0:
            layers.append(nn.Dropout(drop))
        self.layers = spconv.SparseSequential(*layers)

    def forward(self, input):
        return self.layers(input)


class _PointwiseConv3d(nn.Conv3d):
    """nn.Conv3d whose 1x1x1 / stride 1 / no padding case (every conv of DCL-Net's dense part) is a batched GEMM
    W (Cout,Cin) @ x (b,Cin,n) instead of a MIOpen convolution -- same parameters, same state_dict keys, differentiable."""

    def forward(self, x):
        if self.kernel_size != (1, 1, 1) or self.stride != (1, 1, 1) or self.padding != (0, 0, 0) or self.groups != 1:
            return super().forward(x)
        b, c = x.shape[0], x.shape[1]
        y = torch.matmul(self.weight.view(self.out_channels, c), x.reshape(b, c, -1))
        if self.bias is not None:
            y = y + self.bias.view(1, -1, 1)
        return y.view(b, self.out_channels, *x.shape[2:])


class _PointwiseConv1d(nn.Conv1d):
    """nn.Conv1d with kernel 1 as a batched GEMM (see _PointwiseConv3d)."""

    def forward(self, x):
        if self.kernel_size != (1,) or self.stride != (1,) or self.padding != (0,) or self.groups != 1:
            return super().forward(x)
        y = torch.matmul(self.weight.view(self.out_channels, self.in_channels), x)
        return y if self.bias is None else y + self.bias.view(1, -1, 1)


class BasicBlock_3DCONV(nn.Module):
    """Conv3d -> BatchNorm3d -> act on (b,C,n,1,1) tensors (reference Modules.py:58-97)."""

    def __init__(self, dim_in, dim_out, bias, size, stride, padding, norm, act, drop):
        super().__init__()
        layers = [_PointwiseConv3d(dim_in, dim_out, size, stride, padding, bias=bias)]
        if norm:
            layers.append(nn.BatchNorm3d(dim_out))
        a = _act_layer(act)
        if a is not None:
            layers.append(a)
        if drop > 0:
            layers.append(nn.Dropout(drop))
        self.layers = nn.Sequential(*layers)

    def forward(self, input):
        return self.layers(input)


class Backbone_SPCONV(nn.Module):
    """4 x [SparseConv3d(k3,s1,p1)+BN+ReLU ; SubMConv3d(k3)+BN+ReLU] each followed by the shared
    SparseAvgPool3d(k3,s2,p1,use_gs=False) (reference Modules.py:100-159)."""

    def __init__(self, dims, stride_layers, cfg, norm=True):
        super().__init__()
        self.downsample_by_pooling = cfg.downsample_by_pooling
        self.dims, self.stride_layers = dims, stride_layers
        block = partial(BasicBlock_SPCONV, bias=False, act="relu", drop=0.0, norm=norm)
        groups = [[] for _ in range(len(stride_layers) + 1)]
        g = 0
        for i in range(len(dims) - 1):
            first_of_group = (i == 0) or ((i - 1) in stride_layers)
            key = ("spconv_" if first_of_group else "subm_spconv_") + str(g)
            # the reference always passes stride=1: down-sampling is done by the pool (Modules.py:138)
            groups[g].append(block(subm=not first_of_group, dim_in=dims[i], dim_out=dims[i + 1],
                                   size=cfg.kernel_size, stride=1, padding=cfg.kernel_size // 2, indice_key=key))
            if i in stride_layers:
                g += 1
        self.module1 = nn.Sequential(*groups[0])
        self.module2 = nn.Sequential(*groups[1])
        self.module3 = nn.Sequential(*groups[2])
        self.module4 = nn.Sequential(*groups[3])
        self.pool = SparseAvgPool3d(kernel_size=cfg.kernel_size, stride=2, padding=cfg.kernel_size // 2, use_gs=False)

    def forward(self, inputs):
        feats1 = self.pool(self.module1(inputs))
        feats2 = self.pool(self.module2(feats1))
        feats3 = self.pool(self.module3(feats2))
        feats4 = self.pool(self.module4(feats3))
        return feats1, feats2, feats3, feats4


class Aligner(nn.Module):
    """Cross-attention without scaling (reference Modules.py:162-169).  This compatibility forward
    returns the attention map, so it has to materialise it; the fused path never does."""

    def __init__(self):
        super().__init__()
        self.softmax = nn.Softmax(dim=1)

    def forward(self, RI_1, RI_2, RE_2):
        attention_map = self.softmax(torch.bmm(RI_2.transpose(1, 2), RI_1))
        return torch.bmm(RE_2, attention_map), attention_map


class Head_MultiLayerPerceptron(nn.Module):
    """Conv1d(k=1) -> act -> [BatchNorm1d] -> [Dropout] per layer (reference Modules.py:173-201)."""

    def __init__(self, list_dim, list_act, list_bn, list_drop):
        super().__init__()
        layers = []
        d_in = list_dim[0]
        for d, act, bn, drop in zip(list_dim[1:], list_act, list_bn, list_drop):
            layers.append(_PointwiseConv1d(d_in, d, 1))
            a = _act_layer(act)
            if a is not None:
                layers.append(a)
            if bn:
                layers.append(nn.BatchNorm1d(d))
            if drop > 0.0:
                layers.append(nn.Dropout(drop))
            d_in = d
        self.layers = nn.Sequential(*layers)

    def forward(self, input):
        return self.layers(input)


def Ops_tensor2points(tensor, offset=(0., -40., -3.), voxel_extent=(.1, .1, .2)):
    """voxel rows -> (features, [b, centre xyz]) (reference Modules.py:204-211)."""
    indices = tensor.indices.float()
    offset = torch.Tensor(offset).to(indices.device)
    voxel_extent = torch.Tensor(voxel_extent).to(indices.device)
    indices[:, 1:] = indices[:, 1:] * voxel_extent + offset + .5 * voxel_extent
    return tensor.features, indices


def Ops_nearest_neighbor_interpolate(target_points, query_points, query_feats, known_seg=None):
    """3-NN inverse-distance interpolation (reference Modules.py:213-227).  known_seg (optional, i32[nbatch+1]): row ranges
    of `query_points` per batch id -- the search then scans a point's own crop only (same results)."""
    dist, idx = pointnet2_utils_sp.three_nn(target_points, query_points, known_seg)
    dist_recip = 1.0 / (dist + 1e-8)
    norm = torch.sum(dist_recip, dim=1, keepdim=True)
    weight = dist_recip / norm
    return pointnet2_utils_sp.three_interpolate(query_feats.contiguous(), idx, weight)


class Ops_GetPointFeat_spconv(nn.Module):
    """per-point multi-scale voxel features (reference Modules.py:228-251)."""

    def __init__(self, scale_lists=[2, 4, 8, 16], unit_voxel_extent=np.array([0.015, 0.015, 0.015]),
                 voxel_num_limit=np.array([64, 64, 64])):
        super().__init__()
        self.scale_lists = scale_lists
        self.unit_voxel_extent = np.asarray(unit_voxel_extent, dtype=np.float64)
        self.voxel_num_limit = voxel_num_limit
        self.offset = -0.5 * self.unit_voxel_extent * np.asarray(voxel_num_limit)

    def forward(self, points, batch_ids, feats1, feats2, feats3, feats4):
        points = torch.cat([batch_ids.view(-1, 1).float(), points], 1).contiguous()
        outs = []
        for scale, feats in zip(self.scale_lists, (feats1, feats2, feats3, feats4)):
            vx_feats, vx_points = Ops_tensor2points(feats, self.offset, self.unit_voxel_extent * scale)
            # voxel rows of a SparseConvTensor are sorted by crop: per-crop row ranges, computed on the device
            edges = torch.arange(int(feats.batch_size) + 1, device=vx_points.device, dtype=vx_points.dtype)
            seg = torch.searchsorted(vx_points[:, 0].contiguous(), edges).int()
            outs.append(Ops_nearest_neighbor_interpolate(points, vx_points.contiguous(), vx_feats, seg))
        return torch.cat(outs, dim=1)
