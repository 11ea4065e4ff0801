"""torch.autograd Functions of the ops on DCL-Net's path, mirroring the reference's Function classes
(libs/spconv/spconv/functional.py:20-166, libs/pointnet_sp/pointnet2_utils.py:41-86,
libs/pointgroup_ops/functions/pointgroup_ops.py:42-75): forward = the inference kernels, backward = csrc/backward.hip.
Used by the module mirrors (spconv/, libs/) so that `Network(cfg, mode='train')` is trainable on the GPU."""
import torch
from torch.autograd import Function

from . import ops as _ops


class SparseConvFn(Function):
    """SparseConvFunction / SubMConvFunction (functional.py:20-88)."""

    @staticmethod
    def forward(ctx, features, W, nbr, n_out, subm):
        ctx.save_for_backward(features, W, nbr)
        ctx.n_out, ctx.subm = int(n_out), bool(subm)
        return _ops.sparse_conv(features, nbr, n_out, W, subm)

    @staticmethod
    def backward(ctx, grad_output):
        features, W, nbr = ctx.saved_tensors
        dx, dW = _ops.sparse_conv_backward(features, W, grad_output, nbr, ctx.n_out, ctx.subm,
                                           need_dx=ctx.needs_input_grad[0])
        return dx, dW.view_as(W), None, None, None


class SparseAvgPoolFn(Function):
    """SparseAvgPoolFunction with use_gs=False (functional.py:137-166)."""

    @staticmethod
    def forward(ctx, features, nbr, n_out):
        out, rf = _ops.sparse_avgpool(features, nbr, n_out, want_rf=True)
        ctx.save_for_backward(nbr, rf)
        ctx.n_out, ctx.n_in = int(n_out), features.shape[0]
        return out

    @staticmethod
    def backward(ctx, grad_output):
        nbr, rf = ctx.saved_tensors
        return _ops.sparse_avgpool_backward(grad_output, nbr, ctx.n_out, ctx.n_in, rf), None, None


class ThreeInterpolateFn(Function):
    """ThreeInterpolate of libs/pointnet_sp (pointnet2_utils.py:41-86)."""

    @staticmethod
    def forward(ctx, features, idx, weight):
        ctx.save_for_backward(idx, weight)
        ctx.m = features.shape[0]
        return _ops.three_interpolate_sp(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        return _ops.three_interpolate_grad_sp(grad_out, idx, weight, ctx.m), None, None


class VoxelizationFn(Function):
    """Voxelization (pointgroup_ops.py:42-75)."""

    @staticmethod
    def forward(ctx, feats, map_rule, mode=4):
        ctx.save_for_backward(map_rule)
        ctx.mode, ctx.n = mode, feats.shape[0]
        return _ops.voxelize_fp(feats, map_rule, mode)

    @staticmethod
    def backward(ctx, d_out):
        (map_rule,) = ctx.saved_tensors
        return _ops.voxelize_bp(d_out, map_rule, ctx.n, ctx.mode), None, None
