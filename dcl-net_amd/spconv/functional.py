"""spconv.functional subset (libs/spconv/spconv/functional.py:20-174): the autograd Functions the reference's conv.py /
pool.py apply -- `indice_conv`, `indice_subm_conv`, `indice_avgpool` -- with the reference's argument lists, on top of
spconv.ops (pair-format rulebooks).  The module mirrors in this package use the gather-table Functions of autograd.py
instead (no format conversion); these exist so that a caller which keeps the reference's own conv.py / pool.py binds."""
from torch.autograd import Function

from . import ops


class SparseConvFunction(Function):
    """functional.py:22-43"""

    @staticmethod
    def forward(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out):
        ctx.save_for_backward(indice_pairs, indice_pair_num, features, filters)
        return ops.indice_conv(features, filters, indice_pairs, indice_pair_num, num_activate_out, False)

    @staticmethod
    def backward(ctx, grad_output):
        indice_pairs, indice_pair_num, features, filters = ctx.saved_tensors
        input_bp, filters_bp = ops.indice_conv_backward(features, filters, grad_output.contiguous(), indice_pairs,
                                                        indice_pair_num, False)
        return input_bp, filters_bp, None, None, None


class SubMConvFunction(Function):
    """functional.py:69-91"""

    @staticmethod
    def forward(ctx, features, filters, indice_pairs, indice_pair_num, num_activate_out):
        ctx.save_for_backward(indice_pairs, indice_pair_num, features, filters)
        return ops.indice_conv(features, filters, indice_pairs, indice_pair_num, num_activate_out, False, True)

    @staticmethod
    def backward(ctx, grad_output):
        indice_pairs, indice_pair_num, features, filters = ctx.saved_tensors
        input_bp, filters_bp = ops.indice_conv_backward(features, filters, grad_output.contiguous(), indice_pairs,
                                                        indice_pair_num, False, True)
        return input_bp, filters_bp, None, None, None


class SparseAvgPoolFunction(Function):
    """functional.py:137-166: use_gs=False divides by the receptive-field count, use_gs=True by the kernel volume"""

    @staticmethod
    def forward(ctx, features, indice_pairs, indice_pair_num, num_activate_out, use_gs=True):
        if not use_gs:
            summaryrf = ops.get_indice_summaryrf(indice_pairs, indice_pair_num, num_activate_out)
        else:
            summaryrf = indice_pairs.new_zeros(num_activate_out) + int(indice_pairs.shape[0])
        out = ops.indice_avgpool(features, indice_pairs, indice_pair_num, num_activate_out, summaryrf)
        ctx.save_for_backward(indice_pairs, indice_pair_num, features, out, summaryrf)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        indice_pairs, indice_pair_num, features, out, summaryrf = ctx.saved_tensors
        input_bp = ops.indice_avgpool_backward(features, out, grad_output.contiguous(), indice_pairs, indice_pair_num,
                                               summaryrf)
        return input_bp, None, None, None, None


indice_conv = SparseConvFunction.apply
indice_subm_conv = SubMConvFunction.apply
indice_avgpool = SparseAvgPoolFunction.apply
