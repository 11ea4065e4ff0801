"""spconv.ops subset (libs/spconv/spconv/ops.py:19-118,168-188): the op-level boundary of the reference's python package --
what its own conv.py:149-166 / pool.py:237-242 / functional.py:22-174 call, i.e. the Python face of
torch.ops.spconv.{get_indice_pairs_3d, indice_conv_fp32, indice_conv_backward_fp32, indiceSummaryRF, indice_avgpool_fp32,
indice_avgpool_backward_fp32} (src/spconv/all.cc:19-42).  Rulebooks cross this boundary in the REFERENCE format
(indice_pairs (kvol,2,V), indice_pair_num (kvol)); each op converts them to the gather table the kernels walk
(dcl_rulebook_from_pairs, one extra launch, no host read-back)."""
import torch

from .. import ops as _ops


def get_conv_output_size(input_size, kernel_size, stride, padding, dilation):
    out = []
    for i in range(len(input_size)):
        size = (input_size[i] + 2 * padding[i] - dilation[i] * (kernel_size[i] - 1) - 1) // stride[i] + 1
        out.append(1 if kernel_size[i] == -1 else size)
    return out


def _uniform(v, ndim, name):
    v = list(v) if isinstance(v, (list, tuple)) else [v] * ndim
    if any(x != v[0] for x in v):
        raise NotImplementedError("%s must be the same on every axis" % name)
    return int(v[0])


def build_rulebook(aset, ksize, stride, padding, subm):
    """-> (out ActiveSet with host row count, nbr gather table)."""
    if subm:
        return aset, _ops.rulebook_gather(aset, aset, ksize, 1, ksize // 2)
    out = _ops.conv_out_grid(aset, ksize, stride, padding)
    n = int(out.n_dev.item())                       # the one host read-back of this layer
    out.n, out.cap = n, max(n, 1)
    out.indices = out.indices[:n]
    return out, _ops.rulebook_gather(out, aset, ksize, stride, padding)


def get_indice_pairs(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1,
                     out_padding=0, subm=False, transpose=False, grid=None):
    """torch.ops.spconv.get_indice_pairs_3d equivalent: (outids, indice_pairs (27,2,V), indice_num (27))
    in the reference's format (pair order inside an offset unspecified, as in the reference)."""
    if transpose:
        raise NotImplementedError("transposed convolution is not on DCL-Net's path")
    ndim = indices.shape[1] - 1
    if _uniform(dilation, ndim, "dilation") != 1:
        raise NotImplementedError("dilation != 1 is not on DCL-Net's path")
    k, s, p = _uniform(ksize, ndim, "ksize"), _uniform(stride, ndim, "stride"), _uniform(padding, ndim, "padding")
    S = _uniform(spatial_shape, ndim, "spatial_shape")
    aset = _ops.grid_from_indices(indices.int().contiguous(), int(batch_size), S)
    out, nbr = build_rulebook(aset, k, s, p, subm)
    n_out = indices.shape[0] if subm else out.n
    pairs, num = _ops.rulebook_to_pairs(nbr, n_out, indices.shape[0])
    return (indices if subm else out.indices), pairs, num


def _w3(filters):
    """(k,k,k,Cin,Cout) -> (kvol,Cin,Cout) view"""
    return filters.reshape(-1, filters.shape[-2], filters.shape[-1]).contiguous()


def indice_conv(features, filters, indice_pairs, indice_pair_num, num_activate_out, inverse=False, subm=False):
    """ops.py:102-118 -> torch.ops.spconv.indice_conv_fp32 (spconv_ops.h:253-349)."""
    if inverse:
        raise NotImplementedError("inverse sparse convolution is not on DCL-Net's path")
    if filters.dtype != torch.float32:
        raise NotImplementedError("fp32 only (the path computes in fp32)")
    nbr = _ops.rulebook_from_pairs(indice_pairs, indice_pair_num, features.shape[0], num_activate_out)
    return _ops.sparse_conv(features.contiguous(), nbr, int(num_activate_out), _w3(filters), bool(subm))


def indice_conv_backward(features, filters, out_bp, indice_pairs, indice_pair_num, inverse=False, subm=False):
    """ops.py:121-135 -> indice_conv_backward_fp32 (spconv_ops.h:351-438): (input_bp, filters_bp)."""
    if inverse:
        raise NotImplementedError("inverse sparse convolution is not on DCL-Net's path")
    n_out = out_bp.shape[0]
    nbr = _ops.rulebook_from_pairs(indice_pairs, indice_pair_num, features.shape[0], n_out)
    dx, dW = _ops.sparse_conv_backward(features.contiguous(), _w3(filters), out_bp, nbr, n_out, bool(subm))
    return dx, dW.view_as(filters)


def get_indice_summaryrf(indice_pairs, indice_pair_num, num_activate_out):
    """ops.py:168-169 -> torch.ops.spconv.indiceSummaryRF (summaryRF.cu:26-68)."""
    return _ops.indice_summary_rf(indice_pairs, indice_pair_num, num_activate_out)


def indice_avgpool(features, indice_pairs, indice_pair_num, num_activate_out, summaryrf):
    """ops.py:171-179 -> indice_avgpool_fp32 (pool_ops.h:170-208; avgpool.cu:96-176)."""
    if features.dtype != torch.float32:
        raise NotImplementedError("fp32 only (the path computes in fp32)")
    nbr = _ops.rulebook_from_pairs(indice_pairs, indice_pair_num, features.shape[0], num_activate_out)
    return _ops.sparse_avgpool_rf(features, nbr, int(num_activate_out), summaryrf)


def indice_avgpool_backward(features, out_features, out_bp, indice_pairs, indice_pair_num, summaryrf):
    """ops.py:180-188 -> indice_avgpool_backward_fp32 (avgpool.cu:178-206)."""
    n_out = out_bp.shape[0]
    nbr = _ops.rulebook_from_pairs(indice_pairs, indice_pair_num, features.shape[0], n_out)
    return _ops.sparse_avgpool_backward(out_bp, nbr, n_out, features.shape[0], summaryrf.int().contiguous())
