"""spconv.ops subset (libs/spconv/spconv/ops.py:19-118,168-188)."""
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
