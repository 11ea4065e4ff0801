"""SparseConvolution / SparseConv3d / SubMConv3d (libs/spconv/spconv/conv.py:51-174,208-262)."""
import math

import numpy as np
import torch
from torch.nn import init
from torch.nn.parameter import Parameter

from .. import ops as _ops
from ..autograd import SparseConvFn
from . import ops
from .modules import SparseModule
from .tensor import SparseConvTensor


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, output_padding=0, transposed=False, inverse=False, indice_key=None):
        super(SparseConvolution, self).__init__()
        assert groups == 1
        if transposed or inverse:
            raise NotImplementedError("transposed / inverse sparse convolution is not on DCL-Net's path")

        def tolist(v):
            return list(v) if isinstance(v, (list, tuple)) else [v] * ndim
        self.ndim = ndim
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = tolist(kernel_size), tolist(stride)
        self.padding, self.dilation = tolist(padding), tolist(dilation)
        self.output_padding = tolist(output_padding)
        self.conv1x1 = int(np.prod(self.kernel_size)) == 1
        self.transposed, self.inverse, self.groups = transposed, inverse, groups
        self.subm = subm
        self.indice_key = indice_key
        self.weight = Parameter(torch.Tensor(*self.kernel_size, in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.weight.size(-2) * int(np.prod(self.kernel_size))     # HWIO layout
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)

    def forward(self, input):
        assert isinstance(input, SparseConvTensor)
        features, indices = input.features, input.indices
        k = ops._uniform(self.kernel_size, self.ndim, "kernel_size")
        s = ops._uniform(self.stride, self.ndim, "stride")
        p = ops._uniform(self.padding, self.ndim, "padding")
        if ops._uniform(self.dilation, self.ndim, "dilation") != 1:
            raise NotImplementedError("dilation != 1 is not on DCL-Net's path")
        if self.conv1x1:
            input.features = torch.mm(features, self.weight.view(self.in_channels, self.out_channels))
            if self.bias is not None:
                input.features += self.bias
            return input
        if self.subm:
            out_shape = input.spatial_shape
        else:
            out_shape = ops.get_conv_output_size(input.spatial_shape, self.kernel_size, self.stride, self.padding,
                                                 self.dilation)
        datas = input.find_indice_pair(self.indice_key)
        if self.indice_key is not None and datas is not None:
            out_set, _, nbr, _, _ = datas
        else:
            out_set, nbr = ops.build_rulebook(input.active_set(), k, s, p, self.subm)
            input.indice_dict[self.indice_key] = (out_set, indices, nbr, None, input.spatial_shape)
        n_out = indices.shape[0] if self.subm else out_set.n
        W = self.weight.view(-1, self.in_channels, self.out_channels)
        out_features = SparseConvFn.apply(features.contiguous(), W.contiguous(), nbr, n_out, self.subm)
        if self.bias is not None:
            out_features += self.bias
        out_tensor = SparseConvTensor(out_features, indices if self.subm else out_set.indices, out_shape,
                                      input.batch_size)
        out_tensor._aset = out_set
        out_tensor.indice_dict = input.indice_dict
        out_tensor.grid = input.grid
        return out_tensor


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super(SparseConv3d, self).__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                                           bias, indice_key=indice_key)


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super(SubMConv3d, self).__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                                         bias, True, indice_key=indice_key)
