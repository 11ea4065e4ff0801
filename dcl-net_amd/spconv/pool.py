"""SparseAvgPool / SparseAvgPool3d (the DCL-Net authors' addition, libs/spconv/spconv/pool.py:198-279)."""
from .. import ops as _ops
from ..autograd import SparseAvgPoolFn
from . import ops
from .modules import SparseModule
from .tensor import SparseConvTensor


class SparseAvgPool(SparseModule):
    def __init__(self, ndim, kernel_size, stride=1, padding=0, dilation=1, subm=False, use_gs=True):
        super(SparseAvgPool, self).__init__()

        def tolist(v):
            return list(v) if isinstance(v, (list, tuple)) else [v] * ndim
        self.ndim = ndim
        self.kernel_size, self.stride = tolist(kernel_size), tolist(stride)
        self.padding, self.dilation = tolist(padding), tolist(dilation)
        self.subm = subm
        self.use_gs = use_gs

    def forward(self, input):
        assert isinstance(input, SparseConvTensor)
        if self.use_gs:
            raise NotImplementedError("use_gs=True (divide by kernel volume) is not on DCL-Net's path "
                                      "(models/Modules.py:151 passes use_gs=False)")
        k = ops._uniform(self.kernel_size, self.ndim, "kernel_size")
        s = ops._uniform(self.stride, self.ndim, "stride")
        p = ops._uniform(self.padding, self.ndim, "padding")
        if self.subm:
            out_shape = input.spatial_shape
        else:
            out_shape = ops.get_conv_output_size(input.spatial_shape, self.kernel_size, self.stride, self.padding,
                                                 self.dilation)
        out_set, nbr = ops.build_rulebook(input.active_set(), k, s, p, self.subm)
        n_out = input.indices.shape[0] if self.subm else out_set.n
        out_features = SparseAvgPoolFn.apply(input.features.contiguous(), nbr, n_out)
        out_tensor = SparseConvTensor(out_features, input.indices if self.subm else out_set.indices, out_shape,
                                      input.batch_size)
        out_tensor._aset = out_set
        out_tensor.indice_dict = input.indice_dict
        out_tensor.grid = input.grid
        return out_tensor


class SparseAvgPool3d(SparseAvgPool):
    def __init__(self, kernel_size, stride=1, padding=0, dilation=1, use_gs=True):
        super(SparseAvgPool3d, self).__init__(3, kernel_size, stride, padding, dilation, use_gs=use_gs)
