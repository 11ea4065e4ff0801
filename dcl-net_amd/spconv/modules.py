"""Containers of the spconv-shaped front end: SparseModule (marker) and SparseSequential, the subset of
libs/spconv/spconv/modules.py:40-130 that DCL-Net's BasicBlock_SPCONV relies on (models/Modules.py:36-56)."""
from collections import OrderedDict

from torch import nn

from .tensor import SparseConvTensor


class SparseModule(nn.Module):
    """Base class of layers that consume and produce a SparseConvTensor."""


def is_spconv_module(module):
    return isinstance(module, SparseModule)


class SparseSequential(SparseModule):
    """Runs its children in order.  Sparse layers get the tensor object; dense layers (BatchNorm1d, ReLU, ...) are
    applied to `.features` only, and are skipped while the tensor has no active voxel -- the reference's rule
    (modules.py:118-130), which keeps BatchNorm from seeing an empty batch."""

    def __init__(self, *layers, **named_layers):
        super().__init__()
        if len(layers) == 1 and isinstance(layers[0], OrderedDict):
            named = list(layers[0].items())
        else:
            named = [(str(i), m) for i, m in enumerate(layers)]
        for name, m in named + list(named_layers.items()):
            if name in self._modules:
                raise ValueError("duplicate layer name %r" % name)
            self.add_module(name, m)

    def __len__(self):
        return len(self._modules)

    def __getitem__(self, i):
        return list(self._modules.values())[i]

    def forward(self, x):
        for layer in self._modules.values():
            if is_spconv_module(layer):
                if not isinstance(x, SparseConvTensor):
                    raise TypeError("sparse layer fed with a dense tensor")
                x = layer(x)
            elif isinstance(x, SparseConvTensor):
                if x.indices.shape[0]:
                    x.features = layer(x.features)
            else:
                x = layer(x)
        return x
