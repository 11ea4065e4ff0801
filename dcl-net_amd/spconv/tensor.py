"""SparseConvTensor (libs/spconv/spconv/__init__.py:46-85)."""
import numpy as np
import torch

from .. import ops as _ops


class SparseConvTensor(object):
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        self.features = features
        self.indices = indices if indices.dtype == torch.int32 else indices.int()
        self.spatial_shape = [int(s) for s in np.asarray(spatial_shape).reshape(-1)]
        self.batch_size = int(batch_size)
        self.indice_dict = {}
        self.grid = grid
        self._aset = None          # device-side ActiveSet of `indices` (built lazily)

    @property
    def spatial_size(self):
        return int(np.prod(self.spatial_shape))

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key, None)

    def active_set(self):
        """Bitmask/prefix description of `indices` on the (cubic) grid."""
        if self._aset is None:
            S = self.spatial_shape[0]
            if any(s != S for s in self.spatial_shape) or len(self.spatial_shape) != 3:
                raise NotImplementedError("only cubic 3-D grids are supported (DCL-Net uses 64^3..4^3)")
            self._aset = _ops.grid_from_indices(self.indices.contiguous(), self.batch_size, S)
        return self._aset

    def dense(self, channels_first=True):
        shape = [self.batch_size] + list(self.spatial_shape) + [self.features.shape[1]]
        res = torch.zeros(shape, dtype=self.features.dtype, device=self.features.device)
        i = self.indices.long()
        res[i[:, 0], i[:, 1], i[:, 2], i[:, 3]] = self.features
        return res.permute(0, 4, 1, 2, 3).contiguous() if channels_first else res

    @property
    def sparity(self):
        return self.indices.shape[0] / np.prod(self.spatial_shape) / self.batch_size
