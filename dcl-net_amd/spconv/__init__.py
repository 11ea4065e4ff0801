"""`spconv`-shaped front end over the HIP rulebook / sparse-conv kernels.

Mirrors the part of the reference's modified spconv v1.0 python package that DCL-Net uses
(libs/spconv/spconv/__init__.py:46-85, conv.py:51-174, pool.py:198-279, modules.py:40-130):
SparseConvTensor, SparseConv3d, SubMConv3d, SparseAvgPool3d, SparseSequential, SparseModule, with
the same constructor signatures, weight shape (k,k,k,Cin,Cout) and state_dict keys, so that
`import spconv` can be pointed at this package.  Forward only.

Differences by design: rulebooks are device-side gather tables (ops.rulebook_gather) instead of
(27,2,V) pair lists -- `ops.get_indice_pairs` still returns the reference format for callers that
want it; one host read-back per non-submanifold layer (the output row count, needed to size the
returned tensor) instead of the reference's per-layer indiceNum.to(CPU) + 27-iteration host loop.
"""
import sys
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

from .. import ops as _ops
from . import ops  # noqa: F401  (spconv.ops.get_indice_pairs, indice_conv, indice_avgpool, get_indice_summaryrf ...)
from . import functional  # noqa: F401  (spconv.functional.indice_conv / indice_subm_conv / indice_avgpool)
from .conv import SparseConv3d, SparseConvolution, SubMConv3d  # noqa: F401
from .modules import SparseModule, SparseSequential, is_spconv_module  # noqa: F401
from .pool import SparseAvgPool, SparseAvgPool3d  # noqa: F401
from .tensor import SparseConvTensor  # noqa: F401
