"""spconv.functional: the three autograd entry points the reference's conv.py / pool.py apply (`indice_conv`,
`indice_subm_conv`, `indice_avgpool`; libs/spconv/spconv/functional.py:22-43, 69-91, 137-166), generated from ONE small
factory over spconv.ops (pair-format rulebooks) instead of one hand-written Function shell per op.  Nothing in this package
calls them -- the module mirrors use the gather-table Functions of autograd.py (no format conversion) -- and a caller that
keeps the reference's own functional.py can bind that file to this package's spconv.ops unchanged;
they exist so that `import spconv.functional as Fsp` resolves for callers that only swap the package."""
from torch.autograd import Function

from . import ops


def _function(name, run, grad):
    """autograd Function `name`: run(*args) -> (out, tensors to keep); grad(grad_out, *kept) -> gradients of the leading
    inputs (the rest -- rulebooks, counts, flags -- get None)"""

    def forward(ctx, *args):
        out, keep = run(*args)
        ctx.save_for_backward(*keep)
        ctx.n_inputs = len(args)
        return out

    def backward(ctx, grad_out):
        grads = tuple(grad(grad_out.contiguous(), *ctx.saved_tensors))
        return grads + (None,) * (ctx.n_inputs - len(grads))

    return type(name, (Function,), {"forward": staticmethod(forward), "backward": staticmethod(backward)})


def _conv(subm):
    tail = (False, True) if subm else (False,)                 # ops.indice_conv(..., inverse[, subm])

    def run(features, filters, pairs, pair_num, n_out):
        return ops.indice_conv(features, filters, pairs, pair_num, n_out, *tail), (pairs, pair_num, features, filters)

    def grad(g, pairs, pair_num, features, filters):
        return ops.indice_conv_backward(features, filters, g, pairs, pair_num, *tail)

    return run, grad


def _avgpool_run(features, pairs, pair_num, n_out, use_gs=True):
    # use_gs=False divides by the receptive-field count (what DCL-Net configures), use_gs=True by the kernel volume
    rf = (ops.get_indice_summaryrf(pairs, pair_num, n_out) if not use_gs
          else pairs.new_zeros(n_out) + int(pairs.shape[0]))
    out = ops.indice_avgpool(features, pairs, pair_num, n_out, rf)
    return out, (pairs, pair_num, features, out, rf)


def _avgpool_grad(g, pairs, pair_num, features, out, rf):
    return (ops.indice_avgpool_backward(features, out, g, pairs, pair_num, rf),)


SparseConvFunction = _function("SparseConvFunction", *_conv(False))
SubMConvFunction = _function("SubMConvFunction", *_conv(True))
SparseAvgPoolFunction = _function("SparseAvgPoolFunction", _avgpool_run, _avgpool_grad)

indice_conv = SparseConvFunction.apply
indice_subm_conv = SubMConvFunction.apply
indice_avgpool = SparseAvgPoolFunction.apply
