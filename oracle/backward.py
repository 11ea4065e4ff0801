"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the training-side (backward) kernels of the ops on DCL-Net's path.
Only tests/ may import this.  Parity unpinned: the reference ships no gradient fixtures and its CUDA code cannot run
here; each function follows the cited reference lines statement by statement (fp32 throughout).

  indice_conv_backward     libs/spconv/include/spconv/spconv_ops.h:351-438
  indice_avgpool_backward  libs/spconv/include/spconv/pool_ops.h:211-246, src/spconv/avgpool.cu:178-206
  three_interpolate_grad   libs/pointnet_sp/src/interpolate_gpu.cu:124-148
  voxelize_bp              libs/pointgroup_ops/src/voxelize/voxelize.cu:35-50
"""
import numpy as np


def indice_conv_backward(features, filters, out_grad, indice_pairs, indice_num, subm=False):
    """-> (inputGrad (V_in,Cin), filtersGrad like filters).  indice_pairs (kvol,2,V) [k][0]=in rows, [k][1]=out rows."""
    features = np.asarray(features, np.float32)
    out_grad = np.asarray(out_grad, np.float32)
    cin, cout = filters.shape[-2], filters.shape[-1]
    W = np.asarray(filters, np.float32).reshape(-1, cin, cout)
    kvol = W.shape[0]
    input_grad = np.zeros_like(features)                                   # :373
    filters_grad = np.zeros_like(W)                                        # :374
    kmax = int(np.argmax(indice_num))                                      # :364-367 (first maximum)
    if subm:                                                               # :381-385: centre offset without gather
        filters_grad[kmax] = features.T @ out_grad
        input_grad[:] = out_grad @ W[kmax].T
    for i in range(kvol):                                                  # :386-436
        n_hot = int(indice_num[i])
        if n_hot <= 0 or (subm and i == kmax):
            continue
        rows_in, rows_out = indice_pairs[i, 0, :n_hot], indice_pairs[i, 1, :n_hot]
        x_buf, g_buf = features[rows_in], out_grad[rows_out]               # gather (:392-409)
        filters_grad[i] = x_buf.T @ g_buf                                  # :416
        in_buf = (g_buf @ W[i].T).astype(np.float32)                       # :417
        np.add.at(input_grad, rows_in, in_buf)                             # scatter-add (:418-430); rows_in unique per offset
    return input_grad, filters_grad.reshape(filters.shape)


def indice_avgpool_backward(n_in, out_grad, indice_pairs, indice_num, summaryrf):
    """din[in] += dout[out] / (float)rf[out], offsets ascending (pool_ops.h:222-244, avgpool.cu:204)."""
    out_grad = np.asarray(out_grad, np.float32)
    din = np.zeros((n_in, out_grad.shape[1]), np.float32)
    for i in range(indice_pairs.shape[0]):
        n_hot = int(indice_num[i])
        if n_hot <= 0:
            continue
        rows_in, rows_out = indice_pairs[i, 0, :n_hot], indice_pairs[i, 1, :n_hot]
        din[rows_in] = din[rows_in] + out_grad[rows_out] / summaryrf[rows_out].astype(np.float32)[:, None]
    return din


def three_interpolate_grad(grad_out, idx, weight, m):
    """grad_points[idx[n][j]][c] += grad_out[n][c] * weight[n][j] (atomic in the reference: order unspecified)."""
    grad_out = np.asarray(grad_out, np.float32)
    gp = np.zeros((m, grad_out.shape[1]), np.float32)
    for j in range(3):
        np.add.at(gp, idx[:, j], grad_out * weight[:, j:j + 1])
    return gp


def voxelize_bp(d_out, rules, n_points, average=True):
    """d_feats[rules[v][1+i]] += multiplier * d_out[v], multiplier = 1/nActive in float32 (voxelize.cu:41-47)."""
    d_out = np.asarray(d_out, np.float32)
    d_feats = np.zeros((n_points, d_out.shape[1]), np.float32)
    for v in range(rules.shape[0]):
        na = int(rules[v, 0])
        mult = np.float32(1.0) / np.float32(na) if (average and na > 0) else np.float32(1.0)
        for i in range(1, na + 1):
            d_feats[rules[v, i]] += mult * d_out[v]
    return d_feats
