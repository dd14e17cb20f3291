"""Up-sampling transposed convolution with kernel == stride as one GEMM + depth-to-space.

nnU-Net's decoder up-samples with ``ConvTranspose{2,3}d(C_in, C_out, kernel_size=stride, stride=stride)``
(reference ``NexToU_Encoder_Decoder.py:272-276``).  With kernel == stride the output blocks do not
overlap, so the operator is exactly

    out[b, co, s*ks+i, h*kh+j, w*kw+k] = bias[co] + sum_ci x[b, ci, s, h, w] * W[ci, co, i, j, k]

i.e. ONE dense GEMM ``(B*S*H*W, C_in) x (C_in, C_out*ks*kh*kw)`` followed by a pure permutation.  MIOpen
runs it as its generic backward-data algorithm (GEMM + col2im scatter) instead, which on MI355X takes
several times longer for the cfg-2 decoder (tools/transpconv_bench.py).  The GEMM goes to
rocBLAS / hipBLASLt through ``torch.matmul`` (MFMA); autograd differentiates the three plain ops.
The module keeps ``nn.ConvTranspose*d`` as the parameter holder, so ``state_dict`` keys and shapes are
untouched.
"""
from __future__ import annotations

from typing import Sequence

import torch


def transposed_conv_as_gemm(x: torch.Tensor, weight: torch.Tensor, bias, stride: Sequence[int]) -> torch.Tensor:
    """x (B, C_in, *sp), weight (C_in, C_out, *stride) -> (B, C_out, *(sp * stride))."""
    dim = x.dim() - 2
    b, c_in = x.shape[:2]
    sp = tuple(x.shape[2:])
    c_out = weight.shape[1]
    k = tuple(int(s) for s in stride)
    assert tuple(weight.shape[2:]) == k, "kernel size must equal the stride"
    rows = x.reshape(b, c_in, -1).transpose(1, 2)                      # (B, S*H*W, C_in) view
    y = torch.matmul(rows, weight.reshape(c_in, -1))                   # (B, S*H*W, C_out*prod(k))
    y = y.reshape(b, *sp, c_out, *k)                                   # (B, s, h, w, co, i, j, k)
    # -> (B, co, s, i, h, j, w, k)
    order = [0, 1 + dim]
    for d in range(dim):
        order += [1 + d, 2 + dim + d]
    y = y.permute(order).reshape(b, c_out, *[s * kk for s, kk in zip(sp, k)])
    if bias is not None:
        y = y + bias.view(1, -1, *([1] * dim))
    return y
