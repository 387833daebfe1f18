"""Linear -> ELU pairs of the policy MLPs with a fused backward (go2sim_elu_backward_bias): the activation gradient and the
Linear's bias gradient come out of ONE pass over the [B, C] activations instead of an elu_backward pass plus a column-sum
pass.  Forward and the two GEMMs of the backward stay on PyTorch-ROCm (hipBLASLt via TunableOp); same fp32 arithmetic,
parameters and state-dict names untouched (the container is still an nn.Sequential of Linear / ELU children).

The weight gradients dW = gz^T x (a [C, K] output reduced over the 24576 rows of a mini-batch) are computed as an explicit S-way
split over the rows — one batched GEMM + a sum over S — instead of one mm: for these shapes hipBLASLt's own split-K launches ~60
workgroups on 256 CUs (tools/wgrad_bench.py: 166 -> 70 us at 512x263, 325 -> 58 us at 256x512).  Fixed order: deterministic.

Per process through set_library(lib) — the algorithms call it when they run on the GPU with the HIP library."""
import ctypes as C
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

_LIB = None
_WGRAD_SPLIT = int(os.environ.get("GO2_WGRAD_SPLIT", "8"))       # 1 = plain mm
_WGRAD_MIN_ROWS = 256       # rows per split below which the plain mm is used (tests lower it to drive the split path with small goldens)


def _wgrad(gz, x):
    """dW[C, K] = gz[B, C]^T x[B, K]"""
    B, S = gz.shape[0], _WGRAD_SPLIT
    # narrow outputs (the 12 / 1-column heads) keep the tuned mm: splitting them gave nothing (measured)
    if S > 1 and gz.is_cuda and B % S == 0 and B // S >= _WGRAD_MIN_ROWS and gz.shape[1] >= 32:
        return torch.bmm(gz.reshape(S, B // S, -1).transpose(1, 2), x.reshape(S, B // S, -1)).sum(0)
    return gz.t().mm(x)


def set_library(lib):
    global _LIB
    _LIB = lib


class _LinearELU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        y = F.elu_(torch.addmm(bias, x, weight.t()))
        ctx.save_for_backward(x, weight, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        gy = gy.contiguous()
        B, Cn = y.shape
        gz, gb = torch.empty_like(y), torch.empty(Cn, device=y.device, dtype=y.dtype)
        ws = torch.empty(Cn * ((B + 63) // 64), device=y.device, dtype=y.dtype)      # one row of column partials per 64-row tile
        p = lambda t: C.c_void_p(t.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(y.device).cuda_stream) if y.is_cuda else None
        rc = _LIB.go2sim_elu_backward_bias(p(gy), p(y), p(gz), p(gb), p(ws), B, Cn, stream)
        if rc != 0:
            raise RuntimeError("go2sim_elu_backward_bias failed: %s" % _LIB.go2sim_last_error().decode())
        gx = gz.mm(weight) if ctx.needs_input_grad[0] else None
        return gx, _wgrad(gz, x), gb


class _Linear(torch.autograd.Function):
    """The MLP's output layer (no activation): plain addmm forward; backward with the row-split weight gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gy.mm(weight) if ctx.needs_input_grad[0] else None
        return gx, _wgrad(gy, x), gy.sum(0)


class FusedSequential(nn.Sequential):
    """nn.Sequential whose (Linear, ELU(alpha=1)) pairs take the fused path when gradients are being recorded."""

    def forward(self, x):
        mods = list(self)
        fuse = (_LIB is not None and torch.is_grad_enabled() and x.dim() == 2 and x.dtype == torch.float32
                and (x.is_cuda or _LIB.go2sim_is_device_library() == 0))
        i = 0
        while i < len(mods):
            m = mods[i]
            if (fuse and isinstance(m, nn.Linear) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ELU) and mods[i + 1].alpha == 1.0
                    and m.bias is not None and m.out_features % 4 == 0 and m.weight.requires_grad):
                x = _LinearELU.apply(x if x.is_contiguous() else x.contiguous(), m.weight, m.bias)
                i += 2
            elif fuse and x.is_cuda and isinstance(m, nn.Linear) and m.bias is not None and m.weight.requires_grad and _WGRAD_SPLIT > 1 and m.out_features >= 32:
                x = _Linear.apply(x if x.is_contiguous() else x.contiguous(), m.weight, m.bias)
                i += 1
            else:
                x = m(x)
                i += 1
        return x
