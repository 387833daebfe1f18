"""The policy MLPs' forward and backward pass as explicit kernel calls (PPO.update, rsl_rl/rsl_rl/algorithms/ppo.py:120-187 -> autograd over
rsl_rl/rsl_rl/modules/actor_critic.py:50-75).  Parameters and state-dict names are untouched: the container is still an nn.Sequential of Linear / ELU
children; only what runs when gradients are recorded changes.

* _FusedMLP (default when the MLP is [Linear, ELU] x H + a narrow Linear): ONE autograd node per MLP.  Every product of the two passes is a call the node
  issues itself, so each goes to the faster of two kernels (the fp32-MFMA GEMMs of include/go2nn.h with their ELU / ELU' + bias-gradient epilogues, or
  hipBLASLt + the element-wise kernels; _own below holds the measured choice), the backward pass starts with go2nn_head_backward (the narrow head's input,
  weight and bias gradients, the ELU backward and the last hidden layer's bias gradient in one streaming pass instead of two degenerate GEMMs — 58 us for the
  value head's [1,128] weight gradient —, a split-K fix-up, two column-sum launches and an element-wise pass at the head of the critic's chain), and every
  fixed-order reduction of the pass (those partials, the input gradients' column partials, the sums over the weight gradients' row splits) is finished by ONE
  go2nn_sum_rows launch at the end instead of six small launches along the chain of dependent GEMMs.
* _LinearELU / _Linear / _LinearELUHead (MLPs of other shapes, GO2_MLP_NODE=0): per-layer nodes; the ELU backward and the Linear's bias gradient in one pass
  over the activations (go2sim_elu_backward_bias) instead of an elu_backward pass plus a column-sum pass.

The weight gradients dW = gz^T x (a [C, K] output reduced over the 24576 rows of a mini-batch) on the vendor path are an explicit S-way split over the
rows — one batched GEMM + a sum over S — instead of one mm: for these shapes hipBLASLt's own split-K launches ~60 workgroups on 256 CUs
(tools/wgrad_bench.py: 166 -> 70 us at 512x263, 325 -> 58 us at 256x512).  All sums have a fixed order: deterministic, replicas stay bit-identical.

Per process through set_library(lib) — the algorithms call it when they run on the GPU with the HIP library."""
import ctypes as C
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

_LIB = None
_NN = None
_WGRAD_SPLIT = int(os.environ.get("GO2_WGRAD_SPLIT", "8"))       # 1 = plain mm
_MLP_NODE = os.environ.get("GO2_MLP_NODE", "1") == "1"       # 0: per-layer autograd nodes (_LinearELU / _LinearELUHead) instead of the whole-MLP node
_WGRAD_MIN_ROWS = 256       # rows per split below which the plain mm is used (tests lower it to drive the split path with small goldens)


def _wgrad(gz, x):
    """dW[C, K] = gz[B, C]^T x[B, K]"""
    B, S = gz.shape[0], _WGRAD_SPLIT
    # narrow outputs (the 12 / 1-column heads) keep the tuned mm: splitting them gave nothing (measured)
    if S > 1 and gz.is_cuda and B % S == 0 and B // S >= _WGRAD_MIN_ROWS and gz.shape[1] >= 32:
        return torch.bmm(gz.reshape(S, B // S, -1).transpose(1, 2), x.reshape(S, B // S, -1)).sum(0)
    return gz.t().mm(x)


def set_library(lib):
    """lib: the go2sim library (HIP on the GPU; tests pass the oracle).  With the HIP library the learner-side go2nn kernels come along —
    load_nn raises when libgo2nn_hip.so is missing (no silent fallback to the GEMM formulation on a GPU box)."""
    global _LIB, _NN
    _LIB = lib
    if lib is not None and lib.go2sim_is_device_library() == 1 and _NN is None and os.environ.get("GO2_FUSED_HEAD", "1") == "1":
        from ..._nn import load_nn
        _NN = load_nn()


def set_nn_library(nn_lib):
    """The go2nn library for the MLP tails (tests hand in the host build; None switches the fused tail off)."""
    global _NN
    _NN = nn_lib


def _check(rc, what, lib):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, lib.go2nn_last_error().decode()))


class _LinearELUHead(torch.autograd.Function):
    """x -> Linear(w1, b1) -> ELU -> Linear(w2, b2) with out_features(w2) <= 16, as one node (see the module docstring)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        y = F.elu_(torch.addmm(b1, x, w1.t()))
        ctx.save_for_backward(x, w1, y, w2)
        return torch.addmm(b2, y, w2.t())

    @staticmethod
    def backward(ctx, gout):
        x, w1, y, w2 = ctx.saved_tensors
        gout = gout.contiguous()
        B, K = y.shape
        Cn = w2.shape[0]
        n = _NN.go2nn_head_backward_workspace(B, Cn, K)
        if n < 0:
            raise RuntimeError("go2nn_head_backward_workspace: %s" % _NN.go2nn_last_error().decode())
        gz, sums, ws = torch.empty_like(y), torch.empty((Cn + 1) * K + Cn, device=y.device, dtype=y.dtype), torch.empty(int(n), device=y.device, dtype=y.dtype)
        p = lambda t: C.c_void_p(t.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(y.device).cuda_stream) if y.is_cuda else None
        _check(_NN.go2nn_head_backward(p(gout), p(y), p(w2.contiguous()), p(gz), p(sums), p(ws), B, Cn, K, stream), "go2nn_head_backward", _NN)
        gx = gz.mm(w1) if ctx.needs_input_grad[0] else None
        return gx, _wgrad(gz, x), sums[Cn * K:(Cn + 1) * K], sums[:Cn * K].view(Cn, K), sums[(Cn + 1) * K:]


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


def _is_tail(l1, act, l2):
    return (isinstance(l1, nn.Linear) and isinstance(act, nn.ELU) and act.alpha == 1.0 and isinstance(l2, nn.Linear) and l1.bias is not None and l2.bias is not None
            and l1.weight.requires_grad and l2.weight.requires_grad and l2.out_features <= 16 and l1.out_features % 4 == 0 and l1.out_features <= 512)


# Which of a hidden layer's three products go to the go2nn MFMA kernels (include/go2nn.h go2nn_linear_*) and which stay on hipBLASLt + the element-wise
# kernels.  Measured per shape at M = 24576 on one MI355X, alone and as the actor / critic pair on two streams (tools/gemm_bench.py), and — what
# decides — as the whole job with one rule switched at a time, in ONE session on one GPU (profiles/r3_mlp_kernel_choice.txt; boxes differ by a few %):
#   forward:      own for the 256 -> 128 layer (23 us against addmm + elu_ 28; as the pair 58-64 against 84-130) and for the two input layers, whose rows
#                 (45 / 263 floats) are not a multiple of 16 bytes: 27 against 52 us and 79 against 89, 130 against 133 as the pair — gfx950 takes a 16-byte
#                 load from a 4-byte-aligned address as ONE instruction (before that: four loads per quad, and the vendor kernels won).  The 512 -> 256
#                 layer is at parity alone (65 us) and loses as the pair (147 against 139): vendor GEMM + elu_
#   input grad:   own (128 x 128 tiles, 16-deep k-tiles for the 512-wide output: 76 us against mm + go2sim_elu_backward_bias 95; 64 x 128 for the 256-wide
#                 one: 34 against 42; as pairs 159 against 191 and 70 against 81-108): whole job 5.10 -> 5.18 M env-steps/s.  (Before the kernels were
#                 held to 3 waves per SIMD — 184 registers — the same choice cost 4 %.)
#   weight grad:  hipBLASLt's row-split bmm (60 us against 88); its sum over the splits joins the deferred reductions
# GO2_MLP_OWN_F / _I / _W = all | none | auto | k256 override the three rules (A/B runs).
_OWN = {k: os.environ.get("GO2_MLP_OWN_" + k.upper(), "auto") for k in ("f", "i", "w")}


def _own(kind, K, N):
    """kind 'f': y[M,N] = elu(x[M,K] W^T + b); 'i': input gradient of a layer with N outputs, K inputs; 'w': its weight gradient"""
    mode = _OWN[kind]
    if mode == "k256":
        return K <= 256
    if mode == "l3a1":
        return N <= 128 or K <= 64
    if mode == "l3l1":
        return N <= 128 or K % 4 != 0
    if mode != "auto":
        return mode == "all"
    if kind == "f":
        return N <= 128 or K % 4 != 0 or K <= 64
    return kind == "i"


class _FusedMLP(torch.autograd.Function):
    """x -> [Linear -> ELU] x H -> Linear (narrow) as ONE autograd node: every product of the forward and backward pass is an explicit kernel call,
    so each can go to the kernel that is fastest for its shape, and the element-wise work between the layers rides in GEMM epilogues where the
    GEMM is ours (ELU in the forward; ELU' + the bias gradient's column sums in the input gradient) or in the one fused pass that follows a
    vendor GEMM (go2sim_elu_backward_bias).  args: x, w1, b1, ..., wH, bH, w_out, b_out."""

    @staticmethod
    def forward(ctx, x, *params):
        ws, bs = params[0::2], params[1::2]
        H = len(ws) - 1
        p = lambda t: C.c_void_p(t.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream) if x.is_cuda else None
        acts = [x]
        for l in range(H):
            h, w, b = acts[-1], ws[l], bs[l]
            if _own("f", w.shape[1], w.shape[0]):
                y = torch.empty(h.shape[0], w.shape[0], device=h.device, dtype=h.dtype)
                _check(_NN.go2nn_linear_elu_forward(p(h), p(w), p(b), p(y), h.shape[0], w.shape[1], w.shape[0], stream), "go2nn_linear_elu_forward", _NN)
            else:
                y = F.elu_(torch.addmm(b, h, w.t()))
            acts.append(y)
        ctx.save_for_backward(*acts, *ws)
        ctx.H = H
        return torch.addmm(bs[H], acts[-1], ws[H].t())

    @staticmethod
    def backward(ctx, gout):
        from ..._nn import Go2nnSumJob
        H = ctx.H
        acts, ws = ctx.saved_tensors[:H + 1], ctx.saved_tensors[H + 1:]
        dev, dt = gout.device, gout.dtype
        p = lambda t: C.c_void_p(t.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if gout.is_cuda else None
        new = lambda *shape: torch.empty(*shape, device=dev, dtype=dt)
        gout = gout.contiguous()
        B = gout.shape[0]
        jobs = []          # (partial rows, result, nrows, ncols): every fixed-order reduction of this pass, finished by ONE go2nn_sum_rows launch at the end
        # the head: input gradient, ELU', weight / bias gradients and the last hidden layer's bias gradient in one pass
        y, w_out = acts[H], ws[H]
        Cn, K = w_out.shape
        n = _NN.go2nn_head_backward_workspace(B, Cn, K)
        if n < 0:
            raise RuntimeError("go2nn_head_backward_workspace: %s" % _NN.go2nn_last_error().decode())
        gz, sums, wk = torch.empty_like(y), new((Cn + 1) * K + Cn), new(int(n))
        _check(_NN.go2nn_head_backward(p(gout), p(y), p(w_out), p(gz), None, p(wk), B, Cn, K, stream), "go2nn_head_backward", _NN)
        jobs.append((wk, sums, _NN.go2nn_head_backward_rows(B, Cn, K), (Cn + 1) * K + Cn))
        grads = [None] * (2 * (H + 1))
        grads[2 * H], grads[2 * H + 1] = sums[:Cn * K].view(Cn, K), sums[(Cn + 1) * K:]
        gb = sums[Cn * K:(Cn + 1) * K]
        for l in range(H - 1, -1, -1):          # gz: gradient at layer l's pre-activation; gb: its column sums
            w, h = ws[l], acts[l]                # h: the layer's input (= the ELU output of the layer before, or x)
            Co, Ki = w.shape
            S = _WGRAD_SPLIT
            if _own("w", Ki, Co):
                dw, wk = torch.empty_like(w), new(int(_NN.go2nn_linear_backward_workspace(B, Co, Ki)))
                _check(_NN.go2nn_linear_backward_weight(p(gz), p(h), p(dw), p(wk), B, Co, Ki, stream), "go2nn_linear_backward_weight", _NN)
            elif S > 1 and gz.is_cuda and B % S == 0 and B // S >= _WGRAD_MIN_ROWS:
                parts = torch.bmm(gz.view(S, B // S, Co).transpose(1, 2), h.view(S, B // S, Ki))       # the row splits of _wgrad, summed with the rest below
                dw = torch.empty_like(w)
                jobs.append((parts, dw, S, Co * Ki))
            else:
                dw = gz.t().mm(h)
            grads[2 * l], grads[2 * l + 1] = dw, gb
            if l > 0:
                gzp, gbp = torch.empty_like(h), new(Ki)
                if _own("i", Ki, Co):
                    rows_i = _NN.go2nn_linear_backward_input_rows(B, Co, Ki)
                    wk = new(rows_i * Ki)          # (the column partials only: go2nn_linear_backward_workspace is sized for the weight gradient's row splits, ~40 x this)
                    _check(_NN.go2nn_linear_backward_input(p(gz), p(w), p(h), p(gzp), None, p(wk), B, Co, Ki, stream), "go2nn_linear_backward_input", _NN)
                    jobs.append((wk, gbp, rows_i, Ki))
                else:
                    gx = gz.mm(w)
                    wk2 = new(Ki * ((B + 63) // 64))
                    rc = _LIB.go2sim_elu_backward_bias(p(gx), p(h), p(gzp), p(gbp), p(wk2), B, Ki, stream)
                    if rc != 0:
                        raise RuntimeError("go2sim_elu_backward_bias failed: %s" % _LIB.go2sim_last_error().decode())
                gz, gb = gzp, gbp
        gx = gz.mm(ws[0]) if ctx.needs_input_grad[0] else None
        for k in range(0, len(jobs), 16):
            chunk = jobs[k:k + 16]
            arr = (Go2nnSumJob * len(chunk))(*[Go2nnSumJob(t[0].data_ptr(), t[1].data_ptr(), t[2], t[3]) for t in chunk])
            _check(_NN.go2nn_sum_rows(arr, len(chunk), stream), "go2nn_sum_rows", _NN)
        return (gx, *grads)


def _whole_mlp(mods):
    """[(Linear, ELU)] * H + [Linear narrow] with H >= 1 -> the Linear modules, else None"""
    if len(mods) < 3 or len(mods) % 2 == 0 or not _is_tail(mods[-3], mods[-2], mods[-1]):
        return None
    lins = []
    for k in range(0, len(mods) - 1, 2):
        m, a = mods[k], mods[k + 1]
        if not (isinstance(m, nn.Linear) and isinstance(a, nn.ELU) and a.alpha == 1.0 and m.bias is not None and m.weight.requires_grad and m.out_features % 4 == 0):
            return None
        lins.append(m)
    return lins + [mods[-1]]


class _Launch:
    """The go2nn calls of one mini-batch on one device / stream; collects the fixed-order reductions of the pass for ONE go2nn_sum_rows launch."""

    def __init__(self, dev):
        self.dev, self.nn = dev, _NN
        self.stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else None
        self.sums = []          # (partial rows, result, nrows, ncols[, acc, nacc])

    def new(self, *shape):
        return torch.empty(*shape, device=self.dev, dtype=torch.float32)

    def check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, self.nn.go2nn_last_error().decode()))

    def images(self, lins):
        """split images of the layers' weights (both orientations), ONE launch; [None] * n when the split-operand kernels are off"""
        from ..._nn import Go2nnSplitJob
        if not _SPLIT:
            return [None] * len(lins)
        imgs, jobs = [], []
        for m in lins:
            N, K = m.weight.shape
            n = self.nn.go2nn_split_weights_bytes(N, K)
            if n <= 0:
                raise RuntimeError("go2nn_split_weights_bytes: %s" % self.nn.go2nn_last_error().decode())
            imgs.append(torch.empty(int(n), device=self.dev, dtype=torch.uint8))
            jobs.append(Go2nnSplitJob(m.weight.data_ptr(), imgs[-1].data_ptr(), N, K))
        for k in range(0, len(jobs), 16):
            chunk = jobs[k:k + 16]
            self.check(self.nn.go2nn_split_weights((Go2nnSplitJob * len(chunk))(*chunk), len(chunk), self.stream), "go2nn_split_weights")
        return imgs

    def forward(self, jobs, act=0):
        """jobs: [(x [M, K], Linear, image)] (1 or 2, one launch) -> [y [M, N]];  act 0: ELU, 1: none"""
        from ..._nn import Go2nnFwdJob
        ys = [self.new(x.shape[0], m.out_features) for x, m, _ in jobs]
        arr = (Go2nnFwdJob * len(jobs))(*[Go2nnFwdJob(x.data_ptr(), m.weight.data_ptr(), m.bias.data_ptr(), y.data_ptr(), x.shape[0], m.in_features, m.out_features, act,
                                                      img.data_ptr() if img is not None else None) for (x, m, img), y in zip(jobs, ys)])
        self.check(self.nn.go2nn_linear_elu_forward_group(arr, len(jobs), self.stream), "go2nn_linear_elu_forward_group")
        return ys

    def wgrad(self, jobs, sink=None, tag=None):
        """jobs: [(gz [M, C], x [M, Kin], Linear)] with one M: the layers' weight gradients (row-slice partials now, .grad = their sum after finish()).
        sink: a dict that receives the gradient tensors under (tag, "w") instead of the parameters' .grad (inside an autograd node)"""
        from ..._nn import Go2nnBwdWJob
        arr = (Go2nnBwdWJob * len(jobs))(*[Go2nnBwdWJob(gz.data_ptr(), x.data_ptr(), None, gz.shape[0], m.out_features, m.in_features, 1 if _SPLIT else 0) for gz, x, m in jobs])
        rows = self.nn.go2nn_linear_backward_weight_group_rows(arr, len(jobs))
        if rows <= 0:
            raise RuntimeError("go2nn_linear_backward_weight_group_rows: %s" % self.nn.go2nn_last_error().decode())
        for j, (gz, x, m) in enumerate(jobs):
            n = m.out_features * m.in_features
            wk, dw = self.new(rows * n), torch.empty_like(m.weight)
            arr[j].workspace = wk.data_ptr()
            self.sums.append((wk, dw, rows, n))
            if sink is None:
                m.weight.grad = dw
            else:
                sink[(tag, "w")] = dw
        self.check(self.nn.go2nn_linear_backward_weight_group(arr, len(jobs), self.stream), "go2nn_linear_backward_weight_group")

    def bwd_in(self, jobs, plain=False):
        """jobs: [(gz [M, C], Linear, y_prev [M, Kin] or None, image)] -> ([gz_prev [M, Kin]], [gb_prev [Kin]] (valid after finish(); None when plain))"""
        from ..._nn import Go2nnBwdInJob
        arr, outs, gbs = (Go2nnBwdInJob * len(jobs))(), [], []
        for j, (gz, m, yp, img) in enumerate(jobs):
            M, Co, Ki = gz.shape[0], m.out_features, m.in_features
            o = self.new(M, Ki)
            wk = gb = None
            if not plain:
                r = self.nn.go2nn_linear_backward_input_group_rows(M, Co, Ki)
                wk, gb = self.new(r * Ki), self.new(Ki)
                self.sums.append((wk, gb, r, Ki))
            arr[j] = Go2nnBwdInJob(gz.data_ptr(), m.weight.data_ptr(), yp.data_ptr() if yp is not None else None, o.data_ptr(), wk.data_ptr() if wk is not None else None,
                                   M, Co, Ki, 1 if plain else 0, img.data_ptr() if img is not None else None)
            outs.append(o); gbs.append(gb)
        self.check(self.nn.go2nn_linear_backward_input_group(arr, len(jobs), self.stream), "go2nn_linear_backward_input_group")
        return outs, gbs

    def chain_backward(self, chains, sink=None):
        """chains: 1 or 2 dicts {lins, acts, gz, gb, imgs} of equally many layers and one M: gz / gb = the gradient at lins[-1]'s output and its column sums;
        acts[l] = the input of lins[l] (acts[l > 0] an ELU output).  Sets .grad of every weight and bias; -> the gradients at lins[0]'s pre-activation."""
        n = len(chains[0]["lins"])
        gz, gb = [c["gz"] for c in chains], [c["gb"] for c in chains]
        for l in range(n - 1, -1, -1):
            self.wgrad([(gz[j], c["acts"][l], c["lins"][l]) for j, c in enumerate(chains)], sink, l)
            for j, c in enumerate(chains):
                if sink is None:
                    c["lins"][l].bias.grad = gb[j]
                else:
                    sink[(l, "b")] = gb[j]
            if l > 0:
                gz, gb = self.bwd_in([(gz[j], c["lins"][l], c["acts"][l], c["imgs"][l]) for j, c in enumerate(chains)])
        return gz

    def finish(self):
        from ..._nn import Go2nnSumJob
        for k in range(0, len(self.sums), 32):
            chunk = self.sums[k:k + 32]
            arr = (Go2nnSumJob * len(chunk))(*[Go2nnSumJob(t[0].data_ptr(), t[1].data_ptr(), t[2], t[3], t[4].data_ptr() if len(t) > 4 and t[4] is not None else None,
                                                           t[5] if len(t) > 4 and t[4] is not None else 0, 0) for t in chunk])
            self.check(self.nn.go2nn_sum_rows(arr, len(chunk), self.stream), "go2nn_sum_rows")
        self.sums = []



class _FusedChain(torch.autograd.Function):
    """x -> [Linear -> ELU] x n (an activation behind EVERY layer: the experts' backbone of the MoE encoders, rsl_rl/rsl_rl/modules/utils.py:47-62 with last_activation=True)
    as ONE autograd node on the split-operand kernels: forward = n single-job launches with the ELU in the epilogue; backward = one pass for the top layer's ELU' + bias
    gradient (go2sim_elu_backward_bias), then weight gradients and input gradients with the ELU' of the layer below in the epilogue, one go2nn_sum_rows.
    args: x, w1, b1, ..., wn, bn."""

    @staticmethod
    def forward(ctx, x, *params):
        ws, bs = params[0::2], params[1::2]
        k = _Launch(x.device)
        lins = [_LinearView(w, b) for w, b in zip(ws, bs)]
        imgs = k.images(lins)
        acts = [x]
        for l, m in enumerate(lins):
            acts.append(k.forward([(acts[-1], m, imgs[l])])[0])
        ctx.save_for_backward(*acts, *ws)
        ctx.n, ctx.imgs = len(ws), imgs
        return acts[-1]

    @staticmethod
    def backward(ctx, gy):
        n = ctx.n
        acts, ws = ctx.saved_tensors[:n + 1], ctx.saved_tensors[n + 1:]
        k = _Launch(gy.device)
        gy = gy.contiguous()
        y = acts[n]
        B, Cn = y.shape
        gz, gb = torch.empty_like(y), k.new(Cn)
        wk = k.new(Cn * ((B + 63) // 64))
        p = lambda t: C.c_void_p(t.data_ptr())
        rc = _LIB.go2sim_elu_backward_bias(p(gy), p(y), p(gz), p(gb), p(wk), B, Cn, k.stream)
        if rc != 0:
            raise RuntimeError("go2sim_elu_backward_bias failed: %s" % _LIB.go2sim_last_error().decode())
        lins = [_LinearView(w, None) for w in ws]
        grads = {}
        gz0 = k.chain_backward([{"lins": lins, "acts": acts, "gz": gz, "gb": gb, "imgs": ctx.imgs}], sink=grads)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = k.bwd_in([(gz0[0], lins[0], None, ctx.imgs[0])], plain=True)[0][0]
        k.finish()
        out = []
        for l in range(n):
            out += [grads[(l, "w")], grads[(l, "b")]]
        return (gx, *out)


class _LinearView:
    """what _Launch needs of a Linear layer, around bare tensors (inside an autograd node there are no modules)"""

    def __init__(self, weight, bias):
        self.weight, self.bias = weight, bias
        self.out_features, self.in_features = weight.shape


def _all_elu_chain(mods):
    """[Linear, ELU(1)] x n with n >= 1 and nothing else -> the Linear modules, else None"""
    if len(mods) < 2 or len(mods) % 2:
        return None
    lins = []
    for k in range(0, len(mods), 2):
        m, a = mods[k], mods[k + 1]
        if not (isinstance(m, nn.Linear) and isinstance(a, nn.ELU) and a.alpha == 1.0 and m.bias is not None and m.weight.requires_grad and m.out_features % 4 == 0
                and m.in_features >= 4 and m.weight.dtype == torch.float32):
            return None
        lins.append(m)
    return lins


class FusedSequential(nn.Sequential):
    """nn.Sequential whose (Linear, ELU(alpha=1)) pairs take the fused path when gradients are being recorded."""

    def forward(self, x):
        mods = list(self)
        fuse = (_LIB is not None and torch.is_grad_enabled() and x.dim() == 2 and x.dtype == torch.float32
                and (x.is_cuda or _LIB.go2sim_is_device_library() == 0))
        if fuse and _NN is not None and (x.is_cuda or _NN.go2nn_is_device_library() == 0) and _MLP_NODE:
            lins = _whole_mlp(mods)
            if lins is not None:
                return _FusedMLP.apply(x if x.is_contiguous() else x.contiguous(), *[t for m in lins for t in (m.weight, m.bias)])
            lins = _all_elu_chain(mods) if (_SPLIT or _NN.go2nn_is_device_library() == 0) and _CHAIN else None
            if lins is not None:          # (the sink's single-chain keys: one chain per node)
                return _FusedChain.apply(x if x.is_contiguous() else x.contiguous(), *[t for m in lins for t in (m.weight, m.bias)])
        i = 0
        while i < len(mods):
            m = mods[i]
            if (fuse and _NN is not None and i + 3 == len(mods) and _is_tail(mods[i], mods[i + 1], mods[i + 2]) and (x.is_cuda or _NN.go2nn_is_device_library() == 0)):
                x = _LinearELUHead.apply(x if x.is_contiguous() else x.contiguous(), m.weight, m.bias, mods[i + 2].weight, mods[i + 2].bias)
                i += 3
            elif (fuse and isinstance(m, nn.Linear) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ELU) and mods[i + 1].alpha == 1.0
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


# ---- both networks as ONE autograd node (round 4) --------------------------------------------------------------------------------------------------
# PPO.update evaluates the actor and the critic on the same mini-batch rows (ppo.py:131-133).  As two nodes their layers are separate launches — on two HIP
# streams the pair takes twice one network's time (profiles/r3_gemm_bench.txt) and needs the second stream.  _FusedPair issues every layer of BOTH networks as
# one grouped launch (include/go2nn.h ABI 3: go2nn_linear_elu_forward_group / _backward_input_group / _backward_weight_group; the weight gradients read their
# operands straight from global memory into MFMA registers), and finishes every fixed-order reduction of both backward passes with ONE go2nn_sum_rows launch.
_PAIR = os.environ.get("GO2_MLP_PAIR", "1") == "1"       # 0: one node per network (round 3)
# (Measured and rejected, round 4: the weight gradients on a second HIP stream beside the chain of input gradients — nothing depends on them until the optimizer
# step — cost 8 % of the whole job: a 2-workgroup-per-CU weight-gradient kernel and a 3-per-CU input-gradient kernel take each other's occupancy.)


# The hidden layers' products on the bf16 matrix pipe with fp32 operands (include/go2nn.h ABI 4: every fp32 value split exactly into three bf16 planes, six MFMA terms,
# fp32 accumulation — as close to float64 as the fp32-MFMA kernels, tests/test_gpu_mlp_tail.py).  0: the fp32-MFMA kernels.
_SPLIT = os.environ.get("GO2_GEMM_SPLIT", "1") == "1"
_CHAIN = os.environ.get("GO2_MLP_CHAIN", "1") == "1"       # 0: per-layer autograd nodes (hipBLASLt) for the all-ELU chains (the MoE encoders' backbones) instead of _FusedChain (A/B)


def _split_images(ws, H, dev, stream):
    """-> images[j][l]: the split image of hidden layer l's weight of network j (one go2nn_split_weights launch for all of them), or None when switched off.
    The weights change with every optimizer step, so this runs at the head of every mini-batch's forward pass (a few microseconds: the matrices are small)."""
    from ..._nn import Go2nnSplitJob
    if not _SPLIT or H == 0 or 2 * H > 8:
        return None
    imgs, jobs = [[None] * H for _ in range(2)], []
    for j in range(2):
        for l in range(H):
            N, K = ws[j][l].shape
            n = _NN.go2nn_split_weights_bytes(N, K)
            if n <= 0:
                raise RuntimeError("go2nn_split_weights_bytes: %s" % _NN.go2nn_last_error().decode())
            imgs[j][l] = torch.empty(int(n), device=dev, dtype=torch.uint8)
            jobs.append(Go2nnSplitJob(ws[j][l].data_ptr(), imgs[j][l].data_ptr(), N, K))
    _check(_NN.go2nn_split_weights((Go2nnSplitJob * len(jobs))(*jobs), len(jobs), stream), "go2nn_split_weights", _NN)
    return imgs


class _FusedPair(torch.autograd.Function):
    """(x_a, x_c) -> (actor(x_a), critic(x_c)) for two MLPs [Linear -> ELU] x H -> Linear(narrow) with the same hidden widths.
    args: x_a, x_c, H, then the actor's w1, b1, ..., w_out, b_out and the critic's."""

    @staticmethod
    def forward(ctx, xa, xc, H, *params):
        from ..._nn import Go2nnFwdJob
        n = 2 * (H + 1)
        P = (params[:n], params[n:])
        ws = [q[0::2] for q in P]
        bs = [q[1::2] for q in P]
        p = lambda t: t.data_ptr()
        stream = C.c_void_p(torch.cuda.current_stream(xa.device).cuda_stream) if xa.is_cuda else None
        acts = [[xa], [xc]]
        imgs = _split_images(ws, H, xa.device, stream)
        sp = lambda j, l: imgs[j][l].data_ptr() if imgs is not None else None
        for l in range(H):
            ys = [torch.empty(acts[j][-1].shape[0], ws[j][l].shape[0], device=xa.device, dtype=xa.dtype) for j in range(2)]
            jobs = (Go2nnFwdJob * 2)(*[Go2nnFwdJob(p(acts[j][-1]), p(ws[j][l]), p(bs[j][l]), p(ys[j]), acts[j][-1].shape[0], ws[j][l].shape[1], ws[j][l].shape[0], 0, sp(j, l)) for j in range(2)])
            _check(_NN.go2nn_linear_elu_forward_group(jobs, 2, stream), "go2nn_linear_elu_forward_group", _NN)
            for j in range(2):
                acts[j].append(ys[j])
        ctx.save_for_backward(*acts[0], *acts[1], *ws[0], *ws[1])
        ctx.H, ctx.imgs = H, imgs
        return torch.addmm(bs[0][H], acts[0][-1], ws[0][H].t()), torch.addmm(bs[1][H], acts[1][-1], ws[1][H].t())

    @staticmethod
    def backward(ctx, ga, gc):
        from ..._nn import Go2nnBwdInJob, Go2nnBwdWJob, Go2nnSumJob
        H = ctx.H
        sv = ctx.saved_tensors
        acts = (sv[:H + 1], sv[H + 1:2 * (H + 1)])
        ws = (sv[2 * (H + 1):3 * (H + 1)], sv[3 * (H + 1):])
        gouts = (ga.contiguous(), gc.contiguous())
        dev, dt = ga.device, ga.dtype
        p = lambda t: t.data_ptr()
        vp = lambda t: C.c_void_p(t.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if ga.is_cuda else None
        new = lambda *shape: torch.empty(*shape, device=dev, dtype=dt)
        B = gouts[0].shape[0]
        sums = []          # (partial rows, result, nrows, ncols) of both networks: ONE go2nn_sum_rows launch at the end
        grads = [[None] * (2 * (H + 1)) for _ in range(2)]
        gz, gb = [None, None], [None, None]
        for j in range(2):      # the narrow heads: input gradient, ELU', weight / bias gradients, the last hidden layer's bias gradient in one pass each
            y, w_out = acts[j][H], ws[j][H]
            Cn, K = w_out.shape
            n = _NN.go2nn_head_backward_workspace(B, Cn, K)
            if n < 0:
                raise RuntimeError("go2nn_head_backward_workspace: %s" % _NN.go2nn_last_error().decode())
            gz[j], tot, wk = torch.empty_like(y), new((Cn + 1) * K + Cn), new(int(n))
            _check(_NN.go2nn_head_backward(vp(gouts[j]), vp(y), vp(w_out), vp(gz[j]), None, vp(wk), B, Cn, K, stream), "go2nn_head_backward", _NN)
            sums.append((wk, tot, _NN.go2nn_head_backward_rows(B, Cn, K), (Cn + 1) * K + Cn))
            grads[j][2 * H], grads[j][2 * H + 1] = tot[:Cn * K].view(Cn, K), tot[(Cn + 1) * K:]
            gb[j] = tot[Cn * K:(Cn + 1) * K]
        for l in range(H - 1, -1, -1):          # gz[j]: gradient at layer l's pre-activation; gb[j]: its column sums
            shp = [ws[j][l].shape for j in range(2)]
            wj = (Go2nnBwdWJob * 2)(*[Go2nnBwdWJob(p(gz[j]), p(acts[j][l]), None, B, shp[j][0], shp[j][1], 1 if ctx.imgs is not None else 0) for j in range(2)])
            rows = _NN.go2nn_linear_backward_weight_group_rows(wj, 2)
            if rows <= 0:
                raise RuntimeError("go2nn_linear_backward_weight_group_rows: %s" % _NN.go2nn_last_error().decode())
            for j in range(2):
                wk, dw = new(rows * shp[j][0] * shp[j][1]), torch.empty_like(ws[j][l])
                wj[j].workspace = p(wk)
                sums.append((wk, dw, rows, shp[j][0] * shp[j][1]))
                grads[j][2 * l], grads[j][2 * l + 1] = dw, gb[j]
            _check(_NN.go2nn_linear_backward_weight_group(wj, 2, stream), "go2nn_linear_backward_weight_group", _NN)
            if l > 0:
                gzp = [torch.empty_like(acts[j][l]) for j in range(2)]
                ij = (Go2nnBwdInJob * 2)()
                for j in range(2):
                    r = _NN.go2nn_linear_backward_input_group_rows(B, shp[j][0], shp[j][1])
                    wk, gbp = new(r * shp[j][1]), new(shp[j][1])
                    ij[j] = Go2nnBwdInJob(p(gz[j]), p(ws[j][l]), p(acts[j][l]), p(gzp[j]), p(wk), B, shp[j][0], shp[j][1], 0, ctx.imgs[j][l].data_ptr() if ctx.imgs is not None else None)
                    sums.append((wk, gbp, r, shp[j][1]))
                    gb[j] = gbp
                _check(_NN.go2nn_linear_backward_input_group(ij, 2, stream), "go2nn_linear_backward_input_group", _NN)
                gz = gzp
        gx = [gz[j].mm(ws[j][0]) if ctx.needs_input_grad[j] else None for j in range(2)]
        for k in range(0, len(sums), 16):
            chunk = sums[k:k + 16]
            arr = (Go2nnSumJob * len(chunk))(*[Go2nnSumJob(t[0].data_ptr(), t[1].data_ptr(), t[2], t[3]) for t in chunk])
            _check(_NN.go2nn_sum_rows(arr, len(chunk), stream), "go2nn_sum_rows", _NN)
        return (gx[0], gx[1], None, *grads[0], *grads[1])


def pair_forward(seq_a, seq_c, xa, xc):
    """-> (seq_a(xa), seq_c(xc)) through _FusedPair, or None when the two modules are not two fusable MLPs with the same hidden widths (the caller then
    evaluates them one by one)."""
    if not (_PAIR and _MLP_NODE and _LIB is not None and _NN is not None and torch.is_grad_enabled() and isinstance(seq_a, FusedSequential) and isinstance(seq_c, FusedSequential)):
        return None
    if not (xa.dim() == 2 and xc.dim() == 2 and xa.dtype == torch.float32 and xc.dtype == torch.float32 and xa.shape[0] == xc.shape[0] and xa.device == xc.device
            and (xa.is_cuda or (_LIB.go2sim_is_device_library() == 0 and _NN.go2nn_is_device_library() == 0))):
        return None
    la, lc = _whole_mlp(list(seq_a)), _whole_mlp(list(seq_c))
    if la is None or lc is None or len(la) != len(lc) or any(a.out_features != c.out_features for a, c in zip(la[:-1], lc[:-1])):
        return None
    if any(m.in_features < 4 for m in la + lc) or any(m.out_features < 2 for m in la[:-1] + lc[:-1]):
        return None
    cont = lambda t: t if t.is_contiguous() else t.contiguous()
    return _FusedPair.apply(cont(xa), cont(xc), len(la) - 1, *[t for m in la for t in (m.weight, m.bias)], *[t for m in lc for t in (m.weight, m.bias)])


# ---- PPO's whole mini-batch gradient without autograd (round 4) -----------------------------------------------------------------------------------------
# With both networks in one node, what is left around the hidden layers' grouped GEMMs is the chain  heads forward (two degenerate GEMMs) -> loss head
# (go2sim_ppo_loss) -> heads backward (two go2nn_head_backward launches): all per row of the mini-batch, all reading / writing the last hidden activations.
# go2nn_ppo_heads does it in one streaming pass, and since the loss kernel hands out the analytic gradients anyway, the update needs no autograd graph:
# ppo_pair_grads runs forward, loss and backward as explicit launches (12 per mini-batch instead of 20) and installs the gradients as the parameters' .grad.
_HEADS = os.environ.get("GO2_PPO_HEADS", "1") == "1"       # 0: the autograd node + go2sim_ppo_loss (A/B)


def ppo_pair_applicable(ac, xa, xc):
    if not (_HEADS and _PAIR and _MLP_NODE and _LIB is not None and _NN is not None and hasattr(ac, "actor") and hasattr(ac, "critic") and hasattr(ac, "std")):
        return False
    if not (isinstance(ac.actor, FusedSequential) and isinstance(ac.critic, FusedSequential) and xa.dim() == 2 and xc.dim() == 2 and xa.dtype == torch.float32
            and xc.dtype == torch.float32 and xa.shape[0] == xc.shape[0] and (xa.is_cuda or (_LIB.go2sim_is_device_library() == 0 and _NN.go2nn_is_device_library() == 0))):
        return False
    la, lc = _whole_mlp(list(ac.actor)), _whole_mlp(list(ac.critic))
    if la is None or lc is None or len(la) != len(lc) or any(a.out_features != c.out_features for a, c in zip(la[:-1], lc[:-1])):
        return False
    if any(m.in_features < 4 for m in la + lc) or any(m.out_features < 2 for m in la[:-1] + lc[:-1]):
        return False
    K = la[-1].in_features
    return (lc[-1].out_features == 1 and la[-1].out_features <= 16 and K % 4 == 0 and K <= 256 and ac.std.dim() == 1 and ac.std.shape[0] == la[-1].out_features
            and all(p.requires_grad for m in la + lc for p in (m.weight, m.bias)) and ac.std.requires_grad)


def ppo_pair_grads(ac, xa, xc, actions, old_values, adv, returns, old_logp, old_mu, old_sigma, clip, vcoef, ecoef, use_clipped_value_loss, acc=None):
    """One PPO mini-batch gradient (ppo.py:131-170 + loss.backward()) of a plain ActorCritic as explicit launches: grouped hidden layers forward, go2nn_ppo_heads,
    grouped hidden layers backward, ONE go2nn_sum_rows; every parameter's .grad is set (replaced, as after zero_grad(set_to_none=True)).
    acc: optional float32[>= 2] device tensor — the mini-batch's surrogate and value loss are ADDED to acc[0:2] by the same go2nn_sum_rows launch (the update's running sums).
    -> stats [surrogate, value loss, KL, entropy] (means, device tensor)"""
    from ..._nn import Go2nnBwdInJob, Go2nnBwdWJob, Go2nnFwdJob, Go2nnPpoHeads, Go2nnSumJob
    la, lc = _whole_mlp(list(ac.actor)), _whole_mlp(list(ac.critic))
    H = len(la) - 1
    lins = (la, lc)
    ws = [[m.weight for m in l] for l in lins]
    bs = [[m.bias for m in l] for l in lins]
    dev, dt = xa.device, xa.dtype
    p = lambda t: t.data_ptr()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if xa.is_cuda else None
    new = lambda *shape: torch.empty(*shape, device=dev, dtype=dt)
    cont = lambda t: t.detach() if t.is_contiguous() else t.detach().contiguous()
    flat = lambda t: cont(t).reshape(-1)
    with torch.no_grad():
        B = xa.shape[0]
        acts = [[cont(xa)], [cont(xc)]]
        imgs = _split_images(ws, H, dev, stream)
        sp = lambda j, l: imgs[j][l].data_ptr() if imgs is not None else None
        for l in range(H):
            ys = [new(B, ws[j][l].shape[0]) for j in range(2)]
            jobs = (Go2nnFwdJob * 2)(*[Go2nnFwdJob(p(acts[j][-1]), p(ws[j][l]), p(bs[j][l]), p(ys[j]), B, ws[j][l].shape[1], ws[j][l].shape[0], 0, sp(j, l)) for j in range(2)])
            _check(_NN.go2nn_linear_elu_forward_group(jobs, 2, stream), "go2nn_linear_elu_forward_group", _NN)
            for j in range(2):
                acts[j].append(ys[j])
        A, K = ws[0][H].shape
        rows, cols = _NN.go2nn_ppo_heads_rows(B, A, K), _NN.go2nn_ppo_heads_cols(A, K)
        if rows <= 0 or cols <= 0:
            raise RuntimeError("go2nn_ppo_heads: %s" % _NN.go2nn_last_error().decode())
        gz = [new(B, K), new(B, K)]
        part, tot = new(rows * cols), new(cols)
        keep = [cont(actions), cont(old_mu), cont(old_sigma), flat(old_logp), flat(adv), flat(old_values), flat(returns), cont(ac.std)]
        h = Go2nnPpoHeads(p(acts[0][H]), p(acts[1][H]), p(ws[0][H]), p(bs[0][H]), p(ws[1][H]), p(bs[1][H]), p(keep[7]), p(keep[0]), p(keep[1]), p(keep[2]), p(keep[3]), p(keep[4]),
                          p(keep[5]), p(keep[6]), p(gz[0]), p(gz[1]), p(part), B, A, K, int(bool(use_clipped_value_loss)), float(clip), float(vcoef), float(ecoef))
        _check(_NN.go2nn_ppo_heads(C.byref(h), stream), "go2nn_ppo_heads", _NN)
        sums = [(part, tot, rows, cols, acc)]
        o = 4 + A
        ac.std.grad = tot[4:o].view_as(ac.std)
        ws[0][H].grad, gb_a, bs[0][H].grad = tot[o:o + A * K].view(A, K), tot[o + A * K:o + (A + 1) * K], tot[o + (A + 1) * K:o + (A + 1) * K + A]
        o += (A + 1) * K + A
        ws[1][H].grad, gb_c, bs[1][H].grad = tot[o:o + K].view(1, K), tot[o + K:o + 2 * K], tot[o + 2 * K:o + 2 * K + 1]
        gb = [gb_a, gb_c]
        for l in range(H - 1, -1, -1):
            shp = [ws[j][l].shape for j in range(2)]
            wj = (Go2nnBwdWJob * 2)(*[Go2nnBwdWJob(p(gz[j]), p(acts[j][l]), None, B, shp[j][0], shp[j][1], 1 if imgs is not None else 0) for j in range(2)])
            r = _NN.go2nn_linear_backward_weight_group_rows(wj, 2)
            if r <= 0:
                raise RuntimeError("go2nn_linear_backward_weight_group_rows: %s" % _NN.go2nn_last_error().decode())
            for j in range(2):
                wk, dw = new(r * shp[j][0] * shp[j][1]), torch.empty_like(ws[j][l])
                wj[j].workspace = p(wk)
                sums.append((wk, dw, r, shp[j][0] * shp[j][1]))
                ws[j][l].grad, bs[j][l].grad = dw, gb[j]
            _check(_NN.go2nn_linear_backward_weight_group(wj, 2, stream), "go2nn_linear_backward_weight_group", _NN)
            if l > 0:
                gzp = [torch.empty_like(acts[j][l]) for j in range(2)]
                ij = (Go2nnBwdInJob * 2)()
                for j in range(2):
                    ri = _NN.go2nn_linear_backward_input_group_rows(B, shp[j][0], shp[j][1])
                    wk, gbp = new(ri * shp[j][1]), new(shp[j][1])
                    ij[j] = Go2nnBwdInJob(p(gz[j]), p(ws[j][l]), p(acts[j][l]), p(gzp[j]), p(wk), B, shp[j][0], shp[j][1], 0, sp(j, l))
                    sums.append((wk, gbp, ri, shp[j][1]))
                    gb[j] = gbp
                _check(_NN.go2nn_linear_backward_input_group(ij, 2, stream), "go2nn_linear_backward_input_group", _NN)
                gz = gzp
        for k in range(0, len(sums), 16):
            chunk = sums[k:k + 16]
            arr = (Go2nnSumJob * len(chunk))(*[Go2nnSumJob(t[0].data_ptr(), t[1].data_ptr(), t[2], t[3], t[4].data_ptr() if len(t) > 4 and t[4] is not None else None,
                                                             2 if len(t) > 4 and t[4] is not None else 0, 0) for t in chunk])
            _check(_NN.go2nn_sum_rows(arr, len(chunk), stream), "go2nn_sum_rows", _NN)
    return tot[:4]
