"""The MLPs' forward and backward passes as explicit launches of include/go2nn.h (PPO.update, rsl_rl/rsl_rl/algorithms/ppo.py:120-187 -> autograd over
rsl_rl/rsl_rl/modules/actor_critic.py:50-75; the CTS family: algorithms/cts.py:167-285 over modules/actor_critic_cts.py, modules/utils.py).  Parameters and state-dict
names are untouched: the container is still an nn.Sequential of Linear / ELU children; only what runs changes.

Two formulations of the learner exist (round 5; rounds 2-4 had stacked five of them):

* the DEFAULT on a GPU with the library pair loaded —
    - PPO's and CTS's mini-batch gradients without autograd: ppo_pair_grads below, modules/fused_cts.py (grouped split-operand GEMMs, go2nn_ppo_heads, ONE go2nn_sum_rows);
    - for what those do not cover (networks of other shapes: the AC-MoE / MCP heads, the MoE encoders inside autograd) ONE autograd node per Linear / ELU stack on the
      same kernels: _FusedMLP ([Linear, ELU] x H -> Linear), _FusedChain ([Linear, ELU] x n);
* the REFERENCE formulation for A/B runs — plain PyTorch autograd on hipBLASLt: GO2_FUSED_MLP=0 (the algorithms then never call set_library) and, independently,
  GO2_GEMM_SPLIT=0 keeps the nodes but runs their products on the fp32-MFMA kernels instead of the split-operand (3 x bf16 planes) ones.

All sums have a fixed order: deterministic, replicas stay bit-identical.  Per process through set_library(lib) — the algorithms call it when they run on the GPU with the
HIP library."""
import ctypes as C
import os

import torch
import torch.nn as nn

_LIB = None
_NN = None
# The hidden layers' products on the bf16 matrix pipe with fp32 operands (include/go2nn.h ABI 4: every fp32 value split exactly into three bf16 planes, six MFMA terms,
# fp32 accumulation — as close to float64 as the fp32-MFMA kernels, tests/test_gpu_mlp_tail.py).  0: the fp32-MFMA kernels.
_SPLIT = os.environ.get("GO2_GEMM_SPLIT", "1") == "1"
_WG_BELOW = os.environ.get("GO2_WGRAD_BELOW", "1") == "1"      # tools (A/B): 0 = the first layer's weight gradient as a launch of its own (round 5)
_IMAGES = None             # own_forward(images=...): the caller's cache of split weight images, by weight address
_OWN_FORWARD = False       # inside own_forward(): Linear / ELU stacks evaluated WITHOUT gradient also run on the library's kernels (see FusedSequential.forward)


def set_library(lib):
    """lib: the go2sim library (HIP on the GPU; tests pass the oracle).  With the HIP library the go2nn kernels come along —
    load_nn raises when libgo2nn_hip.so is missing (no silent fallback to the GEMM formulation on a GPU box)."""
    global _LIB, _NN
    _LIB = lib
    if lib is not None and lib.go2sim_is_device_library() == 1 and _NN is None:
        from ..._nn import load_nn
        _NN = load_nn()


def set_nn_library(nn_lib):
    """The go2nn library (tests hand in the host build; None switches the nodes off)"""
    global _NN
    _NN = nn_lib


class own_forward:
    """with own_forward(): Linear / ELU stacks evaluated under torch.no_grad() run on the library's kernels too (the update's once-per-update student latents, the
    rollout's MoE student encoder) — opt-in, so that a test's torch-module reference never silently becomes the kernels it is the reference for"""

    def __init__(self, images=None):
        """images: a dict the caller owns — the split images of the layers' weights are kept in it across calls (the 24 steps of a rollout evaluate the same
        parameters: one go2nn_split_weights launch per network and ROLLOUT instead of per step); the caller empties it whenever the parameters may have changed"""
        self.images = images

    def __enter__(self):
        global _OWN_FORWARD, _IMAGES
        self._was, _OWN_FORWARD = (_OWN_FORWARD, _IMAGES), True
        _IMAGES = self.images

    def __exit__(self, *exc):
        global _OWN_FORWARD, _IMAGES
        _OWN_FORWARD, _IMAGES = self._was
        return False


def _check(rc, what, lib):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, lib.go2nn_last_error().decode()))


class _LinearView:
    """what _Launch needs of a Linear layer, around bare tensors (inside an autograd node there are no modules)"""

    def __init__(self, weight, bias):
        self.weight, self.bias = weight, bias
        self.out_features, self.in_features = weight.shape


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
        cache = _IMAGES if (_OWN_FORWARD and not torch.is_grad_enabled()) else None
        for m in lins:
            N, K = m.weight.shape
            if cache is not None and m.weight.data_ptr() in cache:
                imgs.append(cache[m.weight.data_ptr()])
                continue
            n = self.nn.go2nn_split_weights_bytes(N, K)
            if n <= 0:
                raise RuntimeError("go2nn_split_weights_bytes: %s" % self.nn.go2nn_last_error().decode())
            imgs.append(torch.empty(int(n), device=self.dev, dtype=torch.uint8))
            jobs.append(Go2nnSplitJob(m.weight.data_ptr(), imgs[-1].data_ptr(), N, K))
            if cache is not None:
                cache[m.weight.data_ptr()] = imgs[-1]
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

    def wgrad(self, jobs, sink=None, tag=None, lead=None):
        """jobs: [(gz [M, C], x [M, Kin], Linear)] with one M: the layers' weight gradients (row-slice partials now, .grad = their sum after finish()).
        sink: a dict that receives the gradient tensors under (tag, "w") instead of the parameters' .grad (inside an autograd node).
        lead: [(x0, dw)] per job — only the first x0 columns of x (whole 128-column tiles; pitch Kin) into columns [0, x0) of the given dw [C, Kin]: the other columns'
        gradient came out of the layer above's input-gradient launch (bwd_in(below=...))"""
        from ..._nn import Go2nnBwdWJob
        arr = (Go2nnBwdWJob * len(jobs))(*[Go2nnBwdWJob(gz.data_ptr(), x.data_ptr(), None, gz.shape[0], m.out_features, m.in_features if lead is None else lead[j][0], 1 if _SPLIT else 0,
                                                        0 if lead is None else m.in_features) for j, (gz, x, m) in enumerate(jobs)])
        rows = self.nn.go2nn_linear_backward_weight_group_rows(arr, len(jobs))
        if rows <= 0:
            raise RuntimeError("go2nn_linear_backward_weight_group_rows: %s" % self.nn.go2nn_last_error().decode())
        for j, (gz, x, m) in enumerate(jobs):
            if lead is not None:
                x0, dw = lead[j]
                wk = self.new(rows * m.out_features * x0)
                arr[j].workspace = wk.data_ptr()
                self.sums.append((wk, dw, rows, m.out_features * x0, None, 0, (0, x0, m.in_features)))
                continue
            n = m.out_features * m.in_features
            wk, dw = self.new(rows * n), torch.empty_like(m.weight)
            arr[j].workspace = wk.data_ptr()
            self.sums.append((wk, dw, rows, n))
            if sink is None:
                m.weight.grad = dw
            else:
                sink[(tag, "w")] = dw
        self.check(self.nn.go2nn_linear_backward_weight_group(arr, len(jobs), self.stream), "go2nn_linear_backward_weight_group")

    def bwd_in(self, jobs, plain=False, below=None, sink=None, tag=None):
        """jobs: [(gz [M, C], Linear, y_prev [M, Kin] or None, image)] -> ([gz_prev [M, Kin]], [gb_prev [Kin]] (valid after finish(); None when plain)).
        below: [(x [M, K0], Linear K0 -> Kin, x0, keep)] per job — the layer whose ELU output y_prev is: the weight gradient of its input columns [x0, K0) (at most 64)
        comes out of the same launch (include/go2nn.h ABI 6, Go2nnBwdInJob.x_in), formed from the gz_prev tile while it is on the chip; keep False: gz_prev is not
        written at all (-> None).  -> additionally [dw [Kin, K0]] per job: complete after finish() when x0 == 0, else columns [0, x0) are the caller's (wgrad(lead=...))."""
        from ..._nn import Go2nnBwdInJob
        arr, outs, gbs, dws = (Go2nnBwdInJob * len(jobs))(), [], [], []
        for j, (gz, m, yp, img) in enumerate(jobs):
            M, Co, Ki = gz.shape[0], m.out_features, m.in_features
            o = self.new(M, Ki) if below is None or below[j][3] else None
            wk = gb = None
            if not plain:
                r = self.nn.go2nn_linear_backward_input_group_rows(M, Co, Ki)
                wk, gb = self.new(r * Ki), self.new(Ki)
                self.sums.append((wk, gb, r, Ki))
            arr[j] = Go2nnBwdInJob(gz.data_ptr(), m.weight.data_ptr(), yp.data_ptr() if yp is not None else None, o.data_ptr() if o is not None else None,
                                   wk.data_ptr() if wk is not None else None, M, Co, Ki, 1 if plain else 0, img.data_ptr() if img is not None else None)
            if below is not None:
                x, mb, x0, _ = below[j]
                K0 = mb.in_features
                kx = K0 - x0
                rows = self.nn.go2nn_linear_backward_input_fused_rows(M)
                if rows <= 0:
                    raise RuntimeError("go2nn_linear_backward_input_fused_rows: %s" % self.nn.go2nn_last_error().decode())
                dwk, dw = self.new(rows * Ki * kx), torch.empty_like(mb.weight)
                arr[j].Kx, arr[j].x_in, arr[j].dw_workspace, arr[j].ldx = kx, x.data_ptr() + 4 * x0, dwk.data_ptr(), K0
                self.sums.append((dwk, dw, rows, Ki * kx, None, 0, (4 * x0, kx, K0)))
                dws.append(dw)
                if sink is None:
                    mb.weight.grad = dw
                else:
                    sink[(tag, "w")] = dw
            outs.append(o); gbs.append(gb)
        self.check(self.nn.go2nn_linear_backward_input_group(arr, len(jobs), self.stream), "go2nn_linear_backward_input_group")
        return (outs, gbs) if below is None else (outs, gbs, dws)

    def below_plan(self, chains, need_gz0):
        """[x0] per chain when the first layers' weight gradients (their input columns [x0, K0)) can come out of the second layers' input-gradient launch, else None:
        split-operand kernels, dense inputs, K0 <= 64 (x0 = 0: everything) or K0 = whole 128-column tiles + at most 64 columns (the 263-wide critic input: x0 = 256 —
        the tiles go to the weight-gradient kernel without padding, which needs C in whole 128-row tiles and 8 tiles in its group)"""
        if not (_SPLIT and _WG_BELOW) or len(chains[0]["lins"]) < 2:
            return None
        plan, t128 = [], 0
        dev = self.nn.go2nn_is_device_library() != 0          # (the host build takes any shape: the tile conditions are the device kernels')
        for c in chains:
            m0, x = c["lins"][0], c["acts"][0]
            K0 = m0.in_features
            if c["imgs"][1] is None or c["lins"][1].out_features < 4 or not x.is_contiguous() or x.dim() != 2 or x.shape[1] != K0:
                return None
            x0 = 0 if K0 <= 64 else K0 - K0 % 128
            if not 1 <= K0 - x0 <= 64:
                return None
            if x0 and dev and m0.out_features % 128:
                return None
            t128 += (m0.out_features // 128) * (x0 // 128)
            plan.append(x0)
        return plan if (t128 == 0 or t128 >= 8 or not dev) else None

    def bwd_in_blocks(self, do, w, y, imgs):
        """The E heads of a grouped layer (GroupedHeads: Conv1d(groups = E)) back into the shared matrix they read: do [E, M, C] (expert-major, dense), w [E, C, Kin],
        y [M, E * Kin] the ELU outputs the heads read -> (gz [M, E * Kin] = (do[e] w[e]) * elu'(y) in block e, gb [E * Kin] its column sums, valid after finish()).
        Pitched jobs of the grouped input gradient (Go2nnBwdInJob.ld), GO2NN_MAX_GROUP per launch: no [E, M, Kin] intermediate, no transposing copy, no separate ELU' pass."""
        from ..._nn import Go2nnBwdInJob, GO2NN_MAX_GROUP
        E, M, Co = do.shape
        Ki = w.shape[2]
        gz, gb = torch.empty_like(y), self.new(E * Ki)
        r = self.nn.go2nn_linear_backward_input_group_rows(M, Co, Ki)
        if r <= 0:
            raise RuntimeError("go2nn_linear_backward_input_group_rows: %s" % self.nn.go2nn_last_error().decode())
        wk = self.new(E, r * Ki)
        for e0 in range(0, E, GO2NN_MAX_GROUP):
            n = min(GO2NN_MAX_GROUP, E - e0)
            arr = (Go2nnBwdInJob * n)()
            for j in range(n):
                e = e0 + j
                arr[j] = Go2nnBwdInJob(do[e].data_ptr(), w[e].data_ptr(), y.data_ptr() + 4 * e * Ki, gz.data_ptr() + 4 * e * Ki, wk[e].data_ptr(), M, Co, Ki, 0,
                                       imgs[e].data_ptr() if imgs[e] is not None else None, E * Ki)
                self.sums.append((wk[e], gb[e * Ki:(e + 1) * Ki], r, Ki))
            self.check(self.nn.go2nn_linear_backward_input_group(arr, n, self.stream), "go2nn_linear_backward_input_group (pitched)")
        return gz, gb

    def chain_backward(self, chains, sink=None, need_gz0=True):
        """chains: 1 or 2 dicts {lins, acts, gz, gb, imgs} of equally many layers and one M: gz / gb = the gradient at lins[-1]'s output and its column sums;
        acts[l] = the input of lins[l] (acts[l > 0] an ELU output).  Sets .grad of every weight and bias; -> the gradients at lins[0]'s pre-activation
        (need_gz0 False, or one flag per chain: the caller does not read them — where the first layer's weight gradient comes out of the second layer's input-gradient
        launch whole (an input of at most 64 columns) they then never exist in HBM and the entry is None)."""
        n = len(chains[0]["lins"])
        need = list(need_gz0) if isinstance(need_gz0, (list, tuple)) else [need_gz0] * len(chains)
        gz, gb = [c["gz"] for c in chains], [c["gb"] for c in chains]
        for l in range(n - 1, -1, -1):
            self.wgrad([(gz[j], c["acts"][l], c["lins"][l]) for j, c in enumerate(chains)], sink, l)
            for j, c in enumerate(chains):
                if sink is None:
                    c["lins"][l].bias.grad = gb[j]
                else:
                    sink[(l, "b")] = gb[j]
            if l > 0:
                jobs = [(gz[j], c["lins"][l], c["acts"][l], c["imgs"][l]) for j, c in enumerate(chains)]
                plan = self.below_plan(chains, need) if l == 1 else None
                if plan is not None:
                    gz, gb, dws = self.bwd_in(jobs, below=[(c["acts"][0], c["lins"][0], plan[j], need[j] or plan[j] > 0) for j, c in enumerate(chains)], sink=sink, tag=0)
                    for j, c in enumerate(chains):
                        if sink is None:
                            c["lins"][0].bias.grad = gb[j]
                        else:
                            sink[(0, "b")] = gb[j]
                    wide = [j for j in range(len(chains)) if plan[j] > 0]
                    if wide:
                        self.wgrad([(gz[j], chains[j]["acts"][0], chains[j]["lins"][0]) for j in wide], lead=[(plan[j], dws[j]) for j in wide])
                    return [g if nd else None for g, nd in zip(gz, need)]
                gz, gb = self.bwd_in(jobs)
        return gz

    def finish(self):
        from ..._nn import Go2nnSumJob
        for k in range(0, len(self.sums), 32):
            chunk = self.sums[k:k + 32]
            arr = (Go2nnSumJob * len(chunk))()
            for i, t in enumerate(chunk):          # (part, out, rows, columns[, acc, nacc[, (byte offset into out, out_w, out_ld)]])
                acc = t[4] if len(t) > 4 else None
                off, ow, old = t[6] if len(t) > 6 else (0, 0, 0)
                arr[i] = Go2nnSumJob(t[0].data_ptr(), t[1].data_ptr() + off, t[2], t[3], acc.data_ptr() if acc is not None else None, t[5] if acc is not None else 0, 0, ow, old)
            self.check(self.nn.go2nn_sum_rows(arr, len(chunk), self.stream), "go2nn_sum_rows")
        self.sums = []


def _narrow(N, K):
    """the output layer goes through go2nn_head_backward (one streaming pass instead of two degenerate GEMMs: the 12-wide action mean, the 1-wide value, an 8-wide gate)"""
    return N <= 16 and K % 4 == 0 and K <= 512


def _stack_forward(k, x, lins, imgs, H, last_linear):
    """[Linear, ELU] x H (-> Linear) on the grouped kernels (single-job launches) -> the activations [x, h1, ..., hH(, out)]"""
    acts = [x]
    for l in range(H):
        acts.append(k.forward([(acts[-1], lins[l], imgs[l])])[0])
    if last_linear:
        m = lins[H]
        acts.append(torch.addmm(m.bias, acts[-1], m.weight.t()) if _narrow(m.out_features, m.in_features) else k.forward([(acts[-1], m, imgs[H])], act=1)[0])
    return acts


def _input_grad(k, gz0, lin0, img0):
    """d loss / d x = gz0 W0 (no activation in front of the first layer)"""
    if img0 is not None or (lin0.in_features % 4 == 0 and lin0.out_features >= 4):
        return k.bwd_in([(gz0, lin0, None, img0)], plain=True)[0][0]
    return gz0.mm(lin0.weight)          # (fp32-MFMA kernels, an input width that is not a multiple of 4: the one product left to the vendor GEMM)


class _FusedMLP(torch.autograd.Function):
    """x -> [Linear -> ELU] x H -> Linear as ONE autograd node on the grouped kernels of include/go2nn.h: the hidden layers' forward with the ELU in the epilogue;
    backward = go2nn_head_backward for a narrow output layer (its input / weight / bias gradients, the ELU' and the last hidden bias gradient in one streaming pass)
    or a column sum for a wide one (an encoder's 32-wide latent), then weight gradients and input gradients with the ELU' of the layer below in the epilogue, and ONE
    go2nn_sum_rows for every fixed-order reduction of the pass.  args: x, w1, b1, ..., wH, bH, w_out, b_out."""

    @staticmethod
    def forward(ctx, x, *params):
        ws, bs = params[0::2], params[1::2]
        H = len(ws) - 1
        k = _Launch(x.device)
        lins = [_LinearView(w, b) for w, b in zip(ws, bs)]
        narrow = _narrow(lins[H].out_features, lins[H].in_features)
        imgs = k.images(lins[:H] if narrow else lins) + ([None] if narrow else [])
        acts = _stack_forward(k, x, lins, imgs, H, True)
        ctx.save_for_backward(*acts[:H + 1], *ws)
        ctx.H, ctx.imgs, ctx.narrow = H, imgs, narrow
        return acts[H + 1]

    @staticmethod
    def backward(ctx, gout):
        H = ctx.H
        acts, ws = ctx.saved_tensors[:H + 1], ctx.saved_tensors[H + 1:]
        k = _Launch(gout.device)
        gout = gout.contiguous()
        B = gout.shape[0]
        lins = [_LinearView(w, None) for w in ws]
        grads = {}
        if ctx.narrow:
            y, w_out = acts[H], ws[H]
            Cn, K = w_out.shape
            n = _NN.go2nn_head_backward_workspace(B, Cn, K)
            if n < 0:
                raise RuntimeError("go2nn_head_backward_workspace: %s" % _NN.go2nn_last_error().decode())
            gz, sums, wk = torch.empty_like(y), k.new((Cn + 1) * K + Cn), k.new(int(n))
            p = lambda t: C.c_void_p(t.data_ptr())
            _check(_NN.go2nn_head_backward(p(gout), p(y), p(w_out), p(gz), None, p(wk), B, Cn, K, k.stream), "go2nn_head_backward", _NN)
            k.sums.append((wk, sums, _NN.go2nn_head_backward_rows(B, Cn, K), (Cn + 1) * K + Cn))
            grads[(H, "w")], grads[(H, "b")] = sums[:Cn * K].view(Cn, K), sums[(Cn + 1) * K:]
            gz0 = k.chain_backward([{"lins": lins[:H], "acts": acts, "gz": gz, "gb": sums[Cn * K:(Cn + 1) * K], "imgs": ctx.imgs}], sink=grads, need_gz0=ctx.needs_input_grad[0])
        else:
            gb = k.new(gout.shape[1])
            k.sums.append((gout, gb, B, gout.shape[1]))          # the output layer's bias gradient: column sums of gout (go2nn_sum_rows' tall shape)
            gz0 = k.chain_backward([{"lins": lins, "acts": acts, "gz": gout, "gb": gb, "imgs": ctx.imgs}], sink=grads, need_gz0=ctx.needs_input_grad[0])
        gx = _input_grad(k, gz0[0], lins[0], ctx.imgs[0]) if ctx.needs_input_grad[0] else None
        k.finish()
        out = []
        for l in range(H + 1):
            out += [grads[(l, "w")], grads[(l, "b")]]
        return (gx, *out)


class _FusedChain(torch.autograd.Function):
    """x -> [Linear -> ELU] x n (an activation behind EVERY layer: the experts' backbone of the MoE encoders, rsl_rl/rsl_rl/modules/utils.py:47-62 with last_activation=True)
    as ONE autograd node: forward = n single-job launches with the ELU in the epilogue; backward = one pass for the top layer's ELU' + bias gradient
    (go2sim_elu_backward_bias), then the chain as in _FusedMLP.  args: x, w1, b1, ..., wn, bn."""

    @staticmethod
    def forward(ctx, x, *params):
        ws, bs = params[0::2], params[1::2]
        k = _Launch(x.device)
        lins = [_LinearView(w, b) for w, b in zip(ws, bs)]
        imgs = k.images(lins)
        acts = _stack_forward(k, x, lins, imgs, len(lins), False)
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
        gz0 = k.chain_backward([{"lins": lins, "acts": acts, "gz": gz, "gb": gb, "imgs": ctx.imgs}], sink=grads, need_gz0=ctx.needs_input_grad[0])
        gx = _input_grad(k, gz0[0], lins[0], ctx.imgs[0]) if ctx.needs_input_grad[0] else None
        k.finish()
        out = []
        for l in range(n):
            out += [grads[(l, "w")], grads[(l, "b")]]
        return (gx, *out)


class _FusedChainHeads(torch.autograd.Function):
    """The experts of a MoE encoder (rsl_rl/rsl_rl/modules/utils.py:64-93): x -> [Linear -> ELU] x n (the shared backbone, E * hidden wide at its top) -> E linear heads
    hidden -> out WITHOUT their bias (the mixing kernel adds it) as ONE node -> outs [E, B, out], expert-major.  Forward = _FusedChain's + one batched product; backward:
    the heads' weight gradient as one batched product, their input gradient as pitched jobs of the grouped input-gradient kernel that write (do[e] W[e]) * elu'(y) straight
    into block e of the backbone's top gradient with its bias partials (round 5: this replaced a batched product, the transposing copy of its [E, B, hidden] result into
    [B, E * hidden] and a separate ELU' + column-sum pass: 135 -> 45 us of the student step at 8192 envs), then the chain as in _FusedChain.
    args: x, heads weight [E * out, hidden, 1], E, w1, b1, ..., wn, bn."""

    @staticmethod
    def forward(ctx, x, hw, E, *params):
        ws, bs = params[0::2], params[1::2]
        k = _Launch(x.device)
        lins = [_LinearView(w, b) for w, b in zip(ws, bs)]
        imgs = k.images(lins)
        acts = _stack_forward(k, x, lins, imgs, len(lins), False)
        y = acts[-1]
        B, hid = y.shape[0], y.shape[1] // E
        w3 = hw.view(E, -1, hid)                                                        # [E, out, hidden]
        wt, nout = w3.transpose(1, 2), w3.shape[1]
        if nout < 32 and y.is_cuda:          # (a strided-batched product with fewer than 32 output columns is ~100x slower on ROCm 7 — GroupedHeads._heads pads too)
            outs = torch.bmm(y.view(B, E, hid).transpose(0, 1), torch.nn.functional.pad(wt, (0, 32 - nout)))[..., :nout].contiguous()
        else:
            outs = torch.bmm(y.view(B, E, hid).transpose(0, 1), wt)         # [E, B, out]
        ctx.save_for_backward(*acts, *ws, hw)
        ctx.n, ctx.imgs, ctx.E = len(ws), imgs, E
        return outs

    @staticmethod
    def backward(ctx, do):
        n, E = ctx.n, ctx.E
        acts, ws, hw = ctx.saved_tensors[:n + 1], ctx.saved_tensors[n + 1:2 * n + 1], ctx.saved_tensors[2 * n + 1]
        k = _Launch(do.device)
        do = do.contiguous()
        y = acts[n]
        B, hid = y.shape[0], y.shape[1] // E
        w3 = hw.view(E, -1, hid)
        ghw = torch.bmm(do.transpose(1, 2), y.view(B, E, hid).transpose(0, 1)).reshape(hw.shape)          # d / d W[e] = do[e]^T y[:, block e]
        himgs = k.images([_LinearView(w3[e], None) for e in range(E)])
        gz, gb = k.bwd_in_blocks(do, w3, y, himgs)
        lins = [_LinearView(w, None) for w in ws]
        grads = {}
        gz0 = k.chain_backward([{"lins": lins, "acts": acts, "gz": gz, "gb": gb, "imgs": ctx.imgs}], sink=grads, need_gz0=ctx.needs_input_grad[0])
        gx = _input_grad(k, gz0[0], lins[0], ctx.imgs[0]) if ctx.needs_input_grad[0] else None
        k.finish()
        out = []
        for l in range(n):
            out += [grads[(l, "w")], grads[(l, "b")]]
        return (gx, ghw, None, *out)


def chain_heads(backbone, heads, x):
    """outs [E, B, out] (no bias) of GroupedHeads `heads` over the [Linear, ELU] x n Sequential `backbone` as one _FusedChainHeads node, or None when the pair is not
    covered (the library is not bound, no gradient is being recorded, other modules in the backbone, a hidden width that is not a multiple of 4, heads without gradient)."""
    if not isinstance(backbone, FusedSequential) or _LIB is None or _NN is None or not torch.is_grad_enabled() or x.dim() != 2 or x.dtype != torch.float32:
        return None
    if not (x.is_cuda or (_LIB.go2sim_is_device_library() == 0 and _NN.go2nn_is_device_library() == 0)):
        return None
    mods = list(backbone)
    st = _stack(mods)
    if st is None or st[1] or st[2] != len(mods):
        return None
    lins = st[0]
    E, hid, out = heads.groups, heads.cin, heads.cout
    if lins[-1].out_features != E * hid or hid % 4 or out < 4 or not heads.weight.requires_grad or heads.weight.dtype != torch.float32 or not heads.weight.is_contiguous():
        return None
    if not _SPLIT and _NN.go2nn_is_device_library() != 0 and hid % 4:
        return None
    x = x if x.is_contiguous() else x.contiguous()
    return _FusedChainHeads.apply(x, heads.weight, E, *[t for m in lins for t in (m.weight, m.bias)])


def _stack(mods):
    """The leading [Linear, ELU(1)] x H (+ Linear) run of a module list that the nodes cover -> (Linear modules, ends with a Linear, modules consumed), or None.
    Hidden widths are multiples of 4 (16-byte rows of the activations), every layer has a bias and trainable float32 parameters, at least 4 inputs; a narrow output
    layer (<= 16 wide) needs H >= 1 and a last hidden width up to 512 (go2nn_head_backward), a 1-wide one exists only in that form."""
    lins, k = [], 0
    ok = lambda m: (isinstance(m, nn.Linear) and m.bias is not None and m.weight.requires_grad and m.bias.requires_grad and m.weight.dtype == torch.float32 and m.in_features >= 4
                    and m.weight.is_contiguous())
    while k + 1 < len(mods) and ok(mods[k]) and isinstance(mods[k + 1], nn.ELU) and mods[k + 1].alpha == 1.0 and mods[k].out_features % 4 == 0:
        lins.append(mods[k]); k += 2
    if k < len(mods) and ok(mods[k]) and lins:
        m = mods[k]
        if _narrow(m.out_features, m.in_features) or (m.out_features >= 2 and (_SPLIT or m.out_features % 4 == 0 or (_NN is not None and _NN.go2nn_is_device_library() == 0))):
            return lins + [m], True, k + 1
    return (lins, False, k) if lins else None


class FusedSequential(nn.Sequential):
    """nn.Sequential whose leading Linear / ELU(alpha=1) stack runs as one autograd node on the library's kernels when gradients are being recorded (and, inside
    own_forward(), on the same kernels without a node when they are not); whatever follows the stack (a normaliser, a softmax) runs as the modules it is."""

    def forward(self, x):
        mods = list(self)
        on = (_LIB is not None and _NN is not None and x.dim() == 2 and x.dtype == torch.float32 and (x.is_cuda or (_LIB.go2sim_is_device_library() == 0 and _NN.go2nn_is_device_library() == 0)))
        grad = torch.is_grad_enabled()
        st = _stack(mods) if on and (grad or _OWN_FORWARD) else None
        if st is None:
            return super().forward(x)
        lins, last_linear, used = st
        x = x if x.is_contiguous() else x.contiguous()
        if grad:
            x = (_FusedMLP if last_linear else _FusedChain).apply(x, *[t for m in lins for t in (m.weight, m.bias)])
        else:
            k = _Launch(x.device)
            H = len(lins) - (1 if last_linear else 0)
            narrow = last_linear and _narrow(lins[H].out_features, lins[H].in_features)
            x = _stack_forward(k, x, lins, k.images(lins[:H] if narrow else lins) + ([None] if narrow else []), H, last_linear)[-1]
        for m in mods[used:]:
            x = m(x)
        return x



# ---- PPO's whole mini-batch gradient without autograd -----------------------------------------------------------------------------------------------------
# PPO.update evaluates the actor and the critic on the same mini-batch rows (ppo.py:131-133): every hidden layer of BOTH networks is one grouped launch, the narrow
# heads + the loss head + the heads' backward are go2nn_ppo_heads (one streaming pass over the last hidden activations), and since the loss kernel hands out the
# analytic gradients the update needs no autograd graph: forward, loss and backward are 14 explicit launches with the optimizer, the gradients are installed as .grad.


def pair_lins(actor, critic):
    """(actor Linears, critic Linears) when the two modules are [Linear, ELU] x H -> Linear stacks with the same hidden widths and heads go2nn_ppo_heads covers, else None"""
    sa, sc = _stack(list(actor)) if isinstance(actor, nn.Sequential) else None, _stack(list(critic)) if isinstance(critic, nn.Sequential) else None
    if sa is None or sc is None or not sa[1] or not sc[1] or sa[2] != len(list(actor)) or sc[2] != len(list(critic)):
        return None
    la, lc = sa[0], sc[0]
    if len(la) != len(lc) or len(la) < 2 or any(a.out_features != c.out_features for a, c in zip(la[:-1], lc[:-1])):
        return None
    K = la[-1].in_features
    if not (lc[-1].out_features == 1 and la[-1].out_features <= 16 and K % 4 == 0 and K <= 256):
        return None
    return la, lc


def ppo_pair_applicable(ac, xa, xc):
    if not (_LIB is not None and _NN is not None and hasattr(ac, "actor") and hasattr(ac, "critic") and hasattr(ac, "std")):
        return False
    if not (xa.dim() == 2 and xc.dim() == 2 and xa.dtype == torch.float32 and xc.dtype == torch.float32 and xa.shape[0] == xc.shape[0]
            and (xa.is_cuda or (_LIB.go2sim_is_device_library() == 0 and _NN.go2nn_is_device_library() == 0))):
        return False
    pl = pair_lins(ac.actor, ac.critic)
    return pl is not None and ac.std.dim() == 1 and ac.std.shape[0] == pl[0][-1].out_features and ac.std.requires_grad


def pair_grads(k, la, lc, xa, xc, std, batch, clip, vcoef, ecoef, use_clipped_value_loss, surrogate_split=0, acc=None, imgs=None, need_gz0=True):
    """Forward of two MLPs with grouped hidden layers, go2nn_ppo_heads, backward of the hidden layers; the reductions are queued on `k` (the caller finishes).
    batch: actions, old values, advantages, returns, old log-probs, old mu, old sigma.  imgs: (actor images, critic images) of the hidden layers when the caller split
    them together with other networks' weights.  Sets .grad of every parameter of both networks and of std.
    -> (sums [surrogate, value loss, KL, entropy | ...] — valid after k.finish() —, the gradients at the first layers' pre-activations)"""
    from ..._nn import Go2nnPpoHeads
    actions, old_values, adv, returns, old_logp, old_mu, old_sigma = batch
    H, B = len(la) - 1, xa.shape[0]
    cont = lambda t: t.detach() if t.is_contiguous() else t.detach().contiguous()
    flat = lambda t: cont(t).reshape(-1)
    if imgs is None:
        both = k.images(la[:H] + lc[:H])
        imgs = (both[:H], both[H:])
    ia, ic = imgs
    acts = [[cont(xa)], [cont(xc)]]
    for l in range(H):
        ys = k.forward([(acts[0][-1], la[l], ia[l]), (acts[1][-1], lc[l], ic[l])])
        acts[0].append(ys[0]); acts[1].append(ys[1])
    A, K = la[H].weight.shape
    rows, cols = _NN.go2nn_ppo_heads_rows(B, A, K), _NN.go2nn_ppo_heads_cols(A, K)
    if rows <= 0 or cols <= 0:
        raise RuntimeError("go2nn_ppo_heads: %s" % _NN.go2nn_last_error().decode())
    gz = [k.new(B, K), k.new(B, K)]
    part, tot = k.new(rows * cols), k.new(cols)
    keep = [cont(actions), cont(old_mu), cont(old_sigma), flat(old_logp), flat(adv), flat(old_values), flat(returns), cont(std)]
    p = lambda t: t.data_ptr()
    h = Go2nnPpoHeads(p(acts[0][H]), p(acts[1][H]), p(la[H].weight), p(la[H].bias), p(lc[H].weight), p(lc[H].bias), p(keep[7]), p(keep[0]), p(keep[1]), p(keep[2]), p(keep[3]),
                      p(keep[4]), p(keep[5]), p(keep[6]), p(gz[0]), p(gz[1]), p(part), B, A, K, int(bool(use_clipped_value_loss)), float(clip), float(vcoef), float(ecoef), int(surrogate_split))
    k.check(_NN.go2nn_ppo_heads(C.byref(h), k.stream), "go2nn_ppo_heads")
    k.keep = keep          # (alive until the launches have run)
    k.sums.append((part, tot, rows, cols, acc, 4 if acc is not None and acc.numel() >= 4 else (2 if acc is not None else 0)))
    o = 4 + A
    std.grad = tot[4:o].view_as(std)
    la[H].weight.grad, gb_a, la[H].bias.grad = tot[o:o + A * K].view(A, K), tot[o + A * K:o + (A + 1) * K], tot[o + (A + 1) * K:o + (A + 1) * K + A]
    o += (A + 1) * K + A
    lc[H].weight.grad, gb_c, lc[H].bias.grad = tot[o:o + K].view(1, K), tot[o + K:o + 2 * K], tot[o + 2 * K:o + 2 * K + 1]
    gz1 = k.chain_backward([{"lins": la[:H], "acts": acts[0], "gz": gz[0], "gb": gb_a, "imgs": ia}, {"lins": lc[:H], "acts": acts[1], "gz": gz[1], "gb": gb_c, "imgs": ic}], need_gz0=need_gz0)
    return tot, gz1


def ppo_pair_grads(ac, xa, xc, actions, old_values, adv, returns, old_logp, old_mu, old_sigma, clip, vcoef, ecoef, use_clipped_value_loss, acc=None):
    """One PPO mini-batch gradient (ppo.py:131-170 + loss.backward()) of a plain ActorCritic as explicit launches; every parameter's .grad is set (replaced, as after
    zero_grad(set_to_none=True)).  acc: optional float32[>= 2] device tensor — the mini-batch's surrogate and value loss are ADDED to acc[0:2] by the pass's
    go2nn_sum_rows launch (the update's running sums).  -> stats [surrogate, value loss, KL, entropy] (means, device tensor)"""
    la, lc = pair_lins(ac.actor, ac.critic)
    k = _Launch(xa.device)
    with torch.no_grad():
        tot, _ = pair_grads(k, la, lc, xa, xc, ac.std, (actions, old_values, adv, returns, old_logp, old_mu, old_sigma), clip, vcoef, ecoef, use_clipped_value_loss, 0, acc, need_gz0=False)
        k.finish()
    return tot[:4]
