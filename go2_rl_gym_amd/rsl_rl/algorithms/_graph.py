"""A step function captured into a HIP graph after a few eager warm-up calls (allocator / lazy-init settle on a side stream),
then replayed.  If the capture fails the step keeps running eagerly — slower, never wrong.

With more than one rank a mini-batch step is TWO captured halves with the gradient all-reduce issued eagerly between their replays
(ReducedStep + GradBucket): forward/backward/pack | RCCL all-reduce | unpack/LR decision/clip/Adam.  No collective is ever recorded
into a HIP graph (no graph-captured communicator state, no mixing of captured and eager collectives on one communicator), at the
price of one extra graph launch per mini-batch (recording it inside the graph measured 0.6 % slower still with one rank, round 2, and is gone)."""
import os

import torch
import torch.distributed as dist


class GradBucket:
    """One flat fp32 buffer [grad of every parameter | extras] at a fixed address: written by the captured front half, all-reduced
    eagerly, read by the captured back half through per-parameter views that become the parameters' .grad."""

    def __init__(self, params, n_extra=0):
        self.params = [p for p in params if p.grad is not None]
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n + n_extra, device=self.params[0].device, dtype=self.params[0].dtype)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p)); off += p.numel()
        self.extra = self.flat[n:]

    def pack(self, extra=None):
        parts = [p.grad.reshape(-1) for p in self.params]
        if self.extra.numel():
            parts.append(extra.detach().reshape(-1))
        torch.cat(parts, out=self.flat)
        for p in self.params:
            p.grad = None              # the back half installs the bucket views

    def reduce(self):
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)

    def reduce_async(self):
        """-> work handle; the collective runs on the communicator's own stream behind what the current stream has enqueued so far"""
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True)

    def unpack(self, world):
        """-> the extras (shard mean); the parameters' .grad are the shard-mean gradients"""
        self.flat.div_(world)
        for p, v in zip(self.params, self.views):
            p.grad = v
        return self.extra


class FusedClipAdam:
    """`[adaptive-KL learning rate;] clip_grad_norm_(params, max_norm); optimizer.step()` of one mini-batch step (ppo.py:140-155,178-181) as ONE
    library call (go2sim_adam_clip_step: two kernels) instead of ~30 element-wise / multi-tensor launches — with HIP-graph replay the update's
    tail is launch-bound, 180 us of 1.1 ms per mini-batch.  Works on the torch optimizer's OWN state tensors (exp_avg, exp_avg_sq, step), so
    optimizer.state_dict() / load_state_dict() and the checkpoint format are untouched.  `usable` is False (callers keep the torch path) for
    anything the kernel does not cover: no library, GO2_FUSED_ADAM=0, weight decay, amsgrad, maximize, differing groups, too many tensors."""

    def __init__(self, lib, optimizer, params, max_grad_norm):
        self.lib, self.opt, self.params, self.max_norm = lib, optimizer, list(params), float(max_grad_norm)
        self._ws = None
        gs = optimizer.param_groups
        g0 = gs[0]
        same = all(g["betas"] == g0["betas"] and g["eps"] == g0["eps"] and g["lr"] is g0["lr"] for g in gs)
        plain = all(g.get("weight_decay", 0) == 0 and not g.get("amsgrad", False) and not g.get("maximize", False) for g in gs)
        self.usable = bool(lib is not None and hasattr(lib, "go2sim_adam_clip_step") and os.environ.get("GO2_FUSED_ADAM", "1") == "1" and same and plain
                           and torch.is_tensor(g0["lr"]) and g0["lr"].dtype == torch.float32
                           and {id(p) for g in gs for p in g["params"]} >= {id(p) for p in self.params})

    def step(self, kl_mean=None, desired_kl=0.0):
        """-> False if this call could not be served (the caller then runs the torch formulation)"""
        import ctypes as C
        ps = [p for p in self.params if p.grad is not None]
        abi = self.lib.abi
        if not ps or len(ps) > abi.GO2_ADAM_MAX_TENSORS or any(p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous() for p in ps):
            return False
        g0 = self.opt.param_groups[0]
        t = abi.AdamTensors()
        t.count = len(ps)
        fp = C.POINTER(self.lib.abi.real)
        for i, p in enumerate(ps):
            st = self.opt.state[p]
            if len(st) == 0:                    # torch.optim.Adam._init_group (capturable: the step counter is a device tensor)
                st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if not (torch.is_tensor(st["step"]) and st["step"].device == p.device and st["step"].dtype == torch.float32):
                return False
            t.numel[i] = p.numel()
            for name, ten in (("param", p.data), ("grad", p.grad), ("exp_avg", st["exp_avg"]), ("exp_avg_sq", st["exp_avg_sq"]), ("step", st["step"])):
                getattr(t, name)[i] = C.cast(C.c_void_p(ten.data_ptr()), fp)
        n = self.lib.go2sim_adam_workspace_len(C.byref(t))
        if self._ws is None or self._ws.numel() < n or self._ws.device != ps[0].device:
            self._ws = torch.empty(max(n, 256), dtype=torch.float32, device=ps[0].device)
        lr = g0["lr"]
        stream = C.c_void_p(torch.cuda.current_stream(lr.device).cuda_stream) if lr.is_cuda else None
        klp = None
        if kl_mean is not None:
            self._kl = kl_mean.detach().reshape(-1)[:1].contiguous().float()          # (kept alive until the enqueued kernels have run)
            klp = C.c_void_p(self._kl.data_ptr())
        rc = self.lib.go2sim_adam_clip_step(C.byref(t), C.c_void_p(lr.data_ptr()), klp, float(desired_kl), self.max_norm, float(g0["betas"][0]), float(g0["betas"][1]),
                                            float(g0["eps"]), C.c_void_p(self._ws.data_ptr()), stream)
        if rc != 0:
            raise RuntimeError("go2sim_adam_clip_step failed: %s" % self.lib.go2sim_last_error().decode())
        return True


def load_optimizer_state(optimizer, state_dict):
    """optimizer.load_state_dict(state_dict) that KEEPS the addresses of state tensors which already exist (exp_avg, exp_avg_sq, step): the
    loaded values are copied into them.  torch's load_state_dict installs fresh tensors; a HIP graph captured earlier (the fused clip+Adam
    kernels, or torch's capturable Adam) would go on updating the old, by then freed, buffers."""
    old = {p: dict(st) for p, st in optimizer.state.items()}
    optimizer.load_state_dict(state_dict)
    for p, st in optimizer.state.items():
        for k, v in old.get(p, {}).items():
            new = st.get(k)
            if torch.is_tensor(v) and torch.is_tensor(new) and new is not v and new.shape == v.shape:
                v.copy_(new)
                st[k] = v


class no_gc:
    """Python's cyclic garbage collector off for the duration of a stream capture: a collection that happens to run inside the capture may
    finalise device objects of EARLIER captures / runners (graphs, streams, events), which the HIP runtime answers with an abort.
    (torch.cuda.graph collects once on entry; garbage created while capturing — ctypes argument structs, autograd nodes — can trigger more.)"""

    def __enter__(self):
        import gc
        self._was = gc.isenabled()
        gc.collect()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self._was:
            gc.enable()
        return False


def strict_graphs():
    """GO2_STRICT_GRAPHS=1 (bench.py sets it): a failed HIP-graph capture raises instead of degrading to eager execution."""
    return os.environ.get("GO2_STRICT_GRAPHS", "0") == "1"


def all_captured(steps):
    """True iff every CapturedStep / ReducedStep of `steps` (nested lists allowed) is being replayed from a HIP graph."""
    if steps is None:
        return False
    if isinstance(steps, (list, tuple)):
        return len(steps) > 0 and all(all_captured(s) for s in steps)
    if isinstance(steps, ReducedStep):
        return steps.front.graph is not None and steps.back.graph is not None
    return steps.graph is not None


class CapturedStep:
    def __init__(self, fn, enabled=True, warmup=3, name="step", optional=False):
        """optional: a failed capture silently keeps this step eager even under GO2_STRICT_GRAPHS (a convenience graph, not one of the halves
        bench.py reports on)."""
        self.fn, self.enabled, self.warmup, self.name, self.optional = fn, enabled, warmup, name, optional
        self.graph, self.calls = None, 0

    def __call__(self):
        if not self.enabled:
            return self.fn()
        if self.graph is not None:
            return self.graph.replay()
        if self.calls >= self.warmup:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            try:
                with no_gc(), torch.cuda.graph(g):
                    self.fn()
            except Exception as e:      # noqa: BLE001 — any capture problem degrades to eager execution (unless GO2_STRICT_GRAPHS=1)
                if strict_graphs() and not self.optional:
                    raise RuntimeError("HIP-graph capture of %s failed (%s: %s)" % (self.name, type(e).__name__, e)) from e
                print("[go2_rl_gym_amd] HIP-graph capture of %s failed (%s: %s); continuing eagerly" % (self.name, type(e).__name__, e))
                self.enabled = False
                torch.cuda.synchronize()
                return self.fn()
            self.graph = g              # the capture executed nothing: this call's work is the first replay
            return g.replay()
        cur = torch.cuda.current_stream()
        s = torch.cuda.Stream()
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            self.fn()
        cur.wait_stream(s)
        self.calls += 1


class ReducedStep:
    """front() | bucket().reduce() | back() — the two halves captured separately, the collective eager between the replays."""

    def __init__(self, front, back, bucket, enabled=True, warmup=3, name="step"):
        self.front = CapturedStep(front, enabled, warmup, name + " (forward/backward)")
        self.back = CapturedStep(back, enabled, warmup, name + " (optimizer)")
        self.bucket = bucket

    def __call__(self):
        self.front()
        self.bucket().reduce()
        self.back()
