"""A step function captured into a HIP graph after a few eager warm-up calls (allocator / lazy-init settle on a side stream),
then replayed.  If the capture fails the step keeps running eagerly — slower, never wrong.

With more than one rank a mini-batch step is TWO captured halves with the gradient all-reduce issued eagerly between their replays
(ReducedStep + GradBucket): forward/backward/pack | RCCL all-reduce | unpack/LR decision/clip/Adam.  No collective is ever recorded
into a HIP graph (no graph-captured communicator state, no mixing of captured and eager collectives on one communicator), at the
price of one extra graph launch per mini-batch.  GO2_GRAPH_COLLECTIVES=1 records the all-reduce inside a single graph instead."""
import os

import torch
import torch.distributed as dist


def collectives_in_graph():
    return os.environ.get("GO2_GRAPH_COLLECTIVES", "0") == "1"


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

    def unpack(self, world):
        """-> the extras (shard mean); the parameters' .grad are the shard-mean gradients"""
        self.flat.div_(world)
        for p, v in zip(self.params, self.views):
            p.grad = v
        return self.extra


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
    def __init__(self, fn, enabled=True, warmup=3, name="step"):
        self.fn, self.enabled, self.warmup, self.name = fn, enabled, warmup, name
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
                with torch.cuda.graph(g):
                    self.fn()
            except Exception as e:      # noqa: BLE001 — any capture problem degrades to eager execution (unless GO2_STRICT_GRAPHS=1)
                if strict_graphs():
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
