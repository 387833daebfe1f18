"""A step function captured into a HIP graph after a few eager warm-up calls (allocator / lazy-init settle on a side stream),
then replayed.  If the capture fails the step keeps running eagerly — slower, never wrong."""
import torch


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
            except Exception as e:      # noqa: BLE001 — any capture problem degrades to eager execution
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
