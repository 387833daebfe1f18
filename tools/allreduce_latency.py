#!/usr/bin/env python3
"""TOOL: latency of the update's gradient all-reduce (one flat bucket of 489 k floats + KL, sum) on THIS box's RCCL group — with the pool's 1-GPU boxes a 1-rank group:
the fixed cost of a collective (launch, RCCL's own kernel) without any wire time.  DESIGN.md section 7 builds the 8-rank prediction on it.
   python tools/allreduce_latency.py [floats]      (GPU; under torchrun it measures the real group)"""
import os, sys, time
import torch
import torch.distributed as dist
n = int(sys.argv[1]) if len(sys.argv) > 1 else 489_000
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
dist.init_process_group("nccl", rank=rank, world_size=world)
x = torch.randn(n, device="cuda")
for _ in range(20):
    dist.all_reduce(x)
torch.cuda.synchronize()
for reps in (1, 50):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record()
    for _ in range(reps):
        dist.all_reduce(x)
    b.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
    if rank == 0:
        print("all_reduce of %d floats (%.2f MB), %d rank(s), %d back to back: %.1f us each on the stream, %.1f us each wall clock" % (n, n * 4 / 1e6, world, reps, a.elapsed_time(b) * 1e3 / reps, (t1 - t0) * 1e6 / reps))
dist.destroy_process_group()
