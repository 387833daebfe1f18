#!/usr/bin/env python3
"""Timeline of one rollout step and one mini-batch update out of a rocprofv3 kernel trace of bench.py.
On the GPU box:  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 6 --no-cpu-baseline
                 python tools/trace_timeline.py /tmp/kt/kt_kernel_trace.csv > gpurun_out/timeline.txt"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
k = sorted(((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]) for r in rows), key=lambda r: r[1])[-7000:]
idx = [i for i, r in enumerate(k) if "go2_step_kernel<3>" in r[0]]
def show(seg, title):
    t0 = seg[0][1]
    print("==== %s: %d kernels, %.1f us" % (title, len(seg), (seg[-1][2] - t0) / 1e3))
    busy = 0.0
    for r in seg:
        print("%8.1f %8.1f d %6.1f q%s %s" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[0][:100]))
for a, b in zip(idx, idx[1:]):
    if a > idx[3] and (k[b][1] - k[a][1]) < 400e3:
        show(k[a:b + 1], "one rollout step (step kernel .. next step kernel)"); break
for a, b in zip(idx, idx[1:]):
    if (k[b][1] - k[a][1]) > 5e6:
        seg = k[a:b]
        L = [i for i, r in enumerate(seg) if r[0].startswith("go2_ppo_loss_kernel") or "go2nn_ppo_heads_kernel" in r[0]]
        if len(L) < 5:
            continue          # (a pause that is not an update: warm-up, graph capture)
        print("update: %.2f ms, %d kernels, %d mini-batches" % ((seg[-1][2] - seg[0][1]) / 1e6, len(seg), len(L)))
        show(seg[L[3]:L[4] + 1], "one mini-batch (loss / heads kernel .. the next one)")
        pre = seg[:L[0]]
        show(pre[-60:], "before the first mini-batch (GAE, permutation gathers, first forward)")
        # the CTS family's student phase: optimizer steps that follow no loss-head kernel; one step = Adam step .. the next Adam step
        A = [i for i, r in enumerate(seg) if r[0].startswith("go2_adam_step_kernel")]
        stu = [(a, b) for a, b in zip(A, A[1:]) if a > L[-1]]
        if len(stu) > 4:
            a, b = stu[len(stu) // 2]
            print("student phase: %d steps, %.2f ms" % (len(stu) + 1, (seg[A[-1]][2] - seg[stu[0][0]][2]) / 1e6))
            show(seg[a + 1:b + 1], "one student step (behind an Adam step .. the next Adam step)")
        break
