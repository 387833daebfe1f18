#!/usr/bin/env python3
"""TOOL: register / scratch / LDS / occupancy table of the kernels of one HIP source as hipcc reports them (-Rpass-analysis=kernel-resource-usage).
usage: kres.py [-D...] [filter]        (go2nn_impl.cpp; the library is written to /tmp)"""
import re, subprocess, sys
defs = [a for a in sys.argv[1:] if a.startswith("-")]
flt = [a for a in sys.argv[1:] if not a.startswith("-")]
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-cuid=go2nn", "-Rpass-analysis=kernel-resource-usage",
                    "-o", "/tmp/kres.so", "go2_rl_gym_amd/csrc/go2nn_impl.cpp"] + defs, capture_output=True, text=True)
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark: (.*?)\s*\[-Rpass", line)
    if not m:
        if "error" in line: print(line)
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
for c in rows:
    n = subprocess.run(["c++filt", c["name"]], capture_output=True, text=True).stdout.strip()
    if flt and not any(f in n for f in flt): continue
    print("%-70s VGPR %4s AGPR %3s scratch %4s occ %s LDS %6s" % (n[:70], c.get("VGPRs"), c.get("AGPRs"), c.get("ScratchSize [bytes/lane]"), c.get("Occupancy [waves/SIMD]"), c.get("LDS Size [bytes/block]")))
