#!/usr/bin/env python3
"""Static instruction profile of go2_step_kernel<PHYS|POST>: compiles the kernel with phase markers (GO2_ISA_MARKS) and counts the
instructions between them.  No GPU needed.   python tools/isa_profile.py [extra hipcc flags]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/go2_isa.s"
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffast-math", "-fno-slp-vectorize", "-DGO2_ISA_MARKS", "-S", "--cuda-device-only", "-o", out,
                os.path.join(ROOT, "go2_rl_gym_amd", "csrc", "go2sim_impl.cpp")] + sys.argv[1:], check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
m = re.search(r'^_Z15go2_step_kernelILi3EEvPK11Go2DevBlockPKfi14Go2StepOutputs:.*?\n(.*?)s_endpgm', s, re.S | re.M)
cur, acc = "pre", collections.OrderedDict()
for l in m.group(1).split("\n"):
    t = l.strip()
    mk = re.search(r"GO2MARK (\d+)", t)
    if mk:
        cur = "after mark " + mk.group(1); continue
    if not t or t[0] in ";." or re.match(r"^\.?LBB", t):
        continue
    a = acc.setdefault(cur, collections.Counter())
    op = t.split()[0]
    a["total"] += 1
    a["valu"] += op.startswith("v_")
    a["dpp"] += "dpp" in t
    a["accvgpr"] += "accvgpr" in op
    a["scratch"] += op.startswith("scratch_")
    a["ds"] += op.startswith("ds_")
    a["global"] += op.startswith("global_")
    a["salu"] += op.startswith("s_")
    a["trans"] += bool(re.match(r"v_(sin|cos|sqrt|rsq|rcp|exp|log)", op))
for k, a in acc.items():
    print("%-16s total %5d valu %5d (dpp %3d accvgpr %3d trans %3d) scratch %3d ds %3d global %3d salu %4d" % (k, a["total"], a["valu"], a["dpp"], a["accvgpr"], a["trans"], a["scratch"], a["ds"], a["global"], a["salu"]))
for k in ("vgpr_count", "agpr_count", "private_segment_fixed_size", "vgpr_spill_count", "sgpr_count"):
    mm = re.findall(r"\.%s:\s+(\d+)" % k, s)
    print(k, mm[:6])
