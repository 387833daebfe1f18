#!/bin/bash
# SQ instruction-mix / stall counters of go2_step_kernel<3> over tools/step_only.py (two 8-counter passes).
# Usage on the GPU box:  bash tools/sq_pass.sh [num_envs] [steps]   -> gpurun_out/pmc/sq_step_kernel.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-4096}; S=${2:-60}; T=${3:-go2_flat}
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_FLAT"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf /tmp/sq_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/sq_$i -o sq -- python $R/tools/step_only.py $N $S --task $T > /tmp/sq_$i.log 2>&1 || tail -5 /tmp/sq_$i.log
done
python3 - "$N" "$R/gpurun_out/pmc/${T}_sq_step_kernel.json" "$T" "$R" <<'PY'
import csv, glob, json, sys, collections
import hashlib
out = {"num_envs": int(sys.argv[1]), "task": sys.argv[3], "lib_sha256_16": hashlib.sha256(open(sys.argv[4] + "/go2_rl_gym_amd/libgo2sim_hip.so", "rb").read()).hexdigest()[:16]}
for i in (1, 2):
    fs = glob.glob("/tmp/sq_%d/*counter_collection.csv" % i)
    if not fs:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "go2_step_kernel<3>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            out["vgpr"], out["agpr"], out["scratch_bytes_per_lane"] = int(r["VGPR_Count"]), int(r["Accum_VGPR_Count"]), int(r["Scratch_Size"])
    for k, v in acc.items():
        out[k] = sum(v) / len(v)
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
PY
