#!/bin/bash
# Memory-path counters (TA / TCP / TCC) of one learner GEMM (tools/gemm_one.py).  Usage on the GPU box: bash tools/gemm_mem_pmc.sh <tag> f|i|w K N -> gpurun_out/pmc/gemm_mem_<tag>.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
PASSES=("TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TA_BUSY_sum" \
        "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_avr" \
        "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
        "TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
        "GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum")
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1)); rm -rf /tmp/gm_$i
  timeout 100 rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/gm_$i -o gm -- python $R/tools/gemm_one.py "$@" > /tmp/gm_$i.log 2>&1 || { echo "pass $i failed:"; tail -3 /tmp/gm_$i.log; }
done
python3 - "$R/gpurun_out/pmc/gemm_mem_$TAG.json" "$@" <<'PY'
import csv, glob, json, sys, collections
out = {"args": sys.argv[2:]}
for i in range(1, 6):
    fs = glob.glob("/tmp/gm_%d/*counter_collection.csv" % i)
    if not fs:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "go2nn_gemm_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"])); out["kernel"] = r["Kernel_Name"][:80]
    for k, v in acc.items():
        out[k] = sum(v) / len(v)
    fs = glob.glob("/tmp/gm_%d/*kernel_trace.csv" % i)
    if fs:
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(fs[0])) if "go2nn_gemm_kernel" in r["Kernel_Name"]]
        if d: out["us_pass%d" % i] = sum(d) / len(d) / 1e3
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out))
PY
