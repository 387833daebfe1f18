#!/bin/bash
# TOOL: SQ counters of the rollout's policy kernel (tools/policy_bench.py).  On the GPU box: [PASSES="1 3"] bash tools/policy_pmc.sh <tag> [N]  -> gpurun_out/pmc/policy_<tag>.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; N=${2:-4096}
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
P3="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
P4="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1)); rm -rf /tmp/pp_$i
  case " ${PASSES:-1 2 3 4} " in *" $i "*) ;; *) continue;; esac
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/pp_$i -o pp -- python $R/tools/policy_bench.py $N > /tmp/pp_$i.log 2>&1 || tail -5 /tmp/pp_$i.log
done
python3 - "$R/gpurun_out/pmc/policy_$TAG.json" <<'PY'
import csv, glob, json, sys, collections
out = {}
for i in (1, 2, 3, 4):
    fs = glob.glob("/tmp/pp_%d/*counter_collection.csv" % i)
    if not fs:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "go2nn_mlp" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            out["kernel"], out["vgpr"], out["lds"] = r["Kernel_Name"][:60], int(r["VGPR_Count"]), int(r["LDS_Block_Size"])
    for k, v in acc.items():
        out[k] = sum(v) / len(v)
    fs = glob.glob("/tmp/pp_%d/*kernel_trace.csv" % i)
    if fs:
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(fs[0])) if "go2nn_mlp" in r["Kernel_Name"]]
        out["us_pass%d" % i] = sum(d) / len(d) / 1e3
if "GRBM_GUI_ACTIVE" in out and "us_pass3" in out:
    cyc = out["GRBM_GUI_ACTIVE"] / 8          # summed over the 8 XCDs
    out["gui_active_cycles_per_xcd"] = cyc          # (includes the launch's dispatch and drain: cyc / us_pass3 = 2.8 "GHz" for this 36 us kernel — NOT a clock; the GEMMs' 80-100 us launches give 2.17)
    out["mfma_busy_frac_of_active_cycles"] = out.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc)
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out))
PY
