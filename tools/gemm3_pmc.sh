#!/bin/bash
# TOOL: SQ counters of one grouped learner GEMM (build/gemm3_bench ... one).  On the GPU box: bash tools/gemm3_pmc.sh <tag> f|i|w <layer 1|2|3>  -> gpurun_out/pmc/gemm3_<tag>.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
P3="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1)); rm -rf /tmp/gp_$i
  timeout 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/gp_$i -o gp -- ${GEMM3_BENCH:-$R/build/gemm3_bench} $R/go2_rl_gym_amd/libgo2nn_hip.so 24576 one "$@" > /tmp/gp_$i.log 2>&1 || tail -5 /tmp/gp_$i.log
done
python3 - "$R/gpurun_out/pmc/gemm3_$TAG.json" "$@" <<'PY'
import csv, glob, json, sys, collections
out = {"args": sys.argv[2:]}
for i in (1, 2, 3):
    fs = glob.glob("/tmp/gp_%d/*counter_collection.csv" % i)
    if not fs:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if any(n in r["Kernel_Name"] for n in ("go2nn_gemm3_kernel", "go2nn_wgrad_kernel", "go2nn_bx3_kernel")):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            out["kernel"], out["vgpr"], out["agpr"], out["lds"] = r["Kernel_Name"][:80], int(r["VGPR_Count"]), int(r["Accum_VGPR_Count"]), int(r["LDS_Block_Size"])
    for k, v in acc.items():
        out[k] = sum(v) / len(v)
    fs = glob.glob("/tmp/gp_%d/*kernel_trace.csv" % i)
    if fs:
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(fs[0])) if any(n in r["Kernel_Name"] for n in ("go2nn_gemm3_kernel", "go2nn_wgrad_kernel", "go2nn_bx3_kernel"))]
        out["us_pass%d" % i] = sum(d) / len(d) / 1e3
if "GRBM_GUI_ACTIVE" in out and "us_pass3" in out:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs (round 4's files divided by the launch time only and showed "17.4 GHz" / "5 % busy")
    XCDS, SIMDS = 8, 1024
    cyc = out["GRBM_GUI_ACTIVE"] / XCDS                                   # active cycles of the launch, per XCD
    out["clock_GHz"] = cyc / out["us_pass3"] / 1e3
    out["mfma_busy_frac_of_active_cycles"] = out.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (SIMDS * cyc)          # busy cycles summed over 1024 SIMDs / (SIMDs x active cycles)
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out))
PY
