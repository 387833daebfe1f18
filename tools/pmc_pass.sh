#!/bin/bash
# Two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE cannot share a pass: TCC has 4 slots, MI355X_MICROARCH.md) over
# tools/step_only.py, plus a --stats pass; writes gpurun_out/pmc/{pmc_step_kernel.json,step_only_kernel_stats.csv}.
# Usage on the GPU box:  bash tools/pmc_pass.sh [num_envs] [steps]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-4096}; S=${2:-100}
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o pmc -- python $R/tools/step_only.py $N $S > /tmp/pmc_$c.log 2>&1
done
rm -rf /tmp/pmc_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmc_stats -o step_only -- python $R/tools/step_only.py $N $S > /tmp/pmc_stats.log 2>&1
cp /tmp/pmc_stats/step_only_kernel_stats.csv $R/gpurun_out/pmc/ 2>/dev/null
python3 - "$N" "$R/gpurun_out/pmc/pmc_step_kernel.json" <<'PY'
import csv, glob, json, sys
N = int(sys.argv[1]); out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/pmc_%s/*counter_collection.csv" % c)[0]
    rows = [r for r in csv.DictReader(open(f)) if "go2_step_kernel<3>" in r["Kernel_Name"] and r["Counter_Name"] == c]
    v = [float(r["Counter_Value"]) for r in rows]
    out[c] = {"launches": len(v), "mean_kb": sum(v) / len(v), "min_kb": min(v), "max_kb": max(v)}
    out["vgpr"], out["agpr"], out["scratch_bytes_per_lane"], out["lds_bytes"] = int(rows[0]["VGPR_Count"]), int(rows[0]["Accum_VGPR_Count"]), int(rows[0]["Scratch_Size"]), int(rows[0]["LDS_Block_Size"])
st = [r for r in csv.DictReader(open(glob.glob("/tmp/pmc_stats/*kernel_stats.csv")[0])) if "go2_step_kernel<3>" in r["Name"]][0]
out["kernel_avg_us_rocprof_stats"] = float(st["AverageNs"]) / 1e3
out["num_envs"] = N
# unit: KB (x1024).  gfx950 correction of MI355X_MICROARCH.md section HBM: FETCH_SIZE reports half of a wide coalesced read stream ->
# doubled; WRITE_SIZE is taken as is (uncalibrated for this access pattern, stated so).
out["hbm_bytes_per_launch_raw"] = (out["FETCH_SIZE"]["mean_kb"] + out["WRITE_SIZE"]["mean_kb"]) * 1024
out["hbm_bytes_per_launch_corrected"] = (2 * out["FETCH_SIZE"]["mean_kb"] + out["WRITE_SIZE"]["mean_kb"]) * 1024
out["algorithmic_bytes_per_launch"] = 2936 * N
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
PY
