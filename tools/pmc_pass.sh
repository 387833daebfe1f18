#!/bin/bash
# rocprofv3 PMC passes over tools/step_only.py for the shipped go2_step_kernel<PHYS|POST>: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (TCC has 4 slots, MI355X_MICROARCH.md), each also over the calibration probe (a kernel of KNOWN byte count in the step kernel's access
# pattern, go2sim_debug_traffic_probe), plus a --stats pass.  Writes gpurun_out/pmc/<tag>_pmc_step_kernel.json with the raw counters, the
# measured calibration factors, the traffic corrected by them, and the identity of the profiled binary (sha256 of the .so + the kernel's
# VGPR / AGPR / scratch / LDS as rocprofv3 reports them) — bench.py quotes roofline.traffic only for the library with that sha.
# Usage on the GPU box:  bash tools/pmc_pass.sh [num_envs] [steps] [task: go2_flat|go2]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-4096}; S=${2:-100}; T=${3:-go2_flat}
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c /tmp/cal_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o pmc -- python $R/tools/step_only.py $N $S --task $T > /tmp/pmc_$c.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/cal_$c -o pmc -- python $R/tools/step_only.py $N 50 --probe > /tmp/cal_$c.log 2>&1
done
rm -rf /tmp/pmc_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmc_stats -o step_only -- python $R/tools/step_only.py $N $S --task $T > /tmp/pmc_stats.log 2>&1
cp /tmp/pmc_stats/step_only_kernel_stats.csv $R/gpurun_out/pmc/${T}_step_only_kernel_stats.csv 2>/dev/null
python3 - "$N" "$T" "$R" <<'PY'
import csv, glob, hashlib, json, sys
N, task, R = int(sys.argv[1]), sys.argv[2], sys.argv[3]
out = {"num_envs": N, "task": task}
NR, NW = 200, 540
def rows(d, c, name):
    f = glob.glob("/tmp/%s_%s/*counter_collection.csv" % (d, c))[0]
    return [r for r in csv.DictReader(open(f)) if name in r["Kernel_Name"] and r["Counter_Name"] == c]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rr = rows("pmc", c, "go2_step_kernel<3>")
    v = [float(r["Counter_Value"]) for r in rr][10:]            # skip the settling steps right after reset
    out[c] = {"launches": len(v), "mean_kb": sum(v) / len(v), "min_kb": min(v), "max_kb": max(v)}
    out["vgpr"], out["agpr"], out["scratch_bytes_per_lane"], out["lds_bytes"] = int(rr[0]["VGPR_Count"]), int(rr[0]["Accum_VGPR_Count"]), int(rr[0]["Scratch_Size"]), int(rr[0]["LDS_Block_Size"])
    cc = [float(r["Counter_Value"]) for r in rows("cal", c, "go2_traffic_probe_kernel")][5:]
    known = (NR if c == "FETCH_SIZE" else NW) * N * 4
    out["calibration_" + c] = {"launches": len(cc), "reported_bytes": 1024 * sum(cc) / len(cc), "known_bytes": known, "reported_over_known": 1024 * sum(cc) / len(cc) / known}
st = [r for r in csv.DictReader(open(glob.glob("/tmp/pmc_stats/*kernel_stats.csv")[0])) if "go2_step_kernel<3>" in r["Name"]][0]
out["kernel_avg_us_rocprof_stats"] = float(st["AverageNs"]) / 1e3
# unit: KB (x1024).  Corrected = raw / (reported/known) of the calibration probe, per counter; the probe has the step kernel's pattern
# (16-env workgroups, 4 B per env and field, field-major), so no generic x2 is applied.
kf, kw = out["calibration_FETCH_SIZE"]["reported_over_known"], out["calibration_WRITE_SIZE"]["reported_over_known"]
out["hbm_bytes_per_launch_raw"] = (out["FETCH_SIZE"]["mean_kb"] + out["WRITE_SIZE"]["mean_kb"]) * 1024
out["hbm_bytes_per_launch_corrected"] = (out["FETCH_SIZE"]["mean_kb"] / kf + out["WRITE_SIZE"]["mean_kb"] / kw) * 1024
out["algorithmic_bytes_per_launch"] = (2936 if task == "go2_flat" else 4266) * N
out["lib_sha256_16"] = hashlib.sha256(open(R + "/go2_rl_gym_amd/libgo2sim_hip.so", "rb").read()).hexdigest()[:16]
json.dump(out, open(R + "/gpurun_out/pmc/%s_pmc_step_kernel.json" % task, "w"), indent=1)
print(json.dumps(out))
PY
