#!/bin/bash
# The 128-chain sampler under the profiler: what the prior-factor metric's kernels cost next to the evaluation.
#     bash tools/profile_hmc.sh <tag>      -> gpurun_out/<tag>/hmc128_kernel_stats.csv, hmc128_metric_share.txt
set -e
R=$PWD
O=$R/gpurun_out/$1
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$O/raw_hmc" -o hmc128 --output-format csv -- python3 "$R/bench.py" --no-cpu-baseline \
    --steps 2 --warmup 1 --grad-steps 1 --hmc-samples 2 > "$O/hmc128.json" 2> "$O/hmc128.err"
cp "$(find "$O/raw_hmc" -name '*kernel_stats.csv' | head -1)" "$O/hmc128_kernel_stats.csv"
rm -rf "$O/raw_hmc"
cd "$R"
python3 - "$O/hmc128_kernel_stats.csv" > "$O/hmc128_metric_share.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel statistics of `bench.py --steps 2 --warmup 1 --grad-steps 1 --hmc-samples 2` (128 chains, N = 2048, D = 3; rocprofv3 --kernel-trace --stats)")
print("%-60s %8s %12s %8s" % ("kernel", "calls", "total ms", "share"))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("k_prior_trmm", "k_lowrank", "k_metric_kinetic", "k_hmc_", "k_syrk_lower", "k_svc_adjoint", "k_svc_cov", "k_panel_step")):
        print("%-60s %8s %12.3f %7.3f%%" % (n.split("(")[0][-60:], r["Calls"], float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
met = sum(float(r["TotalDurationNs"]) for r in rows if any(k in r["Name"] for k in ("k_prior_trmm", "k_lowrank", "k_metric_kinetic")))
print("prior-factor metric kernels together: %.3f ms = %.3f %% of all kernel time of the run" % (met / 1e6, 100 * met / tot))
PY
cat "$O/hmc128_metric_share.txt"
