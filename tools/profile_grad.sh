#!/bin/bash
# The 128-chain VALUE+GRADIENT step under the profiler (what every MAP / HMC iteration pays), in one call on the GPU box:
#     bash tools/profile_grad.sh <tag> [chains, default 128]        -> gpurun_out/<tag>/g<chains>_*
# rocprofv3 --kernel-trace --stats of `bench.py --chains B --grad`, the per-kernel summary and launch timeline of its last
# evaluation, the per-K-class table of the update kernel, and the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, no
# trace domains) from which the gradient entry of profiles/traffic.json is made.
set -e
R=$PWD
O=$R/gpurun_out/$1
B=${2:-128}
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
Q="--no-cpu-baseline --hmc-samples 0 --chains $B --grad"
t=g$B
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d "$O/raw_$t" -o "$t" --output-format csv -- python3 "$R/bench.py" $Q --steps 2 --warmup 1 \
    > "$O/$t.json" 2> "$O/$t.err"
cp "$(find "$O/raw_$t" -name '*kernel_stats.csv' | head -1)" "$O/${t}_kernel_stats.csv"
TR="$(find "$O/raw_$t" -name '*kernel_trace.csv' | head -1)"
python3 "$R/tools/trace_summary.py" "$TR" k_svc_grad_final > "$O/${t}_last_eval.txt"
python3 "$R/tools/syrk_classes.py" "$TR" 6144 2048 "$B" 1 > "$O/${t}_syrk_classes.txt"
python3 "$R/tools/timeline.py" "$O/raw_$t" k_svc_grad_final > "$O/${t}_timeline.txt" || true
rm -rf "$O/raw_$t"
echo "traced $t"
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 600 rocprofv3 --pmc $c -d "$O/raw_pmc_$c" -o pmc --output-format csv -- python3 "$R/bench.py" $Q --steps 1 --warmup 1 \
        > "$O/${t}_pmc_$c.json" 2> "$O/${t}_pmc_$c.err"
    cp "$(find "$O/raw_pmc_$c" -name '*counter_collection.csv' | head -1)" "$O/${t}_pmc_$c.csv"
    rm -rf "$O/raw_pmc_$c"
    echo "pmc $c"
done
cd "$R"
python3 tools/pmc_summary.py "$O/${t}_pmc_FETCH_SIZE.csv" "$O/${t}_pmc_WRITE_SIZE.csv" "$O/${t}_pmc_traffic.json" "$B chains value+gradient, $1" k_svc_grad_final
python3 tools/pmc_classes.py "$O/${t}_pmc_FETCH_SIZE.csv" "$O/${t}_pmc_WRITE_SIZE.csv" "$O/${t}_pmc_syrk_classes.json" 6144 2048 "$B" grad
# pin the measured traffic of the value+gradient step (bench.py's grad.roofline.traffic) to the kernel source it was measured on
mkdir -p profiles && cp "$O/${t}_pmc_traffic.json" "profiles/${3:-r04}_${t}_pmc_traffic.json"
python3 tools/pin_traffic.py "profiles/${3:-r04}_${t}_pmc_traffic.json" 2048 3 "$B" 1 chain && cp profiles/traffic.json "$O/traffic.json"
tail -1 "$O/$t.json" | cut -c1-600
