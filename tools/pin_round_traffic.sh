#!/bin/bash
# Re-measure the WHOLE-EVALUATION traffic figures of profiles/traffic.json on the current build (they are pinned to build.tree_id()):
#     bash tools/pin_round_traffic.sh <tag> <profiles prefix>         (on the GPU box; results travel back through gpurun_out/<tag>/)
# Two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, no trace domains) each of
#   bench.py --chains 128 --grad                       (the value+gradient step: bench.py's grad.roofline.traffic)
#   bench.py --workload subjects --N 1024              (config 4's per-GPU shape, value)
#   bench.py --workload separable                      (config 5: 16 chains of N = 4096, D = 5, value)
# The k_syrk_lower figure of the headline's value step follows csrc/nmgp_chol.hip only and is re-measured by tools/profile_round.sh.
set -e
R=$PWD
O=$R/gpurun_out/$1
P=${2:-r05}
mkdir -p "$O"
export TMPDIR=/tmp
Q="--no-cpu-baseline --hmc-samples 0"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 600 rocprofv3 --pmc $c -d "$O/raw_pmc_$c" -o pmc --output-format csv -- python3 "$R/bench.py" $Q --chains 128 --grad --steps 1 --warmup 1 \
        > "$O/g128_pmc_$c.json" 2> "$O/g128_pmc_$c.err"
    cp "$(find "$O/raw_pmc_$c" -name '*counter_collection.csv' | head -1)" "$O/g128_pmc_$c.csv"
    rm -rf "$O/raw_pmc_$c"
    echo "g128 pmc $c"
    timeout -k 10 300 rocprofv3 --pmc $c -d "$O/raw_pmc8_$c" -o pmc --output-format csv -- python3 "$R/bench.py" $Q --grad-steps 0 --workload subjects --N 1024 --all-subjects 0 --steps 2 --warmup 1 \
        > "$O/s8_pmc_$c.json" 2> "$O/s8_pmc_$c.err"
    cp "$(find "$O/raw_pmc8_$c" -name '*counter_collection.csv' | head -1)" "$O/s8_pmc_$c.csv"
    rm -rf "$O/raw_pmc8_$c"
    echo "s8 pmc $c"
    timeout -k 10 300 rocprofv3 --pmc $c -d "$O/raw_pmcs_$c" -o pmc --output-format csv -- python3 "$R/bench.py" $Q --grad-steps 0 --workload separable --steps 2 --warmup 1 \
        > "$O/sep16_pmc_$c.json" 2> "$O/sep16_pmc_$c.err"
    cp "$(find "$O/raw_pmcs_$c" -name '*counter_collection.csv' | head -1)" "$O/sep16_pmc_$c.csv"
    rm -rf "$O/raw_pmcs_$c"
    echo "sep16 pmc $c"
done
cd "$R"
python3 tools/pmc_summary.py "$O/g128_pmc_FETCH_SIZE.csv" "$O/g128_pmc_WRITE_SIZE.csv" "$O/g128_pmc_traffic.json" "128 chains value+gradient, $1" k_svc_grad_final > /dev/null
python3 tools/pmc_summary.py "$O/s8_pmc_FETCH_SIZE.csv" "$O/s8_pmc_WRITE_SIZE.csv" "$O/s8_pmc_traffic.json" "8 subjects x N=1024, $1" > /dev/null
python3 tools/pmc_summary.py "$O/sep16_pmc_FETCH_SIZE.csv" "$O/sep16_pmc_WRITE_SIZE.csv" "$O/sep16_pmc_traffic.json" "16 separable chains N=4096 D=5 value, $1" k_col_sumsq > /dev/null
mkdir -p profiles
cp "$O/sep16_pmc_traffic.json" "profiles/${P}_sep16_pmc_traffic.json"
cp "$O/g128_pmc_traffic.json" "profiles/${P}_g128_pmc_traffic.json"
cp "$O/s8_pmc_traffic.json" "profiles/${P}_s8_pmc_traffic.json"
python3 tools/pin_traffic.py "profiles/${P}_g128_pmc_traffic.json" 2048 3 128 1 chain
python3 tools/pin_traffic.py "profiles/${P}_s8_pmc_traffic.json" 1024 3 8 0 subjects
python3 tools/pin_traffic.py "profiles/${P}_sep16_pmc_traffic.json" 4096 5 16 0 separable
cp profiles/traffic.json "$O/traffic.json"
python3 bench.py --no-cpu-baseline --hmc-samples 0 --steps 3 --grad-steps 2 > "$O/bench_after_pin.json" 2> "$O/bench_after_pin.err"
python3 - <<PY
import json
r = json.loads(open("$O/bench_after_pin.json").read().strip().splitlines()[-1])
print("value traffic", r["roofline"]["traffic"], "| grad traffic", r["grad"]["roofline"]["traffic"], r["grad"]["roofline"]["traffic_note"][:80])
PY
