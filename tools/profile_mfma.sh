#!/bin/bash
# MFMA utilisation counters of the 128-chain value step and value+gradient step, one call on the GPU box:
#     bash tools/profile_mfma.sh <tag>       -> gpurun_out/<tag>/{v128,g128}_pmc_mfma.json (+ the raw counter CSVs)
# rocprofv3 --pmc only (no trace domains), python3 directly after `--`, TMPDIR on /tmp.
set -e
R=$PWD
O=$R/gpurun_out/$1
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
Q="--no-cpu-baseline --hmc-samples 0 --grad-steps 0"
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$O/raw_v" -o pmc --output-format csv -- \
    python3 "$R/bench.py" $Q --steps 2 --warmup 1 > "$O/v128_pmc_mfma_bench.json" 2> "$O/v128_pmc_mfma.err"
cp "$(find "$O/raw_v" -name '*counter_collection.csv' | head -1)" "$O/v128_pmc_MFMA_BUSY.csv"
rm -rf "$O/raw_v"
echo "pmc value"
timeout -k 10 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$O/raw_g" -o pmc --output-format csv -- \
    python3 "$R/bench.py" $Q --grad --steps 1 --warmup 1 > "$O/g128_pmc_mfma_bench.json" 2> "$O/g128_pmc_mfma.err"
cp "$(find "$O/raw_g" -name '*counter_collection.csv' | head -1)" "$O/g128_pmc_MFMA_BUSY.csv"
rm -rf "$O/raw_g"
echo "pmc grad"
cd "$R"
python3 tools/pmc_mfma.py "$O/v128_pmc_MFMA_BUSY.csv" "$O/v128_pmc_mfma.json" k_svc_finalize | head -12
python3 tools/pmc_mfma.py "$O/g128_pmc_MFMA_BUSY.csv" "$O/g128_pmc_mfma.json" k_svc_grad_final | head -12
