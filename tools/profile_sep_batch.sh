#!/bin/bash
# The batched separable evaluation (BASELINE config 5: N = 4096, D = 5) under the profiler, in one call on the GPU box:
#     bash tools/profile_sep_batch.sh <tag> [chains, default 16]     -> gpurun_out/<tag>/sep<chains>_*
# rocprofv3 --kernel-trace --stats of `bench.py --workload separable` (value) and `... --grad` (value+gradient): kernel statistics
# and the per-kernel summary of the last evaluation (end marker: the last kernel an evaluation launches).
set -e
R=$PWD
O=$R/gpurun_out/$1
B=${2:-16}
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
for mode in v g; do
    Q="--workload separable --chains $B --no-cpu-baseline --grad-steps 0 --steps 3 --warmup 1"
    marker=k_col_sumsq
    if [ $mode = g ]; then Q="$Q --grad"; marker=k_sep_grad_sum_b; fi
    t=sep${B}$mode
    timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$O/raw_$t" -o "$t" --output-format csv -- python3 "$R/bench.py" $Q \
        > "$O/$t.json" 2> "$O/$t.err"
    cp "$(find "$O/raw_$t" -name '*kernel_stats.csv' | head -1)" "$O/${t}_kernel_stats.csv"
    TR="$(find "$O/raw_$t" -name '*kernel_trace.csv' | head -1)"
    python3 "$R/tools/trace_summary.py" "$TR" $marker > "$O/${t}_last_eval.txt"
    rm -rf "$O/raw_$t"
    echo "traced $t"
done
cd "$R"
