#!/bin/bash
# Every profile artefact of a round in ONE call on the GPU box, from the repo root:
#     bash tools/profile_round.sh <tag> [profiles prefix, default r02]   -> gpurun_out/<tag>/...   (copy what is to be judged into profiles/)
# kernel-trace + stats runs of the workloads DESIGN section 4b quotes, the default bench line outside the profiler, and the
# two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, no trace domains) that roofline.traffic is derived from.
# rocprofv3 is given python3 itself after `--` (nothing that re-execs), TMPDIR on /tmp as the pool's guide prescribes.
set -e
R=$PWD
O=$R/gpurun_out/$1
mkdir -p "$O"
cd /tmp
export TMPDIR=/tmp
Q="--no-cpu-baseline --grad-steps 0 --hmc-samples 0"

trace() {      # trace <tag> <marker> <program args...>
    local tag=$1 marker=$2
    shift 2
    timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$O/raw_$tag" -o "$tag" --output-format csv -- python3 "$@" \
        > "$O/$tag.json" 2> "$O/$tag.err"
    cp "$(find "$O/raw_$tag" -name '*kernel_stats.csv' | head -1)" "$O/${tag}_kernel_stats.csv"
    if [ "$marker" != "-" ]; then
        python3 "$R/tools/trace_summary.py" "$(find "$O/raw_$tag" -name '*kernel_trace.csv' | head -1)" "$marker" > "$O/${tag}_last_eval.txt"
        python3 "$R/tools/timeline.py" "$O/raw_$tag" "$marker" > "$O/${tag}_timeline.txt"
        if [ "$tag" = "batched128" ]; then
            python3 "$R/tools/syrk_classes.py" "$(find "$O/raw_$tag" -name '*kernel_trace.csv' | head -1)" 6144 2048 128 0 > "$O/${tag}_syrk_classes.txt"
        fi
    fi
    rm -rf "$O/raw_$tag"
    echo "traced $tag"
}

trace c1v k_svc_finalize "$R/bench.py" $Q --chains 1 --steps 10 --warmup 3
trace c1g k_svc_finalize "$R/bench.py" $Q --chains 1 --grad --steps 10 --warmup 3
trace s8v k_svc_finalize "$R/bench.py" $Q --workload subjects --N 1024 --steps 10 --warmup 3
trace s8g k_svc_finalize "$R/bench.py" $Q --workload subjects --N 1024 --grad --steps 10 --warmup 3
trace s64v k_svc_finalize "$R/bench.py" $Q --workload subjects --N 1024 --subjects-per-gpu 64 --steps 5 --warmup 2
trace s64g k_svc_finalize "$R/bench.py" $Q --workload subjects --N 1024 --subjects-per-gpu 64 --grad --steps 3 --warmup 1
trace sep_chol - "$R/tools/sep_bench.py" 4096 5 5
NMGP_SEP=eig trace sep_eig - "$R/tools/sep_bench.py" 4096 5 3
trace batched128 k_svc_finalize "$R/bench.py" $Q --steps 3 --warmup 1

for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 600 rocprofv3 --pmc $c -d "$O/raw_pmc_$c" -o pmc --output-format csv -- python3 "$R/bench.py" $Q --steps 2 --warmup 1 \
        > "$O/pmc_$c.json" 2> "$O/pmc_$c.err"
    cp "$(find "$O/raw_pmc_$c" -name '*counter_collection.csv' | head -1)" "$O/batched128_pmc_$c.csv"
    rm -rf "$O/raw_pmc_$c"
    echo "pmc $c"
done
cd "$R"
python3 tools/pmc_summary.py "$O/batched128_pmc_FETCH_SIZE.csv" "$O/batched128_pmc_WRITE_SIZE.csv" "$O/batched128_pmc_traffic.json" "128 chains, $1"
python3 tools/pmc_classes.py "$O/batched128_pmc_FETCH_SIZE.csv" "$O/batched128_pmc_WRITE_SIZE.csv" "$O/batched128_pmc_syrk_classes.json"
# the default line is measured with the fresh traffic pin in place (bench.py reports roofline.traffic only for the kernel source
# the PMC passes ran on); the pin travels back through gpurun_out/
mkdir -p profiles && cp "$O/batched128_pmc_traffic.json" "profiles/${2:-r02}_batched128_pmc_traffic.json"
python3 tools/pin_traffic.py "profiles/${2:-r02}_batched128_pmc_traffic.json"
# config 4's per-GPU shape (8 subjects x N = 1024, value): the traffic of `bench.py --workload subjects`
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c -d "$O/raw_pmc8_$c" -o pmc --output-format csv -- python3 "$R/bench.py" $Q --workload subjects --N 1024 --steps 2 --warmup 1 \
        > "$O/s8_pmc_$c.json" 2> "$O/s8_pmc_$c.err"
    cp "$(find "$O/raw_pmc8_$c" -name '*counter_collection.csv' | head -1)" "$O/s8_pmc_$c.csv"
    rm -rf "$O/raw_pmc8_$c"
done
cd "$R"
python3 tools/pmc_summary.py "$O/s8_pmc_FETCH_SIZE.csv" "$O/s8_pmc_WRITE_SIZE.csv" "$O/s8_pmc_traffic.json" "8 subjects x N=1024, $1"
cp "$O/s8_pmc_traffic.json" "profiles/${2:-r02}_s8_pmc_traffic.json"
python3 tools/pin_traffic.py "profiles/${2:-r02}_s8_pmc_traffic.json" 1024 3 8 0 subjects
cp profiles/traffic.json "$O/traffic.json"
python3 bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"
tail -1 "$O/bench_default.json" | cut -c1-400
