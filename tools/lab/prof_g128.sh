# kernel trace of one 128-chain value+gradient evaluation; usage: bash tools/lab/prof_g128.sh <tag> [env assignments...]
set -e
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 bench.py --chains 128 --grad --steps 2 --warmup 1 --hmc-samples 0 --no-cpu-baseline > $out/bench.json 2> $out/err.log
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 tools/trace_summary.py $f k_svc_grad_final > $out/last_eval.txt
python3 tools/timeline.py $f k_svc_grad_final > $out/timeline.txt 2>/dev/null || true
cat $out/last_eval.txt
