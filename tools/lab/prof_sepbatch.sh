set -e
out=$GRAFT_REPO_ROOT/gpurun_out/prof_sepb
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 $GRAFT_REPO_ROOT/tools/sep_batch_bench.py 4096 5 16 > $out/bench.json 2> $out/err.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
head -30 $f | cut -c1-150
