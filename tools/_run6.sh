set -e
bash tools/profile_round.sh r3 r03 > gpurun_out/r3_profile_round.log 2>&1 || { tail -20 gpurun_out/r3_profile_round.log; exit 1; }
tail -3 gpurun_out/r3_profile_round.log
bash tools/profile_grad.sh r3 128 r03 > gpurun_out/r3_profile_grad.log 2>&1 || { tail -20 gpurun_out/r3_profile_grad.log; exit 1; }
tail -2 gpurun_out/r3_profile_grad.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3/raw_pred -o pred --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/pred_bench.py 2048 201 5 > $GRAFT_REPO_ROOT/gpurun_out/r3/pred_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3/pred_bench.err
cd $GRAFT_REPO_ROOT
cp "$(find gpurun_out/r3/raw_pred -name '*kernel_stats.csv' | head -1)" gpurun_out/r3/pred_kernel_stats.csv
rm -rf gpurun_out/r3/raw_pred
head -12 gpurun_out/r3/pred_kernel_stats.csv
