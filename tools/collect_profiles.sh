#!/bin/bash
# Copy what tools/profile_round.sh left under gpurun_out/<tag>/ into profiles/ under the round's prefix:
#     bash tools/collect_profiles.sh <tag> [prefix, default r02]
set -e
O=gpurun_out/$1
P=${2:-r02}
for t in c1v c1g s8v s8g s64v s64g; do
    cp $O/${t}_kernel_stats.csv profiles/${P}_${t}_kernel_stats.csv
    cp $O/${t}_last_eval.txt profiles/${P}_${t}_last_eval.txt
    cp $O/${t}_timeline.txt profiles/${P}_${t}_timeline.txt
    tail -1 $O/$t.json > profiles/${P}_${t}_bench_under_rocprof.json
done
for f in chol eig; do
    cp $O/sep_${f}_kernel_stats.csv profiles/${P}_sep_N4096_M5_${f}_kernel_stats.csv
    tail -1 $O/sep_$f.json > profiles/${P}_sep_N4096_M5_$f.json
done
cp $O/batched128_kernel_stats.csv profiles/${P}_batched128_kernel_stats.csv
tail -1 $O/batched128.json > profiles/${P}_batched128_bench_under_rocprof.json
cp $O/batched128_last_eval.txt profiles/${P}_batched128_last_eval.txt
for c in FETCH_SIZE WRITE_SIZE; do cp $O/batched128_pmc_$c.csv profiles/${P}_batched128_pmc_$c.csv; done
cp $O/batched128_pmc_traffic.json profiles/${P}_batched128_pmc_traffic.json
cp $O/batched128_pmc_syrk_classes.json profiles/${P}_batched128_pmc_syrk_classes.json
cp $O/traffic.json profiles/traffic.json
tail -1 $O/bench_default.json > profiles/${P}_default_bench.json
[ -f gpurun_out/parity_${P}.json ] && cp gpurun_out/parity_${P}.json profiles/parity_${P}.json
cp $O/batched128_timeline.txt profiles/${P}_batched128_timeline.txt 2>/dev/null || true
cp $O/batched128_syrk_classes.txt profiles/${P}_batched128_syrk_classes.txt 2>/dev/null || true
for t in v128 g128; do
    [ -f $O/${t}_pmc_mfma.json ] && cp $O/${t}_pmc_mfma.json profiles/${P}_${t}_pmc_mfma.json && cp $O/${t}_pmc_MFMA_BUSY.csv profiles/${P}_${t}_pmc_MFMA_BUSY.csv
done
# the value+gradient step (tools/profile_grad.sh)
if [ -f $O/g128_kernel_stats.csv ]; then
    for f in kernel_stats.csv last_eval.txt syrk_classes.txt timeline.txt pmc_FETCH_SIZE.csv pmc_WRITE_SIZE.csv pmc_traffic.json pmc_syrk_classes.json; do
        cp $O/g128_$f profiles/${P}_g128_$f
    done
    tail -1 $O/g128.json > profiles/${P}_g128_bench_under_rocprof.json
fi
# config 4's per-GPU shape under the counters, prediction at the reference's grid
for f in s8_pmc_FETCH_SIZE.csv s8_pmc_WRITE_SIZE.csv s8_pmc_traffic.json pred_kernel_stats.csv; do
    [ -f $O/$f ] && cp $O/$f profiles/${P}_$f
done
[ -f $O/pred_bench.json ] && tail -1 $O/pred_bench.json > profiles/${P}_pred_bench.json
echo collected
