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
echo collected
