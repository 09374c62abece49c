# latency-regime A/B of NMGP_TRTRI: one chain, 4/16 chains, 8 / 64 subjects x N=1024, separable N=4096 D=5 (value+gradient)
mkdir -p gpurun_out/r4d
for rep in 1 2; do
for v in 0 1; do
  export NMGP_TRTRI=$v
  for args in "--chains 1" "--chains 4" "--chains 16" "--chains 32" "--workload subjects --N 1024" "--workload subjects --N 1024 --subjects-per-gpu 64" "--workload subjects --N 1024 --chains-per-subject 8"; do
    python bench.py $args --grad --steps 6 --warmup 2 --hmc-samples 0 --no-cpu-baseline --grad-steps 0 > gpurun_out/r4d/o.json 2>gpurun_out/r4d/err.log
    python -c "import json;r=json.load(open('gpurun_out/r4d/o.json'));print('TRTRI=$v [$args]', round(r['value'],1), round(r['ms_per_step'],3))"
  done
  python tools/sep_bench.py 2>/dev/null | tail -1
done
done
