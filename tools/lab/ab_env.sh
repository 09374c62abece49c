# A/B of environment variants on the 128-chain value+gradient step: bash tools/lab/ab_env.sh <tag> "<env A>" "<env B>" ... (each twice, interleaved)
tag=$1; shift
mkdir -p gpurun_out/$tag
for rep in 1 2; do
  i=0
  for envs in "$@"; do
    i=$((i+1))
    ( for kv in $envs; do export "$kv"; done; python bench.py --chains 128 --grad --steps 4 --warmup 1 --hmc-samples 0 --no-cpu-baseline > gpurun_out/$tag/v${i}_$rep.json 2>gpurun_out/$tag/err.log )
    python -c "import json;r=json.load(open('gpurun_out/$tag/v${i}_$rep.json'));print('[$envs]', round(r['value'],2), {k:round(v,2) for k,v in r['config']['stage_ms'].items()})"
  done
done
