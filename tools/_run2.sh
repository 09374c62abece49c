set -e
mkdir -p gpurun_out/r3c
export NMGP_ROUND=r03
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3c/pytest.txt 2>&1 || { tail -30 gpurun_out/r3c/pytest.txt; exit 1; }
tail -2 gpurun_out/r3c/pytest.txt
python bench.py > gpurun_out/r3c/bench_default.json 2> gpurun_out/r3c/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3c/bench_default.json').read().strip().splitlines()[-1])
print('value', d['value'], 'grad', d['grad']['value'], d['grad']['stage_ms'])
print({k:v for k,v in d['config'].items() if 'hbm' in k and 'note' not in k}, d['cpu_baseline'])
print(d['hmc'])
PY
for a in "--chains 1 --N 1024" "--chains 1 --N 1024 --grad"; do python bench.py $a --no-cpu-baseline --hmc-samples 0 --grad-steps 0 --steps 50 --warmup 5 > gpurun_out/r3c/cfg2_$(echo $a | tr -d ' -').json; tail -1 gpurun_out/r3c/cfg2_$(echo $a | tr -d ' -').json | cut -c1-200; done
