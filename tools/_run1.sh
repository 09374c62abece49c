set -e
mkdir -p gpurun_out/r3b
export NMGP_ROUND=r03
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3b/pytest.txt 2>&1 || { tail -30 gpurun_out/r3b/pytest.txt; exit 1; }
tail -3 gpurun_out/r3b/pytest.txt
python bench.py --no-cpu-baseline --hmc-samples 0 > gpurun_out/r3b/bench_default.json 2> gpurun_out/r3b/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3b/bench_default.json').read().strip().splitlines()[-1])
print('value', d['value'], 'grad', d['grad']['value'], d['grad']['stage_ms'])
PY
