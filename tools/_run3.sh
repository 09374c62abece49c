set -e
mkdir -p gpurun_out/r3d
python tools/make_map_point.py 1000 gpurun_out/map_N2048_M3_seed2222.npz
timeout -k 10 600 python -m pytest tests/test_drivers.py -x -q -m gpu > gpurun_out/r3d/pytest_drivers.txt 2>&1 || { tail -30 gpurun_out/r3d/pytest_drivers.txt; exit 1; }
tail -2 gpurun_out/r3d/pytest_drivers.txt
mkdir -p tests/golden && cp gpurun_out/map_N2048_M3_seed2222.npz tests/golden/
python bench.py --no-cpu-baseline --steps 3 --warmup 1 --grad-steps 1 --hmc-samples 5 > gpurun_out/r3d/bench_hmc.json 2> gpurun_out/r3d/bench_hmc.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3d/bench_hmc.json').read().strip().splitlines()[-1])
print(json.dumps(d['hmc'], indent=1))
PY
