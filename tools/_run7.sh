set -e
mkdir -p gpurun_out/r3g
python tools/pred_bench.py 2048 201 5 > gpurun_out/r3g/pred_bench.json 2> gpurun_out/r3g/pred_bench.err || { tail -5 gpurun_out/r3g/pred_bench.err; exit 1; }
cat gpurun_out/r3g/pred_bench.json
export NMGP_ROUND=r03
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3g/pytest.txt 2>&1 || { tail -30 gpurun_out/r3g/pytest.txt; exit 1; }
tail -2 gpurun_out/r3g/pytest.txt
