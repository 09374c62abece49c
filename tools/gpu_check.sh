set -e
mkdir -p gpurun_out/r3i
export NMGP_ROUND=r03
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3i/pytest.txt 2>&1 || { tail -30 gpurun_out/r3i/pytest.txt; exit 1; }
tail -2 gpurun_out/r3i/pytest.txt
python bench.py > gpurun_out/r3i/bench_default.json 2> gpurun_out/r3i/bench_default.err
tail -1 gpurun_out/r3i/bench_default.json | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()"
