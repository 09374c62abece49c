set -e
mkdir -p gpurun_out/r3f
export NMGP_ROUND=r03
NMGP_STEP_STAMPS=/tmp/stamps.txt python tools/step_stamps.py 6144 > gpurun_out/r3f/step_stamps_n6144.txt 2>&1
NMGP_STEP_STAMPS=/tmp/stamps2.txt python tools/step_stamps.py 3072 > gpurun_out/r3f/step_stamps_n3072.txt 2>&1
cat gpurun_out/r3f/step_stamps_n6144.txt
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3f/pytest.txt 2>&1 || { tail -30 gpurun_out/r3f/pytest.txt; exit 1; }
tail -2 gpurun_out/r3f/pytest.txt
python bench.py > gpurun_out/r3f/bench_default.json 2> gpurun_out/r3f/bench_default.err
tail -1 gpurun_out/r3f/bench_default.json | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()"
