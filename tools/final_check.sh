set -e
R=${1:-r05}
mkdir -p gpurun_out/${R}_final
export NMGP_ROUND=$R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/${R}_final/pytest.txt 2>&1 || { tail -30 gpurun_out/${R}_final/pytest.txt; exit 1; }
tail -2 gpurun_out/${R}_final/pytest.txt
cp gpurun_out/parity_$R.json gpurun_out/${R}_final/parity_$R.json
python bench.py > gpurun_out/${R}_final/bench_default.json 2> gpurun_out/${R}_final/bench_default.err
tail -1 gpurun_out/${R}_final/bench_default.json | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()"
