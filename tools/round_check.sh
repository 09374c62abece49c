#!/bin/bash
# One call on the GPU box: the GPU test suite, the driver's bench line and the side lines of the round.
#     bash tools/round_check.sh <tag>      -> gpurun_out/<tag>/
set -e
R=${1:-r05}
O=gpurun_out/$R
mkdir -p $O
export NMGP_ROUND=${2:-r05}
timeout -k 10 1000 python -m pytest ${PYTEST_ARGS:-tests} -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
cp gpurun_out/parity_$NMGP_ROUND.json $O/parity_$NMGP_ROUND.json 2>/dev/null || true
python -c "import __graft_entry__ as g; g.smoke()"
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -1 $O/bench_default.json | cut -c1-700
timeout -k 10 600 python bench.py --hmc-mass identity,prior --no-cpu-baseline --steps 4 --grad-steps 2 > $O/bench_hmc_both.json 2> $O/bench_hmc_both.err
timeout -k 10 600 python bench.py --workload separable > $O/bench_sep16.json 2> $O/bench_sep16.err
tail -1 $O/bench_sep16.json | cut -c1-500
timeout -k 10 600 python bench.py --workload separable --chains 32 --no-cpu-baseline > $O/bench_sep32.json 2> $O/bench_sep32.err
timeout -k 10 600 python bench.py --workload subjects --N 1024 > $O/bench_s8.json 2> $O/bench_s8.err
tail -1 $O/bench_s8.json | cut -c1-500
timeout -k 10 600 python bench.py --workload subjects --N 1024 --subjects-total 64 --no-cpu-baseline > $O/bench_s64_strong.json 2> $O/bench_s64_strong.err
echo done
