#!/bin/bash
# A/B of the block-assembly kernel variants of nmgp_sep_batch_eval (NMGP_SEP_BLOCKS=1|2), alternating inside one call:
#     bash tools/ab_sep_blocks.sh <tag>
set -e
O=gpurun_out/$1
mkdir -p $O
NMGP_SEP_BLOCKS=${2:-2} NMGP_ROUND=variants timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "separable or sep_" > $O/tests_v2.log 2>&1 || { tail -20 $O/tests_v2.log; exit 1; }
tail -1 $O/tests_v2.log
for v in ${3:-1} ${2:-2} ${3:-1} ${2:-2}; do
    NMGP_SEP_BLOCKS=$v timeout -k 10 200 python tools/sep_batch_bench.py 4096 5 16 2>/dev/null | grep "^{" > $O/ab_$v.jsonl
    python - "$O/ab_$v.jsonl" $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("variant", sys.argv[2], "value %.2f ms (cov %.3f)  value+grad %.2f ms (cov %.3f)" % (
    d["value"]["ms_per_batch"], d["value"]["stage_ms_per_batch"]["cov"], d["value_grad"]["ms_per_batch"], d["value_grad"]["stage_ms_per_batch"]["cov"]))
PY
done
