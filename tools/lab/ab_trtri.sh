set -e
mkdir -p gpurun_out/r4b
python -m pytest tests/test_gpu_variants.py -q -m gpu -x -k "blocked_triangular" > gpurun_out/r4b/trtri_test.log 2>&1 || { tail -40 gpurun_out/r4b/trtri_test.log; exit 1; }
tail -3 gpurun_out/r4b/trtri_test.log
for v in 0 1 0 1; do
  NMGP_TRTRI=$v python bench.py --chains 128 --grad --steps 4 --warmup 1 --hmc-samples 0 --no-cpu-baseline > gpurun_out/r4b/g128_trtri$v.json 2>gpurun_out/r4b/err.log
  python -c "import json;r=json.load(open('gpurun_out/r4b/g128_trtri$v.json'));print('TRTRI=$v', r['value'], r['config']['stage_ms'])"
done
