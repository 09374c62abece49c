set -e
mkdir -p gpurun_out/r3e
export NMGP_ROUND=r03
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3e/pytest.txt 2>&1 || { tail -30 gpurun_out/r3e/pytest.txt; exit 1; }
tail -2 gpurun_out/r3e/pytest.txt
Q="--no-cpu-baseline --hmc-samples 0 --grad-steps 0 --workload subjects --N 1024"
for a in "" "--grad" "--chains-per-subject 8" "--chains-per-subject 8 --grad" "--chains-per-subject 4" "--chains-per-subject 4 --grad"; do
  python bench.py $Q $a --steps 10 --warmup 2 > gpurun_out/r3e/s8_$(echo $a | tr -d ' -').json
  python - "$a" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r3e/s8_%s.json' % sys.argv[1].replace(' ','').replace('-','')).read().strip().splitlines()[-1])
print(sys.argv[1], '| evals/s %.1f ms/step %.3f frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))
PY
done
