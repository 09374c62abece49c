mkdir -p gpurun_out/r3f
Q="--no-cpu-baseline --hmc-samples 0 --grad-steps 0"
for base in 512 256 128 0; do
  if [ $base = 0 ]; then unset NMGP_CHOL_FUSED_BASE; else export NMGP_CHOL_FUSED_BASE=$base; fi
  for a in "--chains 1" "--chains 1 --grad" "--chains 4" "--chains 8" "--chains 12" "--workload subjects --N 1024" "--workload subjects --N 1024 --grad" "--workload subjects --N 1024 --subjects-per-gpu 16"; do
    python bench.py $Q $a --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base=$base', '$a', '| %.1f evals/s %.3f ms' % (d['value'], d['ms_per_step']))"
  done
  python tools/sep_bench.py 4096 5 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base=$base sep N=4096 D=5: value %.3f ms  value+grad %.3f ms' % (d['value']['ms'], d['value_grad']['ms']))"
done
