set -e
mkdir -p gpurun_out/r3x
python bench.py --chains 1 --N 1024 --steps 50 --warmup 5 --hmc-samples 0 --grad-steps 0 2>/dev/null | tail -1 > gpurun_out/r3x/cfg2.json
python bench.py --chains 1 --N 1024 --grad --steps 30 --warmup 5 --hmc-samples 0 2>/dev/null | tail -1 > gpurun_out/r3x/cfg2g.json
python tools/pred_bench.py 2048 201 5 4096 5 2>/dev/null | tail -1 > gpurun_out/r3x/pred.json
python -c "
import json
for f in ('cfg2','cfg2g'):
    d=json.load(open('gpurun_out/r3x/%s.json'%f)); print(f, d['value'], d['ms_per_step'])
print(open('gpurun_out/r3x/pred.json').read()[-700:])"
