# round-4 side measurements quoted in DESIGN section 5 (one call on the GPU box) -> gpurun_out/r4x/
mkdir -p gpurun_out/r4x
Q="--hmc-samples 0 --no-cpu-baseline"
run() { tag=$1; shift; python bench.py $Q "$@" 2>gpurun_out/r4x/$tag.err | tail -1 > gpurun_out/r4x/$tag.json; python -c "import json;r=json.load(open('gpurun_out/r4x/$tag.json'));print('$tag', round(r['value'],1), round(r['ms_per_step'],3), round(r['roofline']['frac'],3), round(r.get('grad',{}).get('value',0),1))"; }
run cfg2_chains1N1024 --chains 1 --N 1024 --steps 50 --warmup 5 --grad-steps 20
run c1 --chains 1 --steps 30 --warmup 5 --grad-steps 10
run c16 --chains 16 --steps 10 --warmup 2 --grad-steps 4
run c32 --chains 32 --steps 6 --warmup 2 --grad-steps 3
run c64 --chains 64 --steps 4 --warmup 1 --grad-steps 2
run s8 --workload subjects --N 1024 --steps 30 --warmup 5
run s8g --workload subjects --N 1024 --grad --steps 20 --warmup 5
run s8k8 --workload subjects --N 1024 --chains-per-subject 8 --steps 10 --warmup 2
run s8k8g --workload subjects --N 1024 --chains-per-subject 8 --grad --steps 6 --warmup 2
run s16 --workload subjects --N 1024 --subjects-per-gpu 16 --steps 20 --warmup 3
run s64 --workload subjects --N 1024 --subjects-per-gpu 64 --steps 6 --warmup 2
run s64g --workload subjects --N 1024 --subjects-per-gpu 64 --grad --steps 4 --warmup 1
python tools/pred_bench.py 2048 201 5 4096 5 2>/dev/null | tail -1 > gpurun_out/r4x/pred_bench.json; cut -c1-600 gpurun_out/r4x/pred_bench.json
python tools/sep_bench.py 4096 5 5 2>/dev/null | tail -1 > gpurun_out/r4x/sep_N4096_M5.json; cut -c1-400 gpurun_out/r4x/sep_N4096_M5.json
NMGP_STEP_STAMPS=/tmp/stamps.txt python tools/step_stamps.py 6144 > gpurun_out/r4x/step_stamps_n6144.txt 2>&1; tail -4 gpurun_out/r4x/step_stamps_n6144.txt
python tools/map_rate.py 2>/dev/null | tail -2
