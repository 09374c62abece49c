# samples/s of the bench's `hmc` object: identity vs diagonal vs dense mass matrix (momenta and energies on the device)
mkdir -p gpurun_out/r4e
for m in identity diag dense identity dense; do
  python bench.py --steps 1 --warmup 1 --grad-steps 0 --no-cpu-baseline --hmc-samples 3 --hmc-mass $m > gpurun_out/r4e/hmc_$m.json 2> gpurun_out/r4e/hmc_$m.err
  python -c "import json;r=json.load(open('gpurun_out/r4e/hmc_$m.json'))['hmc'];print('$m', round(r['samples_per_s'],3), round(r['grad_evals_per_s'],1), r['accept_rate_mean'], r['median_abs_energy_error'])"
done
