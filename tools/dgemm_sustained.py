"""Sustained rocBLAS DGEMM rate (the practical FP64-MFMA ceiling under DVFS): short bursts vs ~1 s of back-to-back GEMMs."""
import sys
from nonstationary_multivariate_gaussian_process_amd import _lib
ctx = _lib.Context(0)
for n, reps in [(4096, 5), (4096, 50), (4096, 500), (6144, 200)]:
    print(n, reps, ctx.measure_dgemm_tflops(n, reps), flush=True)
