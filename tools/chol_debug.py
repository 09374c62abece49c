"""Debug helper: custom blocked Cholesky against numpy on a small SPD matrix, error per 16-column block."""
import sys
import numpy as np
from nonstationary_multivariate_gaussian_process_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(0)
B = rng.standard_normal((n, n))
A = B @ B.T + n * np.eye(n)
ctx = _lib.Context(0)
L = ctx.cholesky(A)
R = np.linalg.cholesky(A)
E = np.abs(L - R)
print("max err", E.max(), "nan", np.isnan(L).sum())
nb = 16
for bj in range(min(n // nb, 8)):
    print("block col", bj, ["%.1e" % np.nanmax(E[bi * nb:(bi + 1) * nb, bj * nb:(bj + 1) * nb]) for bi in range(bj, min(n // nb, bj + 10))])
