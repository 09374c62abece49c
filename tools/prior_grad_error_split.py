"""Where the gradient's distance to the reference comes from at large N (CPU experiment, NumPy/SciPy only):
    python tools/prior_grad_error_split.py [N ...]
The GP-prior part of the gradient is Sigma_prior^-1 (v - mu) with Sigma_prior = RBF(x; alpha = 10, beta = 1) + 1e-6 I (logpos.py:357-368,
condition number ~1e11).  For the same right-hand side this prints, relative to the norm of the solution:
  * the error of a plain double-precision Cholesky solve (what both the reference and this library do, with different kernels),
  * how far apart two such solves are when the matrices differ by <= 1 ulp per entry of exp() (device exp vs torch exp),
  * how far the EXACT solutions of those two matrices are apart (extended-precision iterative refinement on each).
Result: the distance is the two sides' SOLVE errors (each ~3e-6 at N = 2048), not the 1-ulp differences of the
matrices (3e-7); refining the library's solve would cut the distance by ~30 %, the rest is the reference's own kappa * eps error."""
import sys

import numpy as np
from scipy.linalg import cho_factor, cho_solve


def exact(K, v, cf, x0):
    Kl, vl, c = K.astype(np.longdouble), v.astype(np.longdouble), x0.copy()
    for _ in range(3):
        c = c + cho_solve(cf, (vl - Kl @ c.astype(np.longdouble)).astype(np.float64))
    return c


def main():
    rng = np.random.default_rng(1)
    for N in [int(a) for a in sys.argv[1:]] or [256, 1024, 2048]:
        x = np.sort(rng.random(N))
        E = np.exp(-0.5 * ((x[:, None] ** 2 + x[None, :] ** 2) - 2 * np.outer(x, x)))
        v = 3 * (x - 1) ** 3 - 3 + 1e-3 * rng.standard_normal(N)
        K = 100.0 * E + 1e-6 * np.eye(N)
        U = np.triu(rng.integers(-1, 2, size=(N, N)))
        U = U + U.T - np.diag(np.diag(U))
        K2 = 100.0 * E * (1 + U * 1.1e-16) + 1e-6 * np.eye(N)
        cf, cf2 = cho_factor(K, lower=True), cho_factor(K2, lower=True)
        a, a2 = cho_solve(cf, v), cho_solve(cf2, v)
        c, c2 = exact(K, v, cf, a), exact(K2, v, cf2, a2)
        n = np.linalg.norm(c)
        print("N=%5d  plain solve error %.2e | two plain solves (1-ulp different matrices) apart %.2e | their exact solutions apart %.2e | "
              "refined library vs plain reference %.2e" % (N, np.linalg.norm(a - c) / n, np.linalg.norm(a - a2) / n,
                                                          np.linalg.norm(c - c2) / n, np.linalg.norm(c - a2) / n))


if __name__ == "__main__":
    main()
