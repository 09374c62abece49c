"""Phase timing of the critical workgroup of k_panel_step (developer aid).
    NMGP_STEP_STAMPS=/tmp/stamps.txt python tools/step_stamps.py [n]
Factors one n x n matrix (default 6144) and prints, per phase, the mean over the middle steps of every panel (100 MHz wall
clock -> microseconds): 0 start, 1 operands staged, 2 catch-up of block k, 3 solve + store, 4 catch-up of block k+1,
5 X exchanged, 6 D in LDS, 7 diagonal block factored, 8 stored."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nonstationary_multivariate_gaussian_process_amd import _lib  # noqa: E402

path = os.environ.get("NMGP_STEP_STAMPS")
if not path:
    raise SystemExit("set NMGP_STEP_STAMPS=<file>")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
rng = np.random.default_rng(0)
G = rng.standard_normal((n, 64))
A = G @ G.T / 64 + np.eye(n)
ctx = _lib.Context(0)
for _ in range(3):
    ctx.cholesky(A, rng.standard_normal(n), algo=1)
st = np.loadtxt(path)
names = ["staged", "catchup_k", "solve", "catchup_k1", "x_exchange", "D", "potf2", "store"]
mid = [k for k in range(st.shape[0]) if k % 8 not in (0, 7) and st[k, 8] > 0]
d = np.diff(st[mid, :9], axis=1) * 0.01
print("middle steps (%d): " % len(mid) + ", ".join("%s %.2f" % (nm, v) for nm, v in zip(names, d.mean(0))) +
      "  | total %.2f us" % d.sum(1).mean())
first = [k for k in range(st.shape[0]) if k % 8 == 0 and st[k, 8] > 0]
d = np.diff(st[first, :9], axis=1) * 0.01
print("first steps  (%d): " % len(first) + ", ".join("%s %.2f" % (nm, v) for nm, v in zip(names, d.mean(0))) +
      "  | total %.2f us" % d.sum(1).mean())
starts = st[mid, 0]
print("step-to-step start distance (middle steps, us): %.2f" % (np.diff(st[:, 0])[[k for k in mid if k + 1 < st.shape[0]]].mean() * 0.01))
