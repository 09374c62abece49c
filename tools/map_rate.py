"""Adam iterations per second of BatchedMAP (config 4's per-GPU shape by default: 8 subjects x N = 1024, D = 3), device-resident
update against the host-side one.    python tools/map_rate.py [subjects] [N] [iterations]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nonstationary_multivariate_gaussian_process_amd import sim  # noqa: E402
from nonstationary_multivariate_gaussian_process_amd.drivers import BatchedMAP  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
its = int(sys.argv[3]) if len(sys.argv) > 3 else 40
subs = [sim.simulate_nonseparable(N, 3, seed=s) for s in range(B)]
xs, Ys = np.stack([d["x"] for d in subs]), np.stack([d["Y"] for d in subs])
p0 = np.stack([sim.perturb(d["pars_true"], 0.05, 0.3) for d in subs])
for dev in (True, False):
    bm = BatchedMAP(xs, Ys, sim.HYPER_SVC_MPISIM, p0, lr=1e-2, device_resident=dev)
    bm.run(3)
    t0 = time.perf_counter()
    bm.run(its)
    dt = time.perf_counter() - t0
    print("device_resident=%s: %.1f iterations/s (%.3f ms per iteration of %d subjects, N=%d)" % (dev, its / dt, 1e3 * dt / its, B, N))
