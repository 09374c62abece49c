"""The convergence diagnostics tools/hmc_1000.py reports for BASELINE config 3 (CPU): the multi-chain effective sample size must NOT be
a sum of per-chain figures -- chains that sit at different places have a small ESS however smooth each of them looks."""
import os
import sys

import numpy as np

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_multichain_ess_and_split_rhat_on_known_processes():
    import hmc_1000 as H
    rng = np.random.default_rng(0)
    S, C, K = 600, 8, 4
    iid = rng.standard_normal((S, C, K))
    ess = H.multichain_ess(iid)
    assert np.all(ess > 0.7 * S * C) and np.all(ess < 1.3 * S * C)
    assert np.all(np.abs(H.split_rhat(iid) - 1.0) < 0.01)
    # AR(1) with rho = 0.9: ESS = n (1 - rho) / (1 + rho)
    ar = np.zeros((S, C, K))
    e = rng.standard_normal((S, C, K)) * np.sqrt(1 - 0.81)
    for t in range(1, S):
        ar[t] = 0.9 * ar[t - 1] + e[t]
    ess = H.multichain_ess(ar)
    want = S * C * 0.1 / 1.9
    assert np.all(ess > 0.5 * want) and np.all(ess < 2.0 * want)
    # chains that have not mixed: every chain is white noise around its OWN level -- the per-chain sum (round 4's figure) says
    # "thousands", the multi-chain estimator says "a handful", split-R-hat is far from 1
    stuck = iid + 2.0 * np.arange(C)[None, :, None]
    assert np.all(H.multichain_ess(stuck) < 3 * C) and np.all(H.autocorr_ess(stuck) > 0.5 * S * C)
    assert np.all(H.split_rhat(stuck) > 2.0)
    # a slow common drift in every chain is seen by the split halves
    drift = iid + np.linspace(0, 3, S)[:, None, None]
    assert np.all(H.split_rhat(drift) > 1.2)
