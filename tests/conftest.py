import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def golden(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: d[k] for k in d.files}


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


SVC_KEYS = ["mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_L", "alpha_L", "beta_L", "a", "b"]
SEP_KEYS = ["mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_tilde_sigma", "alpha_tilde_sigma", "beta_tilde_sigma",
            "a", "b", "c"]
STA_KEYS = ["mu_tilde_l", "sigma_tilde_l", "a", "b", "c"]


def hyper_dict(vec, keys):
    return {k: float(v) for k, v in zip(keys, vec)}


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1e-300, np.maximum(np.abs(a), np.abs(b)))))


def vec_relerr(a, b):
    """||a-b|| / ||b|| -- the gradient parity measure."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def prior_component_err_on_the_logdet_scale(a, b, N):
    """RELAXED measure (the name says so in every assertion and in the parity table's column): the verbose GP-prior COMPONENTS are
    checked against max(|component|, N/2 |log 1e-6|), not against |component| alone; the log posterior -- their sum with the
    likelihood -- is always held to the plain 1e-6 relative bar next to it.
    Error measure for the verbose GP-prior components (lp_l, lp_uL) at sizes where they are a small difference of
    large terms: each is -N/2 log 2pi - 1/2 log det - 1/2 q with 1/2 |log det| ~ N/2 |log jitter| (most eigenvalues of
    RBF + 1e-6 I sit at the jitter floor), so the error is taken relative to max(|component|, that scale).  The log
    posterior itself (their sum with the likelihood) is always checked at the plain 1e-6 relative north-star bar."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = np.maximum(np.abs(b), 0.5 * N * abs(np.log(1e-6)))
    return float(np.max(np.abs(a - b) / scale))


# ---- achieved-parity table ------------------------------------------------------------------------------
# GPU parity tests record the error they achieved against every golden vector / oracle value; at the end of a session
# that recorded anything the table is written to gpurun_out/parity_<round>.json (copied to profiles/ and committed), so a
# drift towards the tolerance is visible long before it fails a test.
_PARITY = []


def record_parity(case, **errs):
    """errs: name -> achieved error, or name -> (achieved, tolerance)."""
    row = {"case": case}
    for k, v in errs.items():
        if isinstance(v, tuple):
            row[k] = float(v[0])
            row[k + "_tol"] = float(v[1])
        else:
            row[k] = float(v)
    _PARITY.append(row)


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY:
        return
    import json
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "parity_%s.json" % os.environ.get("NMGP_ROUND", "r05"))
    worst = {}
    for row in _PARITY:
        for k, v in row.items():
            if k != "case" and not k.endswith("_tol") and (k + "_tol") in row and row[k + "_tol"] > 0:
                frac = v / row[k + "_tol"]
                if frac > worst.get(k, (0.0, None))[0]:
                    worst[k] = (frac, row["case"])
    with open(path, "w") as f:
        json.dump({"exitstatus": int(exitstatus), "rows": _PARITY,
                   "worst_fraction_of_tolerance": {k: {"frac": v[0], "case": v[1]} for k, v in worst.items()}}, f, indent=1)
