"""CPU-only checks of the drop-in boundary: the shared object builds/loads and exports every symbol that
include/nmgp.h declares; the Python mirror exposes the reference's names and signatures."""
import inspect
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "nmgp.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nmgp_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    from nonstationary_multivariate_gaussian_process_amd import build as b, _lib
    lib_path = b.build(verbose=False)
    assert os.path.exists(lib_path)
    names = declared_symbols()
    assert len(names) >= 30
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib_path]).decode()
    exported = set(re.findall(r" T (nmgp_[a-z0-9_]+)", out))
    missing = [n for n in names if n not in exported]
    assert not missing, "declared in nmgp.h but not exported: %s" % missing
    # the ctypes table binds exactly the declared surface
    assert sorted(_lib.SIGNATURES) == names
    lib = _lib.load(require_gpu=False)
    assert lib.nmgp_version() == 100


def test_loaded_library_is_the_build_of_this_tree(tmp_path, monkeypatch):
    """Build provenance: nmgp_build_id() == SHA-256 of the sources + headers + flags beside the shared object; a change of
    any source without a rebuild makes _lib.load() refuse the library instead of running stale kernels."""
    from nonstationary_multivariate_gaussian_process_amd import build as b, _lib
    b.build(verbose=False)
    tid = b.tree_id()
    assert len(tid) == 64 and _lib.build_id() == tid
    # the id covers every source, both headers and the flags
    for extra_flag in (["-O2"],):
        monkeypatch.setattr(b, "CODEGEN_FLAGS", b.CODEGEN_FLAGS + extra_flag)
        assert b.tree_id() != tid
        monkeypatch.undo()
    # a stale library is refused: pretend the tree moved on
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(b, "tree_id", lambda: "0" * 64)
    with pytest.raises(_lib.NmgpError, match="stale"):
        _lib.load(require_gpu=False)
    monkeypatch.undo()
    assert _lib.load(require_gpu=False).nmgp_build_id().decode() == tid


def test_no_gpu_means_loud_failure():
    from nonstationary_multivariate_gaussian_process_amd import _lib
    lib = _lib.load(require_gpu=False)
    if lib.nmgp_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.NmgpError):
        _lib.Context(0)
    import torch
    from nonstationary_multivariate_gaussian_process_amd.Utility import logpos
    Y = torch.zeros(4, 2, dtype=torch.float64)
    x = torch.linspace(0, 1, 4, dtype=torch.float64)
    with pytest.raises(_lib.NmgpError):
        logpos.nlogpos_obj_SVC(torch.zeros(4 * 4 + 1, dtype=torch.float64), Y, x)


REFERENCE_SIGNATURES = {
    # module: {function: parameter list as in the reference}
    "kernels": {
        "pairwise_distances": ["x", "y"],
        "RBF_cov": ["X1", "X2", "alpha", "beta"],
        "Nonstationary_RBF_cov": ["X1", "sigma1", "ell1", "X2", "sigma2", "ell2"],
    },
    "kronecker_operation": {
        "kronecker_product": ["t1", "t2"],
        "kronecker_product_diag": ["d1", "d2"],
        "kron_inv": ["sigma2", "B", "K"],
        "kron_logdet": ["sigma2", "B", "K"],
        "kron_mv": ["B", "K", "y"],
    },
    "distributions": {
        "multivariate_normal_logpdf": ["y", "mu", "logdetSigma", "invSigma"],
        "multivariate_normal_logpdf0": ["y", "mu", "B", "K", "sigma2"],
        "multivariate_normal_logpdf1": ["y", "mu", "B", "K", "sigma2"],
        "multivariate_normal_logpdf2": ["y", "mu", "B", "K", "sigma2"],
        "inverse_gamma_logpdf": ["x", "alpha", "beta"],
        "inverse_gamma_logpdf_u": ["x", "alpha", "beta"],
        "gamma_logpdf": ["x", "alpha", "beta"],
    },
    "utils": {
        "uLvec2Lvec": ["uL_vec", "M"], "Lvec2uLvec": ["L_vec", "M"], "uLvecs2Lvecs": ["uL_vecs", "N", "M"],
        "Lvecs2uLvecs": ["L_vecs", "N", "M"], "vec2lowtriangle": ["x", "N"], "lowtriangle2vec": ["L", "N"],
    },
    "logpos": {
        "vec2pars": ["pars", "N", "M"], "vec2pars_SVC": ["pars", "N", "M"], "vec2pars_S": ["pars", "M"],
        "generate_K_index_SVC": ["L_f_list"],
        "nlogpos_obj_SVC": ["pars", "Y", "x", "mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_L", "alpha_L",
                            "beta_L", "a", "b", "verbose", "Prior"],
        "logpos_SVC": ["tilde_l", "uL_vecs", "tilde_sigma2_err", "Y", "x", "mu_tilde_l", "alpha_tilde_l",
                       "beta_tilde_l", "mu_L", "alpha_L", "beta_L", "a", "b", "verbose", "Prior"],
        "nlogpos_obj": ["pars", "Y", "x", "mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_tilde_sigma",
                        "alpha_tilde_sigma", "beta_tilde_sigma", "a", "b", "c", "verbose", "Prior"],
        "logpos": ["tilde_l", "tilde_sigma", "uL_vec", "tilde_sigma2_err", "Y", "x", "mu_tilde_l", "alpha_tilde_l",
                   "beta_tilde_l", "mu_tilde_sigma", "alpha_tilde_sigma", "beta_tilde_sigma", "a", "b", "c", "verbose",
                   "Prior"],
        "nlogpos_obj_S": ["pars", "Y", "x", "mu_tilde_l", "sigma_tilde_l", "a", "b", "c", "verbose", "Prior"],
        "logpos_S": ["tilde_l", "tilde_sigma", "uL_vec", "tilde_sigma2_err", "Y", "x", "mu_tilde_l", "sigma_tilde_l",
                     "a", "b", "c", "verbose", "Prior"],
    },
}

REFERENCE_DEFAULTS = {
    ("logpos", "nlogpos_obj_SVC"): dict(mu_tilde_l=0., alpha_tilde_l=5., beta_tilde_l=1., mu_L=0., alpha_L=5.,
                                        beta_L=1., a=1, b=1, verbose=False, Prior=True),
    ("logpos", "nlogpos_obj"): dict(mu_tilde_l=0., alpha_tilde_l=1., beta_tilde_l=1., mu_tilde_sigma=0.,
                                    alpha_tilde_sigma=1., beta_tilde_sigma=1., a=1, b=1, c=10, verbose=False,
                                    Prior=True),
    ("logpos", "nlogpos_obj_S"): dict(a=1, b=1, c=10, verbose=False, Prior=True),
    ("kernels", "RBF_cov"): dict(X2=None, alpha=1., beta=1.),
}


def test_python_mirror_has_reference_signatures():
    from nonstationary_multivariate_gaussian_process_amd import Utility
    for mod, fns in REFERENCE_SIGNATURES.items():
        m = getattr(Utility, mod)
        for fn, params in fns.items():
            sig = inspect.signature(getattr(m, fn))
            got = [p for p in sig.parameters if p not in ("args", "kwargs")]
            assert got == params, (mod, fn, got, params)
    for (mod, fn), defaults in REFERENCE_DEFAULTS.items():
        sig = inspect.signature(getattr(getattr(Utility, mod), fn))
        for k, v in defaults.items():
            assert sig.parameters[k].default == v, (mod, fn, k)
    assert Utility.settings.jitter == 1e-6 and Utility.settings.precision == 1e-6


def test_utility_alias_and_host_helpers():
    import numpy as np
    import torch
    import nonstationary_multivariate_gaussian_process_amd as pkg
    U = pkg.install_utility_alias()
    from Utility import logpos, utils          # noqa: the reference scripts' import line
    assert logpos is U.logpos
    assert torch.equal(utils.uLvec2Lvec(torch.zeros(6, dtype=torch.float64), 3),
                       torch.tensor([1., 0, 1, 0, 0, 1], dtype=torch.float64))
    assert torch.equal(utils.vec2lowtriangle(torch.arange(1., 7., dtype=torch.float64), 3),
                       torch.tensor([[1., 0, 0], [2, 3, 0], [4, 5, 6]], dtype=torch.float64))
    v = torch.arange(12, dtype=torch.float64) * 0.1 - 0.5
    assert torch.allclose(utils.Lvecs2uLvecs(utils.uLvecs2Lvecs(v, 2, 3), 2, 3), v)
    assert np.allclose(utils.uLvecs2Lvecs(v.numpy(), 2, 3), utils.uLvecs2Lvecs(v, 2, 3).numpy())
    p = torch.arange(20, dtype=torch.float64)
    tl, Lv, s = logpos.vec2pars_SVC(p, 2 + 0, 2)     # N=2 -> wrong length on purpose is still sliced like the reference
    assert tl.tolist() == [0., 1.] and float(s) == 19.
    tl, ts, uL, s = logpos.vec2pars_S(torch.arange(6, dtype=torch.float64), 2)
    assert float(tl) == 0 and float(ts) == 1 and uL.tolist() == [2., 3., 4.] and float(s) == 5
