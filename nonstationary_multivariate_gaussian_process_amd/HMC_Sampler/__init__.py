"""Importable ``HMC_Sampler`` for the reference's model scripts.

The scripts do ``sys.path.append("../..")``, ``import HMC_Sampler`` and call
``HMC_Sampler.HMC_sampler.sampler(sample_size=..., potential_func=logpos.nlogpos_obj_SVC, init_position=..., step_size=1e-4,
num_steps_in_leap=20, x=x, Y=Y, duplicate_samples=True, TensorType=settings.torchType, **hyper_pars).main_hmc_loop()``
(Nonseparable_model.py:24-25,228-231; Separable_model.py:11,209; *_mpiKAISER.py with ``M=`` and ``adaptive_step_size=``).
That package is EXTERNAL to the reference repository (a sibling checkout of the authors, absent from the tree), so the
scripts cannot reach their sampler line without it.  This package provides the same entry point on top of
``drivers.HMCSampler`` -- gradients by ``torch.autograd.grad`` through the fused MI355X objectives.

It is served only when the user has no ``HMC_Sampler`` of their own:

* ``install_utility_alias()`` registers it under the top-level name only if ``import HMC_Sampler`` would otherwise fail;
* reached as a TOP-LEVEL package through ``PYTHONPATH=<repo>/nonstationary_multivariate_gaussian_process_amd`` (scripts untouched), it
  first looks for another ``HMC_Sampler`` further down ``sys.path`` and, if there is one, hands over to it.
"""
import sys as _sys


def _users_own(name, here):
    """Spec of a package `name` found on sys.path OUTSIDE this directory's parent, or None."""
    import importlib.machinery
    import os
    mine = os.path.dirname(here)
    for p in _sys.path:
        ap = os.path.abspath(p or ".")
        if ap == mine:
            continue
        spec = importlib.machinery.PathFinder.find_spec(name, [ap])
        if spec is not None and spec.origin and os.path.dirname(os.path.abspath(spec.origin)) != here:
            return spec
    return None


if __name__ == "HMC_Sampler":
    import importlib.util as _ilu
    import os as _os
    _here = _os.path.dirname(_os.path.abspath(__file__))
    _spec = _users_own("HMC_Sampler", _here)
    if _spec is not None:
        # the user's own sampler wins: load it in our place (the import system returns sys.modules["HMC_Sampler"])
        _mod = _ilu.module_from_spec(_spec)
        _sys.modules["HMC_Sampler"] = _mod
        _spec.loader.exec_module(_mod)
    else:
        _root = _os.path.dirname(_os.path.dirname(_here))
        if _root not in _sys.path:
            _sys.path.append(_root)
        import nonstationary_multivariate_gaussian_process_amd.HMC_Sampler as _canon
        from nonstationary_multivariate_gaussian_process_amd.HMC_Sampler import HMC_sampler as _hs
        _sys.modules["HMC_Sampler"] = _canon
        _sys.modules["HMC_Sampler.HMC_sampler"] = _hs
else:
    from . import HMC_sampler  # noqa: F401
