"""MI355X-native GP log-posterior path (drop-in for the reference's ``Utility`` hot path).

    from nonstationary_multivariate_gaussian_process_amd import Utility        # mirror of the reference package
    from nonstationary_multivariate_gaussian_process_amd import install_utility_alias
    install_utility_alias()          # makes `from Utility import logpos` resolve to this implementation

Importing the package never touches the GPU; the first compute call loads libnmgp_hip.so and raises if the
library or an MI355X is missing (no CPU fallback).
"""
import sys

__version__ = "0.1.0"


def install_utility_alias(reference_utility_dir=None):
    """Register this package's ``Utility`` mirror under the top-level name the reference's scripts import
    (``sys.path.append(".."); from Utility import logpos``, Nonseparable_model.py:27-36).

    The mirror OVERLAYS the user's own reference checkout instead of replacing it: the hot-path modules and functions
    (logpos objectives, kernels, kronecker_operation, distributions, the tril helpers of utils, deterministic prediction)
    are the MI355X ones; every other submodule or name the scripts use (``visualization``, ``posterior_analysis``,
    ``model_validation``, ``preprocess_realdata``, ``empirical_estimation``, ``utils.data_split/MSE/RMSE/LPD``,
    ``prediction.vec2pars/vec2list/*_sampling``) is resolved from the reference's ``Utility`` directory --
    ``reference_utility_dir`` if given, else ``$NMGP_REFERENCE_UTILITY``, else the first ``Utility`` directory found on
    ``sys.path`` when such a name is first needed (the scripts append ".." before importing).  Nothing of the reference
    is copied or shipped."""
    from . import Utility
    from .Utility import _overlay
    sys.modules["Utility"] = Utility
    for name in ("settings", "utils", "kernels", "kronecker_operation", "distributions", "logpos", "prediction"):
        sys.modules["Utility." + name] = getattr(Utility, name)
    if reference_utility_dir is not None:
        _overlay.attach(reference_utility_dir)
    install_hmc_sampler()
    return Utility


def install_hmc_sampler(force=False):
    """Make ``import HMC_Sampler`` work for the scripts' sampler line (Nonseparable_model.py:24-25,228-231).  The reference's
    sampler package is external to its repository; ours (``HMC_Sampler/HMC_sampler.py`` -> ``drivers.HMCSampler``) is registered
    under the top-level name ONLY when the user has no ``HMC_Sampler`` of their own on ``sys.path`` (or with ``force``).
    Returns the module that ``import HMC_Sampler`` will yield, or None when the user's own package is left in charge."""
    import importlib.util
    if not force:
        if "HMC_Sampler" in sys.modules:
            return sys.modules["HMC_Sampler"]
        try:
            if importlib.util.find_spec("HMC_Sampler") is not None:
                return None
        except (ImportError, ValueError):
            pass
    from . import HMC_Sampler as pkg
    sys.modules["HMC_Sampler"] = pkg
    sys.modules["HMC_Sampler.HMC_sampler"] = pkg.HMC_sampler
    return pkg
