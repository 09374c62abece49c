"""MI355X-native GP log-posterior path (drop-in for the reference's ``Utility`` hot path).

    from nonstationary_multivariate_gaussian_process_amd import Utility        # mirror of the reference package
    from nonstationary_multivariate_gaussian_process_amd import install_utility_alias
    install_utility_alias()          # makes `from Utility import logpos` resolve to this implementation

Importing the package never touches the GPU; the first compute call loads libnmgp_hip.so and raises if the
library or an MI355X is missing (no CPU fallback).
"""
import sys

__version__ = "0.1.0"


def install_utility_alias():
    """Register this package's ``Utility`` mirror under the top-level name the reference's scripts import
    (``sys.path.append(".."); from Utility import logpos``, Nonseparable_model.py:27-36)."""
    from . import Utility
    sys.modules["Utility"] = Utility
    for name in ("settings", "utils", "kernels", "kronecker_operation", "distributions", "logpos", "prediction"):
        sys.modules["Utility." + name] = getattr(Utility, name)
    return Utility
