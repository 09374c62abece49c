"""Drop-in mirror of the reference's ``Utility`` package for the log-posterior path.

Same module names, function names, positional order, keyword names and defaults as
``/root/reference/Utility/{settings,kernels,kronecker_operation,distributions,utils,logpos,prediction}.py``;
the arithmetic runs on the MI355X through libnmgp_hip.so (C ABI in include/nmgp.h).  Either call
``nonstationary_multivariate_gaussian_process_amd.install_utility_alias()`` before the scripts' imports or put this
package's parent directory first on ``sys.path``: the reference's model scripts' ``from Utility import logpos`` then
resolves here.  Everything else the scripts import from ``Utility`` (plotting, data splits,
sampling-based prediction, ...) is passed through to the user's own reference checkout, see ``_overlay.py``.
"""
import sys as _sys

if __name__ == "Utility":
    # Reached as a TOP-LEVEL package (PYTHONPATH=<repo>/nonstationary_multivariate_gaussian_process_amd, scripts untouched):
    # hand over to the canonical package so that there is one copy of the mirror (and of the ctypes binding) per process.
    # The import system returns whatever sys.modules["Utility"] holds once this file has run.
    import os as _os
    _root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
    if _root not in _sys.path:
        _sys.path.append(_root)
    import nonstationary_multivariate_gaussian_process_amd as _pkg
    _pkg.install_utility_alias()
else:
    from . import settings  # noqa: F401
    from . import utils  # noqa: F401
    from . import kernels  # noqa: F401
    from . import kronecker_operation  # noqa: F401
    from . import distributions  # noqa: F401
    from . import logpos  # noqa: F401
    from . import prediction  # noqa: F401
    from . import _overlay

    def __getattr__(name):
        return _overlay.package_getattr(__name__, name)
