"""Overlay of the mirror on the user's own checkout of the reference ``Utility`` package.

The mirror re-implements only the log-posterior path.  The reference's model scripts also import
modules that are out of scope here (``visualization``, ``posterior_analysis``, ``model_validation``,
``preprocess_realdata``, ``empirical_estimation``; ``Nonseparable_model.py:28-36``) and call helpers of partly mirrored
modules (``utils.data_split/MSE/RMSE/LPD``, ``prediction.vec2pars/vec2list/*_sampling``).  Those are NOT rebuilt, copied or
shipped: they are resolved at run time from the user's reference checkout --

* a submodule the mirror does not have is imported from the reference's ``Utility`` directory (appended to the mirror
  package's ``__path__``), and, because it runs as a member of the package ``Utility``, its own ``from . import kernels``
  etc. bind to the MI355X mirror;
* a name missing from a mirrored module is looked up in the reference's module of the same name, loaded privately as
  ``Utility._ref_<module>``.

The checkout is found lazily: an explicit directory (``attach``), ``$NMGP_REFERENCE_UTILITY``, or the first ``Utility``
directory on ``sys.path`` that is not this package (the scripts do ``sys.path.append("..")`` before importing).
"""
import importlib
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = __name__.rsplit(".", 1)[0]          # "<...>.Utility"
_state = {"dir": None, "explicit": False}


def _is_reference_dir(d):
    return (os.path.isdir(d) and os.path.abspath(d) != _HERE and os.path.isfile(os.path.join(d, "logpos.py")))


def attach(directory):
    """Use `directory` (the reference checkout's ``Utility`` folder) for everything the mirror does not provide."""
    d = os.path.abspath(directory)
    if not _is_reference_dir(d):
        raise FileNotFoundError("%s is not a checkout of the reference's Utility package (no logpos.py)" % directory)
    _state["dir"] = d
    _state["explicit"] = True
    _extend_path(d)
    return d


def reference_dir():
    """The reference ``Utility`` directory in use, or None when none can be found."""
    if _state["dir"]:
        return _state["dir"]
    cands = []
    env = os.environ.get("NMGP_REFERENCE_UTILITY")
    if env:
        cands.append(env)
    for p in sys.path:
        cands.append(os.path.join(p or ".", "Utility"))
    for c in cands:
        c = os.path.abspath(c)
        if _is_reference_dir(c):
            _state["dir"] = c
            _extend_path(c)
            return c
    return None


def _mirror_packages():
    pk = sys.modules.get(_PKG)
    out = [pk] if pk is not None else []
    alias = sys.modules.get("Utility")
    if alias is not None and alias is not pk:
        out.append(alias)
    return out


def _extend_path(d):
    for pk in _mirror_packages():
        if d not in pk.__path__:
            pk.__path__.append(d)


def package_getattr(pkg_name, name):
    """Module-level ``__getattr__`` of the mirror package: ``from Utility import visualization``."""
    if name.startswith("__"):
        raise AttributeError(name)
    d = reference_dir()
    if d is None or not os.path.isfile(os.path.join(d, name + ".py")):
        raise AttributeError(
            "module 'Utility' (MI355X mirror) has no submodule %r; it is outside the mirrored log-posterior path and no "
            "reference checkout providing it was found (put the reference's parent directory on sys.path, set "
            "NMGP_REFERENCE_UTILITY, or call install_utility_alias(reference_utility_dir=...))" % name)
    return importlib.import_module(pkg_name + "." + name)


def module_getattr(mod_name, name):
    """Module-level ``__getattr__`` of a partly mirrored module: ``utils.data_split``, ``prediction.vec2pars``."""
    if name.startswith("__"):
        raise AttributeError(name)
    pkg_name, short = mod_name.rsplit(".", 1)
    ref = _load_reference_module(pkg_name, short)
    if ref is None or not hasattr(ref, name):
        raise AttributeError(
            "module 'Utility.%s' (MI355X mirror) has no attribute %r; it is outside the mirrored log-posterior path%s"
            % (short, name, "" if ref is not None else " and no reference checkout was found to take it from"))
    return getattr(ref, name)


def _load_reference_module(pkg_name, short):
    private = "%s._ref_%s" % (pkg_name, short)
    if private in sys.modules:
        return sys.modules[private]
    d = reference_dir()
    if d is None:
        return None
    path = os.path.join(d, short + ".py")
    if not os.path.isfile(path):
        return None
    spec = importlib.util.spec_from_file_location(private, path)      # __package__ = pkg_name: `from . import x` -> mirror
    mod = importlib.util.module_from_spec(spec)
    sys.modules[private] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        sys.modules.pop(private, None)
        raise
    return mod
