"""The drop-in recipe of INTEGRATION.md section 1 against the script it cites: the import block of the reference's
``Nonseparable_Model/Nonseparable_model.py`` (lines 22-37) must run unchanged with the mirror installed, the hot-path
names must be the MI355X ones and everything else must come from the user's own reference checkout.

Build-container only: skipped where /root/reference is absent (the GPU box).  The reference's lines are READ from the
checkout at test time and executed in a child process whose working directory is the script's folder (so that its
``sys.path.append("..")`` means what it means for the script); nothing of the reference lives in this repository.
CPU only: importing never touches the GPU."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SCRIPT = os.path.join(REF, "Nonseparable_Model", "Nonseparable_model.py")

pytestmark = pytest.mark.skipif(not os.path.isfile(SCRIPT), reason="reference checkout not present")

CHILD = textwrap.dedent('''
    import os, sys, types
    sys.dont_write_bytecode = True
    # stand-in for what the image lacks: seaborn.  (The external HMC package of line 25 -- absent from the reference tree -- is
    # served by the mirror's own HMC_Sampler package: no stub.)
    sys.modules["seaborn"] = types.ModuleType("seaborn")
    MODE = sys.argv[1]
    if MODE == "alias":
        sys.path.insert(0, {root!r})
        import nonstationary_multivariate_gaussian_process_amd as nmgp_amd
        nmgp_amd.install_utility_alias()
    elif MODE == "alias_explicit":
        sys.path.insert(0, {root!r})
        import nonstationary_multivariate_gaussian_process_amd as nmgp_amd
        nmgp_amd.install_utility_alias(reference_utility_dir={refutil!r})
    else:                       # PYTHONPATH variant: the mirror's parent first on the path, script untouched
        assert any(p.endswith("nonstationary_multivariate_gaussian_process_amd") for p in sys.path), sys.path
    lines = open({script!r}).read().split("\\n")
    block = "\\n".join(lines[21:37])          # lines 22-37: "# import private library" ... "from Utility import empirical_estimation"
    assert "from Utility import logpos" in block and "empirical_estimation" in block
    exec(compile(block, {script!r}, "exec"))
    mirror = os.path.join({root!r}, "nonstationary_multivariate_gaussian_process_amd", "Utility")
    # line 25 `import HMC_Sampler` got the mirror's sampler package, with the entry point the script calls at lines 228-231
    import inspect as _insp
    assert os.path.dirname(os.path.abspath(_insp.getsourcefile(HMC_Sampler.HMC_sampler.sampler))) == os.path.join(
        {root!r}, "nonstationary_multivariate_gaussian_process_amd", "HMC_Sampler"), HMC_Sampler
    _sig = _insp.signature(HMC_Sampler.HMC_sampler.sampler.__init__).parameters
    for _k in ("sample_size", "potential_func", "init_position", "step_size", "num_steps_in_leap", "duplicate_samples", "TensorType",
               "M", "adaptive_step_size"):
        assert _k in _sig, _k
    assert callable(HMC_Sampler.HMC_sampler.sampler.main_hmc_loop)
    def where(obj):
        import inspect
        return os.path.dirname(os.path.abspath(inspect.getsourcefile(obj)))
    # hot path: the mirror's
    assert where(logpos.nlogpos_obj_SVC) == mirror, where(logpos.nlogpos_obj_SVC)
    assert where(logpos.nlogpos_obj) == mirror and where(logpos.nlogpos_obj_S) == mirror
    assert where(prediction.pointwise_predmap_inhomogeneous) == mirror
    assert where(utils.uLvecs2Lvecs) == mirror
    import Utility
    assert where(Utility.kernels.Nonstationary_RBF_cov) == mirror
    assert where(Utility.distributions.multivariate_normal_logpdf0) == mirror
    # everything else: the user's reference checkout
    for obj in (utils.data_split, utils.MSE, utils.RMSE, utils.LPD, prediction.vec2pars, prediction.vec2list,
                prediction.pointwise_predsample_inhomogeneous,
                prediction.test_predmap_inhomogeneous_sampling, logpos.nlogpos_obj_hadamard_SVC,
                visualization.Plot_posterior, posterior_analysis.cov2cor, model_validation.get_AIC,
                preprocess_realdata.orig2adj, empirical_estimation.local_estimation):
        assert where(obj) == {refutil!r}, (obj, where(obj))
    # a passed-through module is a member of the package `Utility`: its own `from . import utils` binds to the mirror
    assert empirical_estimation.utils is utils and posterior_analysis.utils is utils
    assert settings.jitter == 1e-6 and settings.precision == 1e-6
    # host-side helpers work without a GPU, e.g. the data split the script performs at line 88
    import numpy as np, torch
    x = torch.linspace(0, 1, 20).double(); Y = torch.randn(20, 2).double()
    assert utils.uLvec2Lvec(torch.zeros(6).double(), 3).tolist() == [1, 0, 1, 0, 0, 1]
    try:
        Utility.no_such_module
    except AttributeError as e:
        assert "outside the mirrored" in str(e)
    else:
        raise AssertionError("missing submodule must raise AttributeError")
    print("OVERLAY-OK", MODE)
''')


@pytest.mark.parametrize("mode", ["alias", "alias_explicit", "pythonpath"])
def test_reference_script_import_block_runs_on_the_overlay(mode, tmp_path):
    code = CHILD.format(root=ROOT, script=SCRIPT, refutil=os.path.join(REF, "Utility"))
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", MPLBACKEND="Agg")
    env.pop("NMGP_REFERENCE_UTILITY", None)
    if mode == "pythonpath":
        env["PYTHONPATH"] = os.path.join(ROOT, "nonstationary_multivariate_gaussian_process_amd") + os.pathsep + \
            env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-c", code, mode], cwd=os.path.dirname(SCRIPT), env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and "OVERLAY-OK" in r.stdout, r.stdout + r.stderr


def test_mirror_alone_names_what_is_missing():
    """Without any reference checkout the mirror still imports, and a name outside the path fails loudly."""
    code = textwrap.dedent('''
        import sys
        sys.path.insert(0, %r)
        import nonstationary_multivariate_gaussian_process_amd as nmgp_amd
        U = nmgp_amd.install_utility_alias()
        from Utility import logpos, utils
        try:
            utils.data_split
        except AttributeError as e:
            assert "no reference checkout" in str(e), e
        else:
            raise AssertionError
        try:
            from Utility import visualization
        except ImportError as e:
            pass
        else:
            raise AssertionError
        print("ALONE-OK")
    ''' % ROOT)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env.pop("NMGP_REFERENCE_UTILITY", None)
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, "-c", code], cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ALONE-OK" in r.stdout, r.stdout + r.stderr


def test_utility_imports_of_every_reference_script_resolve_on_the_overlay():
    """All 17 scripts of the reference that import from ``Utility`` (the three model families, their MPI / simulation / real-data
    variants, the post-processing and simulation tools): their ``from Utility import ...`` lines -- read from the checkout at
    test time -- run on the overlay from the script's own directory, and every imported module is either a mirror module or
    the user's own file."""
    import glob
    scripts = sorted(p for p in glob.glob(os.path.join(REF, "*", "*.py"))
                     if os.path.dirname(p) != os.path.join(REF, "Utility") and "from Utility import" in open(p).read())
    assert len(scripts) >= 15
    code = textwrap.dedent('''
        import os, sys, types, re
        sys.dont_write_bytecode = True
        for name in ("seaborn", "mpi4py", "pyGPs", "statsmodels", "gpytorch"):
            sys.modules.setdefault(name, types.ModuleType(name))
        sys.path.append("..")              # what every one of these scripts does before its imports
        sys.path.insert(0, %r)
        import nonstationary_multivariate_gaussian_process_amd as nmgp_amd
        nmgp_amd.install_utility_alias()
        mirror = os.path.join(%r, "nonstationary_multivariate_gaussian_process_amd", "Utility")
        refutil = %r
        seen = set()
        for script in sys.argv[1:]:
            os.chdir(os.path.dirname(script))
            lines = [ln.strip() for ln in open(script).read().split("\\n") if re.match(r"\\s*from Utility import ", ln)]
            assert lines, script
            ns = {}
            for ln in lines:
                exec(ln, ns)
            for k, v in ns.items():
                if isinstance(v, types.ModuleType) and k != "__builtins__":
                    d = os.path.dirname(os.path.abspath(v.__file__))
                    assert d in (mirror, refutil), (script, k, d)
                    seen.add((k, d == mirror))
        mirrored = {k for k, m in seen if m}
        assert {"logpos", "kernels", "kronecker_operation", "distributions", "utils", "settings", "prediction"} <= mirrored, mirrored
        assert {"posterior_analysis", "visualization", "empirical_estimation"} <= {k for k, m in seen if not m}
        print("IMPORTS-OK", len(sys.argv) - 1)
    ''') % (ROOT, ROOT, os.path.join(REF, "Utility"))
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", MPLBACKEND="Agg")
    env.pop("NMGP_REFERENCE_UTILITY", None)
    r = subprocess.run([sys.executable, "-c", code] + scripts, cwd=os.path.join(REF, "Nonseparable_Model"), env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "IMPORTS-OK" in r.stdout, r.stdout + r.stderr


def test_the_users_own_hmc_sampler_package_wins(tmp_path):
    """The mirror's HMC_Sampler is a stand-in for a package the user normally HAS (the authors' sibling checkout, line 24 of the
    script appends its directory to sys.path): when one is importable, install_utility_alias() leaves it in charge, and reached
    through PYTHONPATH the mirror's package hands over to it."""
    own = tmp_path / "Hamiltonian_Monte_Carlo" / "HMC_Sampler"
    own.mkdir(parents=True)
    (own / "__init__.py").write_text("MINE = True\nfrom . import HMC_sampler\n")
    (own / "HMC_sampler.py").write_text("class sampler:\n    pass\n")
    code = textwrap.dedent('''
        import sys
        sys.path.append(%r)                  # what line 24 of the script does
        MODE = sys.argv[1]
        if MODE == "alias":
            sys.path.insert(0, %r)
            import nonstationary_multivariate_gaussian_process_amd as nmgp_amd
            nmgp_amd.install_utility_alias()
        import HMC_Sampler
        assert getattr(HMC_Sampler, "MINE", False) is True, HMC_Sampler
        assert HMC_Sampler.HMC_sampler.sampler.__module__ == "HMC_Sampler.HMC_sampler"
        print("OWN-OK", MODE)
    ''') % (str(tmp_path / "Hamiltonian_Monte_Carlo"), ROOT)
    for mode in ("alias", "pythonpath"):
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
        env.pop("PYTHONPATH", None)
        if mode == "pythonpath":
            env["PYTHONPATH"] = os.path.join(ROOT, "nonstationary_multivariate_gaussian_process_amd")
        r = subprocess.run([sys.executable, "-c", code, mode], cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "OWN-OK" in r.stdout, r.stdout + r.stderr
