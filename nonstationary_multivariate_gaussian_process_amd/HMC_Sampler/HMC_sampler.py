"""``HMC_Sampler.HMC_sampler.sampler``: the constructor spelling of the reference's sampler call
(Nonseparable_model.py:228-231) on top of :class:`..drivers.HMCSampler`.

    hmc = sampler(sample_size=N_hmc, potential_func=logpos.nlogpos_obj_SVC, init_position=estPars, step_size=1e-4,
                  num_steps_in_leap=20, x=x, Y=Y, duplicate_samples=True, TensorType=settings.torchType, **hyper_pars)
    sample, _ = hmc.main_hmc_loop()            # sample: [sample_size, P] ndarray

Keywords the reference's callers pass: ``sample_size, potential_func, init_position, step_size, num_steps_in_leap,
adaptive_step_size, M, duplicate_samples, TensorType`` -- everything else (``x``, ``Y``, the hyper-parameters) is forwarded
to ``potential_func(position, **kwargs)``.  ``init_position`` may be an ndarray or a tensor.

Opt-in, for the UNCHANGED script: with ``NMGP_HMC_RECIPE=1`` in the environment a sampler whose potential is the mirror's
``logpos.nlogpos_obj_SVC`` or ``logpos.nlogpos_obj`` (and no mass matrix of the caller's) does not run the call as written -- one chain,
identity mass, step 1e-4: at N = 2048 that chain moves each parameter by 2e-3 in 1000 iterations -- but the recipe under which the
chains converge (``drivers.sample_nonseparable`` / ``sample_separable``: mode from ``init_position``, prior-factor metric, warm-up,
step search, ``NMGP_HMC_CHAINS`` (default 4) chains in lock-step on the GPU).  ``main_hmc_loop()`` still returns
``(samples [sample_size, P], info)``: the samples are chain 0's, ``info["all_chains"]`` holds ``[sample_size, chains, P]``.
"""
import os

import numpy as np

from ..drivers import HMCSampler


class sampler(HMCSampler):      # noqa: N801 -- the reference's spelling
    """See the module docstring; ``main_hmc_loop()`` returns ``(samples [sample_size, P], info)``."""

    def __init__(self, sample_size, potential_func, init_position, step_size=1e-4, num_steps_in_leap=20,
                 adaptive_step_size=False, M=None, duplicate_samples=True, TensorType=None, seed=None, **kwargs):
        if hasattr(init_position, "detach"):
            init_position = init_position.detach().cpu().numpy()
        if M is not None and hasattr(M, "detach"):
            M = M.detach().cpu().numpy()
        extra = {} if TensorType is None else {"TensorType": TensorType}
        super().__init__(sample_size=sample_size, potential_func=potential_func, init_position=init_position,
                         step_size=step_size, num_steps_in_leap=num_steps_in_leap, adaptive_step_size=adaptive_step_size,
                         M=M, duplicate_samples=duplicate_samples, seed=seed, **extra, **kwargs)
        self._recipe = None
        if os.environ.get("NMGP_HMC_RECIPE", "0") not in ("", "0") and M is None and "x" in kwargs and "Y" in kwargs:
            # the mirror's own objectives, under whichever name the script imported them (`Utility.logpos` through the alias, or the
            # package's module): same source file as this package's Utility/logpos.py
            import inspect
            from ..Utility import logpos
            try:
                same_file = os.path.samefile(inspect.getsourcefile(potential_func), inspect.getsourcefile(logpos))
            except (TypeError, OSError):
                same_file = False
            name = getattr(potential_func, "__name__", "")
            if same_file and name == "nlogpos_obj_SVC":
                self._recipe = "svc"
            elif same_file and name == "nlogpos_obj":
                self._recipe = "sep"
        self._seed = seed

    def main_hmc_loop(self):
        if self._recipe is None:
            return super().main_hmc_loop()
        from .. import drivers
        kw = dict(self.kwargs)
        x, Y = kw.pop("x"), kw.pop("Y")
        kw.pop("TensorType", None)
        for k in ("verbose", "Prior"):
            kw.pop(k, None)
        x = x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)
        Y = Y.detach().cpu().numpy() if hasattr(Y, "detach") else np.asarray(Y)
        hyper = {k: float(v) for k, v in kw.items()}
        chains = max(1, int(os.environ.get("NMGP_HMC_CHAINS", "4")))
        fn = drivers.sample_nonseparable if self._recipe == "svc" else drivers.sample_separable
        S, info = fn(x.reshape(-1), Y, hyper, self.q0, chains=chains, iters=self.sample_size, num_steps_in_leap=self.L,
                     seed=1 if self._seed is None else int(self._seed))
        out = {"accept_rate": float(np.mean(info["accept_rate"])), "step_size": info["step_size"], "energy_error": info["energy_error"][:, 0],
               "iterations": self.sample_size, "all_chains": S, "recipe": {k: v for k, v in info.items() if k not in ("energy_error",)}}
        return S[:, 0, :].copy(), out
