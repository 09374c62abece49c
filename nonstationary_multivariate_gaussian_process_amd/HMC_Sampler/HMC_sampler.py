"""``HMC_Sampler.HMC_sampler.sampler``: the constructor spelling of the reference's sampler call
(Nonseparable_model.py:228-231) on top of :class:`..drivers.HMCSampler`.

    hmc = sampler(sample_size=N_hmc, potential_func=logpos.nlogpos_obj_SVC, init_position=estPars, step_size=1e-4,
                  num_steps_in_leap=20, x=x, Y=Y, duplicate_samples=True, TensorType=settings.torchType, **hyper_pars)
    sample, _ = hmc.main_hmc_loop()            # sample: [sample_size, P] ndarray

Keywords the reference's callers pass: ``sample_size, potential_func, init_position, step_size, num_steps_in_leap,
adaptive_step_size, M, duplicate_samples, TensorType`` -- everything else (``x``, ``Y``, the hyper-parameters) is forwarded
to ``potential_func(position, **kwargs)``.  ``init_position`` may be an ndarray or a tensor.
"""
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
