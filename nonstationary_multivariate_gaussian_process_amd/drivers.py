"""The loops the log-posterior path is called from: MAP by Adam and HMC.

* :func:`map_nonseparable` mirrors ``train()`` of ``Nonseparable_Model/Nonseparable_model.py:147-210``: Adam over the
  three leaves ``[tilde_l | uL_vecs | tilde_sigma2_err]`` (lr 0.2 each), ``NegLog.backward()`` per iteration,
  ``target_value_hist[i] = -NegLog``, and the flat parameter vector pickled as ``MAP.dat`` every 100 iterations.
* :class:`HMCSampler` replaces the reference's EXTERNAL, un-vendored ``HMC_Sampler.HMC_sampler.sampler`` (a sibling
  checkout that is not in the reference tree; call sites ``Nonseparable_model.py:228-231``): same constructor keywords
  as those call sites (``sample_size, potential_func, init_position, step_size, num_steps_in_leap, adaptive_step_size,
  M, duplicate_samples, TensorType, **potential kwargs``) and ``main_hmc_loop() -> (samples [S, P], info)``.  Its
  trajectories cannot be pinned against the reference (the package is absent): it is validated by energy conservation
  and by posterior moments on a Gaussian target (tests/test_drivers.py).

Host-side orchestration only; every evaluation of the potential goes to the GPU through ``Utility.logpos``.
"""
from __future__ import annotations

import pickle

import numpy as np
import torch

from .Utility import logpos, settings


def map_nonseparable(x, Y, pars0, hyper_pars, N_opt=1000, lr=2e-1, checkpoint_path=None, checkpoint_every=100,
                     verbose=False, callback=None):
    """MAP estimate of the nonseparable model by Adam; returns (pars [P] ndarray, target_value_hist [N_opt])."""
    x = torch.as_tensor(x, dtype=torch.float64)
    Y = torch.as_tensor(Y, dtype=torch.float64)
    N, M = Y.shape
    T = M * (M + 1) // 2
    p0 = torch.as_tensor(np.asarray(pars0, dtype=np.float64))
    tilde_l = p0[:N].clone().requires_grad_(True)
    uL_vecs = p0[N:N + N * T].clone().requires_grad_(True)
    tilde_sigma2_err = p0[-1:].clone().requires_grad_(True)
    optimizer = torch.optim.Adam([{"params": tilde_l, "lr": lr}, {"params": [uL_vecs, tilde_sigma2_err], "lr": lr}])
    hist = np.zeros(N_opt)

    def flat():
        return torch.cat([tilde_l, uL_vecs, tilde_sigma2_err.view(1)])

    for i in range(N_opt):
        optimizer.zero_grad()
        out = logpos.nlogpos_obj_SVC(flat(), Y, x, **hyper_pars, verbose=True)
        NegLog = out[0]
        NegLog.backward()
        optimizer.step()
        hist[i] = -float(NegLog.detach())
        if verbose:
            print("iter %d: loglik %.6g lp_l %.6g lp_uL %.6g lp_s2 %.6g  -> %.8g" % (
                i, float(out[1]), float(out[2]), float(out[3]), float(out[4]), hist[i]))
        if callback is not None:
            callback(i, hist[i])
        if checkpoint_path and i % checkpoint_every == checkpoint_every - 1:
            with open(checkpoint_path, "wb") as f:          # same on-disk format as the reference's MAP.dat
                pickle.dump(flat().detach().numpy(), f)
    pars = flat().detach().numpy().copy()
    if checkpoint_path:
        with open(checkpoint_path, "wb") as f:
            pickle.dump(pars, f)
    return pars, hist


def map_separable(x, Y, pars0, hyper_pars, N_opt=2000, lr=2e-1, verbose=False):
    """MAP estimate of the separable model by Adam (``Separable_Model/Separable_model.py:147-166``): leaves
    ``[tilde_l | tilde_sigma | uL_vec | tilde_sigma2_err]``, two parameter groups with lr 0.2 each, ``NegLog.backward()`` per
    iteration.  Returns (pars [2N+T+1], target_value_hist [N_opt])."""
    x = torch.as_tensor(x, dtype=torch.float64)
    Y = torch.as_tensor(Y, dtype=torch.float64)
    N, M = Y.shape
    T = M * (M + 1) // 2
    p0 = torch.as_tensor(np.asarray(pars0, dtype=np.float64))
    tilde_l = p0[:N].clone().requires_grad_(True)
    tilde_sigma = p0[N:2 * N].clone().requires_grad_(True)
    uL_vec = p0[2 * N:2 * N + T].clone().requires_grad_(True)
    tilde_sigma2_err = p0[-1:].clone().requires_grad_(True)
    optimizer = torch.optim.Adam([{"params": [tilde_sigma, uL_vec, tilde_sigma2_err], "lr": lr}, {"params": tilde_l, "lr": lr}])
    hist = np.zeros(N_opt)
    for i in range(N_opt):
        optimizer.zero_grad()
        Pars = torch.cat([tilde_l, tilde_sigma, uL_vec, tilde_sigma2_err.view(1)])
        out = logpos.nlogpos_obj(Pars, Y, x, **hyper_pars, verbose=True)
        out[0].backward()
        optimizer.step()
        hist[i] = -float(out[0].detach())
        if verbose and i % 100 == 99:
            print("%d/%d target %.8g" % (i + 1, N_opt, hist[i]))
    return torch.cat([tilde_l, tilde_sigma, uL_vec, tilde_sigma2_err.view(1)]).detach().numpy().copy(), hist


def map_stationary(x, Y, pars0, hyper_pars, N_opt=1000, lr=1e-1, verbose=False):
    """MAP estimate of the stationary (LMC) model by Adam (``Stationary_Model/Stationary_model.py:112-131``): Adam(lr 0.1) over
    ``tilde_l, uL_vec, tilde_sigma2_err``; ``tilde_sigma`` stays at its initial value ("fixed for correlation", :89).  Returns
    (pars [T+3], target_value_hist [N_opt])."""
    x = torch.as_tensor(x, dtype=torch.float64)
    Y = torch.as_tensor(Y, dtype=torch.float64)
    M = Y.shape[1]
    T = M * (M + 1) // 2
    p0 = torch.as_tensor(np.asarray(pars0, dtype=np.float64))
    tilde_l = p0[:1].clone().requires_grad_(True)
    tilde_sigma = p0[1:2].clone()
    uL_vec = p0[2:2 + T].clone().requires_grad_(True)
    tilde_sigma2_err = p0[-1:].clone().requires_grad_(True)
    optimizer = torch.optim.Adam([tilde_l, uL_vec, tilde_sigma2_err], lr=lr)
    hist = np.zeros(N_opt)
    for i in range(N_opt):
        optimizer.zero_grad()
        Pars = torch.cat([tilde_l, tilde_sigma, uL_vec, tilde_sigma2_err.view(1)])
        out = logpos.nlogpos_obj_S(Pars, Y, x, **hyper_pars, verbose=True)
        out[0].backward()
        optimizer.step()
        hist[i] = -float(out[0].detach())
        if verbose:
            print("%dth iteration with target value %.8g" % (i + 1, hist[i]))
    return torch.cat([tilde_l, tilde_sigma, uL_vec, tilde_sigma2_err.view(1)]).detach().numpy().copy(), hist


def valid_rows(out, status):
    """ONE validity predicate for every lock-step driver, host- or device-resident: an evaluation counts when the factorisation
    succeeded (status 0) and both the log posterior and the likelihood are finite -- what k_alive_update / k_hmc_status test on the
    device (nmgp_kernels.hip) and nmgp_svc_batch_fetch folds into the status it returns."""
    out = np.asarray(out)
    return (np.asarray(status) == 0) & np.isfinite(out[:, 0]) & np.isfinite(out[:, 1])


class LockStepMAP:
    """MAP by Adam for B independent subjects (or B restarts of one subject) advanced in lock-step: every iteration asks
    ``value_and_grad(P [B, P])`` for the verbose tuples [B, 5], the gradients d NegLog / d pars [B, P] and a status [B] of ALL
    of them at once, then applies torch.optim.Adam's update rule (default betas / eps, no weight decay) row by row -- the same
    arithmetic, in the same order, as one ``torch.optim.Adam`` per subject, so a row reproduces :func:`map_nonseparable`.
    A subject whose evaluation fails (status != 0) is frozen at its last parameters and reported with NegLog = inf from then
    on (the reference wraps ``train()`` in try/except -> NegLog = inf, ``Nonseparable_model_mpisim.py:330-334``) while the
    others go on.  Subclasses provide ``value_and_grad``."""

    def __init__(self, init_pars, lr=2e-1, betas=(0.9, 0.999), eps=1e-8):
        self.P = np.array(init_pars, dtype=np.float64, copy=True)
        if self.P.ndim != 2:
            raise ValueError("init_pars must be [B, P]")
        self.lr, self.b1, self.b2, self.eps = float(lr), float(betas[0]), float(betas[1]), float(eps)
        self.m = np.zeros_like(self.P)
        self.v = np.zeros_like(self.P)
        self.t = 0
        self.alive = np.ones(self.P.shape[0], dtype=bool)

    def value_and_grad(self, P):
        raise NotImplementedError

    def step(self):
        out, grad, status = self.value_and_grad(self.P)
        ok = self.alive & valid_rows(out, status)
        self.alive = ok
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2_sqrt = np.sqrt(1.0 - self.b2 ** self.t)
        g = grad[ok]
        m = self.m[ok] * self.b1 + (1.0 - self.b1) * g            # exp_avg.lerp_(grad, 1 - beta1)
        v = self.v[ok] * self.b2 + (1.0 - self.b2) * g * g        # exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        denom = np.sqrt(v) / bc2_sqrt + self.eps
        self.P[ok] = self.P[ok] - (self.lr / bc1) * (m / denom)
        self.m[ok], self.v[ok] = m, v
        neglog = np.where(ok, out[:, 0], np.inf)
        return neglog, out

    def run(self, N_opt, callback=None):
        """Returns (pars [B, P], target_value_hist [N_opt, B] = -NegLog per iteration, alive [B])."""
        hist = np.zeros((N_opt, self.P.shape[0]))
        for i in range(N_opt):
            neglog, out = self.step()
            hist[i] = -neglog
            if callback is not None:
                callback(i, hist[i], out)
        return self.P.copy(), hist, self.alive.copy()


class BatchedMAP(LockStepMAP):
    """The MAP loop of ``Nonseparable_model_mpisim.py:330-348`` for ALL subjects of a rank at once: B subjects of the same size
    (own x, Y, own GP-prior factors) form one multi-subject batch on the GPU (``nmgp_svc_batch_set_subjects``) and every Adam
    iteration is ONE batched value+gradient launch sequence -- BASELINE config 4's per-GPU work (8 subjects x N = 1024) in the
    caller's own loop.  ``xs`` [B, N], ``Ys`` [B, N, M], ``init_pars`` [B, N(1+T)+1]."""

    def __init__(self, xs, Ys, hyper_pars, init_pars, lr=2e-1, ctx=None, device_resident=True):
        from . import _lib
        super().__init__(init_pars, lr=lr)
        self.device_resident = bool(device_resident)
        xs = np.ascontiguousarray(xs, dtype=np.float64)
        Ys = np.ascontiguousarray(Ys, dtype=np.float64)
        self.ctx = ctx if ctx is not None else _lib.default_context()
        keys = ("mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_L", "alpha_L", "beta_L", "a", "b")
        self.hyper = np.array([float(hyper_pars[k]) for k in keys])
        self.ctx.set_data(xs[0], Ys[0])
        self.ctx.svc_batch_alloc(xs.shape[0])
        self.ctx.svc_batch_set_subjects(xs, Ys)

    def value_and_grad(self, P):
        self.ctx.svc_batch_set_pars(P)
        self.ctx.svc_batch_eval(self.hyper, True, want_grad=True)
        out, status = self.ctx.svc_batch_fetch()
        return out, self.ctx.svc_batch_fetch_grad(), status

    def step(self):
        """``device_resident=True`` (default): parameters, gradients and Adam's moments stay in HBM
        (``nmgp_svc_batch_adam_step``: the update is an elementwise kernel behind the batched evaluation); per iteration only
        the B verbose tuples come back, ``self.P`` is refreshed at the end of :meth:`run` (or by :meth:`sync_pars`).  Same
        arithmetic as the host-side update (``device_resident=False``), bit for bit."""
        if not self.device_resident:
            return super().step()
        if self.t == 0:
            self.ctx.svc_batch_set_pars(self.P)
            self.ctx.svc_batch_adam_begin()
        out, alive = self.ctx.svc_batch_adam_step(self.hyper, True, self.lr, self.b1, self.b2, self.eps)
        self.t += 1
        self.alive = alive
        return np.where(alive, out[:, 0], np.inf), out

    def sync_pars(self):
        if self.device_resident and self.t > 0:
            self.P = self.ctx.svc_batch_get_pars()
        return self.P

    def run(self, N_opt, callback=None):
        _, hist, alive = super().run(N_opt, callback)
        return self.sync_pars().copy(), hist, alive


class HMCSampler:
    """Hamiltonian Monte Carlo with a (optionally dense) constant mass matrix.

    potential_func(position_tensor, **kwargs) -> scalar tensor (the NEGATIVE log posterior, e.g.
    ``logpos.nlogpos_obj_SVC``); gradients come from ``torch.autograd.grad`` (one fused GPU evaluation each).
    """

    def __init__(self, sample_size, potential_func, init_position, step_size=1e-4, num_steps_in_leap=20,
                 adaptive_step_size=False, M=None, duplicate_samples=True, TensorType=settings.torchType, seed=None,
                 target_accept=0.8, **kwargs):
        self.sample_size = int(sample_size)
        self.potential_func = potential_func
        self.q0 = np.asarray(init_position, dtype=np.float64).reshape(-1).copy()
        self.step_size = float(step_size)
        self.L = int(num_steps_in_leap)
        self.adaptive = bool(adaptive_step_size)
        self.duplicate_samples = bool(duplicate_samples)
        self.kwargs = kwargs
        self.rng = np.random.default_rng(seed)
        self.target_accept = target_accept
        P = self.q0.shape[0]
        if M is None:
            self.Mchol = None
            self.Minv = None
        else:
            M = np.asarray(M, dtype=np.float64)
            if M.shape != (P, P):
                raise ValueError("mass matrix must be [P, P]")
            self.Mchol = np.linalg.cholesky(M)
            self.Minv = np.linalg.inv(M)

    # -- pieces ------------------------------------------------------------------------------------
    def potential_and_grad(self, q):
        qt = torch.from_numpy(np.ascontiguousarray(q)).clone().requires_grad_(True)
        U = self.potential_func(qt, **self.kwargs)
        (g,) = torch.autograd.grad(U, qt)
        return float(U.detach()), g.numpy().copy()

    def kinetic(self, p):
        return 0.5 * float(p @ (p if self.Minv is None else self.Minv @ p))

    def draw_momentum(self, P):
        z = self.rng.standard_normal(P)
        return z if self.Mchol is None else self.Mchol @ z

    def velocity(self, p):
        return p if self.Minv is None else self.Minv @ p

    def leapfrog(self, q, p, g, eps):
        """L leapfrog steps from (q, p) with the gradient g at q; returns (q, p, U, g) at the end point."""
        p = p - 0.5 * eps * g
        U = None
        for step in range(self.L):
            q = q + eps * self.velocity(p)
            U, g = self.potential_and_grad(q)
            if not np.isfinite(U):
                return q, p, np.inf, g
            p = p - (eps if step < self.L - 1 else 0.5 * eps) * g
        return q, p, U, g

    # -- the loop -----------------------------------------------------------------------------------
    def main_hmc_loop(self):
        P = self.q0.shape[0]
        q = self.q0.copy()
        U, g = self.potential_and_grad(q)
        samples = np.zeros((self.sample_size, P))
        accepted = 0
        energy_err = np.zeros(self.sample_size)
        eps = self.step_size
        kept = 0
        it = 0
        while kept < self.sample_size:
            p0 = self.draw_momentum(P)
            H0 = U + self.kinetic(p0)
            try:
                q1, p1, U1, g1 = self.leapfrog(q, p0, g, eps)
                H1 = U1 + self.kinetic(p1)
            except RuntimeError:                  # covariance left the positive definite cone: reject
                U1, H1 = np.inf, np.inf
            with np.errstate(invalid="ignore"):       # inf - inf: start and end potential both undefined
                dH = H1 - H0
            u = np.log(self.rng.random())         # drawn every iteration: the stream does not depend on the outcome
            acc = bool(np.isfinite(dH) and (u < -dH))
            if acc:
                q, U, g = q1, U1, g1
                accepted += 1
            if acc or self.duplicate_samples:
                samples[kept] = q
                energy_err[kept] = dH if np.isfinite(dH) else np.nan
                kept += 1
            it += 1
            if self.adaptive and it <= max(50, self.sample_size // 2):
                # simple Robbins-Monro adaptation towards the target acceptance probability
                a = min(1.0, float(np.exp(-dH))) if np.isfinite(dH) else 0.0
                eps *= float(np.exp((a - self.target_accept) / np.sqrt(it)))
        info = {"accept_rate": accepted / max(it, 1), "step_size": eps, "energy_error": energy_err, "iterations": it}
        return samples, info


def sampler(**kw):
    """Alias with the reference's constructor spelling: ``HMC_Sampler.HMC_sampler.sampler(...)``."""
    return HMCSampler(**kw)


SVC_HYPER_KEYS = ("mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_L", "alpha_L", "beta_L", "a", "b")


def _randomized_eigs(hvp, P, k, rank, power_iters, lam_min, seed):
    """Leading eigenpairs (by |eigenvalue|) of the symmetric operator behind ``hvp(V [k, P]) -> [k, P]`` (row v -> A v) by randomised
    subspace iteration with k probes and ``power_iters`` extra passes.  Returns (U [r, P] orthonormal rows or None, lam [r] = |eig|,
    info).  A NEGATIVE eigenvalue below -lam_min means the reference point is not a mode along that direction (an unconverged MAP
    estimate, a saddle): it enters as |lam| (the SoftAbs rule), which keeps a leapfrog step stable while the chain leaves the region."""
    def orth(Yv):
        Q, _ = np.linalg.qr(Yv.T)
        return np.ascontiguousarray(Q.T)

    rng = np.random.default_rng(seed)
    Yv = hvp(rng.standard_normal((k, P)))
    for _ in range(int(power_iters)):
        Yv = hvp(orth(Yv))
    Q = orth(Yv)
    AQ = hvp(Q)
    Tm = AQ @ Q.T
    asym = float(np.abs(Tm - Tm.T).max() / max(np.abs(Tm).max(), 1e-300))
    ev, W = np.linalg.eigh(0.5 * (Tm + Tm.T))
    order = np.argsort(-np.abs(ev))
    ev, W = ev[order], W[:, order]
    keep = np.flatnonzero(np.abs(ev[:rank]) > lam_min)
    U = np.ascontiguousarray(W[:, keep].T @ Q) if keep.size else None
    lam = np.ascontiguousarray(np.abs(ev[keep])) if keep.size else None
    info = {"probes": int(k), "power_iters": int(power_iters), "kept": int(keep.size),
            "lam_max": float(lam[0]) if keep.size else 0.0, "lam_min_kept": float(lam[-1]) if keep.size else 0.0,
            "first_dropped": float(ev[keep.size]) if keep.size < ev.size else None,
            "negative_kept": [float(v) for v in ev[keep] if v < 0], "most_negative": float(ev.min()),
            "asymmetry_of_projected_hessian": asym, "eigenvalues": [float(v) for v in ev[:min(ev.size, rank + 8)]]}
    return U, lam, info


def _preconditioned_lbfgs(f, to_pars_factory, make_metric, q, maxiter, rounds, history, gtol, verbose, nev):
    """The iteration behind :func:`polish_map` / :func:`polish_map_separable`: L-BFGS in whitened coordinates w (pars = q0 + L w, the
    model's ``apply``), the initial matrix of the two-loop recursion H0 = (I + U diag(lam) U^T)^-1 from ``make_metric(q)`` (rebuilt
    every maxiter / rounds iterations), Armijo backtracking.  ``f(w, to_pars) -> (NegLog, whitened gradient)`` (+inf, None outside the
    domain); ``to_pars_factory(q0) -> to_pars(w)``; ``nev``: one-element evaluation counter shared with the caller."""
    q_cur = q
    P = q.shape[0]
    per_round = max(1, int(np.ceil(maxiter / max(rounds, 1))))
    fk = gn = None
    for rnd in range(max(rounds, 1)):
        met = make_metric(q_cur, rnd)
        U, sc = (met.U, met.lam / (1.0 + met.lam)) if met.rank else (None, None)
        to_pars = to_pars_factory(q_cur)

        def precond(d):
            return d - U.T @ (sc * (U @ d)) if U is not None else d

        xk = np.zeros(P)
        fk, gk = f(xk, to_pars)
        if not np.isfinite(fk):
            raise RuntimeError("polish: the objective is undefined at the start point")
        S, Yv, rho = [], [], []
        stalled = False
        for _ in range(per_round):
            gn = float(np.sqrt(_dot(gk, gk)))
            if gn <= gtol * max(1.0, abs(fk)):
                break
            d = -gk.copy()
            al = []
            for s_, y_, r_ in zip(reversed(S), reversed(Yv), reversed(rho)):
                a_ = r_ * _dot(s_, d)
                al.append(a_)
                d -= a_ * y_
            d = precond(d)                               # H0 = (I + U lam U^T)^-1
            for (s_, y_, r_), a_ in zip(zip(S, Yv, rho), reversed(al)):
                d += (a_ - r_ * _dot(y_, d)) * s_
            slope = _dot(gk, d)
            if slope >= 0:                               # stale pairs: restart from the preconditioned gradient
                S, Yv, rho = [], [], []
                d = precond(-gk.copy())
                slope = _dot(gk, d)
            t = 1.0
            while True:
                xn = xk + t * d
                fn, gnew = f(xn, to_pars)
                if fn <= fk + 1e-4 * t * slope:
                    break
                t *= 0.5
                if t < 1e-10:
                    stalled = True
                    break
            if stalled:
                break
            s_, y_ = xn - xk, gnew - gk
            sy = _dot(s_, y_)
            if sy > 1e-10 * np.sqrt(_dot(s_, s_) * _dot(y_, y_)):
                S.append(s_)
                Yv.append(y_)
                rho.append(1.0 / sy)
                if len(S) > history:
                    S.pop(0), Yv.pop(0), rho.pop(0)
            xk, fk, gk = xn, fn, gnew
        q_cur = to_pars(xk)
        gn = float(np.sqrt(_dot(gk, gk)))
        if verbose is not None:
            verbose("polish round %d: NegLog %.4f, whitened |grad| %.3g, %d evaluations so far, metric rank %d (most negative %.3g)" % (
                rnd, fk, gn, nev[0], met.rank, met.info["most_negative"]))
        if gn <= gtol * max(1.0, abs(fk)):
            break
    return q_cur, fk, gn, nev[0]


def _dot(a, b):
    # (not BLAS: on a 64-core host a threaded ddot of 14,337 elements costs more than the GPU evaluation it sits next to)
    return float(np.sum(a * b))


def polish_map(x, Y, hyper_pars, q, maxiter=300, ctx=None, history=30, gtol=1e-6, rounds=4, rank=64, probes=96, verbose=None):
    """From an Adam MAP estimate to the mode of the nonseparable objective (the reference stops Adam after a fixed number of
    iterations, Nonseparable_model.py:161-175; a sampler metric built from the Hessian wants a point that IS a mode -- at N = 2048
    the committed Adam estimate is 4,000 log-posterior units below it).

    L-BFGS in the coordinates w of the prior-factor metric, pars = q + L_blk w, PRECONDITIONED by the metric's low-rank part: the
    initial matrix of the two-loop recursion is H0 = (I + U diag(lam) U^T)^-1 from :func:`prior_lowrank_metric` at the current point
    (rebuilt every ``maxiter / rounds`` iterations: the likelihood's curvature moves while the point does), so the quasi-Newton
    pairs only have to learn how the Hessian differs from the metric.  In the parameters themselves the GP priors' condition number
    of 1e11 makes a quasi-Newton method crawl (2,000 iterations of SciPy's L-BFGS-B ended with |grad| = 108), and in plain
    whitened coordinates the 26 likelihood directions (eigenvalues up to 4e5 next to ~14,300 ones) do the same.  The change of
    coordinates is the device's (``nmgp_svc_batch_prior_apply``); two-loop recursion and Armijo backtracking in NumPy on the host,
    one single-chain value+gradient evaluation on the GPU per trial point (a point outside the positive definite cone counts as
    +inf).  Returns (pars, NegLog, whitened gradient norm, evaluations incl. the metric's)."""
    from . import _lib
    ctx = ctx if ctx is not None else _lib.default_context()
    hyper = np.array([float(hyper_pars[k]) for k in SVC_HYPER_KEYS])
    x, Y = np.asarray(x, dtype=np.float64), np.asarray(Y, dtype=np.float64)
    ctx.set_data(x, Y)
    nev = [0]
    q_cur = np.array(q, dtype=np.float64, copy=True).reshape(-1)
    P = q_cur.shape[0]
    B = int(min(probes, P))
    buf = np.zeros((B, P))

    def apply(v, trans):
        buf[0] = v
        return ctx.svc_batch_prior_apply(hyper, buf, trans=trans)[0].copy()

    def make_metric(q_at, rnd):
        met = prior_lowrank_metric(x, Y, hyper_pars, q_at, rank=rank, oversample=max(B - rank, 0), power_iters=1, seed=11 + rnd, ctx=ctx,
                                   batch=B)          # (leaves a batch of B chains allocated: `apply` uses it)
        nev[0] += met.info["grad_evals"]
        return met

    def to_pars_factory(q0):
        return lambda w: q0 + apply(w, False)

    def f(w, to_pars):
        nev[0] += 1
        try:
            out, g = ctx.logpos_svc(to_pars(w), hyper, True, True)
        except _lib.NmgpNumericalError:
            return np.inf, None
        v = float(out[0])
        if not (np.isfinite(v) and np.all(np.isfinite(g))):
            return np.inf, None
        return v, apply(g, True)

    return _preconditioned_lbfgs(f, to_pars_factory, make_metric, q_cur, maxiter, rounds, history, gtol, verbose, nev)


class PriorMetric:
    """The prior-factor metric of the device-resident trajectories (``nmgp_svc_batch_traj_set_mass_prior``, csrc/nmgp_metric.hip):

        M^-1 = L_blk (I + U diag(lam) U^T)^-1 L_blk^T,   L_blk = blockdiag(chol Sigma_l, chol Sigma_L per uL column, 1)

    with the Cholesky factors of the GP priors of ``logpos.py:357-365`` (cached on the device per subject) and an optional rank-r
    correction ``U`` [r, P] (or [S, r, P] for S subjects; orthonormal rows), ``lam`` [r] / [S, r] >= 0 for the curvature the
    likelihood adds in the whitened coordinates ``pars = mu + L_blk w`` -- see :func:`prior_lowrank_metric`.  It takes the place of
    the ``M = inv(sample covariance)`` the reference's production runs pass (Nonseparable_model_mpiKAISER.py:398-411), which at
    P = 14,337 would need more draws than a run has."""

    def __init__(self, hyper_pars, U=None, lam=None, info=None):
        self.hyper = np.array([float(hyper_pars[k]) for k in SVC_HYPER_KEYS])
        self.U = None if U is None else np.ascontiguousarray(U, dtype=np.float64)
        self.lam = None if lam is None else np.ascontiguousarray(lam, dtype=np.float64)
        self.info = info or {}

    @property
    def rank(self):
        return 0 if self.U is None else int(self.U.shape[-2])

    @staticmethod
    def stack(metrics):
        """One metric for a multi-subject batch from the subjects' own metrics (same hyper-parameters): U [S, r, P], lam [S, r] with
        r = the largest rank; a subject with fewer directions is padded with lam = 0 rows, which leave its metric unchanged."""
        hyper = metrics[0].hyper
        if any(not np.array_equal(m.hyper, hyper) for m in metrics):
            raise ValueError("the subjects' metrics must share the GP-prior hyper-parameters")
        r = max(m.rank for m in metrics)
        out = PriorMetric.__new__(PriorMetric)
        out.hyper, out.info = hyper.copy(), {"subjects": [m.info for m in metrics]}
        if r == 0:
            out.U = out.lam = None
            return out
        P = next(m.U.shape[-1] for m in metrics if m.U is not None)
        out.U, out.lam = np.zeros((len(metrics), r, P)), np.zeros((len(metrics), r))
        for s_, m in enumerate(metrics):
            if m.rank:
                out.U[s_, :m.rank], out.lam[s_, :m.rank] = m.U, m.lam
        return out


def prior_lowrank_metric(x, Y, hyper_pars, q_ref, rank=96, oversample=32, power_iters=1, h=1e-3, lam_min=0.5, seed=0, ctx=None,
                         batch=None):
    """Build the :class:`PriorMetric` of ONE subject at ``q_ref`` (its MAP estimate): the top eigenpairs of

        A = L_blk^T  Hess(-loglik)(q_ref)  L_blk        (the prior's own Hessian is the identity in these coordinates)

    by randomised subspace iteration (``rank + oversample`` probe vectors, ``power_iters`` extra passes).  Every Hessian-vector
    product is a central difference of the LIKELIHOOD gradient (``Prior=False``: no ill-conditioned prior solve enters), ``batch``
    of them per batched launch sequence (``nmgp_svc_batch_eval``), the change of coordinates runs on the device
    (``nmgp_svc_batch_prior_apply``).  ``h``: largest parameter displacement of a probe.  Eigenvalues with |lam| below ``lam_min``
    are dropped (they change a direction's scale by < 25 %); a negative one beyond that enters as |lam|.  Cost: (2 + power_iters) x 2 x (rank + oversample) gradient evaluations."""
    from . import _lib
    ctx = ctx if ctx is not None else _lib.default_context()
    x, Y = np.asarray(x, dtype=np.float64), np.asarray(Y, dtype=np.float64)
    hyper = np.array([float(hyper_pars[k]) for k in SVC_HYPER_KEYS])
    q_ref = np.asarray(q_ref, dtype=np.float64).reshape(-1)
    P = q_ref.shape[0]
    k = int(min(rank + oversample, P))
    B = int(batch) if batch else min(k, 128)
    ctx.set_data(x, Y)
    ctx.svc_batch_alloc(B)
    n_grad = [0]

    def in_chunks(V, fn):
        out = np.empty_like(V)
        buf = np.zeros((B, P))
        for a in range(0, V.shape[0], B):
            m = min(B, V.shape[0] - a)
            buf[:m] = V[a:a + m]
            buf[m:] = buf[0]                  # pad with a valid row
            out[a:a + m] = fn(buf)[:m]
        return out

    def lik_grad(Q):
        ctx.svc_batch_set_pars(Q)
        ctx.svc_batch_eval(hyper, False, want_grad=True)
        out, status = ctx.svc_batch_fetch()
        if not valid_rows(out, status).all():
            raise RuntimeError("prior_lowrank_metric: the likelihood is undefined at a probe point (status %s): reduce h" % status)
        n_grad[0] += Q.shape[0]
        return ctx.svc_batch_fetch_grad()

    def hvp(V):
        """A V^T for the rows of V [k, P]."""
        D = in_chunks(V, lambda v: ctx.svc_batch_prior_apply(hyper, v, trans=False))
        t = h / np.maximum(np.abs(D).max(1), 1e-300)
        Gp = in_chunks(q_ref[None] + t[:, None] * D, lik_grad)
        Gm = in_chunks(q_ref[None] - t[:, None] * D, lik_grad)
        Hq = (Gp - Gm) / (2.0 * t[:, None])
        return in_chunks(Hq, lambda g: ctx.svc_batch_prior_apply(hyper, g, trans=True))

    U, lam, info = _randomized_eigs(hvp, P, k, rank, power_iters, lam_min, seed)
    info.update(h=float(h), grad_evals=int(n_grad[0]))
    return PriorMetric(hyper_pars, U, lam, info)


SEP_HYPER_KEYS = ("mu_tilde_l", "alpha_tilde_l", "beta_tilde_l", "mu_tilde_sigma", "alpha_tilde_sigma", "beta_tilde_sigma", "a", "b", "c")


class SeparablePriorMetric:
    """The prior-factor metric of the SEPARABLE model (logpos.py:216-296; sampler call Separable_model.py:209-210), host side: its
    parameter vector [tilde_l (N) | tilde_sigma (N) | uL_vec (T) | tilde_sigma2_err] carries GP priors RBF(alpha, beta) + 1e-6 I on
    tilde_l and tilde_sigma (logpos.py:271-281) and Normal(0, c) on uL_vec (:283), so
        L_blk = blockdiag(chol Sigma_l, chol Sigma_sigma, c I_T, 1),     M^-1 = L_blk (I + U diag(lam) U^T)^-1 L_blk^T
    exactly as :class:`PriorMetric` for the nonseparable model.  P = 2N + T + 1 is small enough (8,208 at config 5's size) for the
    leapfrog update itself to stay on the host; its two triangular products per step run on the device with the context's cached
    prior factors (``nmgp_sep_prior_apply``: the nonseparable sampler's ``k_prior_trmm`` with the separable parameter layout)."""

    def __init__(self, L_l, L_s, c, T, U=None, lam=None, info=None, ctx=None, hyper=None, N=None):
        """Either the two factors as host matrices (``L_l``, ``L_s``; tests, or a caller with its own factors) or ``ctx`` + ``hyper``
        [9] + ``N``: the products then run on the device with the context's cached prior factors (``nmgp_sep_prior_apply``)."""
        self.ctx, self.hyper = ctx, (None if hyper is None else np.asarray(hyper, dtype=np.float64))
        self.c, self.T = float(c), int(T)
        if ctx is None:
            self.L_l, self.L_s = np.ascontiguousarray(L_l), np.ascontiguousarray(L_s)
            # both orientations contiguous: the two products of a leapfrog step then are plain row-major GEMMs [B, N] x [N, N]
            self.Lt_l = np.ascontiguousarray(self.L_l.T)
            self.Lt_s = self.Lt_l if L_s is L_l else np.ascontiguousarray(self.L_s.T)
            self.N = self.L_l.shape[0]
        else:
            self.L_l = self.L_s = self.Lt_l = self.Lt_s = None
            self.N = int(N)
        self.P = 2 * self.N + self.T + 1
        self.U = None if U is None else np.ascontiguousarray(U, dtype=np.float64)
        self.lam = None if lam is None else np.ascontiguousarray(lam, dtype=np.float64)
        self.info = info or {}

    @property
    def rank(self):
        return 0 if self.U is None else int(self.U.shape[0])

    def apply(self, v, trans):
        """L_blk v (trans False) or L_blk^T v (trans True) for the rows of v [B, P]."""
        if self.ctx is not None:
            return self.ctx.sep_prior_apply(self.hyper, np.ascontiguousarray(v), trans)
        N, T = self.N, self.T
        out = np.empty_like(v)
        out[:, :N] = np.ascontiguousarray(v[:, :N]) @ (self.L_l if trans else self.Lt_l)
        out[:, N:2 * N] = np.ascontiguousarray(v[:, N:2 * N]) @ (self.L_s if trans else self.Lt_s)
        out[:, 2 * N:2 * N + T] = self.c * v[:, 2 * N:2 * N + T]
        out[:, -1] = v[:, -1]
        return out

    def _lowrank(self, u, w):
        return u if self.U is None else u + ((u @ self.U.T) * w) @ self.U

    def root(self, z):          # u = (I + U lam U^T)^1/2 z
        return self._lowrank(z, None if self.U is None else np.sqrt(1.0 + self.lam) - 1.0)

    def W(self, u):             # (I + U lam U^T)^-1 u
        return self._lowrank(u, None if self.U is None else -self.lam / (1.0 + self.lam))

    def kinetic(self, u):
        k = (u * u).sum(1)
        if self.U is not None:
            c = u @ self.U.T
            k = k - (c * c * (self.lam / (1.0 + self.lam))).sum(1)
        return 0.5 * k


def separable_prior_metric(x, Y, hyper_pars, q_ref, rank=64, oversample=32, power_iters=1, h=1e-3, lam_min=0.5, seed=0, ctx=None,
                           batch=16, factors=None):
    """:class:`SeparablePriorMetric` of one subject at ``q_ref``: the context's cached GP-prior factors (``factors`` is accepted for
    compatibility and ignored) and the leading eigenpairs of L_blk^T Hess(-loglik) L_blk by randomised subspace iteration on central
    differences of the likelihood gradient, ``batch`` chains per ``nmgp_sep_batch_eval``."""
    from . import _lib
    ctx = ctx if ctx is not None else _lib.default_context()
    x, Y = np.asarray(x, dtype=np.float64), np.asarray(Y, dtype=np.float64)
    hyper = np.array([float(hyper_pars[k]) for k in SEP_HYPER_KEYS])
    q_ref = np.asarray(q_ref, dtype=np.float64).reshape(-1)
    N, M = Y.shape
    T = M * (M + 1) // 2
    P = 2 * N + T + 1
    if q_ref.shape[0] != P:
        raise ValueError("q_ref must have 2N + T + 1 = %d entries" % P)
    ctx.set_data(x, Y)
    # Normal(0, c) sees a float32-rounded c in the reference (logpos.py:283 passes a Python number to torch.distributions.Normal)
    met = SeparablePriorMetric(None, None, float(np.float32(hyper[8])), T, ctx=ctx, hyper=hyper, N=N)
    k = int(min(rank + oversample, P))
    n_grad = [0]

    def lik_grad(Q):
        G = np.empty_like(Q)
        for a in range(0, Q.shape[0], batch):
            out, g, st = ctx.sep_batch_eval(Q[a:a + batch], hyper, False, True)
            if (st != 0).any() or not np.all(np.isfinite(out[:, 1])):
                raise RuntimeError("separable_prior_metric: the likelihood is undefined (or needed jitter) at a probe point: reduce h")
            G[a:a + batch] = g
        n_grad[0] += Q.shape[0]
        return G

    def hvp(V):
        D = met.apply(V, False)
        t = h / np.maximum(np.abs(D).max(1), 1e-300)
        Hq = (lik_grad(q_ref[None] + t[:, None] * D) - lik_grad(q_ref[None] - t[:, None] * D)) / (2.0 * t[:, None])
        return met.apply(Hq, True)

    met.U, met.lam, info = _randomized_eigs(hvp, P, k, rank, power_iters, lam_min, seed)
    info.update(h=float(h), grad_evals=int(n_grad[0]))
    met.info = info
    return met


def polish_map_separable(x, Y, hyper_pars, q, maxiter=300, ctx=None, history=30, gtol=1e-6, rounds=4, rank=64, probes=96, batch=16,
                         verbose=None):
    """:func:`polish_map` for the separable objective (``logpos.nlogpos_obj``): metric-preconditioned L-BFGS in the whitened
    coordinates of :class:`SeparablePriorMetric`.  Returns (pars, NegLog, whitened gradient norm, evaluations incl. the metric's)."""
    from . import _lib
    ctx = ctx if ctx is not None else _lib.default_context()
    hyper = np.array([float(hyper_pars[k]) for k in SEP_HYPER_KEYS])
    x, Y = np.asarray(x, dtype=np.float64), np.asarray(Y, dtype=np.float64)
    ctx.set_data(x, Y)
    nev = [0]
    q_cur = np.array(q, dtype=np.float64, copy=True).reshape(-1)
    state = {"met": None}

    def make_metric(q_at, rnd):
        met = separable_prior_metric(x, Y, hyper_pars, q_at, rank=rank, oversample=max(probes - rank, 0), power_iters=1, seed=11 + rnd,
                                     ctx=ctx, batch=batch, factors=state["met"])
        nev[0] += met.info["grad_evals"]
        state["met"] = met
        return met

    def to_pars_factory(q0):
        return lambda w: q0 + state["met"].apply(w[None], False)[0]

    def f(w, to_pars):
        nev[0] += 1
        try:
            out, g = ctx.logpos_sep(to_pars(w), hyper, True, True)
        except _lib.NmgpNumericalError:
            return np.inf, None
        v = float(out[0])
        if not (np.isfinite(v) and np.all(np.isfinite(g))):
            return np.inf, None
        return v, state["met"].apply(g[None], True)[0]

    return _preconditioned_lbfgs(f, to_pars_factory, make_metric, q_cur, maxiter, rounds, history, gtol, verbose, nev)


class LockStepHMC:
    """B independent HMC chains advanced in lock-step: every leapfrog step asks ``potential_and_grad(q [B, P])`` for the
    potentials U [B] and gradients [B, P] of ALL chains at once (U = inf marks a chain whose potential is undefined at
    that point, e.g. a covariance outside the positive definite cone).  Constant mass matrix (identity by default, see
    :meth:`set_mass`).  Chain b draws from its own
    generator in the order of :class:`HMCSampler` (momentum, then the uniform of the accept test), so a chain reproduces
    the single-chain sampler started from the same state.  Subclasses provide ``potential_and_grad``."""

    def __init__(self, init_positions, step_size=1e-4, num_steps_in_leap=20, seed=None, M=None, Minv=None):
        self.q = np.array(init_positions, dtype=np.float64, copy=True)
        if self.q.ndim != 2:
            raise ValueError("init_positions must be [B, P]")
        self.B, self.P = self.q.shape
        self.eps = float(step_size)
        self.L = int(num_steps_in_leap)
        self.rngs = [np.random.default_rng(None if seed is None else seed + b) for b in range(self.B)]
        self.set_mass(M, Minv)

    def set_mass(self, M=None, Minv=None):
        """Constant mass matrix shared by the chains (``M`` of the reference's sampler call, Nonseparable_model_mpiKAISER.py:267-270:
        M = inv(sample covariance)): None = identity, a vector [P] = diagonal, a matrix [P, P] = dense.  ``Minv`` may be given
        alongside (the reference's caller HAS it: it is the sample covariance) to spare the inversion; momenta are drawn as
        chol(M) z, the kinetic energy is 1/2 p^T M^-1 p and the drift q += eps M^-1 p."""
        P = self.P
        self.mass_kind = 0
        self.Mchol = self.Minv = self.metric = None
        if M is None and Minv is None:
            return
        if isinstance(M, PriorMetric):
            # the prior-factor metric lives on the device only (whitened momenta, cached prior factors): BatchedHMC's resident loop
            self.mass_kind = 3
            self.metric = M
            return
        if isinstance(M, SeparablePriorMetric):
            if M.P != P:
                raise ValueError("the metric belongs to a parameter vector of length %d, not %d" % (M.P, P))
            self.mass_kind = 4           # host-side whitened-momentum loop of BatchedHMCSeparable
            self.metric = M
            return
        if M is None:
            Minv = np.asarray(Minv, dtype=np.float64)
            M = 1.0 / Minv if Minv.ndim == 1 else np.linalg.inv(Minv)
        M = np.asarray(M, dtype=np.float64)
        if M.shape == (P,):
            if np.any(M <= 0):
                raise ValueError("diagonal mass matrix must be positive")
            self.mass_kind = 1
            self.Mchol = np.sqrt(M)
            self.Minv = 1.0 / M if Minv is None else np.asarray(Minv, dtype=np.float64).reshape(P)
        elif M.shape == (P, P):
            self.mass_kind = 2
            self.Mchol = np.linalg.cholesky(M)
            self.Minv = np.linalg.inv(M) if Minv is None else np.asarray(Minv, dtype=np.float64).reshape(P, P)
        else:
            raise ValueError("mass matrix must be [P] or [P, P] with P = %d" % P)

    def draw_momenta(self):
        """[B, P]: chain b draws P standard normals from its own generator (the stream of :class:`HMCSampler`), then p = chol(M) z."""
        z = np.stack([r.standard_normal(self.P) for r in self.rngs])
        if self.mass_kind == 0:
            return z
        if self.mass_kind >= 3:
            raise NotImplementedError("a PriorMetric / SeparablePriorMetric runs in its sampler's own whitened-momentum loop")
        return z * self.Mchol if self.mass_kind == 1 else z @ self.Mchol.T

    def velocity(self, p):
        if self.mass_kind == 0:
            return p
        if self.mass_kind >= 3:
            raise NotImplementedError("a PriorMetric / SeparablePriorMetric runs in its sampler's own whitened-momentum loop")
        return p * self.Minv if self.mass_kind == 1 else p @ self.Minv          # (M^-1 is symmetric)

    def kinetic(self, p):
        return 0.5 * (p * self.velocity(p)).sum(1)

    def potential_and_grad(self, q):
        raise NotImplementedError

    def run(self, sample_size):
        B, P = self.B, self.P
        samples = np.zeros((sample_size, B, P))
        U, g = self.potential_and_grad(self.q)
        accepted = np.zeros(B)
        energy_err = np.zeros((sample_size, B))
        for it in range(sample_size):
            p0 = self.draw_momenta()
            H0 = U + self.kinetic(p0)
            q1 = self.q.copy()
            p1 = p0 - 0.5 * self.eps * g
            U1, g1 = U, g
            # A chain whose potential is undefined at ANY intermediate point has left the leapfrog map (its momentum kick
            # was skipped there): it must be rejected even if the trajectory comes back to a valid point, exactly as
            # HMCSampler.leapfrog returns inf at the first non-finite potential.
            failed = np.zeros(B, dtype=bool)
            for step in range(self.L):
                q1 = q1 + self.eps * self.velocity(p1)
                U1, g1 = self.potential_and_grad(q1)
                failed |= ~np.isfinite(U1)
                p1 = p1 - (self.eps if step < self.L - 1 else 0.5 * self.eps) * g1
            U1 = np.where(failed, np.inf, U1)
            H1 = U1 + self.kinetic(p1)
            with np.errstate(invalid="ignore"):       # inf - inf: start and end potential both undefined
                dH = H1 - H0
            u = np.array([np.log(r.random()) for r in self.rngs])
            acc = np.isfinite(dH) & (u < -dH)
            self.q[acc] = q1[acc]
            U = np.where(acc, U1, U)
            g = np.where(acc[:, None], g1, g)
            accepted += acc
            energy_err[it] = np.where(np.isfinite(dH), dH, np.nan)
            samples[it] = self.q
        return samples, {"accept_rate": accepted / sample_size, "energy_error": energy_err}


class BatchedHMC(LockStepHMC):
    """B independent HMC chains of one subject advanced in lock-step on the GPU.

    Every leapfrog step evaluates the potential and its gradient for ALL chains with one batched launch sequence
    (``nmgp_svc_batch_eval(want_grad=1)``): the chains are the reference's embarrassingly-parallel unit
    (one process each there, ``Nonseparable_model_mpisim.py:305-306``); here they share the GPU's launch latency.
    Nonseparable model only (the batched entry point of the C ABI).  ``M`` / ``Minv``: constant mass matrix shared by the
    chains (None: identity; [P]: diagonal; [P, P]: dense, as the reference's production sampler passes it,
    Nonseparable_model_mpiKAISER.py:267-270,398-411) -- resident on the device, the drift of every leapfrog step is one GEMM.
    ``x`` [N], ``Y`` [N, M]: B chains of one subject; ``x`` [S, N], ``Y`` [S, N, M]: S subjects with ``chains_per_subject``
    chains each (default 1: config 4's unit, one chain per subject), ``init_positions`` [S * chains_per_subject, P] subject-major
    (row s * chains_per_subject + k = chain k of subject s).  The chains of a subject share its data and prior factors on the
    device; 8 subjects x 8 chains is config 4's per-GPU subject count at a batch size where the factorisation is
    throughput-bound, and what a between-chain diagnostic (R-hat) needs.
    """

    def __init__(self, x, Y, hyper_pars, init_positions, step_size=1e-4, num_steps_in_leap=20, seed=None, ctx=None,
                 device_resident=True, M=None, Minv=None, chains_per_subject=1, device_momenta=None, step_jitter=0.0):
        from . import _lib
        super().__init__(init_positions, step_size, num_steps_in_leap, seed, M, Minv)
        self.device_resident = bool(device_resident)
        # step_jitter j > 0: iteration i uses eps (1 + j u_i), u_i ~ U(-1, 1) from a generator of its own (the chains' streams are
        # untouched), the same step for all chains of the iteration -- the usual guard against a trajectory length that resonates
        # with the posterior's periods (under a PriorMetric they are all ~2 pi).  Device-resident loop only.
        self.step_jitter = float(step_jitter)
        self.jitter_rng = np.random.default_rng(None if seed is None else 7919 * (seed + 1))
        # device_momenta: the host only draws the standard normals z; p0 = chol(M) z and the end point's kinetic energy
        # 1/2 p1^T M^-1 p1 are formed on the device (nmgp_svc_batch_traj_z).  Default: on whenever a mass matrix is set -- the
        # host's share of a dense-mass sample was two [B, P] x [P, P] NumPy products -- off for the identity (where it would only
        # change the summation order of 1/2 |p|^2 against the host-side loop the tests compare with bit for bit).
        self.device_momenta = (self.mass_kind != 0) if device_momenta is None else bool(device_momenta)
        if self.mass_kind == 3:
            if not self.device_resident or not self.device_momenta:
                raise ValueError("a PriorMetric needs device_resident=True and device_momenta=True")
        self.ctx = ctx if ctx is not None else _lib.default_context()
        self.hyper = np.array([float(hyper_pars[k]) for k in SVC_HYPER_KEYS])
        x, Y = np.asarray(x, dtype=np.float64), np.asarray(Y, dtype=np.float64)
        if x.ndim == 2:
            # one chain per SUBJECT (BASELINE config 4's unit: x [B, N], Y [B, N, M], every subject with its own prior factors)
            k = int(chains_per_subject)
            if k < 1 or x.shape[0] * k != self.B or Y.shape[0] * k != self.B:
                raise ValueError("x [S, N] and Y [S, N, M] must hold B / chains_per_subject = %d / %d subjects" % (self.B, k))
            self.ctx.set_data(x[0], Y[0])
            self.ctx.svc_batch_alloc(self.B)
            self.ctx.svc_batch_set_subjects(x, Y, k)
        else:
            self.ctx.set_data(x, Y)
            self.ctx.svc_batch_alloc(self.B)

    def potential_and_grad(self, q):
        """U [B] and dU/dq [B, P]; a chain whose covariance is not positive definite gets U = inf."""
        self.ctx.svc_batch_set_pars(q)
        self.ctx.svc_batch_eval(self.hyper, True, want_grad=True)
        out, status = self.ctx.svc_batch_fetch()
        g = self.ctx.svc_batch_fetch_grad()
        U = out[:, 0].copy()
        bad = ~valid_rows(out, status)
        U[bad] = np.inf
        g[bad] = 0.0
        return U, g

    def run(self, sample_size):
        """``device_resident=True`` (default): positions, momenta and gradients stay in HBM for the whole trajectory
        (``nmgp_svc_batch_traj``: the leapfrog updates are elementwise kernels between the batched evaluations); per sample
        the host uploads the momenta it drew, reads the end point back and decides acceptance -- the same arithmetic and
        the same random streams as the host-side lock-step loop (``device_resident=False``), bit for bit."""
        if not self.device_resident:
            return super().run(sample_size)
        import time
        B, P = self.B, self.P
        samples = np.zeros((sample_size, B, P))
        t_start = time.perf_counter()
        U, _ = self.potential_and_grad(self.q)           # leaves q and dU/dq resident
        if self.mass_kind == 3:
            self.ctx.svc_batch_traj_set_mass_prior(self.metric.hyper, self.metric.U, self.metric.lam)
        else:
            self.ctx.svc_batch_traj_set_mass(None if self.mass_kind == 0 else self.Minv)
            if self.device_momenta and self.mass_kind != 0:
                self.ctx.svc_batch_traj_set_mass_chol(self.Mchol)
        self.ctx.svc_batch_traj_begin()
        accepted = np.zeros(B)
        energy_err = np.zeros((sample_size, B))
        t_loop = time.perf_counter()
        t_traj = 0.0
        for it in range(sample_size):
            eps = self.eps if self.step_jitter <= 0 else self.eps * (1.0 + self.step_jitter * self.jitter_rng.uniform(-1.0, 1.0))
            if self.device_momenta:
                # the same random stream: chain b draws its P standard normals, then (below) the accept uniform
                z = np.stack([r.standard_normal(P) for r in self.rngs])
                H0 = U + 0.5 * (z * z).sum(1)            # p0 = chol(M) z  =>  1/2 p0^T M^-1 p0 = 1/2 |z|^2
                t0 = time.perf_counter()
                q1, K1, U1, failed = self.ctx.svc_batch_traj_z(self.hyper, True, eps, self.L, z)
                t_traj += time.perf_counter() - t0
            else:
                p0 = self.draw_momenta()
                H0 = U + self.kinetic(p0)
                t0 = time.perf_counter()
                q1, p1, U1, failed = self.ctx.svc_batch_traj(self.hyper, True, eps, self.L, p0)
                t_traj += time.perf_counter() - t0
                K1 = self.kinetic(p1)
            U1 = np.where(failed, np.inf, U1)
            H1 = U1 + K1
            with np.errstate(invalid="ignore"):       # inf - inf: start and end potential both undefined
                dH = H1 - H0
            u = np.array([np.log(r.random()) for r in self.rngs])
            acc = np.isfinite(dH) & (u < -dH)
            self.ctx.svc_batch_traj_commit(acc)
            self.q[acc] = q1[acc]
            U = np.where(acc, U1, U)
            accepted += acc
            energy_err[it] = np.where(np.isfinite(dH), dH, np.nan)
            samples[it] = self.q
        t_end = time.perf_counter()
        # where a sample's wall time goes: the synchronous trajectory calls (upload of the normals, the leapfrog launches, the end
        # point's download) against everything the host does around them (normal draws, energies, accept test, bookkeeping)
        timing = {"setup_seconds": t_loop - t_start, "loop_seconds": t_end - t_loop, "trajectory_call_seconds": t_traj,
                  "device_share": t_traj / max(t_end - t_loop, 1e-12)}
        return samples, {"accept_rate": accepted / sample_size, "energy_error": energy_err, "timing": timing}


class BatchedHMCSeparable(LockStepHMC):
    """B independent HMC chains of the SEPARABLE model of one subject in lock-step: every leapfrog step evaluates
    ``logpos.nlogpos_obj`` and its gradient for all chains with one launch sequence (``nmgp_sep_batch_eval``: the chains' B*M
    blocks ``wB[p] K_x + sigma2 I`` form one batch of the blocked Cholesky).  The sampler call of ``Separable_model.py:209`` /
    ``Separable_model_mpiKAISER.py:281`` for B chains at once; chain b reproduces ``HMCSampler(potential_func=logpos.nlogpos_obj,
    ...)`` started from the same state with the same random stream.  The leapfrog update runs on the host (P = 2N + T + 1).
    ``M=`` a :class:`SeparablePriorMetric` (from :func:`separable_prior_metric`) selects the whitened-momentum loop -- the metric
    under which this model's chains mix; ``step_jitter`` as in :class:`BatchedHMC`."""

    KEYS = SEP_HYPER_KEYS

    def __init__(self, x, Y, hyper_pars, init_positions, step_size=2e-4, num_steps_in_leap=20, seed=None, ctx=None, M=None, Minv=None,
                 step_jitter=0.0):
        from . import _lib
        super().__init__(init_positions, step_size, num_steps_in_leap, seed, M, Minv)
        self.step_jitter = float(step_jitter)
        self.jitter_rng = np.random.default_rng(None if seed is None else 7919 * (seed + 1))
        self.ctx = ctx if ctx is not None else _lib.default_context()
        self.hyper = np.array([float(hyper_pars[k]) for k in self.KEYS])
        self.ctx.set_data(np.asarray(x, dtype=np.float64), np.asarray(Y, dtype=np.float64))

    def potential_and_grad(self, q):
        """U [B] and dU/dq [B, P]; a chain whose covariance stays numerically singular after the jitter retries gets U = inf."""
        out, g, status = self.ctx.sep_batch_eval(q, self.hyper, True, True)
        U = out[:, 0].copy()
        bad = (status < 0) | ~np.isfinite(U)
        U[bad] = np.inf
        g[bad] = 0.0
        return U, g

    def run(self, sample_size):
        """With ``M=`` a :class:`SeparablePriorMetric` the chains carry the whitened momentum u = L_blk^T p (as the device-resident
        nonseparable sampler does): draw u = (I + U lam U^T)^1/2 z, kick u -= c L_blk^T g, drift q += eps L_blk (I + U lam U^T)^-1 u,
        kinetic energy 1/2 u^T (I + U lam U^T)^-1 u -- triangular PRODUCTS with the prior factors only, on the host, around one
        batched evaluation on the GPU per leapfrog step.  Any other mass matrix: the lock-step loop of the base class."""
        if self.mass_kind != 4:
            return super().run(sample_size)
        met = self.metric
        B, P = self.B, self.P
        samples = np.zeros((sample_size, B, P))
        U, g = self.potential_and_grad(self.q)
        accepted = np.zeros(B)
        energy_err = np.zeros((sample_size, B))
        for it in range(sample_size):
            eps = self.eps if self.step_jitter <= 0 else self.eps * (1.0 + self.step_jitter * self.jitter_rng.uniform(-1.0, 1.0))
            z = np.stack([r.standard_normal(P) for r in self.rngs])
            H0 = U + 0.5 * (z * z).sum(1)
            q1 = self.q.copy()
            u1 = met.root(z) - 0.5 * eps * met.apply(g, True)
            U1, g1 = U, g
            failed = np.zeros(B, dtype=bool)
            for step in range(self.L):
                q1 = q1 + eps * met.apply(met.W(u1), False)
                U1, g1 = self.potential_and_grad(q1)
                failed |= ~np.isfinite(U1)
                u1 = u1 - (eps if step < self.L - 1 else 0.5 * eps) * met.apply(g1, True)
            U1 = np.where(failed, np.inf, U1)
            H1 = U1 + met.kinetic(u1)
            with np.errstate(invalid="ignore"):
                dH = H1 - H0
            lu = np.array([np.log(r.random()) for r in self.rngs])
            acc = np.isfinite(dH) & (lu < -dH)
            self.q[acc] = q1[acc]
            U = np.where(acc, U1, U)
            g = np.where(acc[:, None], g1, g)
            accepted += acc
            energy_err[it] = np.where(np.isfinite(dH), dH, np.nan)
            samples[it] = self.q
        return samples, {"accept_rate": accepted / sample_size, "energy_error": energy_err}


# ---- the whole recipe behind one call ---------------------------------------------------------------------------------------------
def _sample_recipe(polish, build_metric, make_sampler, pars0, chains, iters, warm, warm_step, windows, window_iters, step_size,
                   step_candidates, target_accept, progress, segment):
    """Mode -> metric -> thermalising iterations -> (metric rebuilt at the chains' mean) x windows -> step search -> main run.
    polish(pars0) -> (mode, NegLog, |grad|, evaluations); build_metric(point, k) -> metric; make_sampler(positions, metric, eps, seed)
    -> an object with run(n) -> (samples [n, B, P], info).  Returns (samples [iters, B, P], info)."""
    import time
    say = progress if progress is not None else (lambda msg: None)
    info = {}
    t0 = time.time()
    mode, nl, gn, nev = polish(np.asarray(pars0, dtype=np.float64).reshape(-1))
    info["mode"] = {"pars": mode, "log_posterior": -nl, "whitened_gradient_norm": gn, "gradient_evaluations": nev, "seconds": time.time() - t0}
    say("mode: log posterior %.4f, whitened |grad| %.3g, %d evaluations, %.1f s" % (-nl, gn, nev, time.time() - t0))
    t0 = time.time()
    metric = build_metric(mode, 0)
    info["metric_at_the_mode"] = dict({k: v for k, v in metric.info.items() if k != "eigenvalues"}, rank=metric.rank, seconds=time.time() - t0)
    say("metric at the mode: rank %d, lam max %.3g, most negative %.3g, %.1f s" % (
        metric.rank, metric.info["lam_max"], metric.info["most_negative"], time.time() - t0))
    cur = np.repeat(mode[None], chains, 0)
    stages = []

    def stage(eps, n, seed, tag):
        hm = make_sampler(cur, metric, eps, seed)
        t1 = time.time()
        chunks, ees, acc, done = [], [], np.zeros(chains), 0
        while done < n:
            k = min(segment, n - done)
            s_, inf = hm.run(k)
            chunks.append(s_)
            ees.append(inf["energy_error"])
            acc += inf["accept_rate"] * k
            done += k
            if n > segment:
                say("  %s: %d / %d iterations, %.1f s, accept so far %.3f" % (tag, done, n, time.time() - t1, acc.sum() / (done * chains)))
        dt = time.time() - t1
        ee = np.concatenate(ees)
        st = {"stage": tag, "step_size": eps, "iterations": n, "seconds": dt, "accept_rate_mean": float(acc.sum() / (n * chains)),
              "accept_rate_by_chain": (acc / n).tolist(), "median_abs_dH": float(np.nanmedian(np.abs(ee))),
              "samples_per_s": n * chains / dt}
        stages.append(st)
        say("%s: %d iterations at eps %.3g in %.1f s, accept %.3f, median |dH| %.3g" % (tag, n, eps, dt, st["accept_rate_mean"], st["median_abs_dH"]))
        return np.concatenate(chunks), ee

    s = None
    if warm > 0:
        # all chains start AT the mode, where the first trajectories convert P / 2 units of kinetic into potential energy: the leapfrog
        # error of that transfer rejects every step worth having, so a few iterations at a small step come first
        s, _ = stage(warm_step, warm, 300, "warm-up")
        cur = s[-1]
    for wdw in range(int(windows)):
        center = (s[-max(10, s.shape[0] // 2):].mean((0, 1)) if s is not None else mode)
        t0 = time.time()
        metric = build_metric(center, 1 + wdw)
        say("window %d: metric at the chains' mean: rank %d, lam max %.3g, most negative %.3g, %.1f s" % (
            wdw, metric.rank, metric.info["lam_max"], metric.info["most_negative"], time.time() - t0))
        s, _ = stage(min(step_candidates), window_iters, 400 + wdw, "adaptation window %d" % wdw)
        cur = s[-1]
    if step_size == "auto":
        tried, best = [], None
        for eps in step_candidates:
            _, inf = make_sampler(cur, metric, eps, 200).run(6)
            a_ = float(inf["accept_rate"].mean())
            tried.append({"step_size": eps, "accept_rate_mean": a_, "median_abs_dH": float(np.nanmedian(np.abs(inf["energy_error"])))})
            say("step search: eps %.3g accept %.2f median |dH| %.3g" % (eps, a_, tried[-1]["median_abs_dH"]))
            if a_ >= target_accept:
                best = eps
            elif best is not None:
                break
        eps = best if best is not None else min(step_candidates)
        info["step_search"] = tried
    else:
        eps = float(step_size)
    S, ee = stage(eps, iters, 1, "main")
    info.update(step_size=eps, stages=stages, metric=metric, energy_error=ee, accept_rate=np.array(stages[-1]["accept_rate_by_chain"]))
    return S, info


def sample_nonseparable(x, Y, hyper_pars, pars0, chains=8, iters=1000, num_steps_in_leap=20, rank=64, polish=300, warm=40,
                        warm_step=0.03, windows=0, window_iters=50, step_size="auto", step_candidates=(0.05, 0.065, 0.08, 0.1, 0.12),
                        target_accept=0.7, step_jitter=0.2, seed=1, ctx=None, progress=None, segment=50):
    """The sampler call of ``Nonseparable_model.py:228-231`` for ``chains`` chains in lock-step, as the recipe under which it
    converges at N = 2048 (``tools/hmc_1000.py`` is the same recipe with its diagnostics): mode by metric-preconditioned
    L-BFGS from ``pars0`` (:func:`polish_map`), :class:`PriorMetric` there (:func:`prior_lowrank_metric`), ``warm`` thermalising
    iterations at ``warm_step``, optionally ``windows`` re-adaptations of the metric at the chains' mean, a step search for
    ``target_accept``, then ``iters`` iterations of :class:`BatchedHMC` with ``step_jitter``.  ``progress``: a callable for one-line
    messages.  Returns (samples [iters, chains, P], info: mode, metrics, stages, step_size, energy_error, accept_rate)."""
    from . import _lib
    ctx = ctx if ctx is not None else _lib.default_context()
    x, Y = np.asarray(x, dtype=np.float64), np.asarray(Y, dtype=np.float64)

    def make_sampler(pos, metric, eps, sd):
        return BatchedHMC(x, Y, hyper_pars, pos, step_size=eps, num_steps_in_leap=num_steps_in_leap, seed=sd + 1000 * seed, ctx=ctx, M=metric,
                          step_jitter=step_jitter)
    return _sample_recipe(lambda p: polish_map(x, Y, hyper_pars, p, maxiter=polish, rounds=8, rank=rank, probes=rank + 32, ctx=ctx,
                                               verbose=progress),
                          lambda q, k: prior_lowrank_metric(x, Y, hyper_pars, q, rank=rank, oversample=32, seed=7 + k, ctx=ctx,
                                                            batch=min(rank + 32, 128)),
                          make_sampler, pars0, chains, iters, warm, warm_step, windows, window_iters, step_size, step_candidates,
                          target_accept, progress, segment)


def sample_separable(x, Y, hyper_pars, pars0, chains=8, iters=1000, num_steps_in_leap=20, rank=64, polish=400, warm=50, warm_step=0.04,
                     windows=2, window_iters=50, step_size="auto", step_candidates=(0.08, 0.11, 0.15), target_accept=0.8,
                     step_jitter=0.2, seed=1, ctx=None, progress=None, segment=25, batch=16):
    """The same for the separable model (``Separable_model.py:209-210``; :class:`SeparablePriorMetric`, :class:`BatchedHMCSeparable`).
    ``windows`` defaults to 2: this posterior's mass sits far from its mode (the sigma(x) <-> B scale ridge), so the metric is
    rebuilt at the chains' mean after the warm-up (``tools/hmc_sep.py``)."""
    from . import _lib
    ctx = ctx if ctx is not None else _lib.default_context()
    x, Y = np.asarray(x, dtype=np.float64), np.asarray(Y, dtype=np.float64)

    def make_sampler(pos, metric, eps, sd):
        return BatchedHMCSeparable(x, Y, hyper_pars, pos, step_size=eps, num_steps_in_leap=num_steps_in_leap, seed=sd + 1000 * seed, ctx=ctx,
                                   M=metric, step_jitter=step_jitter)
    return _sample_recipe(lambda p: polish_map_separable(x, Y, hyper_pars, p, maxiter=polish, rounds=8, rank=rank, probes=rank + 32,
                                                         batch=batch, ctx=ctx, verbose=progress),
                          lambda q, k: separable_prior_metric(x, Y, hyper_pars, q, rank=rank, oversample=32, seed=3 + k, ctx=ctx,
                                                              batch=batch),
                          make_sampler, pars0, chains, iters, warm, warm_step, windows, window_iters, step_size, step_candidates,
                          target_accept, progress, segment)
