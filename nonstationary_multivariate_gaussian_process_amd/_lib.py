"""ctypes binding of libnmgp_hip.so (the C ABI declared in include/nmgp.h).

The product path has NO CPU fallback: if the shared object is missing, cannot be loaded, or no MI355X is
visible, every compute entry raises.  Only ``load(require_gpu=False)`` is allowed without a GPU and only
resolves symbols (used by the CPU test-suite to check the ABI surface).
"""
from __future__ import annotations

import ctypes
import os
import threading

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libnmgp_hip.so")

c_double_p = ctypes.POINTER(ctypes.c_double)
c_ll_p = ctypes.POINTER(ctypes.c_longlong)
c_void_pp = ctypes.POINTER(ctypes.c_void_p)
I, D, P, V = ctypes.c_int, ctypes.c_double, c_double_p, ctypes.c_void_p

NUM_NAN = 1 << 20
NUM_EIG = (1 << 20) + 1
STAGES = ["cov", "chol", "solve", "reduce", "prior", "inverse", "adjoint", "eig", "kronmv", "syrk"]

# name -> (restype, argtypes); must list every function declared in include/nmgp.h
SIGNATURES = {
    "nmgp_ctx_create": (I, [I, c_void_pp]),
    "nmgp_ctx_destroy": (I, [V]),
    "nmgp_last_error": (ctypes.c_char_p, [V]),
    "nmgp_version": (I, []),
    "nmgp_build_id": (ctypes.c_char_p, []),
    "nmgp_sync": (I, [V]),
    "nmgp_device_count": (I, []),
    "nmgp_set_data": (I, [V, P, P, I, I]),
    "nmgp_logpos_svc": (I, [V, P, P, I, P, P]),
    "nmgp_svc_set_pars": (I, [V, P]),
    "nmgp_svc_pars_dev": (V, [V]),
    "nmgp_svc_grad_dev": (V, [V]),
    "nmgp_svc_eval_resident": (I, [V, P, I, I]),
    "nmgp_svc_fetch": (I, [V, P, P]),
    "nmgp_svc_batch_alloc": (I, [V, I]),
    "nmgp_svc_batch_set_pars": (I, [V, P]),
    "nmgp_svc_batch_set_subjects": (I, [V, P, P]),
    "nmgp_svc_batch_pars_dev": (V, [V]),
    "nmgp_svc_batch_eval": (I, [V, P, I, I]),
    "nmgp_svc_batch_fetch_grad": (I, [V, P]),
    "nmgp_svc_batch_grad_dev": (V, [V]),
    "nmgp_svc_batch_fetch": (I, [V, P, ctypes.POINTER(ctypes.c_int)]),
    "nmgp_svc_batch_set_subjects_chains": (I, [V, P, P, I]),
    "nmgp_svc_batch_traj_begin": (I, [V]),
    "nmgp_svc_batch_traj_set_mass": (I, [V, I, P]),
    "nmgp_svc_batch_traj": (I, [V, P, I, D, I, P, P, P, P, ctypes.POINTER(ctypes.c_int)]),
    "nmgp_svc_batch_traj_set_mass_chol": (I, [V, I, P]),
    "nmgp_svc_batch_traj_z": (I, [V, P, I, D, I, P, P, P, P, ctypes.POINTER(ctypes.c_int)]),
    "nmgp_svc_batch_traj_commit": (I, [V, ctypes.POINTER(ctypes.c_int)]),
    "nmgp_svc_batch_traj_set_mass_prior": (I, [V, P, I, P, P]),
    "nmgp_svc_batch_prior_apply": (I, [V, P, I, P, P]),
    "nmgp_sep_prior_apply": (I, [V, P, I, I, P, P]),
    "nmgp_svc_batch_adam_begin": (I, [V]),
    "nmgp_svc_batch_adam_step": (I, [V, P, I, D, D, D, D, P, ctypes.POINTER(ctypes.c_int)]),
    "nmgp_svc_batch_get_pars": (I, [V, P]),
    "nmgp_svc_covariance": (I, [V, P, P]),
    "nmgp_logpos_sep": (I, [V, P, P, I, P, P]),
    "nmgp_logpos_sta": (I, [V, P, P, I, P, P]),
    "nmgp_sep_batch_eval": (I, [V, P, I, P, I, P, P, ctypes.POINTER(ctypes.c_int)]),
    "nmgp_pairwise_distances": (I, [V, P, I, P, I, I, P]),
    "nmgp_rbf_cov": (I, [V, P, I, P, I, I, D, D, P]),
    "nmgp_nonstat_rbf_cov": (I, [V, P, P, P, I, P, P, P, I, I, P]),
    "nmgp_kron_product": (I, [V, P, I, I, P, I, I, P]),
    "nmgp_kron_mv": (I, [V, P, I, I, P, I, I, P, P]),
    "nmgp_mvn_logpdf": (I, [V, P, P, D, P, I, P]),
    "nmgp_mvn_logpdf_kron": (I, [V, P, P, P, I, P, I, D, P]),
    "nmgp_mvn_logpdf_dense": (I, [V, P, P, P, I, P, I, D, P]),
    "nmgp_kron_inv_logdet": (I, [V, D, P, I, P, I, P, P]),
    "nmgp_cholesky": (I, [V, P, I, P, P, P, I]),
    "nmgp_predict_svc": (I, [V, P, P, P, I, P, P, P]),
    "nmgp_predict_sep": (I, [V, P, P, P, I, P, P]),
    "nmgp_predict_sta": (I, [V, P, P, I, P, P]),
    "nmgp_profile_enable": (I, [V, I]),
    "nmgp_profile_read": (I, [V, P, c_ll_p]),
    "nmgp_profile_reset": (I, [V]),
    "nmgp_profile_read_work": (I, [V, P, c_ll_p, P, P]),
    "nmgp_last_sep_attempts": (I, [V]),
    "nmgp_measure_hbm_gbs": (I, [V, ctypes.c_longlong, I, P]),
    "nmgp_measure_hbm_rates": (I, [V, ctypes.c_longlong, I, P]),
    "nmgp_measure_dgemm_tflops": (I, [V, I, I, P]),
}

_lib = None
_lock = threading.Lock()


class NmgpError(RuntimeError):
    """API misuse or HIP runtime failure reported by libnmgp_hip.so (negative return code)."""


class NmgpNumericalError(RuntimeError):
    """Numerical failure (positive return code): covariance not positive definite, NaN, eigensolver."""

    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def load(require_gpu=True):
    """Load the shared object and bind every symbol of the ABI.  Raises if it is absent: there is no fallback."""
    global _lib
    # torch bundles its own ROCm runtime (libamdhip64 / rocBLAS / rocSOLVER with the same sonames as /opt/rocm).
    # Two HIP runtimes in one process crash, so torch must be loaded FIRST: libnmgp_hip.so then binds to the copies
    # already in the process.  (A pure C/C++ host links /opt/rocm directly; see INTEGRATION.md.)
    import torch  # noqa: F401
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise NmgpError(
                    "libnmgp_hip.so is missing (%s). Build it with `python -m "
                    "nonstationary_multivariate_gaussian_process_amd.build` (needs hipcc); this package has no CPU "
                    "fallback." % LIB_PATH)
            lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)        # AttributeError if the .so does not export it
                fn.restype = res
                fn.argtypes = args
            # provenance: the shared object must be the build of THIS tree (sources + headers + flags), not a stale one
            from . import build as _build
            have, want = lib.nmgp_build_id().decode(), _build.tree_id()
            if have != want:
                raise NmgpError("libnmgp_hip.so is stale: it was built from tree %s..., the sources beside it hash to %s... "
                                "Rebuild with `python -m nonstationary_multivariate_gaussian_process_amd.build`." % (have[:16], want[:16]))
            _lib = lib
    if require_gpu and _lib.nmgp_device_count() <= 0:
        raise NmgpError("no HIP device is visible: the MI355X path cannot run (and there is no CPU fallback)")
    return _lib


def build_id():
    """The loaded library's build id (== build.tree_id() of the tree, or load() would have refused it)."""
    return load(require_gpu=False).nmgp_build_id().decode()


def as_f64(a):
    """Contiguous float64 ndarray view/copy of array-like or CPU torch tensor."""
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def ptr(a):
    return a.ctypes.data_as(c_double_p) if a is not None else None


class Context:
    """One nmgp_ctx: one GPU, one stream, one resident subject.  Not thread-safe."""

    def __init__(self, device=None):
        self.lib = load(require_gpu=True)
        if device is None:
            device = default_device()
        self.device = int(device)
        h = ctypes.c_void_p()
        rc = self.lib.nmgp_ctx_create(self.device, ctypes.byref(h))
        self.h = h
        if rc != 0:
            msg = self.lib.nmgp_last_error(h).decode() if h else "nmgp_ctx_create failed"
            raise NmgpError("nmgp_ctx_create(device=%d) -> %d: %s" % (self.device, rc, msg))
        self.N = self.M = self.T = 0
        self._data_key = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.nmgp_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- error mapping --------------------------------------------------------------------------
    def check(self, rc):
        if rc == 0:
            return
        msg = self.lib.nmgp_last_error(self.h).decode()
        if rc > 0:
            raise NmgpNumericalError(rc, msg)
        raise NmgpError("libnmgp_hip error %d: %s" % (rc, msg))

    # -- data -----------------------------------------------------------------------------------
    def set_data(self, x, Y):
        x = as_f64(x).reshape(-1)
        Y = as_f64(Y)
        if Y.ndim != 2 or Y.shape[0] != x.shape[0]:
            raise NmgpError("Y must be [N, M] with N == len(x); got Y%s x%s" % (Y.shape, x.shape))
        key = (x.shape, Y.shape, hash(x.tobytes()), hash(Y.tobytes()))
        if key == self._data_key:
            return
        self.check(self.lib.nmgp_set_data(self.h, ptr(x), ptr(Y), Y.shape[0], Y.shape[1]))
        self.N, self.M = Y.shape
        self.T = self.M * (self.M + 1) // 2
        self._data_key = key

    def sync(self):
        self.check(self.lib.nmgp_sync(self.h))

    # -- nonseparable ---------------------------------------------------------------------------
    def logpos_svc(self, pars, hyper, prior=True, want_grad=False):
        pars = as_f64(pars).reshape(-1)
        P_ = self.N * (1 + self.T) + 1
        if pars.shape[0] != P_:
            raise NmgpError("parameter vector has length %d, expected N(1+T)+1 = %d" % (pars.shape[0], P_))
        hyper = as_f64(hyper)
        out = np.empty(5)
        grad = np.empty(P_) if want_grad else None
        self.check(self.lib.nmgp_logpos_svc(self.h, ptr(pars), ptr(hyper), int(bool(prior)), ptr(out), ptr(grad)))
        return out, grad

    def svc_set_pars(self, pars):
        pars = as_f64(pars).reshape(-1)
        if pars.shape[0] != self.N * (1 + self.T) + 1:
            raise NmgpError("bad parameter vector length %d" % pars.shape[0])
        self.check(self.lib.nmgp_svc_set_pars(self.h, ptr(pars)))
        self.sync()      # the host buffer may be a temporary

    def svc_eval_resident(self, hyper, prior=True, want_grad=False):
        hyper = as_f64(hyper)
        self.check(self.lib.nmgp_svc_eval_resident(self.h, ptr(hyper), int(bool(prior)), int(bool(want_grad))))

    def svc_fetch(self, want_grad=False):
        out = np.empty(5)
        grad = np.empty(self.N * (1 + self.T) + 1) if want_grad else None
        self.check(self.lib.nmgp_svc_fetch(self.h, ptr(out), ptr(grad)))
        return out, grad

    # -- batched chains ---------------------------------------------------------------------------
    def svc_batch_alloc(self, B):
        self.check(self.lib.nmgp_svc_batch_alloc(self.h, int(B)))
        self.B = int(B)

    def svc_batch_set_pars(self, pars):
        pars = as_f64(pars)
        P_ = self.N * (1 + self.T) + 1
        if pars.shape != (self.B, P_):
            raise NmgpError("batched parameters must be [B=%d, P=%d], got %s" % (self.B, P_, pars.shape))
        self.check(self.lib.nmgp_svc_batch_set_pars(self.h, ptr(pars)))
        self.sync()

    def svc_batch_set_subjects(self, xs, Ys, chains_per_subject=1):
        """The batch holds S = B / chains_per_subject subjects: xs [S, N], Ys [S, N, M] (same N, M as set_data); batch element
        b = s * chains_per_subject + k is chain k of subject s.  With the default of 1 every batch element is its own subject."""
        xs, Ys = as_f64(xs), as_f64(Ys)
        k = int(chains_per_subject)
        if k < 1 or self.B % k:
            raise NmgpError("the batch size %d is not a multiple of chains_per_subject = %d" % (self.B, k))
        S = self.B // k
        if xs.shape != (S, self.N) or Ys.shape != (S, self.N, self.M):
            raise NmgpError("subjects must be xs [S=%d, N=%d], Ys [S, N, M=%d]; got %s, %s"
                            % (S, self.N, self.M, xs.shape, Ys.shape))
        self.check(self.lib.nmgp_svc_batch_set_subjects_chains(self.h, ptr(xs), ptr(Ys), k))

    def svc_batch_eval(self, hyper, prior=True, want_grad=False):
        hyper = as_f64(hyper)
        self.check(self.lib.nmgp_svc_batch_eval(self.h, ptr(hyper), int(bool(prior)), int(bool(want_grad))))

    def svc_batch_fetch_grad(self):
        grad = np.empty((self.B, self.N * (1 + self.T) + 1))
        self.check(self.lib.nmgp_svc_batch_fetch_grad(self.h, ptr(grad)))
        return grad

    def svc_batch_fetch(self):
        out = np.empty((self.B, 5))
        status = np.zeros(self.B, dtype=np.int32)
        self.check(self.lib.nmgp_svc_batch_fetch(self.h, ptr(out), status.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
        return out, status

    # -- device-resident leapfrog trajectories of the batch (HMC) ---------------------------------
    def svc_batch_traj_begin(self):
        """After set_pars + batch_eval(want_grad=True) at the start positions: positions and gradients become the state
        the trajectories start from."""
        self.check(self.lib.nmgp_svc_batch_traj_begin(self.h))

    def svc_batch_traj_set_mass(self, minv=None):
        """Mass matrix of the device-resident trajectories: None (identity), diag(M^-1) [P] or dense M^-1 [P, P]."""
        if minv is None:
            self.check(self.lib.nmgp_svc_batch_traj_set_mass(self.h, 0, None))
            return
        minv = as_f64(minv)
        P = self.N * (1 + self.T) + 1
        if minv.shape == (P,):
            kind = 1
        elif minv.shape == (P, P):
            kind = 2
        else:
            raise ValueError("minv must be [P] or [P, P] with P = %d" % P)
        self.check(self.lib.nmgp_svc_batch_traj_set_mass(self.h, kind, ptr(minv)))

    def svc_batch_traj_set_mass_chol(self, mchol):
        """chol(M) of the mass matrix set by svc_batch_traj_set_mass: sqrt(diag M) [P] or a square root R [P, P] with R R^T = M
        (NumPy row-major; np.linalg.cholesky(M), or e.g. L^-T when M^-1 = L L^T is what one has) -- lets svc_batch_traj_z draw
        the momenta p = R z on the device."""
        mchol = as_f64(mchol)
        P = self.N * (1 + self.T) + 1
        if mchol.shape == (P,):
            self.check(self.lib.nmgp_svc_batch_traj_set_mass_chol(self.h, 1, ptr(mchol)))
        elif mchol.shape == (P, P):
            mt = np.ascontiguousarray(mchol.T)                   # column-major R == row-major of its transpose
            self.check(self.lib.nmgp_svc_batch_traj_set_mass_chol(self.h, 2, ptr(mt)))
        else:
            raise ValueError("mchol must be [P] or [P, P] with P = %d" % P)

    def svc_batch_traj_set_mass_prior(self, hyper, U=None, lam=None):
        """The prior-factor metric M^-1 = L_blk (I + U diag(lam) U^T)^-1 L_blk^T of the trajectories (nmgp.h): L_blk from the cached
        GP-prior factors named by hyper [8]; U [r, P] (one subject) or [S, r, P], lam [r] / [S, r] >= 0 the optional low-rank
        correction in whitened coordinates (drivers.prior_lowrank_metric builds it)."""
        hyper = as_f64(hyper)
        if U is None:
            self.check(self.lib.nmgp_svc_batch_traj_set_mass_prior(self.h, ptr(hyper), 0, None, None))
            return
        U, lam = as_f64(U), as_f64(lam)
        P_ = self.N * (1 + self.T) + 1
        if U.ndim == 2:
            U, lam = U[None], lam.reshape(1, -1)
        if U.ndim != 3 or U.shape[2] != P_ or lam.shape != U.shape[:2]:
            raise NmgpError("U must be [S, r, P=%d] with lam [S, r]; got %s, %s" % (P_, U.shape, lam.shape))
        U, lam = np.ascontiguousarray(U), np.ascontiguousarray(lam)
        self.check(self.lib.nmgp_svc_batch_traj_set_mass_prior(self.h, ptr(hyper), int(U.shape[1]), ptr(U), ptr(lam)))

    def svc_batch_prior_apply(self, hyper, v, trans=False):
        """L_blk v (trans=False) or L_blk^T v (trans=True) for B parameter-shaped vectors v [B, P]: the change of coordinates of the
        prior-factor metric (L_blk = blockdiag of the GP-prior Cholesky factors, 1 for the noise parameter)."""
        hyper, v = as_f64(hyper), as_f64(v)
        P_ = self.N * (1 + self.T) + 1
        if v.shape != (self.B, P_):
            raise NmgpError("v must be [B=%d, P=%d], got %s" % (self.B, P_, v.shape))
        out = np.empty_like(v)
        self.check(self.lib.nmgp_svc_batch_prior_apply(self.h, ptr(hyper), int(bool(trans)), ptr(v), ptr(out)))
        return out

    def sep_prior_apply(self, hyper, v, trans=False):
        """L_blk v / L_blk^T v for B vectors v [B, 2N+T+1] of the SEPARABLE parameter layout: L_blk = blockdiag(chol Sigma_l,
        chol Sigma_sigma, c I_T, 1) from the cached GP-prior factors named by hyper [9]."""
        hyper, v = as_f64(hyper), as_f64(v)
        P_ = 2 * self.N + self.T + 1
        if v.ndim != 2 or v.shape[1] != P_:
            raise NmgpError("v must be [B, 2N+T+1 = %d], got %s" % (P_, v.shape))
        out = np.empty_like(v)
        self.check(self.lib.nmgp_sep_prior_apply(self.h, ptr(hyper), int(bool(trans)), int(v.shape[0]), ptr(v), ptr(out)))
        return out

    def svc_batch_traj_z(self, hyper, prior, eps, nsteps, z):
        """One leapfrog trajectory per chain with momenta p0 = chol(M) z formed on the device from the standard normals z [B, P]:
        returns the end point q1 [B, P], the kinetic energy there kin1 [B] = 1/2 p1^T M^-1 p1, the potential U1 [B] (inf for failed
        chains) and failed [B] (bool)."""
        hyper, z = as_f64(hyper), as_f64(z)
        P_ = self.N * (1 + self.T) + 1
        if z.shape != (self.B, P_):
            raise NmgpError("z must be [B=%d, P=%d], got %s" % (self.B, P_, z.shape))
        q1, kin1, U1 = np.empty((self.B, P_)), np.empty(self.B), np.empty(self.B)
        failed = np.zeros(self.B, dtype=np.int32)
        self.check(self.lib.nmgp_svc_batch_traj_z(self.h, ptr(hyper), int(bool(prior)), float(eps), int(nsteps), ptr(z),
                                                  ptr(q1), ptr(kin1), ptr(U1),
                                                  failed.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
        return q1, kin1, U1, failed.astype(bool)

    def svc_batch_traj(self, hyper, prior, eps, nsteps, p0):
        """One leapfrog trajectory per chain from the resident state with momenta p0 [B, P]: returns the end point
        (q1, p1 [B, P]), the potential there U1 [B] (inf for failed chains) and failed [B] (bool)."""
        hyper, p0 = as_f64(hyper), as_f64(p0)
        P_ = self.N * (1 + self.T) + 1
        if p0.shape != (self.B, P_):
            raise NmgpError("momenta must be [B=%d, P=%d], got %s" % (self.B, P_, p0.shape))
        q1, p1, U1 = np.empty((self.B, P_)), np.empty((self.B, P_)), np.empty(self.B)
        failed = np.zeros(self.B, dtype=np.int32)
        self.check(self.lib.nmgp_svc_batch_traj(self.h, ptr(hyper), int(bool(prior)), float(eps), int(nsteps), ptr(p0),
                                                ptr(q1), ptr(p1), ptr(U1),
                                                failed.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
        return q1, p1, U1, failed.astype(bool)

    def svc_batch_traj_commit(self, accept):
        acc = np.ascontiguousarray(np.asarray(accept).astype(np.int32))
        if acc.shape != (self.B,):
            raise NmgpError("accept must be [B=%d]" % self.B)
        self.check(self.lib.nmgp_svc_batch_traj_commit(self.h, acc.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))

    # -- device-resident Adam over the batch (MAP) ------------------------------------------------
    def svc_batch_adam_begin(self):
        self.check(self.lib.nmgp_svc_batch_adam_begin(self.h))

    def svc_batch_adam_step(self, hyper, prior, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        """One Adam iteration of every batch element on the device: returns the verbose tuples [B, 5] at the parameters the
        iteration started from and alive [B] (bool)."""
        hyper = as_f64(hyper)
        out = np.empty((self.B, 5))
        alive = np.zeros(self.B, dtype=np.int32)
        self.check(self.lib.nmgp_svc_batch_adam_step(self.h, ptr(hyper), int(bool(prior)), float(lr), float(beta1),
                                                     float(beta2), float(eps), ptr(out),
                                                     alive.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
        return out, alive.astype(bool)

    def svc_batch_get_pars(self):
        pars = np.empty((self.B, self.N * (1 + self.T) + 1))
        self.check(self.lib.nmgp_svc_batch_get_pars(self.h, ptr(pars)))
        return pars

    def svc_covariance(self, pars):
        pars = as_f64(pars).reshape(-1)
        n = self.N * self.M
        out = np.empty((n, n))
        self.check(self.lib.nmgp_svc_covariance(self.h, ptr(pars), ptr(out)))
        return out

    # -- separable / stationary -------------------------------------------------------------------
    def logpos_sep(self, pars, hyper, prior=True, want_grad=False):
        pars = as_f64(pars).reshape(-1)
        P_ = 2 * self.N + self.T + 1
        if pars.shape[0] != P_:
            raise NmgpError("parameter vector has length %d, expected 2N+T+1 = %d" % (pars.shape[0], P_))
        hyper = as_f64(hyper)
        out = np.empty(6)
        grad = np.empty(P_) if want_grad else None
        self.check(self.lib.nmgp_logpos_sep(self.h, ptr(pars), ptr(hyper), int(bool(prior)), ptr(out), ptr(grad)))
        return out, grad

    def sep_batch_eval(self, pars, hyper, prior=True, want_grad=False):
        """B chains of the separable model in one launch sequence: pars [B, 2N+T+1] -> (out [B, 6], grad [B, P] or None,
        status [B]: 0 exact, 1..3 jitter retries needed, negative = numerical failure even so (row NaN))."""
        pars = as_f64(pars)
        P_ = 2 * self.N + self.T + 1
        if pars.ndim != 2 or pars.shape[1] != P_:
            raise NmgpError("parameters must be [B, 2N+T+1 = %d], got %s" % (P_, pars.shape))
        hyper = as_f64(hyper)
        B = pars.shape[0]
        out = np.empty((B, 6))
        grad = np.empty((B, P_)) if want_grad else None
        status = np.zeros(B, dtype=np.int32)
        self.check(self.lib.nmgp_sep_batch_eval(self.h, ptr(pars), B, ptr(hyper), int(bool(prior)), ptr(out), ptr(grad),
                                                status.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
        return out, grad, status

    def logpos_sta(self, pars, hyper, prior=True, want_grad=False):
        pars = as_f64(pars).reshape(-1)
        P_ = self.T + 3
        if pars.shape[0] != P_:
            raise NmgpError("parameter vector has length %d, expected T+3 = %d" % (pars.shape[0], P_))
        hyper = as_f64(hyper)
        out = np.empty(5)
        grad = np.empty(P_) if want_grad else None
        self.check(self.lib.nmgp_logpos_sta(self.h, ptr(pars), ptr(hyper), int(bool(prior)), ptr(out), ptr(grad)))
        return out, grad

    # -- primitives -----------------------------------------------------------------------------
    def pairwise_distances(self, x1, x2=None):
        x1 = as_f64(x1)
        x2a = None if x2 is None else as_f64(x2)
        n1, d = x1.shape
        n2 = n1 if x2a is None else x2a.shape[0]
        out = np.empty((n1, n2))
        self.check(self.lib.nmgp_pairwise_distances(self.h, ptr(x1), n1, ptr(x2a), n2, d, ptr(out)))
        return out

    def rbf_cov(self, x1, x2=None, alpha=1.0, beta=1.0):
        x1 = as_f64(x1)
        x2a = None if x2 is None else as_f64(x2)
        n1, d = x1.shape
        n2 = n1 if x2a is None else x2a.shape[0]
        out = np.empty((n1, n2))
        self.check(self.lib.nmgp_rbf_cov(self.h, ptr(x1), n1, ptr(x2a), n2, d, float(alpha), float(beta), ptr(out)))
        return out

    def nonstat_rbf_cov(self, x1, s1=None, l1=None, x2=None, s2=None, l2=None):
        x1 = as_f64(x1)
        n1, d = x1.shape
        s1a = None if s1 is None else as_f64(s1).reshape(-1)
        l1a = None if l1 is None else as_f64(l1).reshape(-1)
        x2a = None if x2 is None else as_f64(x2)
        s2a = None if (x2 is None or s2 is None) else as_f64(s2).reshape(-1)
        l2a = None if (x2 is None or l2 is None) else as_f64(l2).reshape(-1)
        n2 = n1 if x2a is None else x2a.shape[0]
        out = np.empty((n1, n2))
        self.check(self.lib.nmgp_nonstat_rbf_cov(self.h, ptr(x1), ptr(s1a), ptr(l1a), n1, ptr(x2a), ptr(s2a), ptr(l2a),
                                                 n2, d, ptr(out)))
        return out

    def kron_product(self, a, b):
        a, b = as_f64(a), as_f64(b)
        out = np.empty((a.shape[0] * b.shape[0], a.shape[1] * b.shape[1]))
        self.check(self.lib.nmgp_kron_product(self.h, ptr(a), a.shape[0], a.shape[1], ptr(b), b.shape[0], b.shape[1],
                                              ptr(out)))
        return out

    def kron_mv(self, B, K, y):
        B, K, y = as_f64(B), as_f64(K), as_f64(y).reshape(-1)
        if y.shape[0] != B.shape[1] * K.shape[1]:
            raise NmgpError("kron_mv: y has length %d, expected %d" % (y.shape[0], B.shape[1] * K.shape[1]))
        out = np.empty(B.shape[0] * K.shape[0])
        self.check(self.lib.nmgp_kron_mv(self.h, ptr(B), B.shape[0], B.shape[1], ptr(K), K.shape[0], K.shape[1], ptr(y),
                                         ptr(out)))
        return out

    def mvn_logpdf(self, y, mu, logdet, inv):
        y, inv = as_f64(y).reshape(-1), as_f64(inv)
        mua = None if mu is None else as_f64(mu).reshape(-1)
        out = np.empty(1)
        self.check(self.lib.nmgp_mvn_logpdf(self.h, ptr(y), ptr(mua), float(logdet), ptr(inv), y.shape[0], ptr(out)))
        return float(out[0])

    def mvn_logpdf_kron(self, y, mu, B, K, sigma2, dense=False):
        y, B, K = as_f64(y).reshape(-1), as_f64(B), as_f64(K)
        mua = None if mu is None else as_f64(mu).reshape(-1)
        out = np.empty(1)
        fn = self.lib.nmgp_mvn_logpdf_dense if dense else self.lib.nmgp_mvn_logpdf_kron
        self.check(fn(self.h, ptr(y), ptr(mua), ptr(B), B.shape[0], ptr(K), K.shape[0], float(sigma2), ptr(out)))
        return float(out[0])

    def kron_inv_logdet(self, sigma2, B, K, want_inv=True):
        B, K = as_f64(B), as_f64(K)
        n = B.shape[0] * K.shape[0]
        inv = np.empty((n, n)) if want_inv else None
        ld = np.empty(1)
        self.check(self.lib.nmgp_kron_inv_logdet(self.h, float(sigma2), ptr(B), B.shape[0], ptr(K), K.shape[0], ptr(inv),
                                                 ptr(ld)))
        return inv, float(ld[0])

    def cholesky(self, A, rhs=None, algo=1):
        """Lower Cholesky factor of the SPD matrix A (and L^-1 rhs when rhs is given)."""
        A = as_f64(A)
        n = A.shape[0]
        rhsa = None if rhs is None else as_f64(rhs).reshape(-1)
        out = np.empty((n, n))
        z = np.empty(n) if rhsa is not None else None
        self.check(self.lib.nmgp_cholesky(self.h, ptr(A), n, ptr(rhsa), ptr(out), ptr(z), int(algo)))
        L = np.tril(out.T)           # the library's column-major lower triangle == row-major upper
        return (L, z) if rhsa is not None else L

    # -- prediction -----------------------------------------------------------------------------
    def predict_svc(self, pars, hyper, xs):
        pars, hyper, xs = as_f64(pars).reshape(-1), as_f64(hyper), as_f64(xs).reshape(-1)
        S = xs.shape[0]
        mean, var, Ls = np.empty((S, self.M)), np.empty((S, self.M)), np.empty((S, self.T))
        self.check(self.lib.nmgp_predict_svc(self.h, ptr(pars), ptr(hyper), ptr(xs), S, ptr(mean), ptr(var), ptr(Ls)))
        return mean, var, Ls

    def predict_sep(self, pars, hyper, xs):
        pars, hyper, xs = as_f64(pars).reshape(-1), as_f64(hyper), as_f64(xs).reshape(-1)
        S = xs.shape[0]
        mean, var = np.empty((S, self.M)), np.empty((S, self.M))
        self.check(self.lib.nmgp_predict_sep(self.h, ptr(pars), ptr(hyper), ptr(xs), S, ptr(mean), ptr(var)))
        return mean, var

    def predict_sta(self, pars, xs):
        pars, xs = as_f64(pars).reshape(-1), as_f64(xs).reshape(-1)
        S = xs.shape[0]
        mean, var = np.empty((S, self.M)), np.empty((S, self.M))
        self.check(self.lib.nmgp_predict_sta(self.h, ptr(pars), ptr(xs), S, ptr(mean), ptr(var)))
        return mean, var

    # -- measurement ----------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self.check(self.lib.nmgp_profile_enable(self.h, int(on)))   # True/1 stage timers, 2 + per-launch SYRK timers

    def profile_reset(self):
        self.check(self.lib.nmgp_profile_reset(self.h))

    def profile_read(self):
        ms = np.zeros(len(STAGES))
        cnt = np.zeros(len(STAGES), dtype=np.int64)
        self.check(self.lib.nmgp_profile_read(self.h, ptr(ms), cnt.ctypes.data_as(c_ll_p)))
        return {s: (float(ms[k]), int(cnt[k])) for k, s in enumerate(STAGES)}

    def profile_read_work(self):
        """{stage: (ms, launches, algorithmic flop, algorithmic bytes)}; tracked for "syrk", 0 elsewhere."""
        ms = np.zeros(len(STAGES))
        cnt = np.zeros(len(STAGES), dtype=np.int64)
        work = np.zeros(len(STAGES))
        nbytes = np.zeros(len(STAGES))
        self.check(self.lib.nmgp_profile_read_work(self.h, ptr(ms), cnt.ctypes.data_as(c_ll_p), ptr(work), ptr(nbytes)))
        return {s: (float(ms[k]), int(cnt[k]), float(work[k]), float(nbytes[k])) for k, s in enumerate(STAGES)}

    def last_sep_attempts(self):
        """Jitter retries the last separable / stationary evaluation needed (0: the exact covariance was evaluated)."""
        return int(self.lib.nmgp_last_sep_attempts(self.h))

    def measure_hbm_gbs(self, nbytes=1 << 30, reps=10):
        out = np.empty(1)
        self.check(self.lib.nmgp_measure_hbm_gbs(self.h, int(nbytes), int(reps), ptr(out)))
        return float(out[0])

    def measure_hbm_rates(self, nbytes=1 << 30, reps=10):
        """{copy, read, write} GB/s of the library's flat streaming kernels (the HBM ceilings as measured in this run)."""
        out = np.empty(3)
        self.check(self.lib.nmgp_measure_hbm_rates(self.h, int(nbytes), int(reps), ptr(out)))
        return {"copy": float(out[0]), "read": float(out[1]), "write": float(out[2])}

    def measure_dgemm_tflops(self, n=4096, reps=5):
        out = np.empty(1)
        self.check(self.lib.nmgp_measure_dgemm_tflops(self.h, int(n), int(reps), ptr(out)))
        return float(out[0])


def default_device():
    """GPU ordinal for this process: NMGP_DEVICE, else LOCAL_RANK (one process per GPU), else 0."""
    for k in ("NMGP_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(k)
        if v is not None and v != "":
            return int(v)
    return 0


_default_ctx = {}


def default_context(device=None):
    """Process-wide context used by the `Utility` mirror (created on first use)."""
    if device is None:
        device = default_device()
    ctx = _default_ctx.get(device)
    if ctx is None:
        ctx = Context(device)
        _default_ctx[device] = ctx
    return ctx
