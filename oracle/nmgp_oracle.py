"""CPU oracle for the GP log-posterior hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a NumPy/SciPy float64 restatement, in this repository's own words, of the
arithmetic the reference performs on its log-posterior path.  It exists so that the HIP
library can be checked on machines where the reference itself is not present (the GPU box).

    * It is NOT part of the product.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
      ``cpu_baseline`` leg of ``bench.py`` may import it; the product path under
      ``nonstationary_multivariate_gaussian_process_amd/`` never does and raises when the
      HIP extension is missing.
    * Parity is PINNED: ``tests/golden/make_golden.py`` imports the reference from
      ``/root/reference`` (possible in the build container), evaluates it on RNG-free and
      seeded inputs and commits the input/output vectors as ``tests/golden/*.npz``;
      ``tests/test_oracle_golden.py`` asserts this restatement against every one of them.

Every function cites the reference lines (relative to ``/root/reference``) it restates.
Layouts follow the reference:  ``Y`` is ``[N, M]`` row-major, ``y = Y.T.ravel()`` is output-major,
``uL_vecs`` is location-major ``[N, T]`` with the row-major lower-triangle slot order
``(0,0),(1,0),(1,1),(2,0),...`` and ``exp`` applied on the diagonal slots.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.linalg import cho_solve, cholesky, eigh, solve_triangular
from scipy.special import gammaln

JITTER = 1e-6       # Utility/settings.py:3
PRECISION = 1e-6    # Utility/settings.py:6
LOG_2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------------------
# Utility/utils.py:10-88  -- lower-triangle packing
# --------------------------------------------------------------------------------------
def tril_slots(M):
    """Row-major (row, col) index arrays of the lower triangle; utils.py:66-67."""
    return np.tril_indices(M)


def diag_slots(M):
    """Positions of the diagonal entries inside a packed tril vector; utils.py:12."""
    return np.cumsum(np.arange(1, M + 1)) - 1


def uLvec2Lvec(uL_vec, M):
    """exp() on the diagonal slots, identity elsewhere; utils.py:10-22."""
    out = np.array(uL_vec, dtype=np.float64, copy=True)
    d = diag_slots(M)
    out[d] = np.exp(out[d])
    return out


def Lvec2uLvec(L_vec, M):
    """Inverse of :func:`uLvec2Lvec`; utils.py:24-36."""
    out = np.array(L_vec, dtype=np.float64, copy=True)
    d = diag_slots(M)
    out[d] = np.log(out[d])
    return out


def uLvecs2Lvecs(uL_vecs, N, M):
    """Location-major batch of :func:`uLvec2Lvec`; utils.py:38-46."""
    T = M * (M + 1) // 2
    A = np.array(uL_vecs, dtype=np.float64, copy=True).reshape(N, T)
    d = diag_slots(M)
    A[:, d] = np.exp(A[:, d])
    return A.reshape(-1)


def Lvecs2uLvecs(L_vecs, N, M):
    """utils.py:48-54."""
    T = M * (M + 1) // 2
    A = np.array(L_vecs, dtype=np.float64, copy=True).reshape(N, T)
    d = diag_slots(M)
    A[:, d] = np.log(A[:, d])
    return A.reshape(-1)


def vec2lowtriangle(v, M):
    """Packed tril vector -> dense [M, M]; utils.py:56-74."""
    v = np.asarray(v, dtype=np.float64)
    if M * (M + 1) // 2 != v.shape[0]:
        raise ValueError("check the dimension size!")
    out = np.zeros((M, M))
    r, c = tril_slots(M)
    out[r, c] = v
    return out


def lowtriangle2vec(L, M):
    """utils.py:77-88."""
    r, c = tril_slots(M)
    return np.asarray(L)[r, c]


def _L_stack(uL_vecs, N, M):
    """[N, M, M] stack of the per-location lower-triangular factors (logpos.py:339-341)."""
    T = M * (M + 1) // 2
    Lv = uLvecs2Lvecs(uL_vecs, N, M).reshape(N, T)
    out = np.zeros((N, M, M))
    r, c = tril_slots(M)
    out[:, r, c] = Lv
    return out


# --------------------------------------------------------------------------------------
# Utility/kernels.py
# --------------------------------------------------------------------------------------
def pairwise_distances(x, y=None):
    """Expanded-form squared distance ||x_i||^2 + ||y_j||^2 - 2 x_i.y_j; kernels.py:5-21."""
    x = np.asarray(x, dtype=np.float64)
    xn = (x ** 2).sum(1).reshape(-1, 1)
    if y is None:
        y = x
        yn = xn.reshape(1, -1)
    else:
        y = np.asarray(y, dtype=np.float64)
        yn = (y ** 2).sum(1).reshape(1, -1)
    return xn + yn - 2.0 * (x @ y.T)


def RBF_cov(X1, X2=None, alpha=1.0, beta=1.0):
    """alpha^2 exp(-d^2/2) on inputs scaled by 1/beta, + jitter*I when X2 is None; kernels.py:24-43."""
    X1 = np.asarray(X1, dtype=np.float64)
    if X2 is None:
        base = np.eye(X1.shape[0]) * JITTER
        X2 = X1
    else:
        X2 = np.asarray(X2, dtype=np.float64)
        base = np.zeros((X1.shape[0], X2.shape[0]))
    d = pairwise_distances(X1 / beta, X2 / beta)
    return base + np.exp(-0.5 * d) * alpha ** 2


def Nonstationary_RBF_cov(X1, sigma1=None, ell1=None, X2=None, sigma2=None, ell2=None):
    """Gibbs kernel s_i s_j sqrt(2 l_i l_j/(l_i^2+l_j^2)) exp(-d^2/(l_i^2+l_j^2)); kernels.py:46-73."""
    X1 = np.asarray(X1, dtype=np.float64)
    N1 = X1.shape[0]
    sigma1 = np.ones(N1) if sigma1 is None else np.asarray(sigma1, dtype=np.float64)
    ell1 = np.ones(N1) if ell1 is None else np.asarray(ell1, dtype=np.float64)
    if X2 is None:
        X2, sigma2, ell2 = X1, sigma1, ell1
        base = np.eye(N1) * JITTER
    else:
        X2 = np.asarray(X2, dtype=np.float64)
        sigma2 = np.asarray(sigma2, dtype=np.float64)
        ell2 = np.asarray(ell2, dtype=np.float64)
        base = np.zeros((N1, X2.shape[0]))
    d = pairwise_distances(X1, X2)
    A = (ell1 ** 2)[:, None] + (ell2 ** 2)[None, :]
    B = ell1[:, None] * ell2[None, :]
    C = sigma1[:, None] * sigma2[None, :]
    return base + C * np.sqrt(2.0 * B / A) * np.exp(-d / A)


# --------------------------------------------------------------------------------------
# Utility/kronecker_operation.py
# --------------------------------------------------------------------------------------
def kronecker_product(t1, t2):
    """Dense Kronecker product; kronecker_operation.py:5-22 (== numpy.kron)."""
    t1 = np.asarray(t1, dtype=np.float64)
    t2 = np.asarray(t2, dtype=np.float64)
    a, b = t1.shape
    c, d = t2.shape
    return (t1[:, None, :, None] * t2[None, :, None, :]).reshape(a * c, b * d)


def kronecker_product_diag(d1, d2):
    """kron of two diagonals as a vector; kronecker_operation.py:25-33."""
    return (np.asarray(d1, dtype=np.float64)[:, None] * np.asarray(d2, dtype=np.float64)[None, :]).reshape(-1)


def kron_mv(B, K, y):
    """(B kron K) y without forming the product; kronecker_operation.py:72-85."""
    B = np.asarray(B, dtype=np.float64)
    K = np.asarray(K, dtype=np.float64)
    M2, N2 = B.shape[1], K.shape[1]
    Ymat = np.asarray(y, dtype=np.float64).reshape(M2, N2).T      # [N2, M2]
    A = (K @ Ymat) @ B.T                                         # [N1, M1]
    return np.ascontiguousarray(A.T).reshape(-1)


def kron_inv(sigma2, B, K):
    """(sigma2 I + B kron K)^-1 via the two small eigendecompositions; kronecker_operation.py:36-54."""
    wB, vB = eigh(np.asarray(B, dtype=np.float64))
    wK, vK = eigh(np.asarray(K, dtype=np.float64))
    U = kronecker_product(vB, vK)
    t = kronecker_product_diag(wB, wK)
    return (U * (1.0 / (t + sigma2))[None, :]) @ U.T


def kron_logdet(sigma2, B, K):
    """log det(sigma2 I + B kron K); kronecker_operation.py:57-69."""
    wB = eigh(np.asarray(B, dtype=np.float64), eigvals_only=True)
    wK = eigh(np.asarray(K, dtype=np.float64), eigvals_only=True)
    return float(np.log(kronecker_product_diag(wB, wK) + sigma2).sum())


# --------------------------------------------------------------------------------------
# Utility/distributions.py
# --------------------------------------------------------------------------------------
def multivariate_normal_logpdf(y, mu, logdetSigma, invSigma):
    """-0.5 logdet - 0.5 r' S^-1 r; the 2 pi term is dropped (distributions.py:21 is overwritten by :22)."""
    r = np.asarray(y, dtype=np.float64) - mu
    return float(-0.5 * logdetSigma - 0.5 * (r @ (np.asarray(invSigma) @ r)))


def multivariate_normal_logpdf0(y, mu, B, K, sigma2):
    """Same quantity for B kron K + sigma2 I in the joint eigenbasis; distributions.py:26-52."""
    wB, vB = eigh(np.asarray(B, dtype=np.float64))
    wK, vK = eigh(np.asarray(K, dtype=np.float64))
    a = kron_mv(vB.T, vK.T, np.asarray(y, dtype=np.float64) - mu)
    t = kronecker_product_diag(wB, wK)
    return float(-0.5 * np.log(t + sigma2).sum() - 0.5 * np.sum(a * a / (t + sigma2)))


def multivariate_normal_logpdf1(y, mu, B, K, sigma2, jitter_B, jitter_K):
    """The robust variant, distributions.py:55-96: logpdf0 after `precision * u` has been added to the diagonals of B
    (:66) and K (:69).  u are the reference's ``torch.rand`` draws (B first, then K); the oracle takes them as
    arguments (jitter_B [M], jitter_K [N], uniform in [0, 1)) so that a seeded reference call can be reproduced."""
    Bj = np.asarray(B, dtype=np.float64) + np.diag(np.asarray(jitter_B, dtype=np.float64) * PRECISION)
    Kj = np.asarray(K, dtype=np.float64) + np.diag(np.asarray(jitter_K, dtype=np.float64) * PRECISION)
    return multivariate_normal_logpdf0(y, mu, Bj, Kj, sigma2)


def multivariate_normal_logpdf2(y, mu, B, K, sigma2):
    """Dense evaluation of the same density; distributions.py:99-113."""
    S = kronecker_product(B, K) + sigma2 * np.eye(B.shape[0] * K.shape[0])
    sign, ld = np.linalg.slogdet(S)
    return multivariate_normal_logpdf(y, mu, ld, np.linalg.inv(S))


def inverse_gamma_logpdf_u(x, alpha=1.0, beta=1.0):
    """distributions.py:116-124."""
    return (-alpha - 1.0) * math.log(x) - beta / x


def inverse_gamma_logpdf(x, alpha=1.0, beta=1.0):
    """Normalised inverse-gamma log density; distributions.py:126-134."""
    return (-alpha - 1.0) * math.log(x) - beta / x + alpha * math.log(beta) - float(gammaln(alpha))


def gamma_logpdf(x, alpha=1.0, beta=1.0):
    """distributions.py:136-137."""
    return (alpha - 1.0) * math.log(x) - beta * x + alpha * math.log(beta) - float(gammaln(alpha))


def mvn_log_prob(v, mean, cov):
    """torch.distributions.MultivariateNormal(mean, cov).log_prob(v): Cholesky + trsv, with 2 pi
    (as used at logpos.py:274,279,358,365).  Returns (log_prob, cov^-1 (v - mean))."""
    Lc = cholesky(cov, lower=True)
    r = np.asarray(v, dtype=np.float64) - mean
    z = solve_triangular(Lc, r, lower=True)
    lp = -0.5 * (z @ z) - np.log(np.diag(Lc)).sum() - 0.5 * r.shape[0] * LOG_2PI
    return float(lp), solve_triangular(Lc, z, lower=True, trans="T")


def normal_log_prob(v, mean, sd):
    """torch.distributions.Normal(mean, sd).log_prob(v) as called at logpos.py:283,446,450.

    Quirk kept for parity: the reference passes ``mean``/``sd`` as Python numbers, which torch turns into
    float32 tensors (default dtype), so the location, the variance sd^2 and log(sd) are float32-rounded
    before they meet the float64 value (e.g. log(10) becomes 2.3025851249694824)."""
    v = np.asarray(v, dtype=np.float64)
    m32 = np.float32(mean)
    s32 = np.float32(sd)
    var = float(s32 * s32)
    log_sd = float(np.log(s32))
    return -((v - float(m32)) ** 2) / (2.0 * var) - log_sd - math.log(math.sqrt(2.0 * math.pi))


# --------------------------------------------------------------------------------------
# Utility/logpos.py -- parameter vector slicing (17-57)
# --------------------------------------------------------------------------------------
def vec2pars(pars, N, M):
    T = M * (M + 1) // 2
    return pars[:N], pars[N:2 * N], pars[2 * N:2 * N + T], pars[-1]


def vec2pars_SVC(pars, N, M):
    T = M * (M + 1) // 2
    return pars[:N], pars[N:N + N * T], pars[-1]


def vec2pars_S(pars, M):
    T = M * (M + 1) // 2
    return pars[0], pars[1], pars[2:2 + T], pars[-1]


# --------------------------------------------------------------------------------------
# Nonseparable ("SVC") objective: logpos.py:299-380
# --------------------------------------------------------------------------------------
def svc_covariance(tilde_l, uL_vecs, tilde_sigma2_err, x, M, add_noise=True):
    """Dense output-major MN x MN covariance of the nonseparable model (logpos.py:339-353):
    S[mN+i, m'N+j] = (K_x[i,j]) (L_i L_j^T)[m,m'] (+ sigma2 on the diagonal), K_x carrying the 1e-6 jitter."""
    x = np.asarray(x, dtype=np.float64)
    N = x.shape[0]
    Ls = _L_stack(uL_vecs, N, M)
    Kx = Nonstationary_RBF_cov(x.reshape(-1, 1), ell1=np.exp(np.asarray(tilde_l, dtype=np.float64)))
    Lcat = Ls.reshape(N * M, M)                       # location-major rows (logpos.py:117)
    Ki = Lcat @ Lcat.T
    order = np.arange(N * M).reshape(N, M).T.reshape(-1)
    Ki = Ki[:, order][order]                          # -> output-major (logpos.py:347-348)
    S = kronecker_product(np.ones((M, M)), Kx) * Ki   # logpos.py:349
    if add_noise:
        S = S + math.exp(float(tilde_sigma2_err)) * np.eye(N * M)
    return S


def logpos_SVC(tilde_l, uL_vecs, tilde_sigma2_err, Y, x, mu_tilde_l, alpha_tilde_l, beta_tilde_l,
               mu_L, alpha_L, beta_L, a, b, verbose=False, Prior=True, formulation="cholesky", grad=False):
    """Log joint posterior of the nonseparable model; logpos.py:326-380.

    ``formulation="reference"`` follows the reference literally (dense inverse + logdet, :352-354);
    ``"cholesky"`` is the algebraically identical factorisation the HIP path uses.
    With ``grad=True`` also returns d(res)/d(pars) (analytic adjoint, validated against the
    reference's autograd by the golden fixtures)."""
    Y = np.asarray(Y, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    tilde_l = np.asarray(tilde_l, dtype=np.float64)
    uL_vecs = np.asarray(uL_vecs, dtype=np.float64)
    tse = float(tilde_sigma2_err)
    N, M = Y.shape
    T = M * (M + 1) // 2
    n = N * M
    y = Y.T.reshape(-1)
    sigma2 = math.exp(tse)
    S = svc_covariance(tilde_l, uL_vecs, tse, x, M)
    if formulation == "reference":
        invS = np.linalg.inv(S)
        _, logdet = np.linalg.slogdet(S)
        alpha = invS @ y
        loglik = -0.5 * logdet - 0.5 * (y @ alpha)
    else:
        C = cholesky(S, lower=True)
        z = solve_triangular(C, y, lower=True)
        loglik = -np.log(np.diag(C)).sum() - 0.5 * (z @ z)
        invS = None
        alpha = None
    X1 = x.reshape(-1, 1)
    Sig_l = RBF_cov(X1, alpha=alpha_tilde_l, beta=beta_tilde_l)
    lp_l, g_l = mvn_log_prob(tilde_l, mu_tilde_l * np.ones(N), Sig_l)
    Sig_L = RBF_cov(X1, alpha=alpha_L, beta=beta_L)
    U = uL_vecs.reshape(N, T)
    lp_uL = 0.0
    g_uL = np.zeros((N, T))
    for t in range(T):                                 # logpos.py:363-365: stride-T gather per slot
        lp_t, g_t = mvn_log_prob(U[:, t], mu_L * np.ones(N), Sig_L)
        lp_uL += lp_t
        g_uL[:, t] = g_t
    lp_s2 = inverse_gamma_logpdf(sigma2, alpha=a, beta=b)
    res = loglik
    if Prior:
        res = res + lp_l + lp_uL + lp_s2 + tse          # :359,367,373,376 (Jacobian of exp)
    out = (res, float(loglik), lp_l, lp_uL, lp_s2) if verbose else res
    if not grad:
        return out
    # ---- analytic gradient of res wrt [tilde_l | uL_vecs | tilde_sigma2_err] ----
    if invS is None:
        invS = cho_solve((C, True), np.eye(n))
        alpha = cho_solve((C, True), y)
    G = 0.5 * (np.outer(alpha, alpha) - invS)          # d loglik = <G, dS>
    G4 = G.reshape(M, N, M, N)
    Ls = _L_stack(uL_vecs, N, M)
    ell = np.exp(tilde_l)
    D = pairwise_distances(X1)
    A = (ell ** 2)[:, None] + (ell ** 2)[None, :]
    K0 = np.sqrt(2.0 * np.outer(ell, ell) / A) * np.exp(-D / A)
    Kx = K0 + JITTER * np.eye(N)
    # H_ij = <G_ij, L_i L_j^T>
    GL = np.einsum("minj,jnr->imr" + "", G4 * Kx[None, :, None, :], Ls, optimize=True)  # sum_j Kx_ij G_ij L_j
    dL = 2.0 * GL                                        # [N, M, M]
    H = np.einsum("minj,imr,jnr->ij", G4, Ls, Ls, optimize=True)
    e2 = (ell ** 2)[:, None]
    dlogK = 0.5 - e2 / A + 2.0 * e2 * D / (A * A)        # l_i d log K0_ij / d l_i
    W = 2.0 * H * K0 * dlogK
    np.fill_diagonal(W, 0.0)
    g_tl = W.sum(1)
    r, c = tril_slots(M)
    g_Lv = dL[:, r, c]                                   # [N, T]
    dslots = diag_slots(M)
    Lv = uLvecs2Lvecs(uL_vecs, N, M).reshape(N, T)
    g_Lv[:, dslots] *= Lv[:, dslots]
    g_s = sigma2 * np.trace(G)
    if Prior:
        g_tl = g_tl - g_l
        g_Lv = g_Lv - g_uL
        g_s = g_s + (-a - 1.0) + b / sigma2 + 1.0
    gvec = np.concatenate([g_tl, g_Lv.reshape(-1), [g_s]])
    return out, gvec


def nlogpos_obj_SVC(pars, Y, x, mu_tilde_l=0.0, alpha_tilde_l=5.0, beta_tilde_l=1.0, mu_L=0.0, alpha_L=5.0,
                    beta_L=1.0, a=1, b=1, verbose=False, Prior=True, formulation="cholesky", grad=False):
    """Negative log posterior on the flat parameter vector; logpos.py:299-323."""
    pars = np.asarray(pars, dtype=np.float64)
    N, M = np.asarray(Y).shape
    tl, uL, tse = vec2pars_SVC(pars, N, M)
    r = logpos_SVC(tl, uL, tse, Y, x, mu_tilde_l, alpha_tilde_l, beta_tilde_l, mu_L, alpha_L, beta_L, a, b,
                   verbose, Prior, formulation, grad)
    g = None
    if grad:
        r, g = r
        g = -g
    r = (-r[0],) + tuple(r[1:]) if verbose else -r
    return (r, g) if grad else r


# --------------------------------------------------------------------------------------
# Separable and stationary objectives: logpos.py:216-296, 383-462
# --------------------------------------------------------------------------------------
def _kron_loglik_and_adjoints(y, B, Kx, sigma2, want_grad):
    """loglik for S = B kron Kx + sigma2 I in the joint eigenbasis (distributions.py:26-52) and, when asked,
    the adjoints dloglik/dB [M,M], dloglik/dKx [N,N], dloglik/dsigma2."""
    M, N = B.shape[0], Kx.shape[0]
    wB, vB = eigh(B)
    wK, vK = eigh(Kx)
    a = kron_mv(vB.T, vK.T, y)
    t = kronecker_product_diag(wB, wK)
    w = 1.0 / (t + sigma2)
    loglik = float(-0.5 * np.log(t + sigma2).sum() - 0.5 * np.sum(a * a * w))
    if not want_grad:
        return loglik, None
    At = (a * w).reshape(M, N)          # alpha in the eigenbasis, [p, q]
    Wm = w.reshape(M, N)
    # dloglik/dKx = 1/2 V_K [ sum_p wB_p (At_p At_p^T - diag(W_p)) ] V_K^T
    Pk = vK @ (At.T * np.sqrt(np.maximum(wB, 0.0))[None, :])     # only valid for wB>=0; handle sign below
    core = (At.T * wB[None, :]) @ At                              # [N, N] in eigenbasis (rank M)
    core[np.diag_indices(N)] -= (Wm * wB[:, None]).sum(0)
    dK = 0.5 * (vK @ core @ vK.T)
    coreB = (At * wK[None, :]) @ At.T
    coreB[np.diag_indices(M)] -= (Wm * wK[None, :]).sum(1)
    dB = 0.5 * (vB @ coreB @ vB.T)
    ds = 0.5 * (np.sum(At * At) - w.sum())
    del Pk
    return loglik, (dB, dK, ds)


def _gibbs_adjoint(dK, x, ell, sig):
    """Chain dloglik/dKx (symmetric) to (tilde_l, tilde_sigma) for Kx = sig sig^T * K0(ell) + jitter I."""
    X1 = x.reshape(-1, 1)
    D = pairwise_distances(X1)
    A = (ell ** 2)[:, None] + (ell ** 2)[None, :]
    K0 = np.sqrt(2.0 * np.outer(ell, ell) / A) * np.exp(-D / A)
    Ks = np.outer(sig, sig) * K0
    e2 = (ell ** 2)[:, None]
    dlogK = 0.5 - e2 / A + 2.0 * e2 * D / (A * A)
    Wl = 2.0 * dK * Ks * dlogK
    np.fill_diagonal(Wl, 0.0)
    g_tl = Wl.sum(1)
    g_ts = 2.0 * (dK * Ks).sum(1)        # d Ks_ij / d tilde_sigma_i = Ks_ij (and the symmetric partner)
    return g_tl, g_ts


def _B_adjoint_to_uL(dB, uL_vec, M):
    L = vec2lowtriangle(uLvec2Lvec(uL_vec, M), M)
    dL = 2.0 * dB @ L
    r, c = tril_slots(M)
    g = dL[r, c]
    d = diag_slots(M)
    g[d] *= L[np.arange(M), np.arange(M)]
    return g


def logpos(tilde_l, tilde_sigma, uL_vec, tilde_sigma2_err, Y, x, mu_tilde_l, alpha_tilde_l, beta_tilde_l,
           mu_tilde_sigma, alpha_tilde_sigma, beta_tilde_sigma, a, b, c, verbose=False, Prior=True, grad=False):
    """Log joint posterior of the separable nonstationary model; logpos.py:237-296.
    The NaN-retry with random jitter (:267-268) is nondeterministic in the reference and is not restated:
    the first attempt is returned as is (NaN included)."""
    Y = np.asarray(Y, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    tilde_l = np.asarray(tilde_l, dtype=np.float64)
    tilde_sigma = np.asarray(tilde_sigma, dtype=np.float64)
    uL_vec = np.asarray(uL_vec, dtype=np.float64)
    tse = float(tilde_sigma2_err)
    N, M = Y.shape
    y = Y.T.reshape(-1)
    L = vec2lowtriangle(uLvec2Lvec(uL_vec, M), M)
    B = L @ L.T
    ell, sig = np.exp(tilde_l), np.exp(tilde_sigma)
    sigma2 = math.exp(tse)
    X1 = x.reshape(-1, 1)
    Kx = Nonstationary_RBF_cov(X1, sigma1=sig, ell1=ell)
    loglik, adj = _kron_loglik_and_adjoints(y, B, Kx, sigma2, grad)
    lp_l, g_l = mvn_log_prob(tilde_l, mu_tilde_l * np.ones(N), RBF_cov(X1, alpha=alpha_tilde_l, beta=beta_tilde_l))
    lp_s, g_s = mvn_log_prob(tilde_sigma, mu_tilde_sigma * np.ones(N),
                             RBF_cov(X1, alpha=alpha_tilde_sigma, beta=beta_tilde_sigma))
    lp_uL = float(normal_log_prob(uL_vec, 0.0, c).sum())
    lp_s2 = inverse_gamma_logpdf(sigma2, alpha=a, beta=b)
    res = loglik
    if Prior:
        res = res + lp_l + lp_s + lp_uL + lp_s2 + tse
    out = (res, loglik, lp_l, lp_s, lp_uL, lp_s2) if verbose else res
    if not grad:
        return out
    dB, dK, ds = adj
    g_tl, g_ts = _gibbs_adjoint(dK, x, ell, sig)
    g_uL = _B_adjoint_to_uL(dB, uL_vec, M)
    g_e = sigma2 * ds
    if Prior:
        g_tl = g_tl - g_l
        g_ts = g_ts - g_s
        g_uL = g_uL - uL_vec / float(np.float32(c) * np.float32(c))
        g_e = g_e + (-a - 1.0) + b / sigma2 + 1.0
    return out, np.concatenate([g_tl, g_ts, g_uL, [g_e]])


def nlogpos_obj(pars, Y, x, mu_tilde_l=0.0, alpha_tilde_l=1.0, beta_tilde_l=1.0, mu_tilde_sigma=0.0,
                alpha_tilde_sigma=1.0, beta_tilde_sigma=1.0, a=1, b=1, c=10, verbose=False, Prior=True, grad=False):
    """logpos.py:216-234."""
    pars = np.asarray(pars, dtype=np.float64)
    N, M = np.asarray(Y).shape
    tl, ts, uL, tse = vec2pars(pars, N, M)
    r = logpos(tl, ts, uL, tse, Y, x, mu_tilde_l, alpha_tilde_l, beta_tilde_l, mu_tilde_sigma, alpha_tilde_sigma,
               beta_tilde_sigma, a, b, c, verbose, Prior, grad)
    g = None
    if grad:
        r, g = r
        g = -g
    r = (-r[0],) + tuple(r[1:]) if verbose else -r
    return (r, g) if grad else r


def logpos_S(tilde_l, tilde_sigma, uL_vec, tilde_sigma2_err, Y, x, mu_tilde_l, sigma_tilde_l, a, b, c,
             verbose=False, Prior=True, grad=False):
    """Log joint posterior of the stationary (LMC) model; logpos.py:405-462.  As in the reference the
    prior terms only exist when ``Prior`` is true (verbose with Prior=False is an error there too)."""
    Y = np.asarray(Y, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    uL_vec = np.asarray(uL_vec, dtype=np.float64)
    tl, ts, tse = float(tilde_l), float(tilde_sigma), float(tilde_sigma2_err)
    N, M = Y.shape
    y = Y.T.reshape(-1)
    L = vec2lowtriangle(uLvec2Lvec(uL_vec, M), M)
    B = L @ L.T
    ell = np.exp(tl * np.ones(N))
    sig = np.exp(ts * np.ones(N))
    sigma2 = math.exp(tse)
    Kx = Nonstationary_RBF_cov(x.reshape(-1, 1), sigma1=sig, ell1=ell)
    loglik, adj = _kron_loglik_and_adjoints(y, B, Kx, sigma2, grad)
    res = loglik
    if Prior:
        lp_l = float(normal_log_prob(tl, mu_tilde_l, sigma_tilde_l))
        lp_uL = float(normal_log_prob(uL_vec, 0.0, c).sum())
        lp_s2 = inverse_gamma_logpdf(sigma2, alpha=a, beta=b)
        res = res + lp_l + lp_uL + lp_s2 + tse
    elif verbose:
        raise UnboundLocalError("log_prior_tilde_l is undefined when Prior is False (logpos.py:459)")
    out = (res, loglik, lp_l, lp_uL, lp_s2) if verbose else res
    if not grad:
        return out
    dB, dK, ds = adj
    # stationary: every l_i equals exp(tl); summing the per-location adjoints gives the scalar one,
    # but the diagonal of K0 is constant in l only when l_i and l_j move together -> recompute directly.
    D = pairwise_distances(x.reshape(-1, 1))
    l2 = math.exp(2.0 * tl)
    Ks = math.exp(2.0 * ts) * np.exp(-D / (2.0 * l2))          # sqrt(2 l l/(2 l^2)) = 1
    g_tl = float((dK * Ks * (D / l2)).sum())                    # d/dtl exp(-D/(2 e^{2tl})) = D/l2 * (.)
    g_ts = float(2.0 * (dK * Ks).sum())
    g_uL = _B_adjoint_to_uL(dB, uL_vec, M)
    g_e = sigma2 * ds
    if Prior:
        g_tl = g_tl - (tl - float(np.float32(mu_tilde_l))) / float(np.float32(sigma_tilde_l) ** 2)
        g_uL = g_uL - uL_vec / float(np.float32(c) * np.float32(c))
        g_e = g_e + (-a - 1.0) + b / sigma2 + 1.0
    return out, np.concatenate([[g_tl], [g_ts], g_uL, [g_e]])


def nlogpos_obj_S(pars, Y, x, mu_tilde_l, sigma_tilde_l, a=1, b=1, c=10, verbose=False, Prior=True, grad=False):
    """logpos.py:383-402."""
    pars = np.asarray(pars, dtype=np.float64)
    N, M = np.asarray(Y).shape
    tl, ts, uL, tse = vec2pars_S(pars, M)
    r = logpos_S(tl, ts, uL, tse, Y, x, mu_tilde_l, sigma_tilde_l, a, b, c, verbose, Prior, grad)
    g = None
    if grad:
        r, g = r
        g = -g
    r = (-r[0],) + tuple(r[1:]) if verbose else -r
    return (r, g) if grad else r


# --------------------------------------------------------------------------------------
# Deterministic prediction: Utility/prediction.py:912-988, 337-408, 1566-1638
# --------------------------------------------------------------------------------------
def _gp_regress(x, xs, v, mu, alpha, beta):
    """Posterior mean of a GP-with-RBF-prior curve at the points xs (prediction.py:926-941)."""
    X1 = x.reshape(-1, 1)
    Sig = RBF_cov(X1, alpha=alpha, beta=beta)
    k = RBF_cov(X1, xs.reshape(-1, 1), alpha=alpha, beta=beta)          # [N, S]
    proj = np.linalg.solve(Sig, k)                                       # torch.solve (LU)
    return mu + proj.T @ (v - mu)


def predmap_inhomogeneous(tilde_l, uL_vecs, tilde_sigma2_err, Y, x, x_star, mu_tilde_l, alpha_tilde_l,
                          beta_tilde_l, mu_L, alpha_L, beta_L):
    """Predictive mean/variance of the nonseparable model at every point of ``x_star`` (vectorised
    restatement of prediction.py:912-988).  Returns (percentiles [S,3,M], L_star [S,T], mean [S,M], var [S,M])."""
    Y = np.asarray(Y, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    xs = np.atleast_1d(np.asarray(x_star, dtype=np.float64))
    tilde_l = np.asarray(tilde_l, dtype=np.float64)
    uL_vecs = np.asarray(uL_vecs, dtype=np.float64)
    N, M = Y.shape
    T = M * (M + 1) // 2
    S = xs.shape[0]
    y = Y.T.reshape(-1)
    sigma2 = math.exp(float(tilde_sigma2_err))
    tl_star = _gp_regress(x, xs, tilde_l, mu_tilde_l, alpha_tilde_l, beta_tilde_l)          # [S]
    U = uL_vecs.reshape(N, T)
    uL_star = np.stack([_gp_regress(x, xs, U[:, t], mu_L, alpha_L, beta_L) for t in range(T)], 1)   # [S, T]
    Lv_star = uL_star.copy()
    d = diag_slots(M)
    Lv_star[:, d] = np.exp(Lv_star[:, d])
    Sfull = svc_covariance(tilde_l, uL_vecs, tilde_sigma2_err, x, M)
    C = cholesky(Sfull, lower=True)
    alpha = cho_solve((C, True), y)
    Ls = _L_stack(uL_vecs, N, M)
    ell = np.exp(tilde_l)
    r, c = tril_slots(M)
    pct = np.zeros((S, 3, M))
    mean = np.zeros((S, M))
    var = np.zeros((S, M))
    for s in range(S):
        ls = math.exp(tl_star[s])
        kx = Nonstationary_RBF_cov(x.reshape(-1, 1), sigma1=np.ones(N), ell1=ell, X2=xs[s].reshape(1, 1),
                                   sigma2=np.ones(1), ell2=np.array([ls]))[:, 0]              # [N]
        Lstar = np.zeros((M, M))
        Lstar[r, c] = Lv_star[s]
        # k_f[(m,i), m'] = kx_i (L_i Lstar^T)[m, m']
        kf = np.einsum("i,imr,nr->min", kx, Ls, Lstar).reshape(M * N, M)
        mean[s] = kf.T @ alpha
        V = solve_triangular(C, kf, lower=True)
        kss = Nonstationary_RBF_cov(xs[s].reshape(1, 1), sigma1=np.ones(1), ell1=np.array([ls]))[0, 0]
        Sf = kss * (Lstar @ Lstar.T) - V.T @ V
        s2 = np.diag(Sf) + sigma2
        s2 = np.where(s2 <= 0, PRECISION, s2)
        var[s] = s2
        sd = np.sqrt(s2)
        pct[s] = np.stack([mean[s] - 1.96 * sd, mean[s], mean[s] + 1.96 * sd])
    return pct, Lv_star, mean, var


def predmap_separable(tilde_l, tilde_sigma, uL_vec, tilde_sigma2_err, Y, x, x_star, mu_tilde_l, alpha_tilde_l,
                      beta_tilde_l, mu_tilde_sigma, alpha_tilde_sigma, beta_tilde_sigma):
    """Predictive mean/variance of the separable model (prediction.py:337-408), vectorised over x_star."""
    Y = np.asarray(Y, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    xs = np.atleast_1d(np.asarray(x_star, dtype=np.float64))
    N, M = Y.shape
    y = Y.T.reshape(-1)
    sigma2 = math.exp(float(tilde_sigma2_err))
    tl_star = _gp_regress(x, xs, np.asarray(tilde_l), mu_tilde_l, alpha_tilde_l, beta_tilde_l)
    ts_star = _gp_regress(x, xs, np.asarray(tilde_sigma), mu_tilde_sigma, alpha_tilde_sigma, beta_tilde_sigma)
    L = vec2lowtriangle(uLvec2Lvec(uL_vec, M), M)
    B = L @ L.T
    ell, sig = np.exp(np.asarray(tilde_l)), np.exp(np.asarray(tilde_sigma))
    Kx = Nonstationary_RBF_cov(x.reshape(-1, 1), sigma1=sig, ell1=ell)
    Sfull = kronecker_product(B, Kx) + sigma2 * np.eye(N * M)
    C = cholesky(Sfull, lower=True)
    alpha = cho_solve((C, True), y)
    S = xs.shape[0]
    mean = np.zeros((S, M))
    var = np.zeros((S, M))
    pct = np.zeros((S, 3, M))
    for s in range(S):
        ls, ss = math.exp(tl_star[s]), math.exp(ts_star[s])
        kx = Nonstationary_RBF_cov(x.reshape(-1, 1), sigma1=sig, ell1=ell, X2=xs[s].reshape(1, 1),
                                   sigma2=np.array([ss]), ell2=np.array([ls]))                 # [N,1]
        kf = kronecker_product(B, kx)                                                            # [MN, M]
        mean[s] = kf.T @ alpha
        V = solve_triangular(C, kf, lower=True)
        kss = Nonstationary_RBF_cov(xs[s].reshape(1, 1), sigma1=np.array([ss]), ell1=np.array([ls]))[0, 0]
        s2 = np.diag(kss * B - V.T @ V) + sigma2
        s2 = np.where(s2 <= 0, PRECISION, s2)
        var[s] = s2
        sd = np.sqrt(s2)
        pct[s] = np.stack([mean[s] - 1.96 * sd, mean[s], mean[s] + 1.96 * sd])
    return pct, mean, var


def predmap_stationary(tilde_l, tilde_sigma, uL_vec, tilde_sigma2_err, Y, x, x_star):
    """Predictive mean/std of the stationary model at test inputs (prediction.py:1566-1638)."""
    Y = np.asarray(Y, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    xs = np.atleast_1d(np.asarray(x_star, dtype=np.float64))
    N, M = Y.shape
    y = Y.T.reshape(-1)
    sigma2 = math.exp(float(tilde_sigma2_err))
    L = vec2lowtriangle(uLvec2Lvec(uL_vec, M), M)
    B = L @ L.T
    ell = math.exp(float(tilde_l))
    sig = math.exp(float(tilde_sigma))
    Kx = RBF_cov(x.reshape(-1, 1), alpha=sig, beta=ell)                 # prediction.py:1587,1623 (plain RBF)
    C = cholesky(kronecker_product(B, Kx) + sigma2 * np.eye(N * M), lower=True)
    alpha = cho_solve((C, True), y)
    S = xs.shape[0]
    mean = np.zeros((S, M))
    var = np.zeros((S, M))
    for s in range(S):
        kx = RBF_cov(x.reshape(-1, 1), xs[s].reshape(1, 1), alpha=sig, beta=ell)
        kf = kronecker_product(B, kx)
        mean[s] = kf.T @ alpha
        V = solve_triangular(C, kf, lower=True)
        s2 = sig * sig * np.diag(B) - np.einsum("am,am->m", V, V) + sigma2
        var[s] = np.where(s2 < 0, PRECISION, s2)
    return mean, var
