"""The mirrored primitives (kernels.*, kron_*, multivariate_normal_logpdf*) are differentiable like the reference's plain torch ops
(kernels.py:46-73, kronecker_operation.py:5-85, distributions.py:10-52): forward on the MI355X, backward through the host
restatement of the same expression (Utility/_bridge.py).  CPU part: the backward plumbing and the restatements themselves;
GPU part: value = device result, gradient = autograd of the expression, composed the way logpos.py:339-354 composes them."""
import numpy as np
import pytest
import torch

D = torch.float64


def _spd(n, seed):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(n, n, dtype=D, generator=g)
    return A @ A.T / n + torch.eye(n, dtype=D)


def test_host_backward_plumbing_and_restatements_cpu():
    from nonstationary_multivariate_gaussian_process_amd.Utility import _bridge
    from nonstationary_multivariate_gaussian_process_amd.Utility import kernels as k, kronecker_operation as ko, distributions as dist
    g = torch.Generator().manual_seed(1)
    M, N = 3, 5
    B, K = _spd(M, 2), _spd(N, 3)
    y, mu = torch.randn(M * N, dtype=D, generator=g), torch.randn(M * N, dtype=D, generator=g)
    s2 = torch.tensor(0.3, dtype=D)
    S = torch.kron(B, K) + s2 * torch.eye(M * N, dtype=D)
    # the restatements against dense algebra
    assert abs(float(dist._host_mvn_kron(y, mu, B, K, s2) + 0.5 * torch.logdet(S) + 0.5 * (y - mu) @ torch.linalg.solve(S, y - mu))) < 1e-12
    assert float((ko._host_kron_mv(B, K, y) - torch.kron(B, K) @ y).abs().max()) < 1e-12
    assert float((ko._host_kron_inv(s2, B, K) - torch.linalg.inv(S)).abs().max()) < 1e-12
    assert abs(float(ko._host_kron_logdet(s2, B, K) - torch.logdet(S))) < 1e-12
    X1, X2 = torch.randn(4, 2, dtype=D, generator=g), torch.randn(3, 2, dtype=D, generator=g)
    assert float((k._host_sqdist(X1, X2) - torch.cdist(X1, X2) ** 2).abs().max()) < 1e-12
    e1, e2 = torch.rand(4, dtype=D, generator=g) + 0.5, torch.rand(3, dtype=D, generator=g) + 0.5
    G = k._host_gibbs(X1, None, e1, X2, torch.ones(3, dtype=D), e2)
    A = e1[:, None] ** 2 + e2[None, :] ** 2
    assert float((G - torch.sqrt(2 * e1[:, None] * e2[None, :] / A) * torch.exp(-torch.cdist(X1, X2) ** 2 / A)).abs().max()) < 1e-12
    Gs = k._host_gibbs(X1, None, e1, None, None, None)
    assert abs(float(Gs[0, 0]) - (1.0 + 1e-6)) < 1e-14                      # jitter on the diagonal when X2 is None (kernels.py:63)
    # the plumbing: a value computed elsewhere (here: the same expression, detached) becomes differentiable w.r.t. exactly the
    # inputs that require grad; non-tensor and None inputs pass through
    e1g, X1g = e1.clone().requires_grad_(True), X1.clone().requires_grad_(True)
    val = k._host_gibbs(X1, None, e1, X2, torch.ones(3, dtype=D), e2).detach()
    out = _bridge.with_host_backward(val, k._host_gibbs, X1g, None, e1g, X2, torch.ones(3, dtype=D), e2)
    w = torch.randn(4, 3, dtype=D, generator=g)
    (out * w).sum().backward()
    e1r, X1r = e1.clone().requires_grad_(True), X1.clone().requires_grad_(True)
    (k._host_gibbs(X1r, None, e1r, X2, torch.ones(3, dtype=D), e2) * w).sum().backward()
    assert torch.allclose(e1g.grad, e1r.grad, rtol=1e-13, atol=1e-15) and torch.allclose(X1g.grad, X1r.grad, rtol=1e-13, atol=1e-15)
    # nothing requires grad (or grad mode is off): the value comes back as is
    assert _bridge.with_host_backward(val, k._host_gibbs, X1, None, e1, X2, None, e2) is val
    with torch.no_grad():
        assert _bridge.with_host_backward(val, k._host_gibbs, X1g, None, e1g, X2, None, e2) is val


@pytest.mark.gpu
def test_primitives_are_differentiable_on_the_gpu_path():
    from nonstationary_multivariate_gaussian_process_amd.Utility import kernels as k, kronecker_operation as ko, distributions as dist
    g = torch.Generator().manual_seed(5)
    N, M = 40, 2
    x = torch.sort(torch.rand(N, dtype=D, generator=g)).values.view(-1, 1)
    tl = (0.3 * torch.randn(N, dtype=D, generator=g) - 2.0).requires_grad_(True)
    uB = torch.randn(M, M, dtype=D, generator=g)
    Bp = uB.clone().requires_grad_(True)
    y = torch.randn(M * N, dtype=D, generator=g)
    s2 = torch.tensor(0.05, dtype=D, requires_grad=True)

    def objective(mod_k, mod_d):
        K = mod_k(x, ell1=torch.exp(tl))
        B = Bp @ Bp.T + torch.eye(M, dtype=D)
        return mod_d(y, torch.zeros(M * N, dtype=D), B, K, s2)
    val = objective(k.Nonstationary_RBF_cov, dist.multivariate_normal_logpdf0)
    gd = torch.autograd.grad(val, [tl, Bp, s2])
    ref = objective(lambda X1, ell1: k._host_gibbs(X1, None, ell1, None, None, None), dist._host_mvn_kron)
    gr = torch.autograd.grad(ref, [tl, Bp, s2])
    assert abs(float(val - ref)) < 1e-9 * abs(float(ref))
    for a, b in zip(gd, gr):
        assert torch.allclose(a, b, rtol=1e-7, atol=1e-9)
    # each primitive on its own
    Kx = k._host_gibbs(x, None, torch.exp(tl.detach()), None, None, None)
    Bm = (uB @ uB.T + torch.eye(M, dtype=D)).requires_grad_(True)
    Km = Kx.clone().requires_grad_(True)
    for fn, hfn, args in ((ko.kron_mv, ko._host_kron_mv, (Bm, Km, y)), (ko.kron_logdet, ko._host_kron_logdet, (s2, Bm, Km)),
                          (ko.kron_inv, ko._host_kron_inv, (s2, Bm, Km)), (ko.kronecker_product, torch.kron, (Bm, Km)),
                          (k.RBF_cov, k._host_rbf, (x.clone().requires_grad_(True), None, 1.3, 0.7)),
                          (k.pairwise_distances, k._host_sqdist, (x.clone().requires_grad_(True), None))):
        out, href = fn(*args), hfn(*args)
        assert torch.allclose(out, href, rtol=1e-9, atol=1e-11), fn.__name__
        wts = torch.randn(out.shape, dtype=D, generator=g) if out.dim() else torch.tensor(1.0, dtype=D)
        wrt = [a for a in args if isinstance(a, torch.Tensor) and a.requires_grad]
        g1 = torch.autograd.grad((out * wts).sum(), wrt, allow_unused=True)
        g2 = torch.autograd.grad((href * wts).sum(), wrt, allow_unused=True)
        for a, b in zip(g1, g2):
            assert (a is None and b is None) or torch.allclose(a, b, rtol=1e-9, atol=1e-11), fn.__name__
