// Separable / stationary objectives (logpos.py:216-296, 383-462), Kronecker-structured primitives
// (kronecker_operation.py:36-85, distributions.py:10-113) and deterministic prediction
// (prediction.py:337-458, 912-1036, 1566-1638).  Entry points declared in include/nmgp.h.
//
// The N x N eigendecomposition is rocSOLVER dsyevd; the M x M one (M <= 8 for the objectives) is a cyclic Jacobi
// on the host: a 5 x 5 matrix is not GPU work.  Everything that scales with N or N^2 runs in hand-written kernels
// (nmgp_kernels_eig.hip) or rocBLAS GEMM/TRSM.
#include <algorithm>

#include <functional>

#include "nmgp_internal.h"

using namespace nmgpk;

namespace {

const double LOG2PI = 1.8378770664093453;

// scratch slot assignment of this translation unit (slots 0..5 belong to nmgp_api.hip)
enum { SL_A = 6, SL_SMALL = 7, SL_K3 = 8, SL_U = 9, SL_PART = 10, SL_G = 11, SL_X = 12, SL_Y = 13, SL_BIG = 14, SL_BIG2 = 15 };

// ---- small dense helpers on the host ----------------------------------------------------------------
// cyclic Jacobi eigendecomposition of the symmetric matrix whose UPPER triangle is given (row-major A, like
// torch.symeig's default).  w ascending, V[m*M + p] = component m of eigenvector p.
void jacobi_eigh(int M, const double* Ain, std::vector<double>& w, std::vector<double>& V) {
    std::vector<double> A((size_t)M * M);
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < M; ++j) A[(size_t)i * M + j] = (i <= j) ? Ain[(size_t)i * M + j] : Ain[(size_t)j * M + i];
    V.assign((size_t)M * M, 0.0);
    for (int i = 0; i < M; ++i) V[(size_t)i * M + i] = 1.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < M; ++j) (i == j ? diag : off) += A[(size_t)i * M + j] * A[(size_t)i * M + j];
        if (off <= 1e-34 * (diag + 1e-300)) break;
        for (int p = 0; p < M - 1; ++p)
            for (int q = p + 1; q < M; ++q) {
                const double apq = A[(size_t)p * M + q];
                if (apq == 0.0) continue;
                const double app = A[(size_t)p * M + p], aqq = A[(size_t)q * M + q];
                const double tau = (aqq - app) / (2.0 * apq);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
                const double cth = 1.0 / std::sqrt(1.0 + t * t), sth = t * cth;
                for (int k = 0; k < M; ++k) {
                    const double akp = A[(size_t)k * M + p], akq = A[(size_t)k * M + q];
                    A[(size_t)k * M + p] = cth * akp - sth * akq;
                    A[(size_t)k * M + q] = sth * akp + cth * akq;
                }
                for (int k = 0; k < M; ++k) {
                    const double apk = A[(size_t)p * M + k], aqk = A[(size_t)q * M + k];
                    A[(size_t)p * M + k] = cth * apk - sth * aqk;
                    A[(size_t)q * M + k] = sth * apk + cth * aqk;
                }
                for (int k = 0; k < M; ++k) {
                    const double vkp = V[(size_t)k * M + p], vkq = V[(size_t)k * M + q];
                    V[(size_t)k * M + p] = cth * vkp - sth * vkq;
                    V[(size_t)k * M + q] = sth * vkp + cth * vkq;
                }
            }
    }
    std::vector<int> idx(M);
    for (int i = 0; i < M; ++i) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](int a, int b) { return A[(size_t)a * M + a] < A[(size_t)b * M + b]; });
    w.resize(M);
    std::vector<double> Vs((size_t)M * M);
    for (int p = 0; p < M; ++p) {
        w[p] = A[(size_t)idx[p] * M + idx[p]];
        for (int m = 0; m < M; ++m) Vs[(size_t)m * M + p] = V[(size_t)m * M + idx[p]];
    }
    V.swap(Vs);
}

// packed tril (exp on the diagonal slots when `unconstrained`) -> dense L (row-major M x M) and B = L L^T
void build_B(const double* uL, int M, bool unconstrained, std::vector<double>& L, std::vector<double>& B) {
    L.assign((size_t)M * M, 0.0);
    int t = 0;
    for (int r = 0; r < M; ++r)
        for (int c = 0; c <= r; ++c, ++t) L[(size_t)r * M + c] = (c == r && unconstrained) ? std::exp(uL[t]) : uL[t];
    B.assign((size_t)M * M, 0.0);
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < M; ++j) {
            double s = 0.0;
            for (int k = 0; k < M; ++k) s += L[(size_t)i * M + k] * L[(size_t)j * M + k];
            B[(size_t)i * M + j] = s;
        }
}

// float32-rounded Normal(mean, sd).log_prob as torch evaluates it when mean/sd are Python numbers
// (logpos.py:283,446,450; see oracle.normal_log_prob).  Returns log prob and d/dv.
double normal_logprob_f32(double v, double mean, double sd, double* dv) {
    const float m32 = (float)mean, s32 = (float)sd;
    const double var = (double)(s32 * s32);
    const double log_sd = (double)std::log(s32);
    const double r = v - (double)m32;
    if (dv) *dv = -r / var;
    return -(r * r) / (2.0 * var) - log_sd - std::log(std::sqrt(2.0 * M_PI));
}

struct EigWork {
    int N = 0, M = 0;
    double* V = nullptr;    // [N, N] eigenvectors (column-major)
    double* wK = nullptr;   // [N]
    double* a = nullptr;    // [M N] projection of y (then alpha in the eigenbasis)
    double* wB = nullptr;   // [M]      device
    double* VB = nullptr;   // [M, M]   device, row-major
    double* VBt = nullptr;  // [M, M]   device, row-major transpose
    double* sums = nullptr; // [8]
    double* sig2 = nullptr; // device scalar
    std::vector<double> h_wB, h_VB, h_L, h_B;
};

int ensure_eig_buffers(nmgp_ctx* c, int N) {
    const size_t nn = (size_t)N * N;
    if (!c->d_K || c->K_cap < nn) {
        NMGP_TRY(nmgp_dev_alloc(c, &c->d_K, nn));
        NMGP_TRY(nmgp_dev_alloc(c, &c->d_K2, nn));
        NMGP_TRY(nmgp_dev_alloc(c, &c->d_w, (size_t)N));
        NMGP_TRY(nmgp_dev_alloc(c, &c->d_E, (size_t)N));
        c->K_cap = nn;
    }
    return 0;
}

// Eigendecompose the symmetric N x N matrix whose lower triangle (column-major) is in c->d_K.
int eig_K(nmgp_ctx* c, int N) {
    NmgpStage sp(c, NMGP_STAGE_EIG);
    HIP_TRY(c, hipMemsetAsync(c->d_info + 3, 0, sizeof(int), c->stream));
    BLAS_TRY(c, rocsolver_dsyevd(c->blas, rocblas_evect_original, rocblas_fill_lower, N, c->d_K, N, c->d_w, c->d_E,
                                 c->d_info + 3));
    return 0;
}

int check_eig_info(nmgp_ctx* c) {
    HIP_TRY(c, hipMemcpyAsync(c->h_info + 3, c->d_info + 3, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->h_info[3] != 0)
        return nmgp_fail(c, NMGP_NUM_EIG, "symmetric eigensolver did not converge (info=%d)", c->h_info[3]);
    return 0;
}

// Upload the small host-side eigendecomposition of B into device scratch; sets the EigWork pointers.
int setup_small(nmgp_ctx* c, EigWork& w, int M, int N, double sigma2) {
    w.M = M;
    w.N = N;
    double* sm;
    NMGP_TRY(nmgp_scratch_get(c, SL_SMALL, (size_t)3 * M * M + M + 32, &sm));
    w.wB = sm;
    w.VB = sm + M;
    w.VBt = w.VB + (size_t)M * M;
    w.sums = w.VBt + (size_t)M * M;
    w.sig2 = w.sums + 16;
    std::vector<double> h((size_t)2 * M * M + M);
    for (int p = 0; p < M; ++p) h[p] = w.h_wB[p];
    for (int m = 0; m < M; ++m)
        for (int p = 0; p < M; ++p) {
            h[M + (size_t)m * M + p] = w.h_VB[(size_t)m * M + p];
            h[M + (size_t)M * M + (size_t)p * M + m] = w.h_VB[(size_t)m * M + p];
        }
    HIP_TRY(c, hipMemcpyAsync(sm, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(w.sig2, &sigma2, sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));    // h and sigma2 are stack/heap temporaries
    NMGP_TRY(nmgp_scratch_get(c, SL_A, (size_t)M * N, &w.a));
    w.V = c->d_K;
    w.wK = c->d_w;
    return 0;
}

// Work the caller wants queued once the likelihood's launches are out and before the host blocks on their results (the
// separable objective's prior solves: enqueued earlier they would delay the first panel step by the ~0.2 ms the host needs for
// the library's small launches; enqueued here they run on their own stream under the factorisation).  Runs at most once.
static thread_local std::function<int()>* g_before_sync = nullptr;
static int run_before_sync() {
    if (!g_before_sync) return 0;
    std::function<int()>* f = g_before_sync;
    g_before_sync = nullptr;
    return (*f)();
}

// loglik of N(0, B kron K + sigma2 I) at the vector d_yv (output-major, device), K's lower triangle in c->d_K.
// On return: V in d_K, wK, a (scaled by w if want_scaled), sums[0..3] on the host in hs.
int kron_loglik(nmgp_ctx* c, EigWork& w, const double* d_yv, double sigma2, bool want_scaled, double hs[4],
                double* loglik) {
    const int N = w.N, M = w.M;
    NMGP_TRY(eig_K(c, N));
    {
        NmgpStage sp(c, NMGP_STAGE_KRONMV);
        int r = kron_mv(c->stream, w.V, N, N, d_yv, w.VBt, M, M, w.a);   // a = (V_B^T kron V_K^T) y  (distributions.py:43)
        if (r) return nmgp_fail(c, r, "kron_mv: more than 64 outputs are not supported");
    }
    {
        NmgpStage sp(c, NMGP_STAGE_REDUCE);
        eig_reduce(c->stream, w.a, w.wB, M, w.wK, N, w.sig2, want_scaled, w.sums);
    }
    HIP_TRY(c, hipMemcpyAsync(c->h_pin + 64, w.sums, 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    NMGP_TRY(run_before_sync());
    NMGP_TRY(check_eig_info(c));
    for (int k = 0; k < 4; ++k) hs[k] = c->h_pin[64 + k];
    *loglik = -0.5 * hs[0] - 0.5 * hs[1];                                   // distributions.py:51
    return 0;
}

// Adjoint of the Kronecker likelihood: per-location gradients (d_g: [2N] = g_tl | g_ts on the device),
// dB (host, M x M) and d loglik / d sigma2.  Requires kron_loglik(want_scaled=true) to have run.
int kron_adjoint(nmgp_ctx* c, EigWork& w, const double* d_ell, const double* d_sig, const double hs[4], double* d_g,
                 std::vector<double>& dB, double* dsigma2) {
    const int N = w.N, M = w.M;
    hipStream_t s = c->stream;
    const double one = 1.0, zero = 0.0;
    double *C, *U, *part, *coreB;
    NMGP_TRY(nmgp_scratch_get(c, SL_K3, (size_t)N * N, &C));
    NMGP_TRY(nmgp_scratch_get(c, SL_U, (size_t)N * M + (size_t)M * M, &U));
    coreB = U + (size_t)N * M;
    const int NJ = (N + 63) / 64;
    NMGP_TRY(nmgp_scratch_get(c, SL_PART, (size_t)NJ * N * 2, &part));
    {
        NmgpStage sp(c, NMGP_STAGE_INVERSE);
        // C = V diag(dvec) V^T, dvec[q] = sum_p wB[p] w[p,q]
        colscale_d(s, w.V, w.wB, M, w.wK, N, w.sig2, c->d_K2);
        BLAS_TRY(c, rocblas_dgemm(c->blas, rocblas_operation_none, rocblas_operation_transpose, N, N, N, &one, c->d_K2, N,
                                  w.V, N, &zero, C, N));
        // U = V At^T  (At = alpha in the eigenbasis, [M, N] row-major == [N, M] column-major)
        BLAS_TRY(c, rocblas_dgemm(c->blas, rocblas_operation_none, rocblas_operation_none, N, M, N, &one, w.V, N, w.a, N,
                                  &zero, U, N));
    }
    {
        NmgpStage sp(c, NMGP_STAGE_ADJOINT);
        sep_adjoint(s, c->d_x, d_ell, d_sig, U, w.wB, M, C, N, part);
        sep_grad_sum(s, part, NJ, N, d_g);
        sep_coreB(s, w.a, w.wB, M, w.wK, N, w.sig2, coreB);
    }
    std::vector<double> hc((size_t)M * M);
    HIP_TRY(c, hipMemcpyAsync(hc.data(), coreB, hc.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    // dB = 1/2 V_B coreB V_B^T
    dB.assign((size_t)M * M, 0.0);
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < M; ++j) {
            double acc = 0.0;
            for (int p = 0; p < M; ++p)
                for (int q = 0; q < M; ++q) acc += w.h_VB[(size_t)i * M + p] * hc[(size_t)p * M + q] * w.h_VB[(size_t)j * M + q];
            dB[(size_t)i * M + j] = 0.5 * acc;
        }
    *dsigma2 = 0.5 * (hs[2] - hs[3]);
    return 0;
}

// chain dB -> gradient w.r.t. the unconstrained packed factor (exp on the diagonal slots)
void dB_to_guL(const std::vector<double>& dB, const std::vector<double>& L, int M, std::vector<double>& g) {
    g.assign((size_t)M * (M + 1) / 2, 0.0);
    int t = 0;
    for (int r = 0; r < M; ++r)
        for (int cc = 0; cc <= r; ++cc, ++t) {
            double acc = 0.0;
            for (int k = 0; k < M; ++k) acc += (dB[(size_t)r * M + k] + dB[(size_t)k * M + r]) * L[(size_t)k * M + cc];
            g[t] = (cc == r) ? acc * L[(size_t)r * M + r] : acc;
        }
}

// log N(v; mu 1, RBF(x; alpha, beta) + jitter I) for the columns of R (already v - mu), via the cached factor.
// q_out[k] = Mahalanobis term of column k; if R2 != null also Sigma^-1 r (for the gradient).
int prior_solve(nmgp_ctx* c, rocblas_handle hb, hipStream_t sp, PriorFactor* pf, double* R, int ncol, double* R2) {
    const double one = 1.0;
    const int N = c->N;
    const bool subst = c->prior_trsv_all && !c->prior_rocblas && N <= 3500;       // by substitution: see gp_project
    if (subst) prior_trsv(sp, false, pf->L, pf->ld, 0, pf->L, pf->ld, 0, R, N, ncol, 1);
    else
        BLAS_TRY(c, rocblas_dtrsm(hb, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none,
                                  rocblas_diagonal_non_unit, N, ncol, &one, pf->L, pf->ld, R, N));
    if (R2) {
        HIP_TRY(c, hipMemcpyAsync(R2, R, (size_t)N * ncol * sizeof(double), hipMemcpyDeviceToDevice, sp));
        if (subst) prior_trsv(sp, true, pf->L, pf->ld, 0, pf->L, pf->ld, 0, R2, N, ncol, 1);
        else
            BLAS_TRY(c, rocblas_dtrsm(hb, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose,
                                      rocblas_diagonal_non_unit, N, ncol, &one, pf->L, pf->ld, R2, N));
    }
    return 0;
}

// ---- Cholesky formulation of the Kronecker likelihood -------------------------------------------------------
// loglik of N(0, B kron K + sigma2 I) as M independent N x N factorisations S_p = wB[p] K + sigma2 I (one batch of the
// blocked Cholesky; no eigendecomposition of K).  K's lower triangle must be in c->d_K (ld = N) and survives.
// With want_grad the same adjoint quantities as kron_adjoint are produced.
struct CholKron {
    double* alpha = nullptr;   // [M, N]  S_p^-1 yt_p   (== U of the adjoint kernel, column-major [N, M])
    double* Cneg = nullptr;    // [M] x (N x N): -S_p^-1
    std::vector<double> tr, tk, aa;   // per block: tr S_p^-1, <S_p^-1, K>, ||alpha_p||^2
};

int kron_chol_loglik(nmgp_ctx* c, EigWork& w, double sigma2, bool want_grad, double* loglik, CholKron* ck) {
    const int N = w.N, M = w.M;
    hipStream_t s = c->stream;
    const int xpad = (N + 1) & 1, xoff = N + 1 + xpad;
    const int ld = want_grad ? (int)((((size_t)2 * N + 2 + 15) / 16) * 16) : (int)((((size_t)N + 1 + 15) / 16) * 16);
    const long long bs = (long long)ld * N;
    double *S, *sm;
    NMGP_TRY(nmgp_scratch_get(c, SL_BIG, (size_t)M * bs, &S));
    NMGP_TRY(nmgp_scratch_get(c, SL_A, (size_t)3 * M * N + 64 + (size_t)M * 16, &sm));
    double *yt = sm, *z = sm + (size_t)M * N, *alpha = z + (size_t)M * N, *red = alpha + (size_t)M * N;
    int* info = reinterpret_cast<int*>(red + (size_t)M * 4);
    HIP_TRY(c, hipMemsetAsync(info, 0, (size_t)M * sizeof(int), s));
    {
        NmgpStage sp(c, NMGP_STAGE_COV);
        rotate_y(s, c->d_Y, w.VB, N, M, yt);                          // yt_p = (V_B^T kron I) y
        sep_blocks(s, c->d_K, w.wB, w.sig2, N, M, S, ld, bs);
    }
    {
        NmgpStage sp(c, NMGP_STAGE_CHOL);
        set_row(s, S, ld, N, yt, N, M, bs, N);
        if (want_grad) identity_rows(s, S, ld, N + 1, N, xpad, M, bs);
        potrf_lower(s, c->stream2, nmgp_chol_events(c, N), S, ld, N, want_grad ? 1 + xpad : 1, want_grad ? N : 0,
                    c->chol_nb1, info, M, bs, 1, nmgp_syrk_hook(c));
        get_row(s, S, ld, N, z, N, M, bs, N);
    }
    {
        NmgpStage sp(c, NMGP_STAGE_REDUCE);
        chol_logdet_quad(s, S, ld, N, z, red, red + 1, M, bs, 4);
    }
    std::vector<double> hr((size_t)M * 4);
    std::vector<int> hi(M);
    NMGP_TRY(run_before_sync());
    HIP_TRY(c, hipMemcpyAsync(hr.data(), red, hr.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(hi.data(), info, (size_t)M * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    for (int p = 0; p < M; ++p)
        if (hi[p] != 0)
            return nmgp_fail(c, hi[p], "block %d of the separable covariance is not positive definite (leading minor %d)",
                             p, hi[p]);
    double ll = 0.0;
    for (int p = 0; p < M; ++p) ll += -0.5 * hr[(size_t)p * 4] - 0.5 * hr[(size_t)p * 4 + 1];
    *loglik = ll;
    if (!want_grad) return 0;
    double* Cneg;
    NMGP_TRY(nmgp_scratch_get(c, SL_BIG2, (size_t)M * N * N, &Cneg));
    double* part;
    const size_t tri_part = (size_t)N * ((N + 255) / 256);           // block sums of the triangular matrix-vector product, per block
    NMGP_TRY(nmgp_scratch_get(c, SL_PART, std::max((size_t)M * tri_part, (size_t)((N + 63) / 64) * N * 2 + (size_t)M * 128 * 3 + 8), &part));
    {
        NmgpStage sp(c, NMGP_STAGE_INVERSE);
        // alpha_p = X_p z_p over the blocks on and above the diagonal only: X = L^-T is upper triangular, and what lies below
        // its diagonal band was never written (k_xtri_seed)
        tri_gemv_upper(s, S + xoff, ld, N, z, alpha, part, M, bs, (long long)tri_part);
        syrk_lower(s, S + xoff, ld, Cneg, N, N, N, N, M, bs, (long long)N * N, 1);              // -S_p^-1
    }
    const int G = sep_traces(s, Cneg, c->d_K, alpha, N, M, part);
    std::vector<double> hp((size_t)M * G * 3);
    HIP_TRY(c, hipMemcpyAsync(hp.data(), part, hp.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    ck->tr.assign(M, 0.0);
    ck->tk.assign(M, 0.0);
    ck->aa.assign(M, 0.0);
    for (int p = 0; p < M; ++p)
        for (int g = 0; g < G; ++g) {
            ck->tr[p] += hp[((size_t)p * G + g) * 3];
            ck->tk[p] += hp[((size_t)p * G + g) * 3 + 1];
            ck->aa[p] += hp[((size_t)p * G + g) * 3 + 2];
        }
    ck->alpha = alpha;
    ck->Cneg = Cneg;
    return 0;
}

// Adjoint for the Cholesky formulation: per-location gradients d_g ([2N] = g_tl | g_ts), dB (host), d loglik/d sigma2.
int kron_chol_adjoint(nmgp_ctx* c, EigWork& w, CholKron& ck, const double* d_ell, const double* d_sig, double* d_g,
                      std::vector<double>& dB, double* dsigma2) {
    const int N = w.N, M = w.M;
    hipStream_t s = c->stream;
    const double one = 1.0, zero = 0.0;
    double *C, *W, *part;
    NMGP_TRY(nmgp_scratch_get(c, SL_K3, (size_t)N * N, &C));
    NMGP_TRY(nmgp_scratch_get(c, SL_U, (size_t)N * M + (size_t)M * M, &W));
    double* Xi = W + (size_t)N * M;
    const int NJ = (N + 63) / 64;
    NMGP_TRY(nmgp_scratch_get(c, SL_PART, (size_t)NJ * N * 2 + (size_t)M * 128 * 3 + 8, &part));
    {
        NmgpStage sp(c, NMGP_STAGE_ADJOINT);
        weighted_sum_lower(s, ck.Cneg, w.wB, N, M, C);                 // C = sum_p wB[p] S_p^-1 (lower)
        fill_lower_to_full(s, C, N, N);
        sep_adjoint(s, c->d_x, d_ell, d_sig, ck.alpha, w.wB, M, C, N, part);
        sep_grad_sum(s, part, NJ, N, d_g);
        // Xi' = U^T (K U): the M x M quadratic forms alpha_p^T K alpha_p'  (W is scratch: zeroed, so that a beta = 0 that is
        // implemented as a scaling cannot carry stale NaNs over)
        HIP_TRY(c, hipMemsetAsync(W, 0, ((size_t)N * M + (size_t)M * M) * sizeof(double), s));
        BLAS_TRY(c, rocblas_dsymm(c->blas, rocblas_side_left, rocblas_fill_lower, N, M, &one, c->d_K, N, ck.alpha, N, &zero,
                                  W, N));
        BLAS_TRY(c, rocblas_dgemm(c->blas, rocblas_operation_transpose, rocblas_operation_none, M, M, N, &one, ck.alpha, N,
                                  W, N, &zero, Xi, M));
    }
    std::vector<double> hx((size_t)M * M);
    HIP_TRY(c, hipMemcpyAsync(hx.data(), Xi, hx.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    // d loglik / dB = V_B Xi V_B^T,  Xi[p,p'] = 1/2 (alpha_p^T K alpha_p' - d_pp' <S_p^-1, K>)
    std::vector<double> Xs((size_t)M * M);
    for (int p = 0; p < M; ++p)
        for (int q = 0; q < M; ++q) Xs[(size_t)p * M + q] = 0.5 * (hx[(size_t)q * M + p] - (p == q ? ck.tk[p] : 0.0));
    dB.assign((size_t)M * M, 0.0);
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < M; ++j) {
            double acc = 0.0;
            for (int p = 0; p < M; ++p)
                for (int q = 0; q < M; ++q) acc += w.h_VB[(size_t)i * M + p] * Xs[(size_t)p * M + q] * w.h_VB[(size_t)j * M + q];
            dB[(size_t)i * M + j] = acc;
        }
    double ds = 0.0;
    for (int p = 0; p < M; ++p) ds += 0.5 * (ck.aa[p] - ck.tr[p]);
    *dsigma2 = ds;
    return 0;
}

// The reference never lets a NaN likelihood out of the separable / stationary objectives: `while loglik != loglik:
// loglik = multivariate_normal_logpdf1(...)` (logpos.py:267-268, 436-437) retries with `precision`-sized RANDOM jitter on the
// diagonals of B and K (distributions.py:66,69).  The same safety net, made deterministic: after a numerical failure
// (a block that is not positive definite, a non-finite likelihood, the eigensolver) the evaluation is repeated up to
// NMGP_SEP_RETRIES times with attempt * precision added to both diagonals; value and gradient then belong to that slightly
// regularised covariance, like the reference's.  Only if every attempt fails does the failure reach the caller.
#define NMGP_SEP_RETRIES 3
template <class BuildK>
int kron_likelihood_with_retry(nmgp_ctx* c, EigWork& w, int M, int N, double sigma2, bool want_grad, BuildK&& build_K,
                               double hs[4], double* loglik, CholKron* ck, int* attempts) {
    const std::vector<double> B0 = w.h_B;
    int rc = 0;
    for (int attempt = 0; attempt <= NMGP_SEP_RETRIES; ++attempt) {
        const double jit = attempt * NMGP_PRECISION;
        w.h_B = B0;
        for (int i = 0; i < M; ++i) w.h_B[(size_t)i * M + i] += jit;
        jacobi_eigh(M, w.h_B.data(), w.h_wB, w.h_VB);
        NMGP_TRY(setup_small(c, w, M, N, sigma2));
        build_K();
        if (attempt > 0) add_diag(c->stream, c->d_K, N, N, jit);
        *loglik = std::nan("");
        if (c->sep_algo == 1)
            rc = kron_chol_loglik(c, w, sigma2, want_grad, loglik, ck);
        else
            rc = kron_loglik(c, w, c->d_y, sigma2, want_grad, hs, loglik);
        if (rc < 0) return rc;                                    // API / runtime error: not a numerical matter
        if (rc == 0 && std::isfinite(*loglik)) {
            *attempts = attempt;
            return 0;
        }
    }
    *attempts = NMGP_SEP_RETRIES;
    return rc;      // > 0: the last numerical failure; 0 with a non-finite loglik is reported by the caller
}

int require_data(nmgp_ctx* c) {
    if (!c->d_x) return nmgp_fail(c, NMGP_E_STATE, "nmgp_set_data must be called before evaluating");
    return 0;
}

}  // namespace

// =================================================================================================
// separable objective
// =================================================================================================
extern "C" int nmgp_logpos_sep(nmgp_ctx* c, const double* pars, const double hyper[9], int prior, double out6[6],
                               double* grad) {
    if (!c) return NMGP_E_NULL;
    if (!pars || !hyper || !out6) return nmgp_fail(c, NMGP_E_NULL, "pars/hyper/out6 must not be NULL");
    NMGP_TRY(require_data(c));
    HIP_TRY(c, hipSetDevice(c->device));
    const int N = c->N, M = c->M, T = c->T;
    const size_t P = (size_t)2 * N + T + 1;
    const double mu_l = hyper[0], al_l = hyper[1], be_l = hyper[2], mu_s = hyper[3], al_s = hyper[4], be_s = hyper[5];
    const double a = hyper[6], b = hyper[7], cc = hyper[8];
    hipStream_t s = c->stream;
    const double tse = pars[P - 1];
    const double sigma2 = std::exp(tse);
    NMGP_TRY(ensure_eig_buffers(c, N));
    HIP_TRY(c, hipMemcpyAsync(c->d_pars, pars, P * sizeof(double), hipMemcpyHostToDevice, s));
    EigWork w;
    build_B(pars + 2 * N, M, true, w.h_L, w.h_B);
    {
        NmgpStage sp(c, NMGP_STAGE_COV);
        exp_vec(s, c->d_pars, N, c->d_ell);
        exp_vec(s, c->d_pars + N, N, c->d_sig);
    }
    // prior factors (cached; created and synchronised on first use) and the prior solves, which depend on the parameters
    // only: queued on the prior stream so that they run under the likelihood's factorisation
    PriorFactor *pl = nullptr, *ps = nullptr;
    NMGP_TRY(nmgp_get_prior(c, al_l, be_l, &pl));
    NMGP_TRY(nmgp_get_prior(c, al_s, be_s, &ps));
    NMGP_TRY(nmgp_get_prior(c, al_l, be_l, &pl));
    PriorStreamScope pscope(c);          // fork now; the solves are queued by the hook, after the likelihood's launches
    std::function<int()> enqueue_priors = [&]() -> int {
        {
            NmgpStage sp(c, NMGP_STAGE_PRIOR, pscope.sp, 0.0, 0.0);
            two_col_rhs(pscope.sp, c->d_pars, mu_l, c->d_pars + N, mu_s, N, c->d_R);
            double* R2 = (grad && prior) ? c->d_R2 : nullptr;
            if (pl == ps) {
                NMGP_TRY(prior_solve(c, pscope.hb, pscope.sp, pl, c->d_R, 2, R2));
            } else {
                NMGP_TRY(prior_solve(c, pscope.hb, pscope.sp, pl, c->d_R, 1, R2));
                NMGP_TRY(prior_solve(c, pscope.hb, pscope.sp, ps, c->d_R + N, 1, R2 ? R2 + N : nullptr));
            }
            col_sumsq(pscope.sp, c->d_R, N, N, 2, c->d_scal + 2);
        }
        pscope.done();
        return 0;
    };
    g_before_sync = &enqueue_priors;
    struct HookReset {
        ~HookReset() { g_before_sync = nullptr; }
    } hook_reset;
    double hs[4], loglik;
    CholKron ck;
    int attempts = 0;
    NMGP_TRY(kron_likelihood_with_retry(c, w, M, N, sigma2, grad != nullptr, [&] {
        NmgpStage sp(c, NMGP_STAGE_COV);
        gibbs_cov_sym(s, c->d_x, c->d_sig, c->d_ell, N, c->d_K, N, false);   // logpos.py:258
    }, hs, &loglik, &ck, &attempts));
    c->last_sep_attempts = attempts;
    // GP priors on tilde_l and tilde_sigma (logpos.py:271-281): solved on the prior stream under the likelihood
    double q[2], hl[2];
    {
        NMGP_TRY(run_before_sync());      // (not yet queued if the likelihood returned before its first synchronisation)
        pscope.join();
        HIP_TRY(c, hipMemcpyAsync(c->h_pin + 72, c->d_scal + 2, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipMemcpyAsync(c->h_pin + 74, pl->logdet, sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipMemcpyAsync(c->h_pin + 75, ps->logdet, sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        q[0] = c->h_pin[72]; q[1] = c->h_pin[73]; hl[0] = c->h_pin[74]; hl[1] = c->h_pin[75];
    }
    const double lp_l = -0.5 * (N * LOG2PI + q[0]) - hl[0];
    const double lp_s = -0.5 * (N * LOG2PI + q[1]) - hl[1];
    double lp_uL = 0.0;
    std::vector<double> g_uL_prior(T, 0.0);
    for (int t = 0; t < T; ++t) lp_uL += normal_logprob_f32(pars[2 * N + t], 0.0, cc, &g_uL_prior[t]);   // logpos.py:283
    const double lp_s2 = (-a - 1.0) * std::log(sigma2) - b / sigma2 + a * std::log(b) - std::lgamma(a);
    double res = 0.0;
    res += loglik;
    if (prior) { res += lp_l; res += lp_s; res += lp_uL; res += lp_s2; res += tse; }
    out6[0] = -res; out6[1] = loglik; out6[2] = lp_l; out6[3] = lp_s; out6[4] = lp_uL; out6[5] = lp_s2;
    if (grad) {
        double* d_g;
        NMGP_TRY(nmgp_scratch_get(c, SL_G, (size_t)2 * N, &d_g));
        std::vector<double> dB, g_uL;
        double ds;
        if (c->sep_algo == 1)
            NMGP_TRY(kron_chol_adjoint(c, w, ck, c->d_ell, c->d_sig, d_g, dB, &ds));
        else
            NMGP_TRY(kron_adjoint(c, w, c->d_ell, c->d_sig, hs, d_g, dB, &ds));
        dB_to_guL(dB, w.h_L, M, g_uL);
        std::vector<double> hg((size_t)2 * N), hr((size_t)2 * N, 0.0);
        HIP_TRY(c, hipMemcpyAsync(hg.data(), d_g, hg.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        if (prior) HIP_TRY(c, hipMemcpyAsync(hr.data(), c->d_R2, hr.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        for (int i = 0; i < 2 * N; ++i) grad[i] = -(hg[i] - hr[i]);
        for (int t = 0; t < T; ++t) grad[2 * N + t] = -(g_uL[t] + (prior ? g_uL_prior[t] : 0.0));
        double ge = sigma2 * ds;
        if (prior) ge += (-a - 1.0) + b / sigma2 + 1.0;
        grad[P - 1] = -ge;
    }
    NMGP_TRY(nmgp_take_launch_error(c));
    if (!std::isfinite(out6[1])) return nmgp_fail(c, NMGP_NUM_NAN, "non-finite separable likelihood (%g)", out6[1]);
    return 0;
}

// =================================================================================================
// separable objective, B chains of the resident subject per launch sequence
// =================================================================================================
// The chains of the separable model's MCMC (the reference runs them one process each: Separable_model_mpisim.py:299-300;
// every HMC iteration asks for the objective of logpos.py:216-296 once per leapfrog step) as ONE batch of the blocked Cholesky:
// chain b contributes its M blocks S_bp = wB_b[p] K_x,b + sigma2_b I, so B chains are B M matrices of order N in every launch of
// the factorisation, of the triangular matrix-vector product and of the inverse SYRK.  Since round 5 EVERY other piece takes the
// chain as a grid dimension too (nmgp_kernels_sep.hip: parameter unpacking + rotation of y, the blocks written straight from
// (x, ell, sigma) -- K_x itself only when the gradient needs it --, one pass over the blocks of -S^-1 for the traces and the weighted
// sum, the fused adjoint; the M x M quadratic forms as strided-batched library products), and the whole evaluation -- value AND
// gradient half -- is enqueued before the ONE synchronisation: the M x M eigendecompositions of B are host work done before the
// first launch, nothing between the halves needs the host.  One chain's 5 blocks of N = 4096 are latency-bound (0.37 of the FP64
// matrix roofline); 16 chains are 80 matrices on the throughput schedule.  Cholesky formulation only (NMGP_SEP=eig falls back to
// chain-by-chain evaluation).  A chain whose covariance fails numerically is re-evaluated through nmgp_logpos_sep, i.e. with the
// reference's jitter retries (status[b] = the retries it needed; API errors fail the call).
// Memory: the batch is evaluated in chunks of chains whose device slab stays below NMGP_SEP_BATCH_SLAB_GB (default 96 GB:
// 32 chains of N = 4096, D = 5 with gradients need 79 GB); grid dimensions bound a chunk to 65,535 / M chains.
// pars [B, 2N+T+1]; out6 [B, 6]; grad [B, P] or NULL; status [B]: 0 exact, k > 0 evaluated with k jitter retries,
// negative = -(NMGP_NUM_NAN or the leading-minor index) if even those failed (out6 row NaN, gradient row zero).
namespace {

struct SepBatchLayout {
    size_t o_P, o_ell, o_sig, o_K, o_small, o_yt, o_z, o_alpha, o_red, o_info, o_R, o_R2, o_q, o_S;
    size_t o_Cneg = 0, o_part = 0, o_C = 0, o_Xi = 0, o_g = 0, o_tr = 0;
    size_t total = 0, small_per = 0, tri_part = 0;
    int ld = 0;
    long long bs = 0;
};

SepBatchLayout sep_batch_layout(int B, int N, int M, int T, bool want_grad) {
    SepBatchLayout L;
    const size_t P = (size_t)2 * N + T + 1, NN = (size_t)N * N, BM = (size_t)B * M;
    L.ld = want_grad ? (int)((((size_t)2 * N + 2 + 15) / 16) * 16) : (int)((((size_t)N + 1 + 15) / 16) * 16);
    L.bs = (long long)L.ld * N;
    L.small_per = (size_t)M + (size_t)M * M + 2;        // wB | VB (row-major) | sigma2 | pad
    L.tri_part = (size_t)N * ((N + 255) / 256);
    const size_t NJ = (N + 63) / 64;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += (n + 1) & ~(size_t)1; return o; };
    L.o_P = take((size_t)B * P); L.o_ell = take((size_t)B * N); L.o_sig = take((size_t)B * N);
    L.o_K = take(want_grad ? (size_t)B * NN : 2);       // K_x of every chain: only the gradient reads it back
    L.o_small = take((size_t)B * L.small_per); L.o_yt = take(BM * N); L.o_z = take(BM * N);
    L.o_alpha = take(BM * N); L.o_red = take(BM * 4); L.o_info = take(BM);
    L.o_R = take((size_t)N * 2 * B); L.o_R2 = take((size_t)N * 2 * B); L.o_q = take((size_t)2 * B + 2);
    L.o_S = take(BM * (size_t)L.bs);
    if (want_grad) {
        L.o_Cneg = take(BM * NN);
        L.o_part = take(std::max(BM * L.tri_part, (size_t)B * NJ * N * 2 + 8));
        L.o_C = take((size_t)B * NN);
        L.o_Xi = take((size_t)B * NMGP_SEP_TR_G * M * (M + 1) / 2);      // partial quadratic forms alpha_p^T K alpha_q, per chain and group
        L.o_g = take((size_t)B * 2 * N);
        L.o_tr = take(BM * NMGP_SEP_TR_G * 3);
    }
    L.total = off;
    return L;
}

// chains [0, B) of `pars` (already offset by the caller): everything enqueued, one synchronisation, host epilogue.  bad[b] = 1 marks a
// chain whose blocks failed numerically (its out6 / grad rows are then unspecified: the caller re-evaluates it one by one).
int sep_batch_core(nmgp_ctx* c, const double* pars, int B, const double hyper[9], int prior, double* out6, double* grad, int* status,
                   std::vector<char>& bad) {
    const int N = c->N, M = c->M, T = c->T;
    const size_t P = (size_t)2 * N + T + 1;
    const bool want_grad = grad != nullptr;
    const double mu_l = hyper[0], al_l = hyper[1], be_l = hyper[2], mu_s = hyper[3], al_s = hyper[4], be_s = hyper[5];
    const double a = hyper[6], bb = hyper[7], cc = hyper[8];
    hipStream_t s = c->stream;
    PriorFactor *pl = nullptr, *ps = nullptr;
    NMGP_TRY(nmgp_get_prior(c, al_l, be_l, &pl));
    NMGP_TRY(nmgp_get_prior(c, al_s, be_s, &ps));
    NMGP_TRY(nmgp_get_prior(c, al_l, be_l, &pl));           // (re-resolved: the second call may have grown the cache)
    const int BM = B * M;
    const SepBatchLayout L = sep_batch_layout(B, N, M, T, want_grad);
    const int xpad = (N + 1) & 1, xoff = N + 1 + xpad;
    const int ld = L.ld;
    const long long bs = L.bs;
    const size_t NN = (size_t)N * N;
    const int sp_ = (int)L.small_per;
    const int G = NMGP_SEP_TR_G;
    double* slab;
    NMGP_TRY(nmgp_scratch_get(c, SL_BIG, L.total, &slab));      // (the single-chain path's slot: the two never run at the same time)
    double *dP = slab + L.o_P, *d_ell = slab + L.o_ell, *d_sig = slab + L.o_sig, *dK = slab + L.o_K, *d_small = slab + L.o_small;
    double *yt = slab + L.o_yt, *z = slab + L.o_z, *alpha = slab + L.o_alpha, *red = slab + L.o_red;
    int* info = reinterpret_cast<int*>(slab + L.o_info);
    double *R = slab + L.o_R, *R2 = slab + L.o_R2, *dq = slab + L.o_q, *S = slab + L.o_S;
    // ---- host: B = L L^T and its eigendecomposition, per chain (M x M: known before the first launch) ----
    std::vector<EigWork> w(B);
    std::vector<double> hsmall((size_t)B * L.small_per, 0.0), sig2(B), tse(B);
    for (int b = 0; b < B; ++b) {
        const double* pb = pars + (size_t)b * P;
        tse[b] = pb[P - 1];
        sig2[b] = std::exp(tse[b]);
        build_B(pb + 2 * N, M, true, w[b].h_L, w[b].h_B);
        jacobi_eigh(M, w[b].h_B.data(), w[b].h_wB, w[b].h_VB);
        double* h = hsmall.data() + (size_t)b * L.small_per;
        for (int p = 0; p < M; ++p) h[p] = w[b].h_wB[p];
        for (int k = 0; k < M * M; ++k) h[M + k] = w[b].h_VB[k];
        h[M + (size_t)M * M] = sig2[b];
    }
    HIP_TRY(c, hipMemcpyAsync(dP, pars, (size_t)B * P * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(d_small, hsmall.data(), hsmall.size() * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemsetAsync(info, 0, (size_t)BM * sizeof(int), s));
    PriorStreamScope pscope(c);
    {
        NmgpStage sp(c, NMGP_STAGE_COV);
        sep_prep_b(s, dP, (long long)P, c->d_Y, d_small, sp_, N, M, d_ell, d_sig, yt, B);
        sep_blocks_b(s, c->d_x, d_ell, d_sig, d_small, sp_, N, M, S, ld, bs, want_grad ? dK : nullptr, B);
    }
    {
        NmgpStage sp(c, NMGP_STAGE_CHOL);
        set_row(s, S, ld, N, yt, N, BM, bs, N);
        if (want_grad) identity_rows(s, S, ld, N + 1, N, xpad, BM, bs);
        potrf_lower(s, c->stream2, nmgp_chol_events(c, N), S, ld, N, want_grad ? 1 + xpad : 1, want_grad ? N : 0, c->chol_nb1, info,
                    BM, bs, 1, nmgp_syrk_hook(c));
        get_row(s, S, ld, N, z, N, BM, bs, N);
    }
    {
        NmgpStage sp(c, NMGP_STAGE_REDUCE);
        chol_logdet_quad(s, S, ld, N, z, red, red + 1, BM, bs, 4);
    }
    {
        // GP priors on tilde_l and tilde_sigma of every chain: 2 B right-hand sides against the cached factors
        NmgpStage sp(c, NMGP_STAGE_PRIOR, pscope.sp, 0.0, 0.0);
        two_col_rhs_b(pscope.sp, dP, (long long)P, mu_l, mu_s, N, R, B);
        double* r2 = (want_grad && prior) ? R2 : nullptr;
        if (pl == ps) {
            NMGP_TRY(prior_solve(c, pscope.hb, pscope.sp, pl, R, 2 * B, r2));
        } else {
            for (int b = 0; b < B; ++b) {
                NMGP_TRY(prior_solve(c, pscope.hb, pscope.sp, pl, R + (size_t)2 * b * N, 1, r2 ? r2 + (size_t)2 * b * N : nullptr));
                NMGP_TRY(prior_solve(c, pscope.hb, pscope.sp, ps, R + (size_t)(2 * b + 1) * N, 1, r2 ? r2 + (size_t)(2 * b + 1) * N : nullptr));
            }
        }
        col_sumsq(pscope.sp, R, N, N, 2 * B, dq);
    }
    pscope.done();
    pscope.join();
    std::vector<double> hr((size_t)BM * 4), hq((size_t)2 * B);
    std::vector<int> hi(BM);
    double hl[2];
    HIP_TRY(c, hipMemcpyAsync(hr.data(), red, hr.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(hi.data(), info, (size_t)BM * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(hq.data(), dq, hq.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(&hl[0], pl->logdet, sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(&hl[1], ps->logdet, sizeof(double), hipMemcpyDeviceToHost, s));
    // ---- gradient half: enqueued behind the value half without waiting for it (a chain that failed produces garbage here, which the
    // epilogue discards) ----
    std::vector<double> hx, hg, hR2, htr;
    if (want_grad) {
        double *Cneg = slab + L.o_Cneg, *part = slab + L.o_part, *C = slab + L.o_C, *Xi = slab + L.o_Xi, *d_g = slab + L.o_g;
        double* trs = slab + L.o_tr;
        {
            NmgpStage sp(c, NMGP_STAGE_INVERSE);
            tri_gemv_upper(s, S + xoff, ld, N, z, alpha, part, BM, bs, (long long)L.tri_part);
            syrk_lower(s, S + xoff, ld, Cneg, N, N, N, N, BM, bs, (long long)NN, 1);              // -S_bp^-1
        }
        {
            NmgpStage sp(c, NMGP_STAGE_ADJOINT);
            // traces, <S^-1, K>, the weighted sum C and the M x M quadratic forms alpha_p^T K alpha_q: ONE pass over -S^-1 and K_x
            sep_reduce_b(s, Cneg, dK, alpha, d_small, sp_, N, M, G, C, trs, Xi, B);
            fill_lower_to_full(s, C, N, N, B);
            sep_adjoint_b(s, c->d_x, d_ell, d_sig, alpha, d_small, sp_, M, C, N, part, d_g, B);
        }
        hx.assign((size_t)B * G * M * (M + 1) / 2, 0.0);
        hg.assign((size_t)B * 2 * N, 0.0);
        hR2.assign((size_t)N * 2 * B, 0.0);
        htr.assign((size_t)BM * G * 3, 0.0);
        HIP_TRY(c, hipMemcpyAsync(hx.data(), Xi, hx.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipMemcpyAsync(hg.data(), d_g, hg.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipMemcpyAsync(htr.data(), trs, htr.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        if (prior) HIP_TRY(c, hipMemcpyAsync(hR2.data(), R2, hR2.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(c, hipStreamSynchronize(s));          // the one synchronisation of the evaluation
    NMGP_TRY(nmgp_take_launch_error(c));
    // ---- host epilogue ----
    for (int b = 0; b < B; ++b) {
        bad[b] = 0;
        double ll = 0.0;
        for (int p = 0; p < M; ++p) {
            if (hi[(size_t)b * M + p] != 0) bad[b] = 1;
            ll += -0.5 * hr[((size_t)b * M + p) * 4] - 0.5 * hr[((size_t)b * M + p) * 4 + 1];
        }
        if (!std::isfinite(ll)) bad[b] = 1;
        const double* pb = pars + (size_t)b * P;
        const double lp_l = -0.5 * (N * LOG2PI + hq[(size_t)2 * b]) - hl[0];
        const double lp_s = -0.5 * (N * LOG2PI + hq[(size_t)2 * b + 1]) - hl[1];
        double lp_uL = 0.0;
        std::vector<double> g_uL_prior(T, 0.0);
        for (int t = 0; t < T; ++t) lp_uL += normal_logprob_f32(pb[2 * N + t], 0.0, cc, &g_uL_prior[t]);
        const double lp_s2 = (-a - 1.0) * std::log(sig2[b]) - bb / sig2[b] + a * std::log(bb) - std::lgamma(a);
        double res = ll;
        if (prior) { res += lp_l; res += lp_s; res += lp_uL; res += lp_s2; res += tse[b]; }
        double* o = out6 + (size_t)b * 6;
        o[0] = -res; o[1] = ll; o[2] = lp_l; o[3] = lp_s; o[4] = lp_uL; o[5] = lp_s2;
        status[b] = 0;
        if (!want_grad || bad[b]) continue;
        std::vector<double> tr(M, 0.0), tk(M, 0.0), aa(M, 0.0);
        const double* tb = htr.data() + (size_t)b * M * G * 3;
        for (int p = 0; p < M; ++p)
            for (int g = 0; g < G; ++g) {
                tr[p] += tb[((size_t)p * G + g) * 3];
                tk[p] += tb[((size_t)p * G + g) * 3 + 1];
                aa[p] += tb[((size_t)p * G + g) * 3 + 2];
            }
        // d loglik / dB = V_B Xi V_B^T,  Xi[p,p'] = 1/2 (alpha_p^T K alpha_p' - d_pp' <S_p^-1, K>)
        std::vector<double> Xs((size_t)M * M), dB((size_t)M * M, 0.0), g_uL, aKa((size_t)M * M, 0.0);
        {
            const int NX = M * (M + 1) / 2;
            const double* hxb = hx.data() + (size_t)b * G * NX;
            int e = 0;
            for (int p = 0; p < M; ++p)
                for (int q = p; q < M; ++q, ++e) {
                    double acc = 0.0;
                    for (int g = 0; g < G; ++g) acc += hxb[(size_t)g * NX + e];
                    aKa[(size_t)p * M + q] = aKa[(size_t)q * M + p] = acc;
                }
        }
        for (int p = 0; p < M; ++p)
            for (int q = 0; q < M; ++q) Xs[(size_t)p * M + q] = 0.5 * (aKa[(size_t)q * M + p] - (p == q ? tk[p] : 0.0));
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < M; ++j) {
                double acc = 0.0;
                for (int p = 0; p < M; ++p)
                    for (int q = 0; q < M; ++q) acc += w[b].h_VB[(size_t)i * M + p] * Xs[(size_t)p * M + q] * w[b].h_VB[(size_t)j * M + q];
                dB[(size_t)i * M + j] = acc;
            }
        double ds = 0.0;
        for (int p = 0; p < M; ++p) ds += 0.5 * (aa[p] - tr[p]);
        dB_to_guL(dB, w[b].h_L, M, g_uL);
        double* gb = grad + (size_t)b * P;
        for (int i = 0; i < 2 * N; ++i) {
            // (R2 holds the chain's two solved columns [tilde_l | tilde_sigma] back to back: the layout of the 2N leading parameters)
            const double r = prior ? hR2[(size_t)2 * b * N + i] : 0.0;
            gb[i] = -(hg[(size_t)b * 2 * N + i] - r);
        }
        for (int t = 0; t < T; ++t) gb[2 * N + t] = -(g_uL[t] + (prior ? g_uL_prior[t] : 0.0));
        double ge = sig2[b] * ds;
        if (prior) ge += (-a - 1.0) + bb / sig2[b] + 1.0;
        gb[P - 1] = -ge;
    }
    return 0;
}

}  // namespace

extern "C" int nmgp_sep_batch_eval(nmgp_ctx* c, const double* pars, int B, const double hyper[9], int prior, double* out6,
                                   double* grad, int* status) {
    if (!c) return NMGP_E_NULL;
    if (!pars || !hyper || !out6 || !status) return nmgp_fail(c, NMGP_E_NULL, "pars/hyper/out6/status must not be NULL");
    if (B <= 0) return nmgp_fail(c, NMGP_E_SHAPE, "B must be positive");
    NMGP_TRY(require_data(c));
    HIP_TRY(c, hipSetDevice(c->device));
    const int N = c->N, M = c->M, T = c->T;
    const size_t P = (size_t)2 * N + T + 1;
    const bool want_grad = grad != nullptr;
    auto one_by_one = [&](int b) -> int {
        int rc = nmgp_logpos_sep(c, pars + (size_t)b * P, hyper, prior, out6 + (size_t)b * 6, want_grad ? grad + (size_t)b * P : nullptr);
        if (rc < 0) return rc;
        if (rc > 0) {
            for (int k = 0; k < 6; ++k) out6[(size_t)b * 6 + k] = std::nan("");
            if (want_grad) for (size_t k = 0; k < P; ++k) grad[(size_t)b * P + k] = 0.0;
            status[b] = -rc;                  // (negative: a leading-minor index must not read as a retry count)
        } else {
            status[b] = c->last_sep_attempts;
        }
        return 0;
    };
    // (the batched block kernel addresses a block with 32-bit byte offsets: (2N + 2) N 8 < 2^31, i.e. N <= 11,584 -- beyond that, and
    // for the eigen formulation, the chains are evaluated one by one)
    const bool blocks_fit = (long long)(2 * (long long)N + 18) * N * 8 < 0x7fffffffLL;
    if (c->sep_algo != 1 || B == 1 || !blocks_fit) {
        for (int b = 0; b < B; ++b) NMGP_TRY(one_by_one(b));
        return 0;
    }
    NMGP_TRY(ensure_eig_buffers(c, N));
    // chunk of chains: the device slab below the cap, the chain x block index within the grid's z limit
    double cap_gb = 96.0;
    if (const char* e = std::getenv("NMGP_SEP_BATCH_SLAB_GB")) cap_gb = std::max(1.0, std::atof(e));
    const size_t per_chain = sep_batch_layout(1, N, M, T, want_grad).total * sizeof(double);
    int Bc = (int)std::min<double>((double)B, std::floor(cap_gb * 1e9 / (double)per_chain));
    Bc = std::min(Bc, 65535 / std::max(M, 1));
    if (Bc < 1)
        return nmgp_fail(c, NMGP_E_SHAPE, "one chain of the separable model at N = %d, D = %d needs %.1f GB of device workspace, above the "
                         "NMGP_SEP_BATCH_SLAB_GB cap of %.0f GB", N, M, per_chain / 1e9, cap_gb);
    std::vector<char> bad(B, 0);
    for (int b0 = 0; b0 < B; b0 += Bc) {
        const int nb = std::min(Bc, B - b0);
        std::vector<char> badc(nb, 0);
        NMGP_TRY(sep_batch_core(c, pars + (size_t)b0 * P, nb, hyper, prior, out6 + (size_t)b0 * 6, want_grad ? grad + (size_t)b0 * P : nullptr,
                                status + b0, badc));
        for (int k = 0; k < nb; ++k) bad[b0 + k] = badc[k];
    }
    // chains that failed numerically: the single-chain entry, with the reference's jitter retries
    for (int b = 0; b < B; ++b)
        if (bad[b]) NMGP_TRY(one_by_one(b));
    c->last_sep_attempts = 0;
    for (int b = 0; b < B; ++b)
        if (status[b] > c->last_sep_attempts) c->last_sep_attempts = status[b];
    return 0;
}

// =================================================================================================
// stationary objective
// =================================================================================================
extern "C" int nmgp_logpos_sta(nmgp_ctx* c, const double* pars, const double hyper[5], int prior, double out5[5],
                               double* grad) {
    if (!c) return NMGP_E_NULL;
    if (!pars || !hyper || !out5) return nmgp_fail(c, NMGP_E_NULL, "pars/hyper/out5 must not be NULL");
    NMGP_TRY(require_data(c));
    HIP_TRY(c, hipSetDevice(c->device));
    const int N = c->N, M = c->M, T = c->T;
    const size_t P = (size_t)T + 3;
    const double mu_l = hyper[0], sd_l = hyper[1], a = hyper[2], b = hyper[3], cc = hyper[4];
    hipStream_t s = c->stream;
    const double tl = pars[0], ts = pars[1], tse = pars[P - 1];
    const double sigma2 = std::exp(tse);
    NMGP_TRY(ensure_eig_buffers(c, N));
    EigWork w;
    build_B(pars + 2, M, true, w.h_L, w.h_B);
    {
        NmgpStage sp(c, NMGP_STAGE_COV);
        // l = exp(tilde_l * ones(N)), sigma = exp(tilde_sigma * ones(N))  (logpos.py:424-425)
        fill_vec(s, c->d_R, N, tl);
        fill_vec(s, c->d_R + N, N, ts);
        exp_vec(s, c->d_R, N, c->d_ell);
        exp_vec(s, c->d_R + N, N, c->d_sig);
    }
    double hs[4], loglik;
    CholKron ck;
    int attempts = 0;
    NMGP_TRY(kron_likelihood_with_retry(c, w, M, N, sigma2, grad != nullptr, [&] {
        NmgpStage sp(c, NMGP_STAGE_COV);
        gibbs_cov_sym(s, c->d_x, c->d_sig, c->d_ell, N, c->d_K, N, false);   // logpos.py:429
    }, hs, &loglik, &ck, &attempts));
    c->last_sep_attempts = attempts;
    double dl = 0.0, lp_l = 0.0, lp_uL = 0.0, lp_s2 = 0.0;
    std::vector<double> g_uL_prior(T, 0.0);
    // the reference only evaluates the prior terms when Prior is true (logpos.py:445-458)
    lp_l = normal_logprob_f32(tl, mu_l, sd_l, &dl);
    for (int t = 0; t < T; ++t) lp_uL += normal_logprob_f32(pars[2 + t], 0.0, cc, &g_uL_prior[t]);
    lp_s2 = (-a - 1.0) * std::log(sigma2) - b / sigma2 + a * std::log(b) - std::lgamma(a);
    double res = 0.0;
    res += loglik;
    if (prior) { res += lp_l; res += lp_uL; res += lp_s2; res += tse; }
    out5[0] = -res; out5[1] = loglik; out5[2] = lp_l; out5[3] = lp_uL; out5[4] = lp_s2;
    if (grad) {
        double* d_g;
        NMGP_TRY(nmgp_scratch_get(c, SL_G, (size_t)2 * N + 8, &d_g));
        std::vector<double> dB, g_uL;
        double ds;
        if (c->sep_algo == 1)
            NMGP_TRY(kron_chol_adjoint(c, w, ck, c->d_ell, c->d_sig, d_g, dB, &ds));
        else
            NMGP_TRY(kron_adjoint(c, w, c->d_ell, c->d_sig, hs, d_g, dB, &ds));
        dB_to_guL(dB, w.h_L, M, g_uL);
        std::vector<double> hg((size_t)2 * N);
        HIP_TRY(c, hipMemcpyAsync(hg.data(), d_g, hg.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        double gtl = 0.0, gts = 0.0;       // the scalar curves move every location together
        for (int i = 0; i < N; ++i) { gtl += hg[i]; gts += hg[N + i]; }
        if (prior) gtl += dl;
        grad[0] = -gtl;
        grad[1] = -gts;
        for (int t = 0; t < T; ++t) grad[2 + t] = -(g_uL[t] + (prior ? g_uL_prior[t] : 0.0));
        double ge = sigma2 * ds;
        if (prior) ge += (-a - 1.0) + b / sigma2 + 1.0;
        grad[P - 1] = -ge;
    }
    NMGP_TRY(nmgp_take_launch_error(c));
    if (!std::isfinite(out5[1])) return nmgp_fail(c, NMGP_NUM_NAN, "non-finite stationary likelihood (%g)", out5[1]);
    return 0;
}

// =================================================================================================
// Kronecker primitives
// =================================================================================================
extern "C" int nmgp_kron_mv(nmgp_ctx* c, const double* B, int m1, int m2, const double* K, int n1, int n2,
                            const double* y, double* out) {
    if (!c) return NMGP_E_NULL;
    if (!B || !K || !y || !out) return nmgp_fail(c, NMGP_E_NULL, "B/K/y/out must not be NULL");
    if (m1 <= 0 || m2 <= 0 || n1 <= 0 || n2 <= 0) return nmgp_fail(c, NMGP_E_SHAPE, "bad shape");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    double *dK, *dy, *dB, *dout;
    NMGP_TRY(nmgp_scratch_get(c, SL_BIG, (size_t)n1 * n2, &dK));
    NMGP_TRY(nmgp_scratch_get(c, SL_Y, (size_t)m2 * n2 + (size_t)m1 * m2 + (size_t)m1 * n1, &dy));
    dB = dy + (size_t)m2 * n2;
    dout = dB + (size_t)m1 * m2;
    HIP_TRY(c, hipMemcpyAsync(dK, K, (size_t)n1 * n2 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(dy, y, (size_t)m2 * n2 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(dB, B, (size_t)m1 * m2 * sizeof(double), hipMemcpyHostToDevice, s));
    {
        NmgpStage sp(c, NMGP_STAGE_KRONMV);
        int r = kron_mv(s, dK, n1, n2, dy, dB, m1, m2, dout);
        if (r) return nmgp_fail(c, r, "kron_mv supports at most 64 columns in B (got %d)", m2);
    }
    HIP_TRY(c, hipMemcpyAsync(out, dout, (size_t)m1 * n1 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return 0;
}

extern "C" int nmgp_mvn_logpdf(nmgp_ctx* c, const double* y, const double* mu, double logdet, const double* inv, int n,
                               double* out) {
    if (!c) return NMGP_E_NULL;
    if (!y || !inv || !out) return nmgp_fail(c, NMGP_E_NULL, "y/invSigma/out must not be NULL");
    if (n <= 0) return nmgp_fail(c, NMGP_E_SHAPE, "n must be positive");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    double *dI, *dv;
    NMGP_TRY(nmgp_scratch_get(c, SL_BIG, (size_t)n * n, &dI));
    NMGP_TRY(nmgp_scratch_get(c, SL_Y, (size_t)4 * n + 8, &dv));
    double *dy = dv, *dmu = dv + n, *dr = dv + 2 * n, *dt = dv + 3 * n, *dres = dv + 4 * n;
    HIP_TRY(c, hipMemcpyAsync(dI, inv, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(dy, y, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
    if (mu) HIP_TRY(c, hipMemcpyAsync(dmu, mu, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
    sub_vec(s, dy, mu ? dmu : nullptr, n, dr);
    const double one = 1.0, zero = 0.0;
    // row-major invSigma times r == (column-major view)^T r   (torch.mv, distributions.py:22)
    BLAS_TRY(c, rocblas_dgemv(c->blas, rocblas_operation_transpose, n, n, &one, dI, n, dr, 1, &zero, dt, 1));
    dot(s, dr, dt, n, dres);
    double h = 0.0;
    HIP_TRY(c, hipMemcpyAsync(&h, dres, sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    *out = -0.5 * logdet - 0.5 * h;
    return 0;
}

// shared front end of the Kronecker-density primitives: uploads K (lower triangle of the column-major view ==
// upper triangle of the row-major matrix, torch.symeig's default) and y - mu.
static int kron_density_setup(nmgp_ctx* c, const double* y, const double* mu, const double* K, int M, int N, double** dr) {
    hipStream_t s = c->stream;
    NMGP_TRY(ensure_eig_buffers(c, N));
    const size_t n = (size_t)M * N;
    double* dv;
    NMGP_TRY(nmgp_scratch_get(c, SL_Y, 3 * n, &dv));
    HIP_TRY(c, hipMemcpyAsync(c->d_K, K, (size_t)N * N * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(dv, y, n * sizeof(double), hipMemcpyHostToDevice, s));
    if (mu) HIP_TRY(c, hipMemcpyAsync(dv + n, mu, n * sizeof(double), hipMemcpyHostToDevice, s));
    sub_vec(s, dv, mu ? dv + n : nullptr, (int)n, dv + 2 * n);
    *dr = dv + 2 * n;
    return 0;
}

extern "C" int nmgp_mvn_logpdf_kron(nmgp_ctx* c, const double* y, const double* mu, const double* B, int M,
                                    const double* K, int N, double sigma2, double* out) {
    if (!c) return NMGP_E_NULL;
    if (!y || !B || !K || !out) return nmgp_fail(c, NMGP_E_NULL, "y/B/K/out must not be NULL");
    if (M <= 0 || N <= 0) return nmgp_fail(c, NMGP_E_SHAPE, "bad shape M=%d N=%d", M, N);
    if (M > 64) return nmgp_fail(c, NMGP_E_UNSUPPORTED, "M=%d > 64 outputs", M);
    HIP_TRY(c, hipSetDevice(c->device));
    double* dr;
    NMGP_TRY(kron_density_setup(c, y, mu, K, M, N, &dr));
    EigWork w;
    jacobi_eigh(M, B, w.h_wB, w.h_VB);
    NMGP_TRY(setup_small(c, w, M, N, sigma2));
    double hs[4], loglik;
    int r = kron_loglik(c, w, dr, sigma2, false, hs, &loglik);
    *out = loglik;
    if (r) return r;
    if (!std::isfinite(loglik)) return nmgp_fail(c, NMGP_NUM_NAN, "non-finite log density");
    return 0;
}

// Builds S = kron(B, K) + sigma2 I (row-major == column-major, symmetric inputs assumed) in scratch SL_BIG2.
static int dense_kron_cov(nmgp_ctx* c, const double* B, int M, const double* K, int N, double sigma2, double** dS) {
    hipStream_t s = c->stream;
    const size_t n = (size_t)M * N;
    double *dB, *dK;
    NMGP_TRY(nmgp_scratch_get(c, SL_X, (size_t)M * M, &dB));
    NMGP_TRY(nmgp_scratch_get(c, SL_BIG, (size_t)N * N, &dK));
    NMGP_TRY(nmgp_scratch_get(c, SL_BIG2, n * n, dS));
    HIP_TRY(c, hipMemcpyAsync(dB, B, (size_t)M * M * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(dK, K, (size_t)N * N * sizeof(double), hipMemcpyHostToDevice, s));
    kron_product(s, dB, M, M, dK, N, N, *dS);
    add_diag(s, *dS, (int)n, (int)n, sigma2);
    return 0;
}

extern "C" int nmgp_mvn_logpdf_dense(nmgp_ctx* c, const double* y, const double* mu, const double* B, int M,
                                     const double* K, int N, double sigma2, double* out) {
    if (!c) return NMGP_E_NULL;
    if (!y || !B || !K || !out) return nmgp_fail(c, NMGP_E_NULL, "y/B/K/out must not be NULL");
    if (M <= 0 || N <= 0) return nmgp_fail(c, NMGP_E_SHAPE, "bad shape M=%d N=%d", M, N);
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int n = M * N;
    double *dS, *dv;
    NMGP_TRY(dense_kron_cov(c, B, M, K, N, sigma2, &dS));
    NMGP_TRY(nmgp_scratch_get(c, SL_Y, (size_t)3 * n + 8, &dv));
    HIP_TRY(c, hipMemcpyAsync(dv, y, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
    if (mu) HIP_TRY(c, hipMemcpyAsync(dv + n, mu, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
    double* dr = dv + 2 * (size_t)n;
    sub_vec(s, dv, mu ? dv + n : nullptr, n, dr);
    HIP_TRY(c, hipMemsetAsync(c->d_info + 4, 0, sizeof(int), s));
    NMGP_TRY(nmgp_chol_factor(c, dS, n, n, 0, c->d_info + 4));
    BLAS_TRY(c, rocblas_dtrsv(c->blas, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, n, dS, n,
                              dr, 1));
    double* dres = dv + 3 * (size_t)n;
    chol_logdet_quad(s, dS, n, n, dr, dres, dres + 1);
    double h[2];
    HIP_TRY(c, hipMemcpyAsync(h, dres, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(c->h_info + 4, c->d_info + 4, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (c->h_info[4] != 0)
        return nmgp_fail(c, c->h_info[4], "B kron K + sigma2 I is not positive definite (leading minor %d)", c->h_info[4]);
    *out = -0.5 * h[0] - 0.5 * h[1];
    return 0;
}

extern "C" int nmgp_kron_inv_logdet(nmgp_ctx* c, double sigma2, const double* B, int M, const double* K, int N,
                                    double* out_inv, double* out_logdet) {
    if (!c) return NMGP_E_NULL;
    if (!B || !K) return nmgp_fail(c, NMGP_E_NULL, "B/K must not be NULL");
    if (M <= 0 || N <= 0) return nmgp_fail(c, NMGP_E_SHAPE, "bad shape M=%d N=%d", M, N);
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const size_t n = (size_t)M * N;
    NMGP_TRY(ensure_eig_buffers(c, N));
    HIP_TRY(c, hipMemcpyAsync(c->d_K, K, (size_t)N * N * sizeof(double), hipMemcpyHostToDevice, s));
    EigWork w;
    jacobi_eigh(M, B, w.h_wB, w.h_VB);
    NMGP_TRY(setup_small(c, w, M, N, sigma2));
    NMGP_TRY(eig_K(c, N));
    NMGP_TRY(check_eig_info(c));
    double* dw;
    NMGP_TRY(nmgp_scratch_get(c, SL_Y, n + 8, &dw));
    kron_w(s, w.wB, M, w.wK, N, sigma2, dw);                      // 1 / (t + sigma2), kronecker_operation.py:52
    if (out_logdet) {
        // sum log(t + sigma2) = - sum log w   (kronecker_operation.py:69)
        std::vector<double> hw(n);
        HIP_TRY(c, hipMemcpyAsync(hw.data(), dw, n * sizeof(double), hipMemcpyDeviceToHost, s));
        std::vector<double> hk(N);
        HIP_TRY(c, hipMemcpyAsync(hk.data(), w.wK, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        double acc = 0.0;
        for (int p = 0; p < M; ++p)
            for (int q = 0; q < N; ++q) acc += std::log(w.h_wB[p] * hk[q] + sigma2);
        *out_logdet = acc;
    }
    if (out_inv) {
        double *U, *Us, *C;
        NMGP_TRY(nmgp_scratch_get(c, SL_BIG, n * n, &U));
        NMGP_TRY(nmgp_scratch_get(c, SL_BIG2, n * n, &Us));
        NMGP_TRY(nmgp_scratch_get(c, SL_K3, n * n, &C));
        kron_eigvec(s, w.VB, M, w.V, N, U);                       // U = V_B kron V_K, row-major (kronecker_operation.py:50)
        // row-major U with columns scaled by w: Us_rm[r, k] = U_rm[r, k] w[k]  ==  column-major rows scaled.
        // C = U_rm diag(w) U_rm^T = (U_cm^T) diag(w) (U_cm): scale the ROWS of the column-major view.
        // rocBLAS has no row-scaling; use dgemm on the transposes: build D = diag(w) implicitly through dgmm.
        BLAS_TRY(c, rocblas_ddgmm(c->blas, rocblas_side_left, (int)n, (int)n, U, (int)n, dw, 1, Us, (int)n));
        const double one = 1.0, zero = 0.0;
        BLAS_TRY(c, rocblas_dgemm(c->blas, rocblas_operation_transpose, rocblas_operation_none, (int)n, (int)n, (int)n,
                                  &one, U, (int)n, Us, (int)n, &zero, C, (int)n));
        HIP_TRY(c, hipMemcpyAsync(out_inv, C, n * n * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
    }
    return 0;
}

// =================================================================================================
// deterministic prediction
// =================================================================================================
// GP-regression of the latent curves at new inputs (prediction.py:926-941): proj[s, k] = (Sigma^-1 k*(xs_s))^T r_k with r_k
// the k-th column of R (already value - mean).  Kstar is built as RBF(xs, x) row-major [S, N] == column-major [N, S].
// proj: column-major [S, ncol].
// The ORDER is the reference's (proj_l = solve(Sigma_l, k_l), then the dot product with the curve): Sigma = RBF + 1e-6 I has a
// condition number of ~1e11, k* lies in its smooth eigenspace and Sigma^-1 k* is a vector of moderate size, whereas
// Sigma^-1 r amplifies whatever roughness the curve r has by 1e6 and leaves the result to cancellation in k*^T (Sigma^-1 r):
// solving for r first put L* 3.5e-6 away from the reference at N = 512 (this order: the level of the CPU oracle).
static int gp_project(nmgp_ctx* c, PriorFactor* pf, const double* d_xs, int S, const double* R /*[N,ncol]*/, int ncol,
                      double* proj) {
    const int N = c->N;
    hipStream_t s = c->stream;
    const double one = 1.0, zero = 0.0;
    double* Ks;
    NMGP_TRY(nmgp_scratch_get(c, SL_X, (size_t)N * S, &Ks));
    rbf_cov_rect(s, d_xs, S, c->d_x, N, 1, pf->alpha, pf->beta, false, Ks);
    // W = Sigma^-1 K* through the cached Cholesky factor (S right-hand sides).  By substitution (k_prior_trsv: true divisions,
    // one workgroup per right-hand side): the library's trsm multiplies by INVERTED 128 x 128 diagonal blocks, which costs
    // digits on a factor of condition number ~1e5.5 (NMGP_PRIOR_SOLVE=rocblas selects it; sizes beyond the kernel's LDS too)
    if (N <= 15000 && !c->prior_rocblas) {        // (the right-hand side lives in LDS: 8 N + 34 KB of the CU's 160 KB)
        prior_trsv(s, false, pf->L, pf->ld, 0, pf->L, pf->ld, 0, Ks, N, S, 1);
        prior_trsv(s, true, pf->L, pf->ld, 0, pf->L, pf->ld, 0, Ks, N, S, 1);
    } else {
        BLAS_TRY(c, rocblas_dtrsm(c->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none,
                                  rocblas_diagonal_non_unit, N, S, &one, pf->L, pf->ld, Ks, N));
        BLAS_TRY(c, rocblas_dtrsm(c->blas, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose,
                                  rocblas_diagonal_non_unit, N, S, &one, pf->L, pf->ld, Ks, N));
    }
    BLAS_TRY(c, rocblas_dgemm(c->blas, rocblas_operation_transpose, rocblas_operation_none, S, ncol, N, &one, Ks, N, R,
                              N, &zero, proj, S));
    return 0;
}

// 0: the last separable / stationary evaluation is that of the exact covariance; k > 0: value and gradient belong to the
// covariance with k x 1e-6 added to the diagonals of B and K_x (the deterministic stand-in for the reference's random-jitter
// retry, logpos.py:267-268)
extern "C" int nmgp_last_sep_attempts(const nmgp_ctx* c) { return c ? c->last_sep_attempts : -1; }

extern "C" int nmgp_predict_svc(nmgp_ctx* c, const double* pars, const double hyper[8], const double* xs, int S,
                                double* mean, double* var, double* Lstar) {
    if (!c) return NMGP_E_NULL;
    if (!pars || !hyper || !xs || !mean || !var) return nmgp_fail(c, NMGP_E_NULL, "null argument");
    if (S <= 0) return nmgp_fail(c, NMGP_E_SHAPE, "S must be positive");
    NMGP_TRY(require_data(c));
    HIP_TRY(c, hipSetDevice(c->device));
    const int N = c->N, M = c->M, T = c->T, n = c->n;
    const long long P = c->P_svc;
    hipStream_t s = c->stream;
    const double mu_l = hyper[0], al_l = hyper[1], be_l = hyper[2], mu_L = hyper[3], al_L = hyper[4], be_L = hyper[5];
    NMGP_TRY(nmgp_ensure_S(c));
    const int ld = c->ldS;
    HIP_TRY(c, hipMemcpyAsync(c->d_pars, pars, (size_t)P * sizeof(double), hipMemcpyHostToDevice, s));
    PriorFactor *pl = nullptr, *pL = nullptr;
    NMGP_TRY(nmgp_get_prior(c, al_l, be_l, &pl));
    NMGP_TRY(nmgp_get_prior(c, al_L, be_L, &pL));
    NMGP_TRY(nmgp_get_prior(c, al_l, be_l, &pl));
    double* sm;
    NMGP_TRY(nmgp_scratch_get(c, SL_Y, (size_t)S * (3 + 2 * T + 3 * M) + 16, &sm));      // xs | proj | tl* | L* | mean | colsq | var
    double* d_xs = sm;
    double* proj = d_xs + S;                      // [S, 1+T]
    double* tl_star = proj + (size_t)S * (1 + T);
    double* Ls = tl_star + S;                     // [S, T]
    double* d_mean = Ls + (size_t)S * T;          // [S*M]
    double* d_colsq = d_mean + (size_t)S * M;
    double* d_var = d_colsq + (size_t)S * M;
    HIP_TRY(c, hipMemcpyAsync(d_xs, xs, (size_t)S * sizeof(double), hipMemcpyHostToDevice, s));
    svc_prior_rhs(s, c->d_pars, N, T, mu_l, mu_L, c->d_R, N);
    if (pl == pL) {
        NMGP_TRY(gp_project(c, pl, d_xs, S, c->d_R, 1 + T, proj));
    } else {
        NMGP_TRY(gp_project(c, pl, d_xs, S, c->d_R, 1, proj));
        NMGP_TRY(gp_project(c, pL, d_xs, S, c->d_R + N, T, proj + S));
    }
    svc_star(s, proj, S, M, mu_l, mu_L, tl_star, Ls);
    // Sigma with y AND the S M cross-covariance vectors as extra rows below it: one blocked factorisation turns row r into
    // r L^-T, i.e. z = L^-1 y and v_e = L^-1 k_e for every grid point and output at once -- mean_e = v_e . z (= k_e^T Sigma^-1 y,
    // prediction.py:973), |v_e|^2 = the diagonal of T T^T (:975-977).  (Until round 3: accuracy-first substitution factorisation,
    // two library dtrsv for Sigma^-1 y and a 3 S-column dtrsm: 16.2 ms at N = 2048, S = 201.)  The buffer has room for n extra
    // rows (the gradient path's L^-T block); more grid outputs than that go through in slices of (n - 2) / M grid points.
    svc_prep(s, c->d_pars, N, M, c->d_ell, c->d_Lv);
    const int smax = std::max(1, (n - 2) / M);              // grid points per factorisation
    double* part;
    NMGP_TRY(nmgp_scratch_get(c, SL_BIG, (size_t)2 * std::min(S, smax) * M * ((n + 127) / 128), &part));
    HIP_TRY(c, hipMemsetAsync(c->d_info, 0, sizeof(int), s));
    for (int s0 = 0; s0 < S; s0 += smax) {
        const int Sc = std::min(smax, S - s0), E = Sc * M;
        int r = svc_cov_build(s, c->d_x, c->d_ell, c->d_Lv, c->d_pars + (P - 1), c->d_S, ld, N, M, false);
        if (r) return nmgp_fail(c, r, "unsupported number of outputs M=%d", M);
        set_row(s, c->d_S, ld, n, c->d_y, n, 1, 0, 0);
        svc_crosscov_rows(s, c->d_x, c->d_ell, c->d_Lv, N, M, d_xs + s0, tl_star + s0, Ls + (size_t)s0 * T, Sc, c->d_S, ld, n + 1);
        if (c->chol_algo == 1) {
            potrf_lower(s, c->stream2, nmgp_chol_events(c, n), c->d_S, ld, n, 1 + E, 0, c->chol_nb1, c->d_info, 1, 0, 0,
                        nmgp_syrk_hook(c));
        } else {
            // comparison path (NMGP_CHOL=rocsolver): library factorisation, the extra rows solved as right-hand sides X L^T = R
            const double one = 1.0;
            BLAS_TRY(c, rocsolver_dpotrf(c->blas, rocblas_fill_lower, n, c->d_S, ld, c->d_info));
            BLAS_TRY(c, rocblas_dtrsm(c->blas, rocblas_side_right, rocblas_fill_lower, rocblas_operation_transpose,
                                      rocblas_diagonal_non_unit, 1 + E, n, &one, c->d_S, ld, c->d_S + n, ld));
        }
        pred_rows_reduce(s, c->d_S, ld, n, n + 1, n, E, part, d_mean + (size_t)s0 * M, d_colsq + (size_t)s0 * M);
    }
    svc_predvar(s, Ls, d_colsq, S, M, c->d_pars + (P - 1), d_var);
    HIP_TRY(c, hipMemcpyAsync(mean, d_mean, (size_t)S * M * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(var, d_var, (size_t)S * M * sizeof(double), hipMemcpyDeviceToHost, s));
    if (Lstar) HIP_TRY(c, hipMemcpyAsync(Lstar, Ls, (size_t)S * T * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(c->h_info, c->d_info, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (c->h_info[0] != 0)
        return nmgp_fail(c, c->h_info[0], "covariance not positive definite (leading minor %d)", c->h_info[0]);
    c->last_kind = 0;
    return 0;
}

// shared by the separable and stationary predictors once K's lower triangle is in c->d_K and the starred
// quantities are on the device
static int eig_predict(nmgp_ctx* c, EigWork& w, double sigma2, int mode, const double* d_xs, const double* tl_star,
                       const double* ts_star, double sig0, double l0, const double* d_kss, bool strict_clip, int S,
                       double* mean, double* var) {
    const int N = c->N, M = c->M;
    hipStream_t s = c->stream;
    double hs[4], loglik;
    NMGP_TRY(kron_loglik(c, w, c->d_y, sigma2, false, hs, &loglik));       // a = projection of y (prediction.py:388)
    double *KX, *Cq, *sm;
    NMGP_TRY(nmgp_scratch_get(c, SL_BIG, (size_t)N * S, &KX));
    NMGP_TRY(nmgp_scratch_get(c, SL_BIG2, (size_t)N * S, &Cq));
    NMGP_TRY(nmgp_scratch_get(c, SL_U, (size_t)2 * S * M + M, &sm));
    double *d_mean = sm, *d_var = sm + (size_t)S * M, *d_Bdiag = d_var + (size_t)S * M;
    std::vector<double> bd(M);
    for (int m = 0; m < M; ++m) bd[m] = w.h_B[(size_t)m * M + m];
    HIP_TRY(c, hipMemcpyAsync(d_Bdiag, bd.data(), M * sizeof(double), hipMemcpyHostToDevice, s));
    sep_crossvec(s, mode, c->d_x, c->d_sig, c->d_ell, N, d_xs, tl_star, ts_star, sig0, l0, S, KX);
    const double one = 1.0, zero = 0.0;
    BLAS_TRY(c, rocblas_dgemm(c->blas, rocblas_operation_transpose, rocblas_operation_none, N, S, N, &one, w.V, N, KX, N,
                              &zero, Cq, N));
    sep_predict(s, Cq, w.a, w.wB, w.VB, M, w.wK, N, sigma2, d_Bdiag, d_kss, strict_clip, S, d_mean, d_var);
    HIP_TRY(c, hipMemcpyAsync(mean, d_mean, (size_t)S * M * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(var, d_var, (size_t)S * M * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return 0;
}

// The same predictor through the Cholesky formulation (the default of the objectives, NMGP_SEP=eig selects the eigen one above):
// M blocks S_p = wB[p] K + sigma2 I as ONE batch of the blocked factorisation, the rotated data yt_p and the S cross-covariance
// vectors riding below every block as extra rows (rows become r L_p^-T) -- no eigendecomposition of K_x (133 ms at N = 4096), no
// triangular solve.  c->d_K holds the lower triangle of K_x and survives.
static int chol_predict(nmgp_ctx* c, EigWork& w, double sigma2, int mode, const double* d_xs, const double* tl_star,
                        const double* ts_star, double sig0, double l0, const double* d_kss, bool strict_clip, int S, double* mean,
                        double* var) {
    const int N = c->N, M = c->M;
    hipStream_t s = c->stream;
    const int smax = std::max(1, N - 2), Sm = std::min(S, smax);
    const int ld = (int)((((size_t)N + 1 + Sm + 15) / 16) * 16);
    const long long bs = (long long)ld * N;
    double *KX, *Sbuf, *sm, *part;
    NMGP_TRY(nmgp_scratch_get(c, SL_BIG2, (size_t)N * S, &KX));
    NMGP_TRY(nmgp_scratch_get(c, SL_BIG, (size_t)M * bs, &Sbuf));
    NMGP_TRY(nmgp_scratch_get(c, SL_U, (size_t)2 * S * M + M + (size_t)2 * M * S + 16, &sm));
    NMGP_TRY(nmgp_scratch_get(c, SL_PART, (size_t)2 * Sm * ((N + 127) / 128), &part));
    double *d_mean = sm, *d_var = sm + (size_t)S * M, *d_Bdiag = d_var + (size_t)S * M, *dots = d_Bdiag + M, *sqs = dots + (size_t)M * S;
    int* info = reinterpret_cast<int*>(sqs + (size_t)M * S);
    std::vector<double> bd(M);
    for (int m = 0; m < M; ++m) bd[m] = w.h_B[(size_t)m * M + m];
    HIP_TRY(c, hipMemcpyAsync(d_Bdiag, bd.data(), M * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemsetAsync(info, 0, (size_t)M * sizeof(int), s));
    sep_crossvec(s, mode, c->d_x, c->d_sig, c->d_ell, N, d_xs, tl_star, ts_star, sig0, l0, S, KX);
    double* yt = w.a;                                              // [M, N]: yt_p = (V_B^T kron I) y
    rotate_y(s, c->d_Y, w.VB, N, M, yt);
    for (int s0 = 0; s0 < S; s0 += smax) {
        const int Sc = std::min(smax, S - s0);
        sep_blocks(s, c->d_K, w.wB, w.sig2, N, M, Sbuf, ld, bs);
        set_row(s, Sbuf, ld, N, yt, N, M, bs, N);
        cols_to_rows(s, KX + (size_t)s0 * N, N, Sc, Sbuf, ld, N + 1, M, bs);
        potrf_lower(s, c->stream2, nmgp_chol_events(c, N), Sbuf, ld, N, 1 + Sc, 0, c->chol_nb1, info, M, bs, 1, nmgp_syrk_hook(c));
        for (int p = 0; p < M; ++p)
            pred_rows_reduce(s, Sbuf + (size_t)p * bs, ld, N, N + 1, N, Sc, part, dots + (size_t)p * S + s0, sqs + (size_t)p * S + s0);
    }
    sep_predict_chol(s, dots, sqs, w.wB, w.VB, M, sigma2, d_Bdiag, d_kss, strict_clip, S, d_mean, d_var);
    std::vector<int> hi(M);
    HIP_TRY(c, hipMemcpyAsync(mean, d_mean, (size_t)S * M * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(var, d_var, (size_t)S * M * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(hi.data(), info, (size_t)M * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    NMGP_TRY(nmgp_take_launch_error(c));
    for (int p = 0; p < M; ++p)
        if (hi[p] != 0)
            return nmgp_fail(c, hi[p], "block %d of the separable covariance is not positive definite (leading minor %d)", p, hi[p]);
    return 0;
}

extern "C" int nmgp_predict_sep(nmgp_ctx* c, const double* pars, const double hyper[9], const double* xs, int S,
                                double* mean, double* var) {
    if (!c) return NMGP_E_NULL;
    if (!pars || !hyper || !xs || !mean || !var) return nmgp_fail(c, NMGP_E_NULL, "null argument");
    if (S <= 0) return nmgp_fail(c, NMGP_E_SHAPE, "S must be positive");
    NMGP_TRY(require_data(c));
    HIP_TRY(c, hipSetDevice(c->device));
    const int N = c->N, M = c->M, T = c->T;
    const size_t P = (size_t)2 * N + T + 1;
    hipStream_t s = c->stream;
    const double mu_l = hyper[0], al_l = hyper[1], be_l = hyper[2], mu_s = hyper[3], al_s = hyper[4], be_s = hyper[5];
    const double sigma2 = std::exp(pars[P - 1]);
    NMGP_TRY(ensure_eig_buffers(c, N));
    HIP_TRY(c, hipMemcpyAsync(c->d_pars, pars, P * sizeof(double), hipMemcpyHostToDevice, s));
    PriorFactor *pl = nullptr, *ps = nullptr;
    NMGP_TRY(nmgp_get_prior(c, al_l, be_l, &pl));
    NMGP_TRY(nmgp_get_prior(c, al_s, be_s, &ps));
    NMGP_TRY(nmgp_get_prior(c, al_l, be_l, &pl));
    double* sm;
    NMGP_TRY(nmgp_scratch_get(c, SL_Y, (size_t)S * 6 + 16, &sm));
    double *d_xs = sm, *proj = sm + S, *tl_star = proj + 2 * (size_t)S, *ts_star = tl_star + S, *d_kss = ts_star + S;
    HIP_TRY(c, hipMemcpyAsync(d_xs, xs, (size_t)S * sizeof(double), hipMemcpyHostToDevice, s));
    two_col_rhs(s, c->d_pars, mu_l, c->d_pars + N, mu_s, N, c->d_R);
    if (pl == ps) {
        NMGP_TRY(gp_project(c, pl, d_xs, S, c->d_R, 2, proj));
    } else {
        NMGP_TRY(gp_project(c, pl, d_xs, S, c->d_R, 1, proj));
        NMGP_TRY(gp_project(c, ps, d_xs, S, c->d_R + N, 1, proj + S));
    }
    sep_star(s, proj, S, mu_l, mu_s, tl_star, ts_star, d_kss);
    EigWork w;
    build_B(pars + 2 * N, M, true, w.h_L, w.h_B);
    jacobi_eigh(M, w.h_B.data(), w.h_wB, w.h_VB);
    NMGP_TRY(setup_small(c, w, M, N, sigma2));
    exp_vec(s, c->d_pars, N, c->d_ell);
    exp_vec(s, c->d_pars + N, N, c->d_sig);
    gibbs_cov_sym(s, c->d_x, c->d_sig, c->d_ell, N, c->d_K, N, false);
    return (c->sep_algo == 1 ? chol_predict : eig_predict)(c, w, sigma2, 0, d_xs, tl_star, ts_star, 0.0, 1.0, d_kss, false, S, mean, var);
}

extern "C" int nmgp_predict_sta(nmgp_ctx* c, const double* pars, const double* xs, int S, double* mean, double* var) {
    if (!c) return NMGP_E_NULL;
    if (!pars || !xs || !mean || !var) return nmgp_fail(c, NMGP_E_NULL, "null argument");
    if (S <= 0) return nmgp_fail(c, NMGP_E_SHAPE, "S must be positive");
    NMGP_TRY(require_data(c));
    HIP_TRY(c, hipSetDevice(c->device));
    const int N = c->N, M = c->M, T = c->T;
    hipStream_t s = c->stream;
    const double l0 = std::exp(pars[0]), sig0 = std::exp(pars[1]);
    const double sigma2 = std::exp(pars[T + 2]);
    NMGP_TRY(ensure_eig_buffers(c, N));
    double* sm;
    NMGP_TRY(nmgp_scratch_get(c, SL_Y, (size_t)S * 2 + 16, &sm));
    double *d_xs = sm, *d_kss = sm + S;
    HIP_TRY(c, hipMemcpyAsync(d_xs, xs, (size_t)S * sizeof(double), hipMemcpyHostToDevice, s));
    fill_vec(s, d_kss, S, sig0 * sig0);                                   // sigma**2 * diag(B_f)  (prediction.py:1593)
    EigWork w;
    build_B(pars + 2, M, true, w.h_L, w.h_B);
    jacobi_eigh(M, w.h_B.data(), w.h_wB, w.h_VB);
    NMGP_TRY(setup_small(c, w, M, N, sigma2));
    rbf_cov_sym(s, c->d_x, N, sig0, l0, c->d_K, N, false);                // RBF_cov(x, alpha=sigma, beta=l) (:1587)
    return (c->sep_algo == 1 ? chol_predict : eig_predict)(c, w, sigma2, 1, d_xs, nullptr, nullptr, sig0, l0, d_kss, true, S, mean, var);
}
