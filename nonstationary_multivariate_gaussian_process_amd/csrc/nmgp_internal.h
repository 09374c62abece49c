// Internal declarations of libnmgp_hip.so (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "nmgp.h"

#define NMGP_JITTER 1e-6     // Utility/settings.py:3
#define NMGP_PRECISION 1e-6  // Utility/settings.py:6

struct DevBuf {
    double* p = nullptr;
    size_t cap = 0;  // elements
};

// Cholesky factor of an N x N GP-prior covariance RBF(x; alpha, beta) + jitter I, cached per (alpha, beta)
// for the current x (logpos.py:357,362 rebuild and re-factorise it on every evaluation).
struct PriorFactor {
    double alpha = 0, beta = 0;
    double* L = nullptr;       // N x N, lower, column-major, leading dimension ld
    int ld = 0;
    double* logdet = nullptr;  // device scalar: log det of the covariance
};

struct StageTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    double ms = 0;
    long long count = 0;
    double work = 0;   // algorithmic flop of the timed launches, where tracked
    double bytes = 0;  // algorithmic HBM bytes of the timed launches, where tracked
};

// optional per-launch profiling of k_syrk_lower: begin() before / end() after each launch, on the launch's stream
struct SyrkHook {
    void* user = nullptr;
    void* (*begin)(void* user, hipStream_t s, double flop, double bytes) = nullptr;
    void (*end)(void* user, void* token) = nullptr;
};

struct nmgp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;           // look-ahead stream of the custom factorisation (CU-masked, see nmgp_ctx_create)
    int stream2_cus = 0;                     // CUs stream2 may use (0 = no mask)
    // The GP-prior solves depend on the parameter vector only: they run on their own stream under the factorisation, whose
    // latency-bound panel steps leave most of the chip idle (NMGP_PRIOR_OVERLAP=0: on the main stream, after the factorisation)
    hipStream_t stream_prior = nullptr;
    rocblas_handle blas_prior = nullptr;      // its own handle (own workspace): no stream switching on the main one
    hipEvent_t ev_prior_fork = nullptr, ev_prior_join = nullptr;
    int prior_overlap = 1;
    std::vector<hipEvent_t> chol_ev;          // events ordering the two streams
    int chol_lookahead = 1;                   // far trailing update on stream2 under the next panel (small batches only)
    SyrkHook syrk_hook;
    int sep_algo = 1;                         // separable/stationary likelihood: 1 = M batched Cholesky blocks, 0 = dsyevd
    rocblas_handle blas = nullptr;
    std::string err;

    // subject data
    int N = 0, M = 0, T = 0, n = 0;
    double* d_x = nullptr;   // [N]
    double* d_Y = nullptr;   // [N, M] row-major
    double* d_y = nullptr;   // [n] output-major
    // nonseparable state
    long long P_svc = 0;
    double* d_pars = nullptr;   // [P]  (largest of the three parameter layouts)
    double* d_grad = nullptr;   // [P]
    double* d_ell = nullptr;    // [N]    exp(tilde_l)
    double* d_sig = nullptr;    // [N]    exp(tilde_sigma) (separable)
    double* d_Lv = nullptr;     // [N, T] packed tril factors, exp applied on the diagonal slots
    double* d_S = nullptr;      // [ldS, n] covariance / factor (lower, column-major) + rows for y and for L^-T
    double* d_Sinv = nullptr;   // [n, n] -Sigma^-1 (gradient path of the custom factorisation)
    size_t S_cap = 0;
    int ldS = 0;
    double* d_z = nullptr;      // [n]   L^-1 y, then alpha = Sigma^-1 y
    double* d_alpha = nullptr;  // [n]
    double* d_R = nullptr;      // [N, 1+T] prior right-hand sides / solutions (column-major)
    double* d_R2 = nullptr;     // [N, 1+T] Sigma_prior^-1 r
    double* d_scal = nullptr;   // small device scalars (see SC_* in nmgp_api.hip)
    double* d_part = nullptr;   // partial sums of the adjoint pass
    size_t part_cap = 0;
    int* d_info = nullptr;      // rocSOLVER status words
    double* h_pin = nullptr;    // pinned host staging (scalars)
    int* h_info = nullptr;
    // generic scratch for the primitive entry points
    DevBuf scratch[16];
    std::vector<PriorFactor> priors;
    // eigen path
    double* d_K = nullptr;      // [N, N] K_x then eigenvectors (separable / stationary)
    double* d_K2 = nullptr;     // [N, N] second N x N workspace
    size_t K_cap = 0;
    double* d_w = nullptr;      // [N] eigenvalues of K_x
    double* d_E = nullptr;      // [N] syevd workspace
    // batched value-only evaluation: B chains of the resident subject, one launch sequence for all of them
    int batch = 0;
    double* b_pars = nullptr;   // [B, P]
    double* b_ell = nullptr;    // [B, N]
    double* b_Lv = nullptr;     // [B, N, T]
    double* b_S = nullptr;      // [B] x (ld x n) covariance / factor buffers
    double* b_z = nullptr;      // [B, n]
    double* b_R = nullptr;      // [N, B (1 + T)]
    double* b_scal = nullptr;   // [B, 16] : logdet, quad, out5...
    double* b_q = nullptr;      // [B (1 + T)]
    int* b_info = nullptr;      // [B]
    // batched gradient state (allocated on the first batched value+gradient evaluation)
    bool b_grad_ready = false;
    double* b_S2 = nullptr;     // [B] x (ld2 x n): covariance + y row + pad + identity rows -> L^-T
    double* b_Sinv = nullptr;   // [B] x (n x n): -Sigma^-1
    double* b_alpha = nullptr;  // [B, n]
    double* b_part = nullptr;   // [B, NJ, N, 1+T]
    double* b_grad = nullptr;   // [B, P]
    double* b_R2 = nullptr;     // [N, B (1+T)]
    double* b_tr = nullptr;     // [B, 2]
    bool b_last_grad = false;
    // device-resident HMC trajectories (nmgp_svc_batch_traj_*): momenta, the state before the trajectory, validity flags
    double* b_mom = nullptr;    // [B, P]
    double* b_q0 = nullptr;     // [B, P]
    double* b_g0 = nullptr;     // [B, P]
    int* b_hmc = nullptr;       // [4, B]: bad (current position), bad0 (before the trajectory), failed, accept
    bool b_traj_ready = false;  // nmgp_svc_batch_traj_begin ran on the current state
    // device-resident Adam (nmgp_svc_batch_adam_*): moments and the alive flags of the subjects
    double* b_am = nullptr;     // [B, P]
    double* b_av = nullptr;     // [B, P]
    int* b_alive = nullptr;     // [B]
    long long b_adam_t = -1;    // iterations done; -1 = not started
    // multi-subject batch: every batch element has its own (x, Y) and its own prior factors
    bool b_multi = false;
    double* b_x = nullptr;      // [B, N]
    double* b_y = nullptr;      // [B, n] output-major
    std::vector<PriorFactor> b_priors;   // L: [B] x (ld x N), logdet: [B]
    int b_cps = 1;              // chains per subject of a multi-subject batch: batch element z belongs to subject z / b_cps
    int b_mass_kind = 0;        // mass matrix of the device-resident trajectories: 0 identity, 1 diagonal, 2 dense (b_minv = M^-1),
                                // 3 prior-factor metric M^-1 = L_blk (I + U diag(lam) U^T)^-1 L_blk^T (nmgp_metric.hip; b_mom then holds
                                // the WHITENED momentum u = L_blk^T p)
    double* b_minv = nullptr;   // [P] or [P, P]
    double* b_vel = nullptr;    // [B, P] velocities M^-1 p (dense mass matrix)
    double* b_mchol = nullptr;  // chol(M): [P] (diagonal: sqrt) or [P, P] lower, column-major -- momenta p = chol(M) z drawn on the device
    double* b_kin = nullptr;    // [B] kinetic energies 1/2 p^T M^-1 p at the end of a trajectory
    // prior-factor metric (kind 3): GP-prior hyper-parameters of the factors, rank-r likelihood correction per subject
    double b_mhyp[4] = {0, 0, 0, 0};   // alpha_tilde_l, beta_tilde_l, alpha_L, beta_L
    int b_mrank = 0;
    double* b_mU = nullptr;     // [S, r, P]: subject s's orthonormal directions as r rows of length P
    double* b_mw = nullptr;     // [3, S, r]: sqrt(1 + lam) - 1 (draw), -lam / (1 + lam) (velocity), lam / (1 + lam) (kinetic energy)
    double* b_mc = nullptr;     // [B, r] projections U^T u
    int last_sep_attempts = 0;  // jitter retries the last separable / stationary evaluation needed (0 = the exact covariance)
    bool last_want_grad = false;
    int last_kind = 0;          // 1 svc

    int chol_algo = 1;          // 0 = rocSOLVER dpotrf + rocBLAS dtrsv, 1 = custom blocked factorisation (nmgp_chol.hip)
    bool prior_rocblas = false;               // NMGP_PRIOR_SOLVE=rocblas: library trsm for every GP-prior solve
    bool prior_trsv_all = false;              // NMGP_PRIOR_SOLVE=trsv: every GP-prior solve of the objectives by substitution
                                              // (k_prior_trsv), also one subject's right-hand sides, which default to the library
                                              // (measured: identical parity figures -- the prior terms' distance to the reference is
                                              // the FACTOR's conditioning, not the solve -- and 4 ms slower per 128-chain step).
                                              // Prediction always solves by substitution: there the library's inverted diagonal
                                              // blocks cost two digits (gp_project)
    int chol_nb1 = 0;           // outer panel width of the custom factorisation; 0 = auto (1024 for batches of large matrices, else 512)
    int profiling = 0;                        // 0 off, 1 stage timers, 2 + one event pair per k_syrk_lower launch
    StageTimer timers[NMGP_STAGE_COUNT];
};

int nmgp_fail(nmgp_ctx* ctx, int code, const char* fmt, ...);
// helpers shared between the translation units (defined in nmgp_api.hip)
int nmgp_dev_alloc(nmgp_ctx* c, double** p, size_t nelem);
int nmgp_scratch_get(nmgp_ctx* c, int slot, size_t nelem, double** out);
int nmgp_get_prior(nmgp_ctx* c, double alpha, double beta, PriorFactor** out);
int nmgp_ensure_S(nmgp_ctx* c);
// Cholesky of the n x n lower triangle (custom gfx950 factorisation or rocSOLVER, per ctx->chol_algo); `extra` rows
// below the matrix are carried along by the custom path only (must be 0 for rocSOLVER).
int nmgp_chol_factor(nmgp_ctx* c, double* A, int ld, int n, int extra, int* d_info);
bool nmgp_poison();
hipEvent_t* nmgp_chol_events(nmgp_ctx* c, int n);
struct NmgpStage {   // RAII HIP-event timer of one stage on the context's stream (or on an explicit stream)
    nmgp_ctx* c; int stage; hipStream_t stream; hipEvent_t e0 = nullptr, e1 = nullptr;
    NmgpStage(nmgp_ctx* ctx, int st);
    NmgpStage(nmgp_ctx* ctx, int st, hipStream_t s, double work, double bytes);
    ~NmgpStage();
};

#define HIP_TRY(ctx, expr)                                                                        \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            return nmgp_fail(ctx, NMGP_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                             __FILE__, __LINE__);                                                 \
    } while (0)

#define BLAS_TRY(ctx, expr)                                                                       \
    do {                                                                                          \
        rocblas_status s__ = (expr);                                                              \
        if (s__ != rocblas_status_success)                                                        \
            return nmgp_fail(ctx, NMGP_E_HIP, "%s failed: rocblas_status %d (%s:%d)", #expr, (int)s__, \
                             __FILE__, __LINE__);                                                 \
    } while (0)

// Kernel launches go through NMGP_LAUNCH: a launch the runtime rejects (block size, LDS or grid limits) would otherwise be
// silent -- the next fetch would copy stale results and return 0.  The failure is recorded with the kernel's name
// (thread-local: a context belongs to one host thread) and turned into NMGP_E_HIP by nmgp_take_launch_error() at the next
// API boundary.
void nmgp_note_launch_error(const char* kernel, hipError_t e);
int nmgp_take_launch_error(nmgp_ctx* c);
// (An error left behind by an EARLIER runtime call whose status was dropped -- an event record, a library internal -- is
// taken off first and recorded under its own label, so that it is not pinned on this kernel.)
#define NMGP_LAUNCH(kern, ...)                                             \
    do {                                                                   \
        hipError_t pe__ = hipGetLastError();                               \
        if (pe__ != hipSuccess) nmgp_note_launch_error("(an earlier HIP call, noticed before " #kern ")", pe__); \
        hipLaunchKernelGGL(kern, __VA_ARGS__);                             \
        hipError_t le__ = hipGetLastError();                               \
        if (le__ != hipSuccess) nmgp_note_launch_error(#kern, le__);       \
    } while (0)

#define NMGP_TRY(expr)          \
    do {                        \
        int r__ = (expr);       \
        if (r__ != 0) return r__; \
    } while (0)

// Fork / join of the prior stream: everything queued on the main stream so far is visible to the prior work, and the main
// stream waits for it before the scalar epilogue (rocBLAS follows the stream it is told).
struct PriorStreamScope {
    nmgp_ctx* c;
    hipStream_t sp;            // stream of the prior work
    rocblas_handle hb;         // ... and the rocBLAS handle bound to it
    bool forked = false, pending = false;
    explicit PriorStreamScope(nmgp_ctx* ctx) : c(ctx), sp(ctx->stream), hb(ctx->blas) {
        if (!c->prior_overlap || !c->stream_prior || !c->blas_prior) return;
        if (hipEventRecord(c->ev_prior_fork, c->stream) != hipSuccess) return;
        if (hipStreamWaitEvent(c->stream_prior, c->ev_prior_fork, 0) != hipSuccess) return;
        sp = c->stream_prior;
        hb = c->blas_prior;
        forked = true;
    }
    void done() {              // end of the prior work: join event recorded
        if (!forked) return;
        hipEventRecord(c->ev_prior_join, c->stream_prior);
        forked = false;
        pending = true;
    }
    void join() {              // before the first consumer on the main stream
        if (pending) hipStreamWaitEvent(c->stream, c->ev_prior_join, 0);
        pending = false;
    }
    ~PriorStreamScope() {
        done();
        join();
    }
};

// ---- kernel launchers (nmgp_kernels.hip) -------------------------------------------------------
const SyrkHook* nmgp_syrk_hook(nmgp_ctx* c);

namespace nmgpk {

// parameter unpacking: ell = exp(tilde_l), Lv = tril factors with exp on the diagonal slots
void svc_prep(hipStream_t s, const double* pars, int N, int M, double* ell, double* Lv, int batch = 1);
// kernel #1: fused nonseparable covariance (lower triangle, column-major, output-major indices)
int svc_cov_build(hipStream_t s, const double* x, const double* ell, const double* Lv, const double* tse,
                  double* S, int ld, int N, int M, bool full, int batch = 1, long long sstride = 0, int xstride = 0, int cps = 1);
// symmetric N x N builds (lower triangle unless full)
void rbf_cov_sym(hipStream_t s, const double* x, int N, double alpha, double beta, double* out, int ld, bool full,
                 int batch = 1);
void gibbs_cov_sym(hipStream_t s, const double* x, const double* sig, const double* ell, int N, double* out, int ld,
                   bool full);
// rectangular d-dimensional primitives, row-major output [n1, n2]
void pairwise_rect(hipStream_t s, const double* x1, int n1, const double* x2, int n2, int d, double* out);
void rbf_cov_rect(hipStream_t s, const double* x1, int n1, const double* x2, int n2, int d, double alpha, double beta,
                  bool sym, double* out);
void gibbs_cov_rect(hipStream_t s, const double* x1, const double* s1, const double* l1, int n1, const double* x2,
                    const double* s2, const double* l2, int n2, int d, bool sym, double* out);
void kron_product(hipStream_t s, const double* a, int ar, int ac, const double* b, int br, int bc, double* out);
// reductions
void chol_logdet_quad(hipStream_t s, const double* L, int ld, int n, const double* z, double* out_logdet,
                      double* out_quad, int batch = 1, long long bstride = 0, int ostride = 0);
void col_sumsq(hipStream_t s, const double* R, int ld, int rows, int cols, double* out);
// op(L) x = r for the columns r of R ([N] each, nrhs per batch element); column 0 uses L0, the others L1; N <= 3500 (LDS)
void prior_trsv(hipStream_t s, bool trans, const double* L0, int ld0, long long s0, const double* L1, int ld1, long long s1,
                double* R, int N, int nrhs, int batch, int cps = 1);
void diag_logsum2(hipStream_t s, const double* L, int ld, int n, double* out);
void fill_lower_to_full(hipStream_t s, double* A, int ld, int n, int batch = 1);
void transpose_y(hipStream_t s, const double* Y, int N, int M, double* y);
void svc_prior_rhs(hipStream_t s, const double* pars, int N, int T, double mu_l, double mu_L, double* R, int ld,
                   int batch = 1);
void stream_copy(hipStream_t s, const double* src, double* dst, size_t nelem, int mode = 0, double* sink = nullptr);
void adam_step(hipStream_t s, double* par, const double* g, double* m, double* v, int* alive, const int* info, const double* scal,
               double b1, double b2, double bc2s, double eps, double step, long long P, int B);
void hmc_status(hipStream_t s, const int* info, const double* scal, int* bad, int* failed, int B);
void hmc_kick_drift(hipStream_t s, double* p, const double* g, double* q, const int* bad, double c, double eps, int drift,
                    long long P, int B);
void hmc_drift(hipStream_t s, double* q, const double* p, const double* vel, const double* minv_diag, double eps, long long P, int B);
void hmc_scale(hipStream_t s, double* p, const double* d, long long P, int B);
void hmc_kinetic(hipStream_t s, const double* p, const double* vel, const double* minv_diag, double* kin, long long P, int B);
void hmc_restore(hipStream_t s, double* q, double* g, const double* q0, const double* g0, int* bad, const int* bad0,
                 const int* accept, long long P, int B);
// ---- nmgp_metric.hip: the prior-factor metric of the trajectories ----
// out[b] (op)= coef * op(L_blk) in[b] over the parameter layout (mode 0 assign, 1 out += coef acc, 2 out -= coef acc unless bad[b])
void prior_trmm(hipStream_t s, bool trans, const double* L0, int ld0, long long s0, const double* L1, int ld1, long long s1,
                const double* in, double* out, int N, int T, long long P, int B, int cps, double coef, int mode, const int* bad);
void prior_trmm_sep(hipStream_t s, bool trans, const double* L0, int ld0, const double* L1, int ld1, const double* in, double* out, int N,
                    int T, long long P, int B, double cscale);
void lowrank_proj(hipStream_t s, const double* U, const double* u, double* c, long long P, int r, int B, int cps);
void lowrank_apply(hipStream_t s, const double* U, const double* wgt, const double* c, const double* in, double* out, long long P,
                   int r, int B, int cps);
void metric_kinetic(hipStream_t s, const double* u, const double* c, const double* sw, double* kin, long long P, int r, int B, int cps);
int svc_adjoint(hipStream_t s, const double* x, const double* ell, const double* Lv, const double* alpha,
                const double* Sinv, int ld, int N, int M, double* part, double ssign = 1.0, int batch = 1,
                int xstride = 0, int cps = 1);
// The rows of L^-T of a gradient evaluation are seeded only in a band (k_xtri_seed, nmgp_chol.hip): the identity in a row's own
// 64-column block and zeros in the NMGP_XTRI_SEED_BLOCKS blocks left of it.  Every reader of those rows reaches a bounded distance
// left of the diagonal; nmgp_chol.hip and nmgp_kernels.hip static_assert their reach against this constant, so that a change of
// tile size or block width cannot silently read unwritten memory (which only NMGP_POISON would show).
#define NMGP_XTRI_SEED_BLOCKS 3
#define NMGP_TRI_GEMV_BLOCK 256     // block edge of tri_gemv_upper (reads the 256 x 256 blocks on and above the diagonal)
// out = W z, W upper triangular (n x n, column-major with leading dimension ld; the zeros left of the diagonal are stored);
// part: n * ceil(n / 256) doubles of scratch per matrix
void tri_gemv_upper(hipStream_t s, const double* W, int ld, int n, const double* z, double* out, double* part, int batch = 1,
                    long long wstride = 0, long long pstride = 0);
void trace_terms(hipStream_t s, const double* alpha, const double* Sinv, int ld, int n, double* out,
                 double ssign = 1.0, int batch = 1);
void svc_grad_final(hipStream_t s, const double* part, int NJ, int N, int M, const double* Lv, const double* R2,
                    int ldR, const double* pars, const double* tr, double a, double b, int prior, double* grad,
                    int batch = 1);
void svc_finalize(hipStream_t s, const double* logdet, const double* quad, const double* q, const double* hl_l,
                  const double* hl_L, const double* pars, long long P, int N, int T, double a, double b,
                  double ig_const, int prior, double* out5, int batch = 1, int sstride = 0, int hstride = 0, int cps = 1);
void half_logdet(hipStream_t s, const double* L, int ld, int n, double* out, int batch = 1);
// ---- nmgp_kernels_eig.hip ----
int kron_mv(hipStream_t s, const double* K, int n1, int n2, const double* y, const double* B, int m1, int m2,
            double* out);
void eig_reduce(hipStream_t s, double* a, const double* wB, int M, const double* wK, int N, const double* sigma2p,
                bool scale, double* out);
void colscale_d(hipStream_t s, const double* V, const double* wB, int M, const double* wK, int N, const double* sigma2p,
                double* Vs);
void colscale(hipStream_t s, const double* V, const double* svec, int rows, int cols, double* Vs);
void sep_coreB(hipStream_t s, const double* At, const double* wB, int M, const double* wK, int N,
               const double* sigma2p, double* coreB);
void sep_adjoint(hipStream_t s, const double* x, const double* ell, const double* sig, const double* U,
                 const double* wB, int M, const double* C, int N, double* part);
void sep_grad_sum(hipStream_t s, const double* part, int NJ, int N, double* g);
void exp_vec(hipStream_t s, const double* in, int n, double* out);
void fill_vec(hipStream_t s, double* out, int n, double v);
void two_col_rhs(hipStream_t s, const double* a, double mu_a, const double* b, double mu_b, int N, double* R);
void sub_vec(hipStream_t s, const double* y, const double* mu, int n, double* out);
void dot(hipStream_t s, const double* a, const double* b, int n, double* out);
void kron_eigvec(hipStream_t s, const double* VB, int M, const double* VK, int N, double* U);
void kron_w(hipStream_t s, const double* wB, int M, const double* wK, int N, double sigma2, double* w);
void svc_star(hipStream_t st, const double* proj, int S, int M, double mu_l, double mu_L, double* tl_star,
              double* Lstar);
void svc_crosscov_rows(hipStream_t st, const double* x, const double* ell, const double* Lv, int N, int M, const double* xs,
                       const double* tl_star, const double* Lstar, int S, double* A, int ld, int R0);
// part: 2 * E * ceil(n / 128) doubles
void pred_rows_reduce(hipStream_t st, const double* A, int ld, int n, int R0, int zrow, int E, double* part, double* mean,
                      double* colsq);
void svc_predvar(hipStream_t st, const double* Lstar, const double* colsq, int S, int M, const double* tse,
                 double* var);
void sep_crossvec(hipStream_t st, int mode, const double* x, const double* sig, const double* ell, int N,
                  const double* xs, const double* tl_star, const double* ts_star, double sig0, double l0, int S,
                  double* KX);
void sep_predict(hipStream_t st, const double* Cq, const double* a, const double* wB, const double* VB, int M,
                 const double* wK, int N, double sigma2, const double* Bdiag, const double* kss, bool strict_clip, int S,
                 double* mean, double* var);
void sep_predict_chol(hipStream_t st, const double* dots, const double* sqs, const double* wB, const double* VB, int M, double sigma2,
                      const double* Bdiag, const double* kss, bool strict_clip, int S, double* mean, double* var);
void cols_to_rows(hipStream_t st, const double* KX, int N, int S, double* A, int ld, int R0, int batch, long long bstride);
void sep_star(hipStream_t st, const double* proj, int S, double mu_l, double mu_s, double* tl_star, double* ts_star,
              double* kss);
void add_diag(hipStream_t s, double* A, int ld, int n, double v);
void rotate_y(hipStream_t s, const double* Y, const double* VB, int N, int M, double* yt);
void sep_blocks(hipStream_t s, const double* K, const double* wB, const double* sigma2p, int N, int M, double* out,
                int ldo, long long bstride);
#define NMGP_SEP_TR_G 128      // partial sums per block that sep_traces leaves for the host: out[(p * NMGP_SEP_TR_G + g) * 3 + {tr, tk, aa}]
int sep_traces(hipStream_t s, const double* Cneg, const double* K, const double* alpha, int N, int M, double* out);
void weighted_sum_lower(hipStream_t s, const double* Cneg, const double* wB, int N, int M, double* C);
// ---- nmgp_kernels_sep.hip: the separable objective's pieces with the chain as a grid dimension (nmgp_sep_batch_eval) ----
void sep_prep_b(hipStream_t s, const double* pars, long long P, const double* Y, const double* small, int small_per, int N, int M,
                double* ell, double* sig, double* yt, int B);
void sep_blocks_b(hipStream_t s, const double* x, const double* ell, const double* sig, const double* small, int small_per, int N, int M,
                  double* S, int ldo, long long bstride, double* Kout, int B);
void sep_reduce_b(hipStream_t s, const double* Cneg, const double* K, const double* alpha, const double* small, int small_per, int N,
                  int M, int G, double* C, double* out, double* xi, int B);
void sep_adjoint_b(hipStream_t s, const double* x, const double* ell, const double* sig, const double* U, const double* small,
                   int small_per, int M, const double* C, int N, double* part, double* g, int B);
void two_col_rhs_b(hipStream_t s, const double* pars, long long P, double mu_a, double mu_b, int N, double* R, int B);
// ---- nmgp_chol.hip ----
void syrk_lower(hipStream_t s, const double* A, int lda, double* C, int ldc, int mrows, int ncols, int K, int batch,
                long long bstride, long long cstride = -1, int ktri = 0, int tri_row0 = 0x7fffffff, int tri_k0 = 0);
void identity_rows(hipStream_t s, double* A, int lda, int row0, int n, int pad, int batch = 1, long long bstride = 0);
void potf2_64(hipStream_t s, double* A, int lda, int nb, int* info, int goff, int batch, long long bstride,
              int istride);
void trsm_64(hipStream_t s, const double* L, int ldl, int nb, double* A, int lda, int rows, int batch,
             long long bstride);
void set_row(hipStream_t s, double* A, int lda, int row, const double* v, int n, int batch, long long bstride,
             long long vstride, int cps = 1);
void get_row(hipStream_t s, const double* A, int lda, int row, double* v, int n, int batch, long long bstride,
             long long vstride);
void potrf_lower(hipStream_t s, hipStream_t s2, hipEvent_t* ev, double* A, int lda, int n, int extra, int xtri,
                 int nb1, int* info, int batch, long long bstride, int istride, const SyrkHook* hook = nullptr,
                 int precise = 0);

// ---- nmgp_trtri.hip ----
bool trtri_post_applies(int n, int xtri, int lda, int batch);
void trtri_upper_post(hipStream_t s, double* S, int ld, int n, int xoff, int batch, long long bs, const SyrkHook* hook);

}  // namespace nmgpk
