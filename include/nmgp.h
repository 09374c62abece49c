/*
 * nmgp.h -- C ABI of libnmgp_hip.so: the MI355X (gfx950) implementation of the multi-output
 * Gaussian-process log-posterior path of Corleno/Nonstationary_Multivariate_Gaussian_Process.
 *
 * The reference has no FFI layer: its boundary is a set of Python module functions
 * (Utility/{kernels,kronecker_operation,distributions,logpos}.py).  Every entry point below names the
 * reference function (file:line under the reference tree) whose arithmetic it replaces; the Python
 * package `nonstationary_multivariate_gaussian_process_amd.Utility` binds these symbols with ctypes
 * behind the reference's own signatures (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - plain C, no torch / HIP types in any signature; all scalars are double / int / long long.
 *   - unless a name ends in `_dev`, array arguments are caller-owned HOST buffers, float64,
 *     C-contiguous (row-major); the library copies in/out and owns all device memory.
 *   - return value: 0 success; <0 API misuse (NMGP_E_*); >0 numerical failure:
 *       k in [1, 1<<20)  : Cholesky failed, leading minor k not positive definite (rocSOLVER info)
 *       NMGP_NUM_NAN     : a NaN/Inf reached the result
 *       NMGP_NUM_EIG     : eigensolver did not converge
 *     nmgp_last_error(ctx) returns a human readable message for the last non-zero return.
 *   - a context is bound to one GPU and one HIP stream and is NOT thread-safe; use one per host thread.
 *   - layouts follow the reference: Y is [N,M] row-major, the stacked observation vector is
 *     output-major y[m*N+i] = Y[i,m] (logpos.py:338); the nonseparable parameter vector is
 *     [tilde_l (N) | uL_vecs (N*T, location-major, tril row-major slots) | tilde_sigma2_err]
 *     (logpos.py:32-43), the separable one [tilde_l (N) | tilde_sigma (N) | uL_vec (T) | tilde_sigma2_err]
 *     (logpos.py:17-29), the stationary one [tilde_l, tilde_sigma, uL_vec (T), tilde_sigma2_err]
 *     (logpos.py:46-57); T = M(M+1)/2.
 */
#ifndef NMGP_H
#define NMGP_H

#ifdef __cplusplus
extern "C" {
#endif

#define NMGP_VERSION 100

/* API-misuse codes */
#define NMGP_E_NULL        (-1)   /* required pointer is NULL */
#define NMGP_E_SHAPE       (-2)   /* non-positive or inconsistent dimension */
#define NMGP_E_STATE       (-3)   /* call order violated (e.g. eval before nmgp_set_data) */
#define NMGP_E_UNSUPPORTED (-4)   /* M larger than NMGP_MAX_OUTPUTS, etc. */
#define NMGP_E_HIP         (-5)   /* HIP / rocBLAS / rocSOLVER runtime error (see nmgp_last_error) */
#define NMGP_E_NOMEM       (-6)
/* numerical-failure codes (positive) */
#define NMGP_NUM_NAN       (1 << 20)
#define NMGP_NUM_EIG       ((1 << 20) + 1)

#define NMGP_MAX_OUTPUTS 8

typedef struct nmgp_ctx nmgp_ctx;

/* ---- context ------------------------------------------------------------------------------- */
/* device = HIP device ordinal. */
int nmgp_ctx_create(int device, nmgp_ctx** out);
int nmgp_ctx_destroy(nmgp_ctx* ctx);
const char* nmgp_last_error(const nmgp_ctx* ctx);
int nmgp_version(void);
/* SHA-256 (hex) of the sources, headers and code-generation flags the library was built from (build.py: tree_id()).
 * The Python binding refuses a shared object whose id differs from the source tree beside it. */
const char* nmgp_build_id(void);
/* Blocks until all work queued on the context's stream is complete. */
int nmgp_sync(nmgp_ctx* ctx);
/* Number of visible HIP devices (does not create a context). */
int nmgp_device_count(void);

/* ---- data residency ------------------------------------------------------------------------ */
/* Upload one subject's inputs (x: [N], Y: [N,M] row-major) and size all workspaces.  Replaces the
 * per-call tensor plumbing of logpos.py:337-338.  May be called again with another subject. */
int nmgp_set_data(nmgp_ctx* ctx, const double* x, const double* Y, int N, int M);

/* ---- nonseparable ("SVC") objective:  logpos.nlogpos_obj_SVC / logpos_SVC, logpos.py:299-380 -- */
/* hyper[8] = {mu_tilde_l, alpha_tilde_l, beta_tilde_l, mu_L, alpha_L, beta_L, a, b}.
 * out5 = {NegLog, loglik, log_prior_tilde_l, log_prior_uL_vecs, log_prior_sigma2_err} (the verbose tuple).
 * prior = the reference's `Prior` flag.  grad (length N(1+T)+1) receives d NegLog / d pars, or pass NULL.
 * Synchronous. */
int nmgp_logpos_svc(nmgp_ctx* ctx, const double* pars, const double hyper[8], int prior,
                    double out5[5], double* grad);

/* Resident form used by MCMC/MAP loops and the benchmark: the parameter vector lives in HBM
 * (nmgp_svc_pars_dev returns its device address, length N(1+T)+1; nmgp_svc_set_pars copies a host
 * vector into it).  nmgp_svc_eval_resident only enqueues work on the context's stream; results stay on
 * the device until nmgp_svc_fetch (which synchronises, checks the factorisation status and copies
 * out5 / grad). */
int nmgp_svc_set_pars(nmgp_ctx* ctx, const double* pars);
double* nmgp_svc_pars_dev(nmgp_ctx* ctx);
double* nmgp_svc_grad_dev(nmgp_ctx* ctx);
int nmgp_svc_eval_resident(nmgp_ctx* ctx, const double hyper[8], int prior, int want_grad);
int nmgp_svc_fetch(nmgp_ctx* ctx, double out5[5], double* grad);

/* Batched form: B chains of the resident subject (same x, Y; B parameter vectors stacked [B, P]) are evaluated by ONE
 * launch sequence -- every kernel takes the chain as a grid dimension, so the latency-bound panel steps of the
 * factorisation are paid once per batch.  This is the throughput path of MCMC with
 * many chains (the reference runs its chains as separate processes, Nonseparable_model_mpisim.py:305-306).
 * nmgp_svc_batch_alloc sizes B covariance buffers (B * 8 * MN * (MN+1) bytes; called again with the SAME B it keeps every buffer --
 * the gradient workspace of 128 chains at the headline size is 116 GB, seconds of allocation -- and only resets the batch's state:
 * chains of the resident subject, identity metric, no trajectory / optimiser, zero parameters); nmgp_svc_batch_eval only enqueues;
 * nmgp_svc_batch_fetch synchronises and returns out[B,5] and per-chain status[B] (0 ok; k>0 leading minor k not
 * positive definite; NMGP_NUM_NAN) -- a failing chain yields NaNs in its row, not a failed call. */
int nmgp_svc_batch_alloc(nmgp_ctx* ctx, int B);
int nmgp_svc_batch_set_pars(nmgp_ctx* ctx, const double* pars);
/* Optional: make every batch element its own SUBJECT (x: [B,N], Y: [B,N,M] row-major; N, M as in nmgp_set_data) with
 * its own GP-prior factors -- BASELINE config 4, the reference's one-process-per-subject pattern in one launch
 * sequence.  Without this call all batch elements are chains of the subject given to nmgp_set_data. */
int nmgp_svc_batch_set_subjects(nmgp_ctx* ctx, const double* x, const double* Y);
/* Several chains per subject: x [S, N], Y [S, N, M] with S = batch / chains_per_subject; batch element b = s * chains_per_subject + k
 * is chain k of subject s (its parameter vector: row b of nmgp_svc_batch_set_pars).  The chains of a subject share its inputs and its
 * GP-prior factors.  (The reference runs one process per subject and chain: Nonseparable_model_mpisim.py:305-348.) */
int nmgp_svc_batch_set_subjects_chains(nmgp_ctx* ctx, const double* x, const double* Y, int chains_per_subject);
double* nmgp_svc_batch_pars_dev(nmgp_ctx* ctx);
int nmgp_svc_batch_eval(nmgp_ctx* ctx, const double hyper[8], int prior, int want_grad);
int nmgp_svc_batch_fetch(nmgp_ctx* ctx, double* out, int* status);
/* gradients d NegLog / d pars of the last batched evaluation (want_grad = 1): [B, P] on the host / in HBM */
int nmgp_svc_batch_fetch_grad(nmgp_ctx* ctx, double* grad);
double* nmgp_svc_batch_grad_dev(nmgp_ctx* ctx);

/* Device-resident leapfrog trajectories of the B chains -- the inner loop of the HMC sampler the reference's scripts hand their
 * potential to (Nonseparable_model.py:228-231: step_size 1e-4, num_steps_in_leap 20; the sampler itself is an external package).
 * Positions (the batch's parameter vectors), momenta and gradients stay in HBM between the `nsteps` batched value+gradient
 * evaluations of a trajectory; the caller draws the momenta and the accept uniforms and keeps the potentials.
 *   nmgp_svc_batch_traj_begin : after a batched value+gradient evaluation of the start positions (set_pars + batch_eval(.., 1))
 *   nmgp_svc_batch_traj       : p0 [B,P] in; end point q1, p1 [B,P], potential U1 [B] (NegLog; +inf where failed) and
 *                               failed [B] (1: the potential was undefined somewhere on the trajectory -> reject) out
 *   nmgp_svc_batch_traj_commit: accept [B]; rejected chains get their pre-trajectory position and gradient back
 *   nmgp_svc_batch_traj_set_mass: constant mass matrix M of the sampler (Nonseparable_model_mpiKAISER.py:267-270,398-411 passes
 *                               M = inv(sample covariance), step size 1e-1, 5 leapfrog steps): kind 0 identity (default),
 *                               1 diagonal (minv = diag(M^-1) [P]), 2 dense (minv = M^-1 [P,P], row-major == column-major: it is
 *                               symmetric).  The drift becomes q += eps M^-1 p (dense: one GEMM per leapfrog step for all chains);
 *                               with nmgp_svc_batch_traj the momenta p ~ N(0, M) and the kinetic energy 1/2 p^T M^-1 p remain
 *                               the caller's.  Changing the metric (or the batch's subjects) invalidates a begun trajectory.
 *   nmgp_svc_batch_traj_set_mass_chol: a square root of M for the same kind -- sqrt(diag M) [P], or R [P,P] COLUMN-major with
 *                               R R^T = M (the lower Cholesky factor, zeros stored above the diagonal, or any other) -- so that
 *                               the device can draw the momenta itself
 *   nmgp_svc_batch_traj_z     : like nmgp_svc_batch_traj, but z [B,P] in are STANDARD NORMALS: p0 = chol(M) z is formed on the
 *                               device (start kinetic energy = 1/2 |z|^2 for any metric) and kin1 [B] = 1/2 p1^T M^-1 p1 comes
 *                               back instead of p1 -- no [B,P] x [P,P] product is left on the host.
 *   nmgp_svc_batch_traj_set_mass_prior: the PRIOR-FACTOR metric (kind 3), the one that makes the sampler mix at N = 2048.  The posterior
 *                               is dominated by the GP priors RBF(alpha, beta) + 1e-6 I on tilde_l and on each stride-T column of uL_vecs
 *                               (logpos.py:357-365; condition number ~1e11), whose Cholesky factors the context caches; in the
 *                               coordinates pars = mu + L_blk w, L_blk = blockdiag(chol Sigma_l, chol Sigma_L x T, 1), the Hessian of
 *                               the potential is I + (a few dozen likelihood-informed directions).  M^-1 = L_blk (I + U diag(lam) U^T)^-1
 *                               L_blk^T with U [S, r, P] (per subject: r orthonormal rows of length P), lam [S, r] >= 0 the rank-r
 *                               correction (rank 0 / NULL: the pure prior metric); hyper as for the objective.  The device carries the
 *                               whitened momentum L_blk^T p: every leapfrog step costs two triangular mat-vecs with the cached
 *                               factors and two [P, r] products per chain, no solve and no [P, P] matrix.  Only nmgp_svc_batch_traj_z
 *                               runs under this metric.  nmgp_svc_batch_set_subjects* resets it to the identity (it belongs to the
 *                               subjects it was built for).
 *   nmgp_svc_batch_prior_apply: out [B,P] = L_blk in (trans 0) or L_blk^T in (trans 1) for B parameter-shaped host vectors -- the
 *                               change of coordinates of that metric, exported so that the caller can build U, lam (Hessian-vector
 *                               products of the likelihood in whitened coordinates: drivers.prior_lowrank_metric).
 * An API-level failure anywhere in a trajectory call (either entry; inside the leapfrog loop or in the end point's reductions and
 * copies) restores the start state and requires a fresh value+gradient evaluation + nmgp_svc_batch_traj_begin. */
int nmgp_svc_batch_traj_begin(nmgp_ctx* ctx);
int nmgp_svc_batch_traj_set_mass(nmgp_ctx* ctx, int kind, const double* minv);
int nmgp_svc_batch_traj_set_mass_chol(nmgp_ctx* ctx, int kind, const double* mchol);
int nmgp_svc_batch_traj_set_mass_prior(nmgp_ctx* ctx, const double hyper[8], int rank, const double* U, const double* lam);
int nmgp_svc_batch_prior_apply(nmgp_ctx* ctx, const double hyper[8], int trans, const double* in, double* out);
/* The same change of coordinates for the SEPARABLE model's parameter vector [tilde_l | tilde_sigma | uL_vec | tilde_sigma2_err]:
 * L_blk = blockdiag(chol Sigma_l, chol Sigma_sigma, c I_T, 1) -- the GP priors of logpos.py:271-281 and the Normal(0, c) prior of
 * :283 (c float32-rounded, as the reference passes it); hyper as for nmgp_logpos_sep.  in / out: B host vectors of length 2N+T+1.
 * The metric of the separable sampler (drivers.SeparablePriorMetric) is built on it. */
int nmgp_sep_prior_apply(nmgp_ctx* ctx, const double hyper[9], int trans, int B, const double* in, double* out);
int nmgp_svc_batch_traj(nmgp_ctx* ctx, const double hyper[8], int prior, double eps, int nsteps, const double* p0,
                        double* q1, double* p1, double* U1, int* failed);
int nmgp_svc_batch_traj_z(nmgp_ctx* ctx, const double hyper[8], int prior, double eps, int nsteps, const double* z,
                          double* q1, double* kin1, double* U1, int* failed);
int nmgp_svc_batch_traj_commit(nmgp_ctx* ctx, const int* accept);

/* Device-resident Adam over the batch -- the MAP loop of Nonseparable_model.py:147-210 (torch.optim.Adam, default betas / eps)
 * for B chains or, with nmgp_svc_batch_set_subjects, for all subjects of a rank at once (Nonseparable_model_mpisim.py:330-348):
 * parameters, gradients and moments stay in HBM; per iteration only the B verbose tuples come back.
 *   nmgp_svc_batch_adam_begin : start points = the batch's parameter vectors (nmgp_svc_batch_set_pars); moments := 0
 *   nmgp_svc_batch_adam_step  : one value+gradient evaluation + update; out [B,5] are the tuples at the parameters the iteration
 *                               started from (NaN row for a subject that is no longer alive); alive [B] turns 0 at a subject's
 *                               first failed evaluation (its parameters stay frozen; the others go on)
 *   nmgp_svc_batch_get_pars   : the parameter vectors [B,P] as they stand */
int nmgp_svc_batch_adam_begin(nmgp_ctx* ctx);
int nmgp_svc_batch_adam_step(nmgp_ctx* ctx, const double hyper[8], int prior, double lr, double beta1, double beta2, double eps,
                             double* out, int* alive);
int nmgp_svc_batch_get_pars(nmgp_ctx* ctx, double* pars);

/* Dense covariance of the nonseparable model as the reference assembles it (logpos.py:339-353:
 * K_x, generate_K_index_SVC, the n-major->m-major permutation, kron(ones, K_x) * K_i, + sigma2 I).
 * out: [MN, MN] full symmetric, output-major.  For tests and for callers that want Sigma itself. */
int nmgp_svc_covariance(nmgp_ctx* ctx, const double* pars, double* out);

/* ---- separable objective:  logpos.nlogpos_obj / logpos, logpos.py:216-296 ------------------- */
/* hyper[9] = {mu_tilde_l, alpha_tilde_l, beta_tilde_l, mu_tilde_sigma, alpha_tilde_sigma,
 *             beta_tilde_sigma, a, b, c};
 * out6 = {NegLog, loglik, lp_tilde_l, lp_tilde_sigma, lp_uL_vec, lp_sigma2_err}; grad length 2N+T+1. */
int nmgp_logpos_sep(nmgp_ctx* ctx, const double* pars, const double hyper[9], int prior,
                    double out6[6], double* grad);

/* ---- stationary objective:  logpos.nlogpos_obj_S / logpos_S, logpos.py:383-462 -------------- */
/* hyper[5] = {mu_tilde_l, sigma_tilde_l, a, b, c}; out5 = {NegLog, loglik, lp_tilde_l, lp_uL_vec,
 * lp_sigma2_err}; grad length T+3. */
int nmgp_logpos_sta(nmgp_ctx* ctx, const double* pars, const double hyper[5], int prior,
                    double out5[5], double* grad);

/* B chains of the SEPARABLE model of the resident subject per launch sequence -- the unit the reference runs as separate processes
 * (Separable_model_mpisim.py:299-300) and the potential evaluations of its sampler (Separable_model.py:209): chain b's M blocks
 * wB_b[p] K_x,b + sigma2_b I join ONE batch of the blocked Cholesky (B M matrices of order N), as do the triangular products and the
 * inverse SYRK of the gradient; objective of logpos.py:216-296 per chain.
 * pars [B, 2N+T+1]; out6 [B, 6] (as nmgp_logpos_sep); grad [B, 2N+T+1] or NULL; status [B]: 0 = exact covariance, k in 1..3 = the
 * chain needed k jitter retries (re-evaluated through nmgp_logpos_sep: the reference's `while loglik != loglik` loop), negative =
 * -(numerical failure code) if it failed even so (its out6 row is NaN, its gradient row zero).  Returns 0 unless an API / runtime
 * error occurred.
 * Every piece of the evaluation takes the chain as a grid dimension and the value and gradient halves are enqueued back to back: one
 * host synchronisation per call.  Device workspace: B M (2N + 2) N doubles for the factorisation with gradients (N + 1 without) +
 * B M N^2 (-S^-1) + 2 B N^2 (K_x, the weighted sum) -- 2.5 GB per chain at N = 4096, D = 5; the batch is evaluated in chunks of
 * chains whose workspace stays below NMGP_SEP_BATCH_SLAB_GB (environment, default 96) and of at most 65,535 / M chains (grid limit);
 * NMGP_E_SHAPE if a single chain does not fit.  Largest batch exercised on hardware: 32 chains x 5 blocks of N = 4096 (B M = 160). */
int nmgp_sep_batch_eval(nmgp_ctx* ctx, const double* pars, int B, const double hyper[9], int prior, double* out6, double* grad,
                        int* status);

/* ---- primitives (host buffers in / out) ----------------------------------------------------- */
/* kernels.pairwise_distances, kernels.py:5-21.  x1: [n1,d], x2: [n2,d] or NULL (=x1). out: [n1,n2]. */
int nmgp_pairwise_distances(nmgp_ctx* ctx, const double* x1, int n1, const double* x2, int n2, int d,
                            double* out);
/* kernels.RBF_cov, kernels.py:24-43.  x2 == NULL selects the symmetric form with jitter*I. */
int nmgp_rbf_cov(nmgp_ctx* ctx, const double* x1, int n1, const double* x2, int n2, int d,
                 double alpha, double beta, double* out);
/* kernels.Nonstationary_RBF_cov, kernels.py:46-73.  s1/l1 (s2/l2) may be NULL (= ones). */
int nmgp_nonstat_rbf_cov(nmgp_ctx* ctx, const double* x1, const double* s1, const double* l1, int n1,
                         const double* x2, const double* s2, const double* l2, int n2, int d,
                         double* out);
/* kronecker_operation.kronecker_product, kronecker_operation.py:5-22.  a: [ar,ac], b: [br,bc]. */
int nmgp_kron_product(nmgp_ctx* ctx, const double* a, int ar, int ac, const double* b, int br, int bc,
                      double* out);
/* kronecker_operation.kron_mv, kronecker_operation.py:72-85.  B: [m1,m2], K: [n1,n2], y: [m2*n2]
 * (output-major), out: [m1*n1]. */
int nmgp_kron_mv(nmgp_ctx* ctx, const double* B, int m1, int m2, const double* K, int n1, int n2,
                 const double* y, double* out);
/* distributions.multivariate_normal_logpdf, distributions.py:10-23: -0.5 logdet - 0.5 (y-mu)' invSigma (y-mu)
 * for a caller-supplied inverse [n,n] and log-determinant (mu may be NULL = zeros). */
int nmgp_mvn_logpdf(nmgp_ctx* ctx, const double* y, const double* mu, double logdetSigma,
                    const double* invSigma, int n, double* out);
/* distributions.multivariate_normal_logpdf0, distributions.py:26-52: log density (without 2 pi) of
 * N(mu, B kron K + sigma2 I) in the joint eigenbasis.  B: [M,M], K: [N,N], y/mu: [MN] (mu may be NULL). */
int nmgp_mvn_logpdf_kron(nmgp_ctx* ctx, const double* y, const double* mu, const double* B, int M,
                         const double* K, int N, double sigma2, double* out);
/* distributions.multivariate_normal_logpdf2, distributions.py:99-113 (dense Cholesky evaluation). */
int nmgp_mvn_logpdf_dense(nmgp_ctx* ctx, const double* y, const double* mu, const double* B, int M,
                          const double* K, int N, double sigma2, double* out);
/* kronecker_operation.kron_inv / kron_logdet, kronecker_operation.py:36-69. out_inv: [MN,MN]. */
int nmgp_kron_inv_logdet(nmgp_ctx* ctx, double sigma2, const double* B, int M, const double* K, int N,
                         double* out_inv, double* out_logdet);

/* Cholesky factorisation A = L L^T of a symmetric positive definite [n,n] matrix (torch.cholesky as used at
 * prediction.py:974; also the entry through which the custom blocked factorisation of the log-posterior path is
 * tested on its own).  out_L: [n,n] row-major with the factor in the UPPER triangle == column-major lower (i.e.
 * out_L^T is the usual lower factor; the other triangle is scratch: the input's values, except that the diagonal 16x16
 * blocks carry the inverted diagonal blocks the panel solve reuses).  rhs (optional, [n]) is carried
 * through the factorisation as an extra row: out_z = L^-1 rhs.  algo: 1 = custom gfx950 factorisation, 0 = rocSOLVER. */
int nmgp_cholesky(nmgp_ctx* ctx, const double* A, int n, const double* rhs, double* out_L, double* out_z,
                  int algo);

/* ---- deterministic prediction (prediction.py:912-988, 337-408, 1566-1638) -------------------- */
/* Nonseparable: predictive mean / variance of y at S new inputs xs given MAP parameters.
 * mean, var: [S,M]; Lstar: [S,T] (predicted L_vec at xs, exp already applied on the diagonal slots). */
int nmgp_predict_svc(nmgp_ctx* ctx, const double* pars, const double hyper[8], const double* xs, int S,
                     double* mean, double* var, double* Lstar);
int nmgp_predict_sep(nmgp_ctx* ctx, const double* pars, const double hyper[9], const double* xs, int S,
                     double* mean, double* var);
int nmgp_predict_sta(nmgp_ctx* ctx, const double* pars, const double* xs, int S, double* mean,
                     double* var);

/* ---- measurement ---------------------------------------------------------------------------- */
/* Per-stage HIP-event timing on the context's stream (bench.py roofline figures).  Stages: */
enum {
    NMGP_STAGE_COV = 0,     /* fused covariance build (kernel #1)            */
    NMGP_STAGE_CHOL = 1,    /* Cholesky factorisation of Sigma               */
    NMGP_STAGE_SOLVE = 2,   /* triangular solve(s) with y                    */
    NMGP_STAGE_REDUCE = 3,  /* log-det / quadratic-form reductions + combine */
    NMGP_STAGE_PRIOR = 4,   /* GP priors (cached factors, trsm)              */
    NMGP_STAGE_INVERSE = 5, /* Sigma^-1 for the gradient                     */
    NMGP_STAGE_ADJOINT = 6, /* fused adjoint contraction (kernel #5)         */
    NMGP_STAGE_EIG = 7,     /* eigendecomposition (separable / stationary)   */
    NMGP_STAGE_KRONMV = 8,  /* Kron-vec contraction (kernel #3)              */
    NMGP_STAGE_SYRK = 9,    /* every k_syrk_lower launch of the blocked Cholesky (FP64 MFMA), timed on its own stream */
    NMGP_STAGE_COUNT = 10
};
int nmgp_profile_enable(nmgp_ctx* ctx, int on);   /* 0 off; 1 one HIP-event pair per stage; 2 additionally one pair per k_syrk_lower launch */
/* Accumulated milliseconds and launch counts per stage since the last reset; synchronises. */
int nmgp_profile_read(nmgp_ctx* ctx, double ms[NMGP_STAGE_COUNT], long long count[NMGP_STAGE_COUNT]);
int nmgp_profile_reset(nmgp_ctx* ctx);
/* As nmgp_profile_read, plus the algorithmic work accumulated per stage; tracked for NMGP_STAGE_SYRK only (0 elsewhere):
 * flop = 2 K per updated lower-trapezoid element of every launch, bytes = 8 * (read + write of those elements + the
 * mrows x K panel read once). */
int nmgp_profile_read_work(nmgp_ctx* ctx, double ms[NMGP_STAGE_COUNT], long long count[NMGP_STAGE_COUNT],
                           double flop[NMGP_STAGE_COUNT], double bytes[NMGP_STAGE_COUNT]);
/* Jitter retries the last nmgp_logpos_sep / nmgp_logpos_sta evaluation needed: 0 = value and gradient are those of the exact
 * covariance, k > 0 = of the covariance with k x 1e-6 added to the diagonals of B and K_x (the reference retries a NaN
 * likelihood with RANDOM jitter of that size, logpos.py:267-268 / distributions.py:55-96; here any numerical failure of the
 * first attempt -- NaN or a non-positive pivot -- triggers the deterministic retry).  -1 for a NULL context. */
int nmgp_last_sep_attempts(const nmgp_ctx* ctx);
/* Micro-benchmarks used to state measured peaks next to the spec ones: HBM stream (GB/s) and a
 * rocBLAS dgemm of size n (TFLOP/s). */
int nmgp_measure_hbm_gbs(nmgp_ctx* ctx, long long bytes, int reps, double* gbs);
/* gbs3 = {copy (read + write bytes counted), read only, write only}: the chip's streaming ceilings as measured in the run, the
 * figures the HBM-bound kernels are judged against (flat 16-byte-per-lane kernels, tools/lab/hbm_lab.hip). */
int nmgp_measure_hbm_rates(nmgp_ctx* ctx, long long bytes, int reps, double gbs3[3]);
int nmgp_measure_dgemm_tflops(nmgp_ctx* ctx, int n, int reps, double* tflops);

#ifdef __cplusplus
}
#endif
#endif /* NMGP_H */
