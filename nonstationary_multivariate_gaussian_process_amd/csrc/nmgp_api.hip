// C ABI of libnmgp_hip.so: context, data residency, the nonseparable objective (value + gradient),
// primitives, profiling.  See include/nmgp.h for the contract and the reference lines each entry replaces.
#include <cstdarg>

#include "nmgp_internal.h"

using namespace nmgpk;

// layout of the small device scalar block
enum {
    SC_LOGDET = 0,
    SC_QUAD = 1,
    SC_PRIORQ = 2,    // 1 + T entries (T <= 36)
    SC_TRACE = 40,    // 2 entries
    SC_OUT = 48,      // 8 entries
    SC_MISC = 56,
    SC_COUNT = 128
};

int nmgp_fail(nmgp_ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

// ---- launch-error bookkeeping (NMGP_LAUNCH, nmgp_internal.h) -----------------------------------------
static thread_local char g_launch_err[256] = {0};

void nmgp_note_launch_error(const char* kernel, hipError_t e) {
    if (g_launch_err[0]) return;                 // keep the first one: later launches usually fail because of it
    snprintf(g_launch_err, sizeof(g_launch_err), "launch of %s was rejected: %s", kernel, hipGetErrorString(e));
}

int nmgp_take_launch_error(nmgp_ctx* c) {
    if (!g_launch_err[0]) return 0;
    int rc = nmgp_fail(c, NMGP_E_HIP, "%s", g_launch_err);
    g_launch_err[0] = 0;
    return rc;
}

// ---- profiling helpers ---------------------------------------------------------------------------
typedef NmgpStage StageScope;
NmgpStage::NmgpStage(nmgp_ctx* ctx, int st, hipStream_t s, double work, double bytes) : c(ctx), stage(st), stream(s) {
        if (!c->profiling) return;
        StageTimer& t = c->timers[stage];
        t.work += work;
        t.bytes += bytes;
        if (!t.pool.empty()) {
            e0 = t.pool.back().first;
            e1 = t.pool.back().second;
            t.pool.pop_back();
        } else {
            hipEventCreate(&e0);
            hipEventCreate(&e1);
        }
        hipEventRecord(e0, stream);
}
NmgpStage::NmgpStage(nmgp_ctx* ctx, int st) : c(ctx), stage(st), stream(ctx->stream) {
        if (!c->profiling) return;
        StageTimer& t = c->timers[stage];
        if (!t.pool.empty()) {
            e0 = t.pool.back().first;
            e1 = t.pool.back().second;
            t.pool.pop_back();
        } else {
            hipEventCreate(&e0);
            hipEventCreate(&e1);
        }
        hipEventRecord(e0, stream);
}
NmgpStage::~NmgpStage() {
        if (!c->profiling || !e0) return;
        hipEventRecord(e1, stream);
        c->timers[stage].pending.push_back({e0, e1});
}

static void* syrk_hook_begin(void* user, hipStream_t s, double flop, double bytes) {
    nmgp_ctx* c = static_cast<nmgp_ctx*>(user);
    if (c->profiling < 2) return nullptr;
    return new NmgpStage(c, NMGP_STAGE_SYRK, s, flop, bytes);
}
static void syrk_hook_end(void*, void* token) { delete static_cast<NmgpStage*>(token); }

const SyrkHook* nmgp_syrk_hook(nmgp_ctx* c) {
    c->syrk_hook.user = c;
    c->syrk_hook.begin = syrk_hook_begin;
    c->syrk_hook.end = syrk_hook_end;
    return &c->syrk_hook;
}

static void profile_collect(nmgp_ctx* c) {
    for (int s = 0; s < NMGP_STAGE_COUNT; ++s) {
        StageTimer& t = c->timers[s];
        for (auto& pr : t.pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                t.ms += ms;
                t.count += 1;
            }
            t.pool.push_back(pr);
        }
        t.pending.clear();
    }
}

// ---- memory helpers ---------------------------------------------------------------------------
int nmgp_dev_alloc(nmgp_ctx* c, double** p, size_t nelem) {
    if (*p) {
        hipFree(*p);
        *p = nullptr;
    }
    if (nelem == 0) nelem = 1;
    hipError_t e = hipMalloc((void**)p, nelem * sizeof(double));
    if (e != hipSuccess) {
        *p = nullptr;
        return nmgp_fail(c, NMGP_E_NOMEM, "hipMalloc of %zu bytes failed: %s", nelem * sizeof(double),
                         hipGetErrorString(e));
    }
    if (nmgp_poison()) hipMemsetAsync(*p, 0xFF, nelem * sizeof(double), c->stream);
    return 0;
}

// NMGP_POISON=1 (debugging aid, exercised by tests/test_gpu_variants.py): every fresh device buffer and every scratch
// hand-out is filled with NaNs (0xFF bytes) first, so that a kernel or library call that reads memory nobody wrote shows
// up as a NaN in the result instead of depending on what the allocator happens to return.
bool nmgp_poison() {
    static const bool on = [] {
        const char* e = std::getenv("NMGP_POISON");
        return e && std::atoi(e) != 0;
    }();
    return on;
}

// leading dimension (doubles) of a column-major factorisation buffer with `rows` rows: a multiple of 16 (64-byte columns)
size_t nmgp_ld(size_t rows) { return ((rows + 15) / 16) * 16; }

int nmgp_scratch_get(nmgp_ctx* c, int slot, size_t nelem, double** out) {
    DevBuf& b = c->scratch[slot];
    if (b.cap < nelem || !b.p) {
        NMGP_TRY(nmgp_dev_alloc(c, &b.p, nelem));
        b.cap = nelem;
    } else if (nmgp_poison()) {
        hipMemsetAsync(b.p, 0xFF, nelem * sizeof(double), c->stream);
    }
    *out = b.p;
    return 0;
}

static void free_priors(nmgp_ctx* c) {
    for (auto& p : c->priors) {
        if (p.L) hipFree(p.L);
        if (p.logdet) hipFree(p.logdet);
    }
    c->priors.clear();
}

static void free_subject(nmgp_ctx* c) {
    double** ptrs[] = {&c->d_x, &c->d_Y, &c->d_y, &c->d_pars, &c->d_grad, &c->d_ell, &c->d_sig, &c->d_Lv,
                       &c->d_S, &c->d_Sinv, &c->d_z, &c->d_alpha, &c->d_R, &c->d_R2, &c->d_part, &c->d_K, &c->d_K2,
                       &c->d_w, &c->d_E};
    for (double** p : ptrs) {
        if (*p) hipFree(*p);
        *p = nullptr;
    }
    c->S_cap = c->K_cap = c->part_cap = 0;
    free_priors(c);
    {
        double** bp[] = {&c->b_pars, &c->b_ell, &c->b_Lv, &c->b_S, &c->b_z, &c->b_R, &c->b_scal, &c->b_q,
                         &c->b_S2, &c->b_Sinv, &c->b_alpha, &c->b_part, &c->b_grad, &c->b_R2, &c->b_tr};
        c->b_grad_ready = false;
        c->b_multi = false;
        if (c->b_x) hipFree(c->b_x);
        if (c->b_y) hipFree(c->b_y);
        c->b_x = c->b_y = nullptr;
        for (auto& pf : c->b_priors) {
            if (pf.L) hipFree(pf.L);
            if (pf.logdet) hipFree(pf.logdet);
        }
        c->b_priors.clear();
        for (double** p : bp) {
            if (*p) hipFree(*p);
            *p = nullptr;
        }
        if (c->b_info) hipFree(c->b_info);
        c->b_info = nullptr;
        c->batch = 0;
    }
}

// ---- context -----------------------------------------------------------------------------------
extern "C" int nmgp_version(void) { return NMGP_VERSION; }

extern "C" int nmgp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int nmgp_ctx_create(int device, nmgp_ctx** out) {
    if (!out) return NMGP_E_NULL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return NMGP_E_HIP;
    if (device < 0 || device >= ndev) return NMGP_E_SHAPE;
    nmgp_ctx* c = new nmgp_ctx();
    c->device = device;
    *out = c;   // returned even on failure below so that nmgp_last_error() is reachable
    if (const char* e = std::getenv("NMGP_CHOL")) c->chol_algo = (std::strcmp(e, "rocsolver") == 0) ? 0 : 1;
    if (const char* e = std::getenv("NMGP_CHOL_NB1")) {
        int v = std::atoi(e);
        if (v >= 64) c->chol_nb1 = (v / 64) * 64;
    }
    HIP_TRY(c, hipSetDevice(device));
    if (const char* e = std::getenv("NMGP_CHOL_LOOKAHEAD")) c->chol_lookahead = std::atoi(e);   // default 1: see potrf_lower
    if (const char* e = std::getenv("NMGP_PRIOR_SOLVE")) {
        c->prior_rocblas = std::strcmp(e, "rocblas") == 0;
        c->prior_trsv_all = std::strcmp(e, "trsv") == 0;
    }
    if (const char* e = std::getenv("NMGP_SEP")) c->sep_algo = (std::strcmp(e, "eig") == 0) ? 0 : 1;
    {
        // Two streams for the look-ahead factorisation: the main stream carries the latency-bound panel steps (and
        // everything else of an evaluation), stream2 the far trailing updates that run under the NEXT panel's steps.
        // An update tile holds a CU for ~100 us and two of them fill a CU's LDS, so a panel-step workgroup that shares the
        // chip with an update in flight would wait for a tile to retire (stream priorities alone: measured, no gain).
        // stream2 is therefore created with a CU MASK that leaves `la_cus` CUs -- the same CUs of every XCD: bit i of the
        // mask is CU i / 8 of XCD i % 8 -- free of update tiles; the unmasked main stream finds them idle whenever an update
        // is running, and uses the whole chip when none is.  NMGP_LOOKAHEAD_CUS=<n> (default 64; 0 = plain stream).
        int lo = 0, hi = 0;     // "greatest" (numerically lowest) priority is hi
        hipDeviceGetStreamPriorityRange(&lo, &hi);
        const bool prio = lo != hi;
        if (prio) HIP_TRY(c, hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi));
        else HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        int la_cus = 64;
        if (const char* e = std::getenv("NMGP_LOOKAHEAD_CUS")) la_cus = std::atoi(e);
        hipDeviceProp_t prop{};
        int ncu = 0;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) ncu = prop.multiProcessorCount;
        // The mask's layout (bit i = CU i / 8 of XCD i % 8) is MI355X's: 8 XCDs x 32 CUs.  On any other device the look-ahead stream
        // is a plain low-priority stream.  (hipExtStreamCreateWithCUMask makes a BLOCKING stream -- it synchronises with the legacy
        // NULL stream, which the library never uses -- and carries no priority; both are irrelevant next to the mask.)
        const bool mi355 = std::strncmp(prop.gcnArchName, "gfx950", 6) == 0 && ncu == 256;
        if (mi355 && la_cus > 0 && la_cus < ncu) {
            std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
            for (int i = la_cus; i < ncu; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
            if (hipExtStreamCreateWithCUMask(&c->stream2, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
                (void)hipGetLastError();
                c->stream2 = nullptr;
            } else {
                c->stream2_cus = ncu - la_cus;
            }
        }
        if (!c->stream2) {
            if (prio) HIP_TRY(c, hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, lo));
            else HIP_TRY(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        }
    }
    if (const char* e = std::getenv("NMGP_PRIOR_OVERLAP")) c->prior_overlap = std::atoi(e);
    HIP_TRY(c, hipStreamCreateWithFlags(&c->stream_prior, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_prior_fork, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_prior_join, hipEventDisableTiming));
    BLAS_TRY(c, rocblas_create_handle(&c->blas));
    BLAS_TRY(c, rocblas_set_stream(c->blas, c->stream));
    BLAS_TRY(c, rocblas_set_pointer_mode(c->blas, rocblas_pointer_mode_host));
    BLAS_TRY(c, rocblas_create_handle(&c->blas_prior));
    BLAS_TRY(c, rocblas_set_stream(c->blas_prior, c->stream_prior));
    BLAS_TRY(c, rocblas_set_pointer_mode(c->blas_prior, rocblas_pointer_mode_host));
    NMGP_TRY(nmgp_dev_alloc(c, &c->d_scal, SC_COUNT));
    HIP_TRY(c, hipMalloc((void**)&c->d_info, 8 * sizeof(int)));
    HIP_TRY(c, hipHostMalloc((void**)&c->h_pin, SC_COUNT * sizeof(double)));
    HIP_TRY(c, hipHostMalloc((void**)&c->h_info, 8 * sizeof(int)));
    HIP_TRY(c, hipMemsetAsync(c->d_info, 0, 8 * sizeof(int), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_scal, 0, SC_COUNT * sizeof(double), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int nmgp_ctx_destroy(nmgp_ctx* c) {
    if (!c) return NMGP_E_NULL;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    free_subject(c);
    for (auto& b : c->scratch)
        if (b.p) hipFree(b.p);
    for (auto& t : c->timers) {
        for (auto& pr : t.pending) {
            hipEventDestroy(pr.first);
            hipEventDestroy(pr.second);
        }
        for (auto& pr : t.pool) {
            hipEventDestroy(pr.first);
            hipEventDestroy(pr.second);
        }
    }
    if (c->d_scal) hipFree(c->d_scal);
    if (c->d_info) hipFree(c->d_info);
    if (c->h_pin) hipHostFree(c->h_pin);
    if (c->h_info) hipHostFree(c->h_info);
    if (c->blas) rocblas_destroy_handle(c->blas);
    if (c->blas_prior) rocblas_destroy_handle(c->blas_prior);
    if (c->stream2) {
        hipStreamSynchronize(c->stream2);
        hipStreamDestroy(c->stream2);
    }
    for (auto e : c->chol_ev) hipEventDestroy(e);
    if (c->stream_prior) {
        hipStreamSynchronize(c->stream_prior);
        hipStreamDestroy(c->stream_prior);
    }
    if (c->ev_prior_fork) hipEventDestroy(c->ev_prior_fork);
    if (c->ev_prior_join) hipEventDestroy(c->ev_prior_join);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

extern "C" const char* nmgp_last_error(const nmgp_ctx* c) { return c ? c->err.c_str() : "null context"; }

extern "C" int nmgp_sync(nmgp_ctx* c) {
    if (!c) return NMGP_E_NULL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

// ---- data residency ----------------------------------------------------------------------------
extern "C" int nmgp_set_data(nmgp_ctx* c, const double* x, const double* Y, int N, int M) {
    if (!c) return NMGP_E_NULL;
    if (!x || !Y) return nmgp_fail(c, NMGP_E_NULL, "nmgp_set_data: x and Y must not be NULL");
    if (N <= 0 || M <= 0) return nmgp_fail(c, NMGP_E_SHAPE, "nmgp_set_data: N=%d, M=%d must be positive", N, M);
    if (M > NMGP_MAX_OUTPUTS)
        return nmgp_fail(c, NMGP_E_UNSUPPORTED, "nmgp_set_data: M=%d exceeds NMGP_MAX_OUTPUTS=%d", M, NMGP_MAX_OUTPUTS);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_subject(c);
    c->N = N;
    c->M = M;
    c->T = M * (M + 1) / 2;
    c->n = N * M;
    const size_t Ns = N, T = c->T, n = c->n;
    c->P_svc = (long long)(Ns * (1 + T) + 1);
    size_t Pmax = Ns * (1 + T) + 1;
    if (2 * Ns + T + 1 > Pmax) Pmax = 2 * Ns + T + 1;
    NMGP_TRY(nmgp_dev_alloc(c, &c->d_x, Ns));
    NMGP_TRY(nmgp_dev_alloc(c, &c->d_Y, n));
    NMGP_TRY(nmgp_dev_alloc(c, &c->d_y, n));
    NMGP_TRY(nmgp_dev_alloc(c, &c->d_pars, Pmax));
    NMGP_TRY(nmgp_dev_alloc(c, &c->d_grad, Pmax));
    NMGP_TRY(nmgp_dev_alloc(c, &c->d_ell, Ns));
    NMGP_TRY(nmgp_dev_alloc(c, &c->d_sig, Ns));
    NMGP_TRY(nmgp_dev_alloc(c, &c->d_Lv, Ns * T));
    NMGP_TRY(nmgp_dev_alloc(c, &c->d_z, n));
    NMGP_TRY(nmgp_dev_alloc(c, &c->d_alpha, n));
    NMGP_TRY(nmgp_dev_alloc(c, &c->d_R, Ns * (1 + T)));
    NMGP_TRY(nmgp_dev_alloc(c, &c->d_R2, Ns * (1 + T)));
    HIP_TRY(c, hipMemcpyAsync(c->d_x, x, Ns * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_Y, Y, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    transpose_y(c->stream, c->d_Y, N, M, c->d_y);
    HIP_TRY(c, hipMemsetAsync(c->d_pars, 0, Pmax * sizeof(double), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

int nmgp_ensure_S(nmgp_ctx* c) {
    const size_t n = c->n;
    // rows: n (matrix) + 1 (right-hand side y) + n (identity -> L^-T, gradient path); even leading dimension
    const size_t ld = ((2 * n + 2 + 15) / 16) * 16;
    if (!c->d_S || c->S_cap < ld * n) {
        NMGP_TRY(nmgp_dev_alloc(c, &c->d_S, ld * n));
        c->S_cap = ld * n;
    }
    c->ldS = (int)ld;
    return 0;
}

// Cached Cholesky factor of RBF(x; alpha, beta) + jitter I (the GP-prior covariances of logpos.py:357,362).
int nmgp_get_prior(nmgp_ctx* c, double alpha, double beta, PriorFactor** out) {
    for (auto& p : c->priors)
        if (p.alpha == alpha && p.beta == beta) {
            *out = &p;
            return 0;
        }
    PriorFactor pf;
    pf.alpha = alpha;
    pf.beta = beta;
    const size_t Ns = c->N;
    pf.ld = (int)(((Ns + 15) / 16) * 16);
    NMGP_TRY(nmgp_dev_alloc(c, &pf.L, (size_t)pf.ld * Ns));
    if (nmgp_dev_alloc(c, &pf.logdet, 1) != 0) {
        hipFree(pf.L);
        return NMGP_E_NOMEM;
    }
    hipMemsetAsync(c->d_info + 1, 0, sizeof(int), c->stream);
    rbf_cov_sym(c->stream, c->d_x, c->N, alpha, beta, pf.L, pf.ld, false);
    // RBF + 1e-6 I has condition numbers ~1e11 at N = 2048: the factorisation must be backward stable
    // (rocSOLVER's inverse-based panel solve reports such matrices as indefinite; LAPACK and this one do not)
    int rc = nmgp_chol_factor(c, pf.L, pf.ld, c->N, 0, c->d_info + 1);
    half_logdet(c->stream, pf.L, pf.ld, c->N, pf.logdet);
    hipMemcpyAsync(c->h_info + 1, c->d_info + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (rc != 0 || e != hipSuccess || c->h_info[1] != 0) {
        int info = c->h_info[1];
        hipFree(pf.L);
        hipFree(pf.logdet);
        if (rc != 0) return rc;
        if (e != hipSuccess) return nmgp_fail(c, NMGP_E_HIP, "prior factorisation failed (%s)", hipGetErrorString(e));
        return nmgp_fail(c, info, "GP prior covariance RBF(alpha=%g, beta=%g)+jitter is not positive definite "
                         "(leading minor %d)", alpha, beta, info);
    }
    c->priors.push_back(pf);
    *out = &c->priors.back();
    return 0;
}

// events for the look-ahead factorisation (created once, reused by every evaluation)
hipEvent_t* nmgp_chol_events(nmgp_ctx* c, int n) {
    if (!c->chol_lookahead) return nullptr;
    const int nbmin = c->chol_nb1 > 0 ? c->chol_nb1 : 512;
    const size_t need = 2 * (size_t)((n + nbmin - 1) / nbmin) + 3;
    while (c->chol_ev.size() < need) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        c->chol_ev.push_back(e);
    }
    return c->chol_ev.data();
}

int nmgp_chol_factor(nmgp_ctx* c, double* A, int ld, int n, int extra, int* d_info) {
    if (c->chol_algo == 1 && (ld % 2 == 0)) {
        // used for the cached prior covariances, the dense-MVN primitive and prediction: accuracy before speed
        potrf_lower(c->stream, c->stream2, nmgp_chol_events(c, n), A, ld, n, extra, 0, c->chol_nb1, d_info, 1, 0, 0, nmgp_syrk_hook(c), 1);
        return 0;
    }
    if (extra != 0) return nmgp_fail(c, NMGP_E_STATE, "rocSOLVER path cannot carry extra rows");
    BLAS_TRY(c, rocsolver_dpotrf(c->blas, rocblas_fill_lower, n, A, ld, d_info));
    return 0;
}

// ---- nonseparable objective --------------------------------------------------------------------
static int svc_enqueue(nmgp_ctx* c, const double hyper[8], int prior, int want_grad) {
    if (!c->d_x) return nmgp_fail(c, NMGP_E_STATE, "nmgp_set_data must be called before evaluating");
    if (!hyper) return nmgp_fail(c, NMGP_E_NULL, "hyper must not be NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    NMGP_TRY(nmgp_ensure_S(c));
    const int N = c->N, M = c->M, T = c->T, n = c->n, ld = c->ldS;
    const long long P = c->P_svc;
    const double mu_l = hyper[0], al_l = hyper[1], be_l = hyper[2], mu_L = hyper[3], al_L = hyper[4], be_L = hyper[5];
    const double a = hyper[6], b = hyper[7];
    hipStream_t s = c->stream;
    double* sc = c->d_scal;
    // priors first: the factor pointers may be created (and synchronised) here
    PriorFactor *pl = nullptr, *pL = nullptr;
    NMGP_TRY(nmgp_get_prior(c, al_l, be_l, &pl));
    NMGP_TRY(nmgp_get_prior(c, al_L, be_L, &pL));
    // get_prior may have grown c->priors: re-resolve the first pointer
    NMGP_TRY(nmgp_get_prior(c, al_l, be_l, &pl));
    if (want_grad) {
        const size_t NJ = (N + 63) / 64;
        // adjoint partial rows; before that pass the same buffer holds the block sums of alpha = L^-T z (tri_gemv_upper)
        const size_t need = std::max(NJ * (size_t)N * (T + 1), (size_t)n * ((n + 255) / 256));
        if (!c->d_part || c->part_cap < need) {
            NMGP_TRY(nmgp_dev_alloc(c, &c->d_part, need));
            c->part_cap = need;
        }
    }
    HIP_TRY(c, hipMemsetAsync(c->d_info, 0, sizeof(int), s));
    if (nmgp_poison()) HIP_TRY(c, hipMemsetAsync(sc + SC_OUT, 0xFF, 8 * sizeof(double), s));   // stale results must not survive
    // The GP priors depend on the parameter vector only: they run on their own stream under the factorisation.  The fork
    // event is recorded HERE (so that stream does not wait for the factorisation), the solves themselves are enqueued after the
    // factorisation's launches: the host needs ~0.2 ms for the library's ~40 small launches, which must not delay the first
    // panel step.
    PriorStreamScope ps(c);
    {
        StageScope sp(c, NMGP_STAGE_COV);
        svc_prep(s, c->d_pars, N, M, c->d_ell, c->d_Lv);
        int r = svc_cov_build(s, c->d_x, c->d_ell, c->d_Lv, c->d_pars + (P - 1), c->d_S, ld, N, M, false);
        if (r) return nmgp_fail(c, r, "unsupported number of outputs M=%d", M);
    }
    const bool custom = (c->chol_algo == 1);
    const int xpad = (n + 1) & 1;            // rows between y and the identity block
    const int xoff = n + 1 + xpad;           // first row of X = L^-T (even)
    if (custom) {
        {
            StageScope sp(c, NMGP_STAGE_CHOL);
            set_row(s, c->d_S, ld, n, c->d_y, n, 1, 0, 0);              // y rides along as row n
            // with gradient: a zero pad row (keeps the next block at an even offset) and n identity rows -> X = L^-T
            if (want_grad) identity_rows(s, c->d_S, ld, n + 1, n, xpad);
            // row n becomes z = L^-1 y
            potrf_lower(s, c->stream2, nmgp_chol_events(c, n), c->d_S, ld, n, want_grad ? 1 + xpad : 1, want_grad ? n : 0,
                        c->chol_nb1, c->d_info, 1, 0, 0, nmgp_syrk_hook(c));
        }
        {
            StageScope sp(c, NMGP_STAGE_SOLVE);
            get_row(s, c->d_S, ld, n, c->d_z, n, 1, 0, 0);
            if (want_grad) {
                // alpha = Sigma^-1 y = L^-T z = X z (d_part is free until the adjoint pass: scratch for the block sums)
                tri_gemv_upper(s, c->d_S + xoff, (int)ld, n, c->d_z, c->d_alpha, c->d_part);
            }
        }
    } else {
        {
            StageScope sp(c, NMGP_STAGE_CHOL);
            BLAS_TRY(c, rocsolver_dpotrf(c->blas, rocblas_fill_lower, n, c->d_S, ld, c->d_info));
        }
        {
            StageScope sp(c, NMGP_STAGE_SOLVE);
            HIP_TRY(c, hipMemcpyAsync(c->d_z, c->d_y, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
            BLAS_TRY(c, rocblas_dtrsv(c->blas, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, n,
                                      c->d_S, ld, c->d_z, 1));
            if (want_grad) {
                HIP_TRY(c, hipMemcpyAsync(c->d_alpha, c->d_z, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
                BLAS_TRY(c, rocblas_dtrsv(c->blas, rocblas_fill_lower, rocblas_operation_transpose,
                                          rocblas_diagonal_non_unit, n, c->d_S, ld, c->d_alpha, 1));
            }
        }
    }
    {
        StageScope sp(c, NMGP_STAGE_REDUCE);
        chol_logdet_quad(s, c->d_S, ld, n, c->d_z, sc + SC_LOGDET, sc + SC_QUAD);
    }
    {
        NmgpStage sp(c, NMGP_STAGE_PRIOR, ps.sp, 0.0, 0.0);
        const double one = 1.0;
        svc_prior_rhs(ps.sp, c->d_pars, N, T, mu_l, mu_L, c->d_R, N);
        // (a single subject's 1 + T columns stay with the library: k_prior_trsv needs tens of workgroups to win, its
        // 7 workgroups took 0.69 ms against 0.26 ms at N = 2048)
        const bool subst = c->prior_trsv_all && !c->prior_rocblas && N <= 3500;
        if (subst) {
            prior_trsv(ps.sp, false, pl->L, pl->ld, 0, pL->L, pL->ld, 0, c->d_R, N, 1 + T, 1);
        } else if (pl == pL) {
            BLAS_TRY(c, rocblas_dtrsm(ps.hb, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none,
                                      rocblas_diagonal_non_unit, N, 1 + T, &one, pl->L, pl->ld, c->d_R, N));
        } else {
            BLAS_TRY(c, rocblas_dtrsm(ps.hb, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none,
                                      rocblas_diagonal_non_unit, N, 1, &one, pl->L, pl->ld, c->d_R, N));
            BLAS_TRY(c, rocblas_dtrsm(ps.hb, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none,
                                      rocblas_diagonal_non_unit, N, T, &one, pL->L, pL->ld, c->d_R + N, N));
        }
        col_sumsq(ps.sp, c->d_R, N, N, 1 + T, sc + SC_PRIORQ);
        if (want_grad && prior) {
            HIP_TRY(c, hipMemcpyAsync(c->d_R2, c->d_R, (size_t)N * (1 + T) * sizeof(double), hipMemcpyDeviceToDevice, ps.sp));
            if (subst) {
                prior_trsv(ps.sp, true, pl->L, pl->ld, 0, pL->L, pL->ld, 0, c->d_R2, N, 1 + T, 1);
            } else if (pl == pL) {
                BLAS_TRY(c, rocblas_dtrsm(ps.hb, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose,
                                          rocblas_diagonal_non_unit, N, 1 + T, &one, pl->L, pl->ld, c->d_R2, N));
            } else {
                BLAS_TRY(c, rocblas_dtrsm(ps.hb, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose,
                                          rocblas_diagonal_non_unit, N, 1, &one, pl->L, pl->ld, c->d_R2, N));
                BLAS_TRY(c, rocblas_dtrsm(ps.hb, rocblas_side_left, rocblas_fill_lower, rocblas_operation_transpose,
                                          rocblas_diagonal_non_unit, N, T, &one, pL->L, pL->ld, c->d_R2 + N, N));
            }
        }
    }
    ps.done();
    ps.join();
    {
        StageScope sp(c, NMGP_STAGE_REDUCE);
        const double ig_const = a * std::log(b) - std::lgamma(a);   // distributions.py:134
        svc_finalize(s, sc + SC_LOGDET, sc + SC_QUAD, sc + SC_PRIORQ, pl->logdet, pL->logdet, c->d_pars, P, N, T, a, b,
                     ig_const, prior, sc + SC_OUT);
    }
    if (want_grad) {
        const double* Sinv = c->d_S;
        int ldi = ld;
        double ssign = 1.0;
        {
            StageScope sp(c, NMGP_STAGE_INVERSE);
            if (custom) {
                // -Sigma^-1 = -(L^-T)(L^-T)^T: one more MFMA SYRK over the triangular X (k-panels left of a tile skipped)
                if (!c->d_Sinv) NMGP_TRY(nmgp_dev_alloc(c, &c->d_Sinv, (size_t)n * n));
                // (ktri = 2: both triangles are written, the adjoint pass reads the full symmetric matrix)
                syrk_lower(s, c->d_S + xoff, ld, c->d_Sinv, n, n, n, n, 1, 0, 0, 2);
                Sinv = c->d_Sinv;
                ldi = n;
                ssign = -1.0;
            } else {
                BLAS_TRY(c, rocsolver_dpotri(c->blas, rocblas_fill_lower, n, c->d_S, ld, c->d_info + 2));
                fill_lower_to_full(s, c->d_S, ld, n);
            }
        }
        {
            StageScope sp(c, NMGP_STAGE_ADJOINT);
            trace_terms(s, c->d_alpha, Sinv, ldi, n, sc + SC_TRACE, ssign);
            int r = svc_adjoint(s, c->d_x, c->d_ell, c->d_Lv, c->d_alpha, Sinv, ldi, N, M, c->d_part, ssign);
            if (r) return nmgp_fail(c, r, "unsupported number of outputs M=%d", M);
            svc_grad_final(s, c->d_part, (N + 63) / 64, N, M, c->d_Lv, c->d_R2, N, c->d_pars, sc + SC_TRACE, a, b,
                           prior, c->d_grad);
        }
    }
    c->last_want_grad = want_grad != 0;
    c->last_kind = 1;
    return nmgp_take_launch_error(c);
}

extern "C" int nmgp_svc_set_pars(nmgp_ctx* c, const double* pars) {
    if (!c) return NMGP_E_NULL;
    if (!pars) return nmgp_fail(c, NMGP_E_NULL, "pars must not be NULL");
    if (!c->d_pars) return nmgp_fail(c, NMGP_E_STATE, "nmgp_set_data must be called first");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(c->d_pars, pars, (size_t)c->P_svc * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return 0;
}

extern "C" double* nmgp_svc_pars_dev(nmgp_ctx* c) { return c ? c->d_pars : nullptr; }
extern "C" double* nmgp_svc_grad_dev(nmgp_ctx* c) { return c ? c->d_grad : nullptr; }

extern "C" int nmgp_svc_eval_resident(nmgp_ctx* c, const double hyper[8], int prior, int want_grad) {
    if (!c) return NMGP_E_NULL;
    return svc_enqueue(c, hyper, prior, want_grad);
}

extern "C" int nmgp_svc_fetch(nmgp_ctx* c, double out5[5], double* grad) {
    if (!c) return NMGP_E_NULL;
    if (c->last_kind != 1) return nmgp_fail(c, NMGP_E_STATE, "no nonseparable evaluation is pending");
    if (grad && !c->last_want_grad)
        return nmgp_fail(c, NMGP_E_STATE, "gradient requested but the last evaluation ran with want_grad=0");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    HIP_TRY(c, hipMemcpyAsync(c->h_pin, c->d_scal + SC_OUT, 8 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(c->h_info, c->d_info, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    if (grad)
        HIP_TRY(c, hipMemcpyAsync(grad, c->d_grad, (size_t)c->P_svc * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    NMGP_TRY(nmgp_take_launch_error(c));
    if (c->h_info[0] != 0)
        return nmgp_fail(c, c->h_info[0], "Cholesky of the %d x %d covariance failed: leading minor %d is not positive "
                         "definite", c->n, c->n, c->h_info[0]);
    if (out5)
        for (int k = 0; k < 5; ++k) out5[k] = c->h_pin[k];
    if (!std::isfinite(c->h_pin[0]) || !std::isfinite(c->h_pin[1]))
        return nmgp_fail(c, NMGP_NUM_NAN, "non-finite log posterior (NegLog=%g, loglik=%g)", c->h_pin[0], c->h_pin[1]);
    return 0;
}

extern "C" int nmgp_logpos_svc(nmgp_ctx* c, const double* pars, const double hyper[8], int prior, double out5[5],
                               double* grad) {
    if (!c) return NMGP_E_NULL;
    if (!out5) return nmgp_fail(c, NMGP_E_NULL, "out5 must not be NULL");
    NMGP_TRY(nmgp_svc_set_pars(c, pars));
    NMGP_TRY(svc_enqueue(c, hyper, prior, grad != nullptr));
    return nmgp_svc_fetch(c, out5, grad);
}

// ---- batched evaluation: B chains of the resident subject per launch sequence --------------------------
// MCMC runs many independent chains per subject; their factorisations have identical shapes, so every kernel of
// the forward path takes the chain index as a grid dimension.  The 96 latency-bound 64-wide panel steps of the
// Cholesky are then paid once per batch instead of once per chain, and the MFMA trailing updates fill the chip.
// back to the identity metric: every buffer of a mass matrix released
static void mass_reset(nmgp_ctx* c) {
    double** ptrs[] = {&c->b_minv, &c->b_vel, &c->b_mchol, &c->b_mU, &c->b_mw, &c->b_mc};
    for (double** p : ptrs) {
        if (*p) hipFree(*p);
        *p = nullptr;
    }
    c->b_mass_kind = 0;
    c->b_mrank = 0;
    c->b_traj_ready = false;             // a trajectory begun under the old metric must be begun again
}

static void free_batch(nmgp_ctx* c) {
    double** ptrs[] = {&c->b_pars, &c->b_ell, &c->b_Lv, &c->b_S, &c->b_z, &c->b_R, &c->b_scal, &c->b_q,
                       &c->b_S2, &c->b_Sinv, &c->b_alpha, &c->b_part, &c->b_grad, &c->b_R2, &c->b_tr,
                       &c->b_mom, &c->b_q0, &c->b_g0, &c->b_am, &c->b_av, &c->b_minv, &c->b_vel, &c->b_mchol, &c->b_kin,
                       &c->b_mU, &c->b_mw, &c->b_mc};
    c->b_mass_kind = 0;
    c->b_mrank = 0;
    c->b_cps = 1;
    if (c->b_alive) hipFree(c->b_alive);
    c->b_alive = nullptr;
    c->b_adam_t = -1;
    c->b_grad_ready = false;
    c->b_traj_ready = false;
    if (c->b_hmc) hipFree(c->b_hmc);
    c->b_hmc = nullptr;
    c->b_multi = false;
    if (c->b_x) hipFree(c->b_x);
    if (c->b_y) hipFree(c->b_y);
    c->b_x = c->b_y = nullptr;
    for (auto& pf : c->b_priors) {
        if (pf.L) hipFree(pf.L);
        if (pf.logdet) hipFree(pf.logdet);
    }
    c->b_priors.clear();
    for (double** p : ptrs) {
        if (*p) hipFree(*p);
        *p = nullptr;
    }
    if (c->b_info) hipFree(c->b_info);
    c->b_info = nullptr;
    c->batch = 0;
}

extern "C" int nmgp_svc_batch_alloc(nmgp_ctx* c, int B) {
    if (!c) return NMGP_E_NULL;
    if (!c->d_x) return nmgp_fail(c, NMGP_E_STATE, "nmgp_set_data must be called first");
    if (B <= 0) return nmgp_fail(c, NMGP_E_SHAPE, "batch size must be positive");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (B == c->batch && c->b_pars) {
        // The same batch size again (every sampler / optimiser object starts with this call): the B-sized buffers -- 116 GB of
        // factorisation and inverse workspace for 128 gradient chains at the headline size, seconds of hipMalloc -- are kept;
        // only the STATE goes back to that of a fresh batch: chains of the resident subject, identity metric, no trajectory, no
        // optimiser, zero parameters.
        mass_reset(c);
        c->b_multi = false;
        c->b_cps = 1;
        if (c->b_x) hipFree(c->b_x);
        if (c->b_y) hipFree(c->b_y);
        c->b_x = c->b_y = nullptr;
        for (auto& pf : c->b_priors) {
            if (pf.L) hipFree(pf.L);
            if (pf.logdet) hipFree(pf.logdet);
        }
        c->b_priors.clear();
        c->b_adam_t = -1;
        c->b_traj_ready = false;
        c->b_last_grad = false;
        if (c->last_kind == 2) c->last_kind = 0;           // no batched evaluation is pending any more
        HIP_TRY(c, hipMemsetAsync(c->b_pars, 0, (size_t)B * c->P_svc * sizeof(double), c->stream));
        return 0;
    }
    free_batch(c);
    const size_t N = c->N, T = c->T, n = c->n, P = (size_t)c->P_svc;
    const size_t ld = nmgp_ld(n + 1);
    NMGP_TRY(nmgp_dev_alloc(c, &c->b_pars, (size_t)B * P));
    NMGP_TRY(nmgp_dev_alloc(c, &c->b_ell, (size_t)B * N));
    NMGP_TRY(nmgp_dev_alloc(c, &c->b_Lv, (size_t)B * N * T));
    NMGP_TRY(nmgp_dev_alloc(c, &c->b_S, (size_t)B * ld * n));
    NMGP_TRY(nmgp_dev_alloc(c, &c->b_z, (size_t)B * n));
    NMGP_TRY(nmgp_dev_alloc(c, &c->b_R, N * (size_t)B * (1 + T)));
    NMGP_TRY(nmgp_dev_alloc(c, &c->b_scal, (size_t)B * 16));
    NMGP_TRY(nmgp_dev_alloc(c, &c->b_q, (size_t)B * (1 + T)));
    HIP_TRY(c, hipMalloc((void**)&c->b_info, (size_t)B * sizeof(int)));
    HIP_TRY(c, hipMemsetAsync(c->b_pars, 0, (size_t)B * P * sizeof(double), c->stream));
    c->batch = B;
    return 0;
}

extern "C" int nmgp_svc_batch_set_pars(nmgp_ctx* c, const double* pars) {
    if (!c) return NMGP_E_NULL;
    if (!pars) return nmgp_fail(c, NMGP_E_NULL, "pars must not be NULL");
    if (c->batch <= 0) return nmgp_fail(c, NMGP_E_STATE, "nmgp_svc_batch_alloc must be called first");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(c->b_pars, pars, (size_t)c->batch * c->P_svc * sizeof(double), hipMemcpyHostToDevice,
                              c->stream));
    c->b_traj_ready = false;
    return 0;
}

extern "C" double* nmgp_svc_batch_pars_dev(nmgp_ctx* c) { return c ? c->b_pars : nullptr; }

// Multi-subject batch (BASELINE config 4: many subjects of the same size, the reference runs one process per subject,
// Nonseparable_model_mpisim.py:305-306): batch element b gets its own inputs x[b], Y[b] and its own GP-prior factors.
// x: [B, N], Y: [B, N, M] row-major.  N and M are those of nmgp_set_data (whose subject is then ignored by the batch).
extern "C" int nmgp_svc_batch_set_subjects(nmgp_ctx* c, const double* x, const double* Y) {
    return nmgp_svc_batch_set_subjects_chains(c, x, Y, 1);
}

// The same with SEVERAL chains per subject: x [S, N], Y [S, N, M] for S = batch / chains_per_subject subjects; batch element
// b = s * chains_per_subject + k is chain k of subject s.  The chains of a subject share its inputs and its GP-prior factors (one
// pair per SUBJECT, not per chain).  8 subjects x 8 chains on one GPU is config 4's per-GPU subject count at the batch size
// where the factorisation is throughput-bound, and several chains per subject is what an R-hat needs anyway.
extern "C" int nmgp_svc_batch_set_subjects_chains(nmgp_ctx* c, const double* x, const double* Y, int chains_per_subject) {
    if (!c) return NMGP_E_NULL;
    if (!x || !Y) return nmgp_fail(c, NMGP_E_NULL, "x/Y must not be NULL");
    if (c->batch <= 0) return nmgp_fail(c, NMGP_E_STATE, "nmgp_svc_batch_alloc must be called first");
    if (chains_per_subject < 1 || c->batch % chains_per_subject != 0)
        return nmgp_fail(c, NMGP_E_SHAPE, "the batch size %d is not a multiple of chains_per_subject = %d", c->batch, chains_per_subject);
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t B = (size_t)c->batch / chains_per_subject, N = c->N, M = c->M, n = c->n;      // B = subjects
    hipStream_t s = c->stream;
    NMGP_TRY(nmgp_dev_alloc(c, &c->b_x, B * N));      // (re-allocated: the number of subjects may differ from the last call)
    NMGP_TRY(nmgp_dev_alloc(c, &c->b_y, B * n));
    double* tmp;
    NMGP_TRY(nmgp_scratch_get(c, 2, B * n, &tmp));
    HIP_TRY(c, hipMemcpyAsync(c->b_x, x, B * N * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(tmp, Y, B * n * sizeof(double), hipMemcpyHostToDevice, s));
    for (size_t b = 0; b < B; ++b) transpose_y(s, tmp + b * n, (int)N, (int)M, c->b_y + b * n);
    HIP_TRY(c, hipStreamSynchronize(s));
    for (auto& pf : c->b_priors) {
        if (pf.L) hipFree(pf.L);
        if (pf.logdet) hipFree(pf.logdet);
    }
    c->b_priors.clear();
    c->b_multi = true;
    c->b_cps = chains_per_subject;
    // new data: the resident gradient and a trajectory begun on the old subjects no longer describe the batch
    c->b_traj_ready = false;
    c->b_last_grad = false;
    if (c->b_mass_kind == 3) mass_reset(c);      // a prior-factor metric belongs to the subjects it was built for: back to identity
    return 0;
}

// per-subject Cholesky factors of RBF(x_b; alpha, beta) + jitter I for the whole batch (one batched factorisation)
static int get_batch_prior(nmgp_ctx* c, double alpha, double beta, PriorFactor** out) {
    for (auto& p : c->b_priors)
        if (p.alpha == alpha && p.beta == beta) {
            *out = &p;
            return 0;
        }
    PriorFactor pf;
    pf.alpha = alpha;
    pf.beta = beta;
    const size_t N = c->N, B = (size_t)c->batch / c->b_cps;      // one factor per SUBJECT
    pf.ld = (int)(((N + 15) / 16) * 16);
    NMGP_TRY(nmgp_dev_alloc(c, &pf.L, B * (size_t)pf.ld * N));
    if (nmgp_dev_alloc(c, &pf.logdet, B) != 0) {
        hipFree(pf.L);
        return NMGP_E_NOMEM;
    }
    int* info;
    HIP_TRY(c, hipMalloc((void**)&info, B * sizeof(int)));
    hipMemsetAsync(info, 0, B * sizeof(int), c->stream);
    rbf_cov_sym(c->stream, c->b_x, c->N, alpha, beta, pf.L, pf.ld, false, (int)B);
    potrf_lower(c->stream, c->stream2, nmgp_chol_events(c, c->N), pf.L, pf.ld, c->N, 0, 0, c->chol_nb1, info, (int)B,
                (long long)pf.ld * N, 1, nmgp_syrk_hook(c), 1);
    half_logdet(c->stream, pf.L, pf.ld, c->N, pf.logdet, (int)B);
    std::vector<int> hi(B);
    hipMemcpyAsync(hi.data(), info, B * sizeof(int), hipMemcpyDeviceToHost, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(info);
    int bad = 0;
    for (size_t b = 0; b < B; ++b)
        if (hi[b] != 0 && bad == 0) bad = hi[b];
    if (e != hipSuccess || bad != 0) {
        hipFree(pf.L);
        hipFree(pf.logdet);
        if (e != hipSuccess) return nmgp_fail(c, NMGP_E_HIP, "batched prior factorisation failed (%s)", hipGetErrorString(e));
        return nmgp_fail(c, bad, "GP prior covariance RBF(alpha=%g, beta=%g)+jitter of a subject is not positive definite "
                         "(leading minor %d)", alpha, beta, bad);
    }
    c->b_priors.push_back(pf);
    *out = &c->b_priors.back();
    return 0;
}

static int batch_grad_alloc(nmgp_ctx* c) {
    if (c->b_grad_ready) return 0;
    const size_t N = c->N, T = c->T, n = c->n, P = (size_t)c->P_svc, B = c->batch;
    const size_t ld2 = nmgp_ld(2 * n + 2);
    const size_t NJ = (N + 63) / 64;
    NMGP_TRY(nmgp_dev_alloc(c, &c->b_S2, B * ld2 * n));
    NMGP_TRY(nmgp_dev_alloc(c, &c->b_Sinv, B * n * n));
    NMGP_TRY(nmgp_dev_alloc(c, &c->b_alpha, B * n));
    // adjoint partial rows (NJ N (T + 1) per chain); before that pass the block sums of alpha = L^-T z (n ceil(n / 256) per chain)
    NMGP_TRY(nmgp_dev_alloc(c, &c->b_part, B * std::max(NJ * N * (T + 1), (size_t)n * ((n + 255) / 256))));
    NMGP_TRY(nmgp_dev_alloc(c, &c->b_grad, B * P));
    NMGP_TRY(nmgp_dev_alloc(c, &c->b_R2, N * B * (1 + T)));
    NMGP_TRY(nmgp_dev_alloc(c, &c->b_tr, B * 2));
    c->b_grad_ready = true;
    return 0;
}

extern "C" int nmgp_svc_batch_eval(nmgp_ctx* c, const double hyper[8], int prior, int want_grad) {
    if (!c) return NMGP_E_NULL;
    if (!hyper) return nmgp_fail(c, NMGP_E_NULL, "hyper must not be NULL");
    if (c->batch <= 0) return nmgp_fail(c, NMGP_E_STATE, "nmgp_svc_batch_alloc must be called first");
    HIP_TRY(c, hipSetDevice(c->device));
    const int N = c->N, M = c->M, T = c->T, n = c->n, B = c->batch;
    const long long P = c->P_svc;
    c->b_traj_ready = false;             // the resident gradient no longer belongs to a trajectory's start point
    if (want_grad) NMGP_TRY(batch_grad_alloc(c));
    // value-only: rows = n + 1 (y); with gradient: + pad + n identity rows (-> L^-T), in the larger buffer
    const int xpad = (n + 1) & 1, xoff = n + 1 + xpad;
    const int ld = want_grad ? (int)nmgp_ld((size_t)2 * n + 2) : (int)nmgp_ld((size_t)n + 1);
    double* S = want_grad ? c->b_S2 : c->b_S;
    const long long bs = (long long)ld * n;
    const double mu_l = hyper[0], al_l = hyper[1], be_l = hyper[2], mu_L = hyper[3], al_L = hyper[4], be_L = hyper[5];
    const double a = hyper[6], b = hyper[7];
    hipStream_t s = c->stream;
    const bool multi = c->b_multi;                  // the batch holds several subjects (b_cps consecutive chains each)
    const int cps = multi ? c->b_cps : 1;
    const double* xs = multi ? c->b_x : c->d_x;
    const int xstride = multi ? N : 0;
    PriorFactor *pl = nullptr, *pL = nullptr;
    // (the factor cache is a vector: the second look-up may grow it and move its elements, so the first pointer is re-resolved)
    if (multi) {
        NMGP_TRY(get_batch_prior(c, al_l, be_l, &pl));
        NMGP_TRY(get_batch_prior(c, al_L, be_L, &pL));
        NMGP_TRY(get_batch_prior(c, al_l, be_l, &pl));
    } else {
        NMGP_TRY(nmgp_get_prior(c, al_l, be_l, &pl));
        NMGP_TRY(nmgp_get_prior(c, al_L, be_L, &pL));
        NMGP_TRY(nmgp_get_prior(c, al_l, be_l, &pl));
    }
    HIP_TRY(c, hipMemsetAsync(c->b_info, 0, (size_t)B * sizeof(int), s));
    if (nmgp_poison()) HIP_TRY(c, hipMemsetAsync(c->b_scal, 0xFF, (size_t)B * 16 * sizeof(double), s));
    PriorStreamScope ps(c);          // fork now, enqueue the prior solves after the factorisation (see svc_enqueue)
    {
        StageScope sp(c, NMGP_STAGE_COV);
        svc_prep(s, c->b_pars, N, M, c->b_ell, c->b_Lv, B);
        int r = svc_cov_build(s, xs, c->b_ell, c->b_Lv, c->b_pars + (P - 1), S, ld, N, M, false, B, bs, xstride, cps);
        if (r) return nmgp_fail(c, r, "unsupported number of outputs M=%d", M);
    }
    {
        StageScope sp(c, NMGP_STAGE_CHOL);
        // one subject: every chain shares y (vstride 0); multi-subject: y of batch element b
        set_row(s, S, ld, n, multi ? c->b_y : c->d_y, n, B, bs, multi ? n : 0, cps);
        if (want_grad) identity_rows(s, S, ld, n + 1, n, xpad, B, bs);
        potrf_lower(s, c->stream2, nmgp_chol_events(c, n), S, ld, n, want_grad ? 1 + xpad : 1, want_grad ? n : 0, c->chol_nb1,
                    c->b_info, B, bs, 1, nmgp_syrk_hook(c));
    }
    {
        StageScope sp(c, NMGP_STAGE_SOLVE);
        get_row(s, S, ld, n, c->b_z, n, B, bs, n);
        if (want_grad) {
            // alpha_b = X_b z_b (b_part is free until the adjoint pass: scratch for the block sums)
            tri_gemv_upper(s, S + xoff, (int)ld, n, c->b_z, c->b_alpha, c->b_part, (int)B, bs,
                           (long long)n * ((n + 255) / 256));
        }
    }
    {
        StageScope sp(c, NMGP_STAGE_REDUCE);
        chol_logdet_quad(s, S, ld, n, c->b_z, c->b_scal, c->b_scal + 1, B, bs, 16);
    }
    {
        NmgpStage sp(c, NMGP_STAGE_PRIOR, ps.sp, 0.0, 0.0);
        const double one = 1.0;
        svc_prior_rhs(ps.sp, c->b_pars, N, T, mu_l, mu_L, c->b_R, N, B);
        const bool same = (pl == pL);
        for (int pass = 0; pass < ((want_grad && prior) ? 2 : 1); ++pass) {
            double* R = pass == 0 ? c->b_R : c->b_R2;
            const rocblas_operation op = pass == 0 ? rocblas_operation_none : rocblas_operation_transpose;
            if (pass == 1)
                HIP_TRY(c, hipMemcpyAsync(c->b_R2, c->b_R, (size_t)N * B * (1 + T) * sizeof(double),
                                          hipMemcpyDeviceToDevice, ps.sp));
            if (N <= 3500 && !c->prior_rocblas && (c->prior_trsv_all || (multi && B * (1 + T) >= 32))) {
                // one streaming pass per right-hand side over its factor (k_prior_trsv; per-subject factors in a multi-subject
                // batch, the subject's shared factors otherwise: stride 0)
                prior_trsv(ps.sp, pass == 1, pl->L, pl->ld, multi ? (long long)pl->ld * N : 0, pL->L, pL->ld,
                           multi ? (long long)pL->ld * N : 0, R, N, 1 + T, (int)B, cps);
            } else if (multi && cps > 1) {
                // several chains per subject through the library: the chains of a subject are consecutive, so their (1 + T) cps
                // columns form one right-hand-side block per subject
                const rocblas_stride sA_l = (rocblas_stride)pl->ld * N, sA_L = (rocblas_stride)pL->ld * N;
                const rocblas_stride sB = (rocblas_stride)(1 + T) * N;
                const int S_ = B / cps;
                for (int sj = 0; sj < S_; ++sj) {
                    double* Rs = R + (size_t)sj * cps * sB;
                    if (same) {
                        BLAS_TRY(c, rocblas_dtrsm(ps.hb, rocblas_side_left, rocblas_fill_lower, op, rocblas_diagonal_non_unit, N,
                                                  cps * (1 + T), &one, pl->L + (size_t)sj * sA_l, pl->ld, Rs, N));
                    } else {
                        BLAS_TRY(c, rocblas_dtrsm_strided_batched(ps.hb, rocblas_side_left, rocblas_fill_lower, op,
                                                                  rocblas_diagonal_non_unit, N, 1, &one, pl->L + (size_t)sj * sA_l,
                                                                  pl->ld, 0, Rs, N, sB, cps));
                        BLAS_TRY(c, rocblas_dtrsm_strided_batched(ps.hb, rocblas_side_left, rocblas_fill_lower, op,
                                                                  rocblas_diagonal_non_unit, N, T, &one, pL->L + (size_t)sj * sA_L,
                                                                  pL->ld, 0, Rs + N, N, sB, cps));
                    }
                }
            } else if (multi) {
                // per-subject factors: strided-batched solves (columns of chain b start at b (1+T) N)
                const rocblas_stride sA_l = (rocblas_stride)pl->ld * N, sA_L = (rocblas_stride)pL->ld * N;
                const rocblas_stride sB = (rocblas_stride)(1 + T) * N;
                if (same) {
                    BLAS_TRY(c, rocblas_dtrsm_strided_batched(ps.hb, rocblas_side_left, rocblas_fill_lower, op,
                                                              rocblas_diagonal_non_unit, N, 1 + T, &one, pl->L, pl->ld, sA_l,
                                                              R, N, sB, B));
                } else {
                    BLAS_TRY(c, rocblas_dtrsm_strided_batched(ps.hb, rocblas_side_left, rocblas_fill_lower, op,
                                                              rocblas_diagonal_non_unit, N, 1, &one, pl->L, pl->ld, sA_l, R,
                                                              N, sB, B));
                    BLAS_TRY(c, rocblas_dtrsm_strided_batched(ps.hb, rocblas_side_left, rocblas_fill_lower, op,
                                                              rocblas_diagonal_non_unit, N, T, &one, pL->L, pL->ld, sA_L,
                                                              R + N, N, sB, B));
                }
            } else if (same) {
                // one multi-right-hand-side solve for the whole batch: the prior factor depends on (x, alpha, beta) only
                BLAS_TRY(c, rocblas_dtrsm(ps.hb, rocblas_side_left, rocblas_fill_lower, op, rocblas_diagonal_non_unit, N,
                                          B * (1 + T), &one, pl->L, pl->ld, R, N));
            } else {
                // two factors shared by all chains (alpha_l / beta_l differ from alpha_L / beta_L, e.g. the reference's
                // _distributed hyper-parameters): column 0 of every chain against the first, its T other columns against the
                // second -- two strided-batched calls with factor stride 0 instead of 2 B library calls
                const rocblas_stride sB = (rocblas_stride)(1 + T) * N;
                BLAS_TRY(c, rocblas_dtrsm_strided_batched(ps.hb, rocblas_side_left, rocblas_fill_lower, op,
                                                          rocblas_diagonal_non_unit, N, 1, &one, pl->L, pl->ld, 0, R, N, sB, B));
                BLAS_TRY(c, rocblas_dtrsm_strided_batched(ps.hb, rocblas_side_left, rocblas_fill_lower, op,
                                                          rocblas_diagonal_non_unit, N, T, &one, pL->L, pL->ld, 0, R + N, N, sB,
                                                          B));
            }
            if (pass == 0) col_sumsq(ps.sp, c->b_R, N, N, B * (1 + T), c->b_q);
        }
    }
    ps.done();
    ps.join();
    {
        StageScope sp(c, NMGP_STAGE_REDUCE);
        const double ig_const = a * std::log(b) - std::lgamma(a);
        svc_finalize(s, c->b_scal, c->b_scal + 1, c->b_q, pl->logdet, pL->logdet, c->b_pars, P, N, T, a, b, ig_const,
                     prior, c->b_scal + 8, B, 16, multi ? 1 : 0, cps);
    }
    if (want_grad) {
        {
            StageScope sp(c, NMGP_STAGE_INVERSE);
            syrk_lower(s, S + xoff, ld, c->b_Sinv, n, n, n, n, B, bs, (long long)n * n, 2);     // -Sigma^-1 = -X X^T, both triangles
        }
        {
            StageScope sp(c, NMGP_STAGE_ADJOINT);
            trace_terms(s, c->b_alpha, c->b_Sinv, n, n, c->b_tr, -1.0, B);
            int r = svc_adjoint(s, xs, c->b_ell, c->b_Lv, c->b_alpha, c->b_Sinv, n, N, M, c->b_part, -1.0, B, xstride, cps);
            if (r) return nmgp_fail(c, r, "unsupported number of outputs M=%d", M);
            svc_grad_final(s, c->b_part, (N + 63) / 64, N, M, c->b_Lv, c->b_R2, N, c->b_pars, c->b_tr, a, b, prior,
                           c->b_grad, B);
        }
    }
    c->b_last_grad = want_grad != 0;
    c->last_kind = 2;
    return nmgp_take_launch_error(c);
}

extern "C" double* nmgp_svc_batch_grad_dev(nmgp_ctx* c) { return c ? c->b_grad : nullptr; }

// grad: [B, P] = d NegLog / d pars of every chain of the last batched evaluation (which must have asked for it)
extern "C" int nmgp_svc_batch_fetch_grad(nmgp_ctx* c, double* grad) {
    if (!c) return NMGP_E_NULL;
    if (!grad) return nmgp_fail(c, NMGP_E_NULL, "grad must not be NULL");
    if (c->last_kind != 2 || !c->b_last_grad)
        return nmgp_fail(c, NMGP_E_STATE, "the last batched evaluation did not compute gradients");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(grad, c->b_grad, (size_t)c->batch * c->P_svc * sizeof(double), hipMemcpyDeviceToHost,
                              c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

// out: [B, 5] verbose tuples; status: [B] (0 ok, k > 0 leading minor k not positive definite, NMGP_NUM_NAN).
// A failing chain does not fail the call: the return value is 0 and its status / NaN row tell.
extern "C" int nmgp_svc_batch_fetch(nmgp_ctx* c, double* out, int* status) {
    if (!c) return NMGP_E_NULL;
    if (!out) return nmgp_fail(c, NMGP_E_NULL, "out must not be NULL");
    if (c->last_kind != 2) return nmgp_fail(c, NMGP_E_STATE, "no batched evaluation is pending");
    HIP_TRY(c, hipSetDevice(c->device));
    const int B = c->batch;
    std::vector<double> h((size_t)B * 16);
    std::vector<int> hi(B);
    HIP_TRY(c, hipMemcpyAsync(h.data(), c->b_scal, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(hi.data(), c->b_info, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    NMGP_TRY(nmgp_take_launch_error(c));
    for (int z = 0; z < B; ++z) {
        int st = hi[z];
        for (int k = 0; k < 5; ++k) out[(size_t)z * 5 + k] = h[(size_t)z * 16 + 8 + k];
        if (st == 0 && (!std::isfinite(out[(size_t)z * 5]) || !std::isfinite(out[(size_t)z * 5 + 1]))) st = NMGP_NUM_NAN;
        if (st != 0)
            for (int k = 0; k < 5; ++k) out[(size_t)z * 5 + k] = std::nan("");
        if (status) status[z] = st;
    }
    return 0;
}

// ---- device-resident leapfrog trajectories of the B chains (drivers.py BatchedHMC) -----------------------------------------
// State between calls: b_pars = positions q, b_grad = dU/dq at q (U = NegLog), validity flags; the host keeps U and draws the
// momenta / the accept uniforms (so a chain reproduces the single-chain sampler's random stream).
extern "C" int nmgp_svc_batch_traj_begin(nmgp_ctx* c) {
    if (!c) return NMGP_E_NULL;
    if (c->batch <= 0 || c->last_kind != 2 || !c->b_last_grad)
        return nmgp_fail(c, NMGP_E_STATE, "a batched value+gradient evaluation of the start positions must come first");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t B = c->batch, P = (size_t)c->P_svc;
    if (!c->b_mom) {
        NMGP_TRY(nmgp_dev_alloc(c, &c->b_mom, B * P));
        NMGP_TRY(nmgp_dev_alloc(c, &c->b_q0, B * P));
        NMGP_TRY(nmgp_dev_alloc(c, &c->b_g0, B * P));
        HIP_TRY(c, hipMalloc((void**)&c->b_hmc, 4 * B * sizeof(int)));
    }
    hmc_status(c->stream, c->b_info, c->b_scal, c->b_hmc, nullptr, (int)B);
    c->b_traj_ready = true;
    return nmgp_take_launch_error(c);
}

// Mass matrix of the device-resident trajectories (the reference's production sampler passes a dense M = inv(sample covariance)
// with step size 1e-1 and 5 leapfrog steps: Nonseparable_model_mpiKAISER.py:267-270,398-411).  kind 0: identity (default);
// 1: diagonal, minv = diag(M^-1) [P]; 2: dense, minv = M^-1 [P, P] (symmetric; P = 14,337 at the headline size: 1.6 GB, shared by
// all chains of the batch).  The device only needs the VELOCITY M^-1 p of the drift q += eps M^-1 p -- one GEMM [P, P] x [P, B] per
// leapfrog step for all chains; the caller draws the momenta p ~ N(0, M) and evaluates the kinetic energy 1/2 p^T M^-1 p.
extern "C" int nmgp_svc_batch_traj_set_mass(nmgp_ctx* c, int kind, const double* minv) {
    if (!c) return NMGP_E_NULL;
    if (c->batch <= 0) return nmgp_fail(c, NMGP_E_STATE, "nmgp_svc_batch_alloc must be called first");
    if (kind < 0 || kind > 2) return nmgp_fail(c, NMGP_E_SHAPE, "mass matrix kind must be 0 (identity), 1 (diagonal) or 2 (dense)");
    if (kind != 0 && !minv) return nmgp_fail(c, NMGP_E_NULL, "minv must not be NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const size_t P = (size_t)c->P_svc, B = c->batch;
    mass_reset(c);
    if (kind == 0) return 0;
    const size_t nelem = kind == 1 ? P : P * P;
    NMGP_TRY(nmgp_dev_alloc(c, &c->b_minv, nelem));
    if (kind == 2) NMGP_TRY(nmgp_dev_alloc(c, &c->b_vel, B * P));
    HIP_TRY(c, hipMemcpyAsync(c->b_minv, minv, nelem * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->b_mass_kind = kind;
    return 0;
}

// A square root of the mass matrix set by nmgp_svc_batch_traj_set_mass (same kind): diagonal: sqrt(diag M) [P]; dense: R with
// R R^T = M, COLUMN-major [P, P] (element (i, j) at mchol[i + j P]; e.g. the lower Cholesky factor with zeros above the diagonal).  With it
// nmgp_svc_batch_traj_z draws the momenta on the device: p = chol(M) z from the standard normals z the caller uploads.
extern "C" int nmgp_svc_batch_traj_set_mass_chol(nmgp_ctx* c, int kind, const double* mchol) {
    if (!c) return NMGP_E_NULL;
    if (c->batch <= 0) return nmgp_fail(c, NMGP_E_STATE, "nmgp_svc_batch_alloc must be called first");
    if (kind != c->b_mass_kind) return nmgp_fail(c, NMGP_E_STATE, "nmgp_svc_batch_traj_set_mass(kind = %d) must come first (current kind %d)", kind, c->b_mass_kind);
    if (kind == 0) return 0;
    if (!mchol) return nmgp_fail(c, NMGP_E_NULL, "mchol must not be NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t P = (size_t)c->P_svc;
    const size_t nelem = kind == 1 ? P : P * P;
    if (c->b_mchol) hipFree(c->b_mchol);
    c->b_mchol = nullptr;
    NMGP_TRY(nmgp_dev_alloc(c, &c->b_mchol, nelem));
    HIP_TRY(c, hipMemcpyAsync(c->b_mchol, mchol, nelem * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

// The cached GP-prior factors the prior-factor metric is made of (per subject in a multi-subject batch).  Looked up by
// (alpha, beta) at every use: the cache is a vector that may have grown since -- pointers into it are not kept.
static int metric_factors(nmgp_ctx* c, PriorFactor** pl, PriorFactor** pL) {
    const double* h = c->b_mhyp;
    if (c->b_multi) {
        NMGP_TRY(get_batch_prior(c, h[0], h[1], pl));
        NMGP_TRY(get_batch_prior(c, h[2], h[3], pL));
        NMGP_TRY(get_batch_prior(c, h[0], h[1], pl));      // (re-resolved: the second call may have grown the cache)
    } else {
        NMGP_TRY(nmgp_get_prior(c, h[0], h[1], pl));
        NMGP_TRY(nmgp_get_prior(c, h[2], h[3], pL));
        NMGP_TRY(nmgp_get_prior(c, h[0], h[1], pl));
    }
    return 0;
}

// out[b] (op)= coef op(L_blk) in[b] for the whole batch (nmgp_metric.hip)
static int metric_trmm(nmgp_ctx* c, bool trans, const double* in, double* out, double coef, int mode, const int* bad) {
    PriorFactor *pl = nullptr, *pL = nullptr;
    NMGP_TRY(metric_factors(c, &pl, &pL));
    const int N = c->N;
    const bool multi = c->b_multi;
    prior_trmm(c->stream, trans, pl->L, pl->ld, multi ? (long long)pl->ld * N : 0, pL->L, pL->ld, multi ? (long long)pL->ld * N : 0,
               in, out, N, c->T, c->P_svc, c->batch, multi ? c->b_cps : c->batch, coef, mode, bad);
    return 0;
}
// (single subject: cps = batch makes every chain "subject 0" of the metric's per-subject tables)
static inline int metric_cps(const nmgp_ctx* c) { return c->b_multi ? c->b_cps : c->batch; }

// p -= c g (chains whose gradient is defined), then -- if drift -- q += eps M^-1 p
static int traj_kick_drift(nmgp_ctx* c, double kick, double eps, int drift) {
    const int B = c->batch;
    const long long P = c->P_svc;
    hipStream_t s = c->stream;
    int* bad = c->b_hmc;
    if (c->b_mass_kind == 0) {
        hmc_kick_drift(s, c->b_mom, c->b_grad, c->b_pars, bad, kick, eps, drift, P, B);
        return 0;
    }
    if (c->b_mass_kind == 3) {
        // whitened momentum u = L_blk^T p:  u -= c L_blk^T g;  q += eps L_blk W u,  W = I - U diag(lam / (1 + lam)) U^T
        NMGP_TRY(metric_trmm(c, true, c->b_grad, c->b_mom, kick, 2, bad));
        if (!drift) return 0;
        const double* v = c->b_mom;
        const int r = c->b_mrank;
        if (r > 0) {
            const size_t Sr = (size_t)(B / metric_cps(c)) * r;
            lowrank_proj(s, c->b_mU, c->b_mom, c->b_mc, P, r, B, metric_cps(c));
            lowrank_apply(s, c->b_mU, c->b_mw + Sr, c->b_mc, c->b_mom, c->b_vel, P, r, B, metric_cps(c));
            v = c->b_vel;
        }
        return metric_trmm(c, false, v, c->b_pars, eps, 1, nullptr);
    }
    hmc_kick_drift(s, c->b_mom, c->b_grad, c->b_pars, bad, kick, eps, 0, P, B);
    if (!drift) return 0;
    if (c->b_mass_kind == 2) {
        // V [P x B] = M^-1 [P x P] * momenta [P x B] (the [B, P] row-major block IS column-major [P, B])
        const double one = 1.0, zero = 0.0;
        BLAS_TRY(c, rocblas_dgemm(c->blas, rocblas_operation_none, rocblas_operation_none, (int)P, B, (int)P, &one, c->b_minv,
                                  (int)P, c->b_mom, (int)P, &zero, c->b_vel, (int)P));
        hmc_drift(s, c->b_pars, c->b_mom, c->b_vel, nullptr, eps, P, B);
    } else {
        hmc_drift(s, c->b_pars, c->b_mom, nullptr, c->b_minv, eps, P, B);
    }
    return 0;
}

// An API-level failure anywhere in a trajectory call puts the start state back (positions, gradients, validity flags) and demands a
// fresh value+gradient evaluation + nmgp_svc_batch_traj_begin; the first failure's message is kept.
static int traj_abort(nmgp_ctx* c, int rc) {
    const int B = c->batch;
    const size_t bytes = (size_t)B * c->P_svc * sizeof(double);
    hipStream_t s = c->stream;
    const std::string msg = c->err;
    hipMemcpyAsync(c->b_pars, c->b_q0, bytes, hipMemcpyDeviceToDevice, s);
    hipMemcpyAsync(c->b_grad, c->b_g0, bytes, hipMemcpyDeviceToDevice, s);
    hipMemcpyAsync(c->b_hmc, c->b_hmc + B, (size_t)B * sizeof(int), hipMemcpyDeviceToDevice, s);
    hipStreamSynchronize(s);
    (void)hipGetLastError();
    c->b_last_grad = false;
    c->b_traj_ready = false;
    c->err = msg;
    return rc;
}

// The prior-factor metric (kind 3; nmgp_metric.hip):  M^-1 = L_blk (I + U diag(lam) U^T)^-1 L_blk^T  with L_blk the block diagonal of
// the cached Cholesky factors of the GP priors named by hyper (logpos.py:357-365: RBF(alpha_tilde_l, beta_tilde_l) + 1e-6 I for
// tilde_l, RBF(alpha_L, beta_L) + 1e-6 I for each stride-T column of uL_vecs, 1 for tilde_sigma2_err) and an optional rank-r
// correction for the curvature the likelihood adds in the whitened coordinates.  U: [S, r, P] (subject s's orthonormal directions as
// r rows of length P; S = number of subjects of the batch, 1 without nmgp_svc_batch_set_subjects), lam: [S, r] >= 0.  rank 0: the
// pure prior metric (U, lam may be NULL).
extern "C" int nmgp_svc_batch_traj_set_mass_prior(nmgp_ctx* c, const double hyper[8], int rank, const double* U, const double* lam) {
    if (!c) return NMGP_E_NULL;
    if (!hyper) return nmgp_fail(c, NMGP_E_NULL, "hyper must not be NULL");
    if (c->batch <= 0) return nmgp_fail(c, NMGP_E_STATE, "nmgp_svc_batch_alloc must be called first");
    if (rank < 0 || rank > 1024) return nmgp_fail(c, NMGP_E_SHAPE, "rank of the metric's correction must be in [0, 1024]");
    if (rank > 0 && (!U || !lam)) return nmgp_fail(c, NMGP_E_NULL, "U / lam must not be NULL for rank > 0");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const size_t P = (size_t)c->P_svc, B = c->batch;
    const size_t S = c->b_multi ? B / c->b_cps : 1;
    for (size_t k = 0; k < S * (size_t)rank; ++k)
        if (!(lam[k] >= 0.0) || !std::isfinite(lam[k]))
            return nmgp_fail(c, NMGP_E_SHAPE, "lam[%zu] = %g: the correction's eigenvalues must be finite and >= 0", k, lam[k]);
    mass_reset(c);
    c->b_mhyp[0] = hyper[1];
    c->b_mhyp[1] = hyper[2];
    c->b_mhyp[2] = hyper[4];
    c->b_mhyp[3] = hyper[5];
    c->b_mass_kind = 3;                          // (metric_factors reads b_mhyp)
    PriorFactor *pl = nullptr, *pL = nullptr;
    int rc = metric_factors(c, &pl, &pL);        // builds the factors now, so that a non-PD prior is reported here
    if (rc == 0 && rank > 0) {
        std::vector<double> w(3 * S * rank);
        for (size_t k = 0; k < S * (size_t)rank; ++k) {
            w[k] = std::sqrt(1.0 + lam[k]) - 1.0;
            w[S * rank + k] = -lam[k] / (1.0 + lam[k]);
            w[2 * S * rank + k] = lam[k] / (1.0 + lam[k]);
        }
        rc = nmgp_dev_alloc(c, &c->b_mU, S * rank * P);
        if (rc == 0) rc = nmgp_dev_alloc(c, &c->b_mw, 3 * S * rank);
        if (rc == 0) rc = nmgp_dev_alloc(c, &c->b_mc, B * rank);
        if (rc == 0) rc = nmgp_dev_alloc(c, &c->b_vel, B * P);
        if (rc == 0) {
            hipError_t e = hipMemcpyAsync(c->b_mU, U, S * rank * P * sizeof(double), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(c->b_mw, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) rc = nmgp_fail(c, NMGP_E_HIP, "upload of the metric's correction failed: %s", hipGetErrorString(e));
        }
    }
    if (rc != 0) {
        const std::string msg = c->err;
        mass_reset(c);
        c->err = msg;
        return rc;
    }
    c->b_mrank = rank;
    return 0;
}

// out[b] = op(L_blk) in[b] for the B parameter-shaped vectors in [B, P] (host): trans = 0: L_blk v (whitened -> parameter
// coordinates, without the prior mean), trans = 1: L_blk^T g (a gradient -> whitened coordinates).  L_blk as above, from hyper.
extern "C" int nmgp_svc_batch_prior_apply(nmgp_ctx* c, const double hyper[8], int trans, const double* in, double* out) {
    if (!c) return NMGP_E_NULL;
    if (!hyper || !in || !out) return nmgp_fail(c, NMGP_E_NULL, "NULL argument");
    if (c->batch <= 0) return nmgp_fail(c, NMGP_E_STATE, "nmgp_svc_batch_alloc must be called first");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t P = (size_t)c->P_svc, B = c->batch;
    double *din = nullptr, *dout = nullptr;
    NMGP_TRY(nmgp_scratch_get(c, 3, B * P, &din));
    NMGP_TRY(nmgp_scratch_get(c, 4, B * P, &dout));
    HIP_TRY(c, hipMemcpyAsync(din, in, B * P * sizeof(double), hipMemcpyHostToDevice, c->stream));
    // (the factors are looked up through b_mhyp: set it for the call, put the trajectory metric's back afterwards)
    double keep[4];
    std::memcpy(keep, c->b_mhyp, sizeof keep);
    c->b_mhyp[0] = hyper[1];
    c->b_mhyp[1] = hyper[2];
    c->b_mhyp[2] = hyper[4];
    c->b_mhyp[3] = hyper[5];
    const int rc = metric_trmm(c, trans != 0, din, dout, 1.0, 0, nullptr);
    std::memcpy(c->b_mhyp, keep, sizeof keep);
    NMGP_TRY(rc);
    HIP_TRY(c, hipMemcpyAsync(out, dout, B * P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return nmgp_take_launch_error(c);
}

// The same change of coordinates for the SEPARABLE model's parameter vector [tilde_l | tilde_sigma | uL_vec | tilde_sigma2_err]
// (logpos.py:17-29): L_blk = blockdiag(chol Sigma_l, chol Sigma_sigma, c I_T, 1) with the cached factors of the GP priors of
// logpos.py:271-281 and the (float32-rounded, as the reference passes it) sd c of the Normal(0, c) prior on uL_vec (:283).
// in / out: B vectors of length 2N + T + 1 (host).  The resident subject's factors; no batch has to be allocated.
extern "C" int nmgp_sep_prior_apply(nmgp_ctx* c, const double hyper[9], int trans, int B, const double* in, double* out) {
    if (!c) return NMGP_E_NULL;
    if (!hyper || !in || !out) return nmgp_fail(c, NMGP_E_NULL, "NULL argument");
    if (!c->d_x) return nmgp_fail(c, NMGP_E_STATE, "nmgp_set_data must be called first");
    if (B <= 0) return nmgp_fail(c, NMGP_E_SHAPE, "B must be positive");
    HIP_TRY(c, hipSetDevice(c->device));
    const int N = c->N, T = c->T;
    const size_t P = (size_t)2 * N + T + 1;
    PriorFactor *pl = nullptr, *ps = nullptr;
    NMGP_TRY(nmgp_get_prior(c, hyper[1], hyper[2], &pl));
    NMGP_TRY(nmgp_get_prior(c, hyper[4], hyper[5], &ps));
    NMGP_TRY(nmgp_get_prior(c, hyper[1], hyper[2], &pl));      // (re-resolved: the second look-up may have grown the cache)
    double *din = nullptr, *dout = nullptr;
    NMGP_TRY(nmgp_scratch_get(c, 3, (size_t)B * P, &din));
    NMGP_TRY(nmgp_scratch_get(c, 4, (size_t)B * P, &dout));
    HIP_TRY(c, hipMemcpyAsync(din, in, (size_t)B * P * sizeof(double), hipMemcpyHostToDevice, c->stream));
    prior_trmm_sep(c->stream, trans != 0, pl->L, pl->ld, ps->L, ps->ld, din, dout, N, T, (long long)P, B, (double)(float)hyper[8]);
    HIP_TRY(c, hipMemcpyAsync(out, dout, (size_t)B * P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return nmgp_take_launch_error(c);
}

// The leapfrog loop shared by nmgp_svc_batch_traj / nmgp_svc_batch_traj_z: the momenta are in b_mom.  ANY API-level failure in
// the middle of a trajectory (an evaluation, a library GEMM of the dense-mass drift -- not a chain's numerical failure: those
// are flags) goes through ONE block that puts the start state back and demands a fresh value+gradient evaluation before the next
// nmgp_svc_batch_traj_begin, so that a caller who retries cannot continue from a half-advanced position.
static int traj_run(nmgp_ctx* c, const double hyper[8], int prior, double eps, int nsteps) {
    const int B = c->batch;
    const long long P = c->P_svc;
    const size_t bytes = (size_t)B * P * sizeof(double);
    hipStream_t s = c->stream;
    int* bad = c->b_hmc;
    int* bad0 = c->b_hmc + B;
    int* fl = c->b_hmc + 2 * B;
    HIP_TRY(c, hipMemcpyAsync(c->b_q0, c->b_pars, bytes, hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->b_g0, c->b_grad, bytes, hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(bad0, bad, (size_t)B * sizeof(int), hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, hipMemsetAsync(fl, 0, (size_t)B * sizeof(int), s));
    int rc = traj_kick_drift(c, 0.5 * eps, eps, 1);
    for (int step = 0; rc == 0 && step < nsteps; ++step) {
        rc = nmgp_svc_batch_eval(c, hyper, prior, 1);
        if (rc) break;
        hmc_status(s, c->b_info, c->b_scal, bad, fl, B);
        const bool last = step == nsteps - 1;
        rc = traj_kick_drift(c, last ? 0.5 * eps : eps, eps, last ? 0 : 1);
    }
    if (rc) return traj_abort(c, rc);
    return 0;
}

// failures AFTER the leapfrog loop (the end point's GEMM / reductions / copies) take the same exit as failures inside it
#define TRAJ_HIP_TRY(ctx, expr)                                                                                            \
    do {                                                                                                                   \
        hipError_t e__ = (expr);                                                                                           \
        if (e__ != hipSuccess)                                                                                             \
            return traj_abort(ctx, nmgp_fail(ctx, NMGP_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                                             __LINE__));                                                                   \
    } while (0)
#define TRAJ_BLAS_TRY(ctx, expr)                                                                                           \
    do {                                                                                                                   \
        rocblas_status s__ = (expr);                                                                                       \
        if (s__ != rocblas_status_success)                                                                                 \
            return traj_abort(ctx, nmgp_fail(ctx, NMGP_E_HIP, "%s failed: rocblas_status %d (%s:%d)", #expr, (int)s__, __FILE__, \
                                             __LINE__));                                                                   \
    } while (0)
#define TRAJ_TRY(ctx, expr)                      \
    do {                                         \
        int r__ = (expr);                        \
        if (r__ != 0) return traj_abort(ctx, r__); \
    } while (0)

// One trajectory for every chain: p0 [B, P] (host) are the momenta drawn by the caller; `nsteps` leapfrog steps of size `eps`
// (mass matrix: nmgp_svc_batch_traj_set_mass, identity by default), one batched value+gradient evaluation per step.  Returns the end point q1, p1 [B, P], the
// potential there U1 [B] and failed [B] = 1 for a chain whose potential was undefined at ANY point of the trajectory (to be
// rejected).  The device then holds the END state; nmgp_svc_batch_traj_commit puts the rejected chains back.
extern "C" int nmgp_svc_batch_traj(nmgp_ctx* c, const double hyper[8], int prior, double eps, int nsteps, const double* p0,
                                   double* q1, double* p1, double* U1, int* failed) {
    if (!c) return NMGP_E_NULL;
    if (!hyper || !p0 || !q1 || !p1 || !U1 || !failed) return nmgp_fail(c, NMGP_E_NULL, "NULL argument");
    if (!c->b_traj_ready) return nmgp_fail(c, NMGP_E_STATE, "nmgp_svc_batch_traj_begin must be called on the current state");
    if (nsteps <= 0) return nmgp_fail(c, NMGP_E_SHAPE, "nsteps must be positive");
    if (c->b_mass_kind == 3)
        return nmgp_fail(c, NMGP_E_STATE, "the prior-factor metric carries the whitened momentum L_blk^T p: use nmgp_svc_batch_traj_z "
                         "(momenta drawn on the device)");
    HIP_TRY(c, hipSetDevice(c->device));
    const int B = c->batch;
    const long long P = c->P_svc;
    const size_t bytes = (size_t)B * P * sizeof(double);
    hipStream_t s = c->stream;
    HIP_TRY(c, hipMemcpyAsync(c->b_mom, p0, bytes, hipMemcpyHostToDevice, s));
    NMGP_TRY(traj_run(c, hyper, prior, eps, nsteps));
    std::vector<double> h((size_t)B * 16);
    TRAJ_HIP_TRY(c, hipMemcpyAsync(q1, c->b_pars, bytes, hipMemcpyDeviceToHost, s));
    TRAJ_HIP_TRY(c, hipMemcpyAsync(p1, c->b_mom, bytes, hipMemcpyDeviceToHost, s));
    TRAJ_HIP_TRY(c, hipMemcpyAsync(h.data(), c->b_scal, h.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    TRAJ_HIP_TRY(c, hipMemcpyAsync(failed, c->b_hmc + 2 * B, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, s));
    TRAJ_HIP_TRY(c, hipStreamSynchronize(s));
    TRAJ_TRY(c, nmgp_take_launch_error(c));
    for (int z = 0; z < B; ++z) U1[z] = failed[z] ? INFINITY : h[(size_t)z * 16 + 8];
    c->b_traj_ready = true;
    return 0;
}

// The same trajectory with the momenta DRAWN ON THE DEVICE: z [B, P] (host) are standard normals, p0 = chol(M) z (identity: p0 = z;
// diagonal / dense: nmgp_svc_batch_traj_set_mass_chol must have uploaded chol(M) -- one GEMM [P, P] x [P, B] for all chains; prior-factor
// metric: the whitened momentum u = (I + U diag(lam) U^T)^1/2 z), so the
// start kinetic energy is 1/2 |z|^2 whatever the metric, and the END kinetic energy kin1 [B] = 1/2 p1^T M^-1 p1 comes back
// instead of the momenta themselves (dense: one more GEMM; a reduction per chain).  This is what makes a dense-mass sample cost
// what an identity-mass one does: the host's share was two [B, P] x [P, P] products per sample in NumPy (P = 14,337).
extern "C" int nmgp_svc_batch_traj_z(nmgp_ctx* c, const double hyper[8], int prior, double eps, int nsteps, const double* z,
                                     double* q1, double* kin1, double* U1, int* failed) {
    if (!c) return NMGP_E_NULL;
    if (!hyper || !z || !q1 || !kin1 || !U1 || !failed) return nmgp_fail(c, NMGP_E_NULL, "NULL argument");
    if (!c->b_traj_ready) return nmgp_fail(c, NMGP_E_STATE, "nmgp_svc_batch_traj_begin must be called on the current state");
    if (nsteps <= 0) return nmgp_fail(c, NMGP_E_SHAPE, "nsteps must be positive");
    if ((c->b_mass_kind == 1 || c->b_mass_kind == 2) && !c->b_mchol)
        return nmgp_fail(c, NMGP_E_STATE, "nmgp_svc_batch_traj_set_mass_chol must upload chol(M) before momenta can be drawn on the device");
    HIP_TRY(c, hipSetDevice(c->device));
    const int B = c->batch;
    const long long P = c->P_svc;
    const size_t bytes = (size_t)B * P * sizeof(double);
    hipStream_t s = c->stream;
    const double one = 1.0, zero = 0.0;
    const int r = c->b_mrank, mcps = metric_cps(c);
    const size_t Sr = (size_t)(B / mcps) * r;
    if (!c->b_kin) NMGP_TRY(nmgp_dev_alloc(c, &c->b_kin, (size_t)B));
    if (c->b_mass_kind == 2) {
        // p0 [P x B] = chol(M) [P x P] z [P x B]   (z staged in the velocity buffer)
        HIP_TRY(c, hipMemcpyAsync(c->b_vel, z, bytes, hipMemcpyHostToDevice, s));
        BLAS_TRY(c, rocblas_dgemm(c->blas, rocblas_operation_none, rocblas_operation_none, (int)P, B, (int)P, &one, c->b_mchol,
                                  (int)P, c->b_vel, (int)P, &zero, c->b_mom, (int)P));
    } else {
        HIP_TRY(c, hipMemcpyAsync(c->b_mom, z, bytes, hipMemcpyHostToDevice, s));
        if (c->b_mass_kind == 1) hmc_scale(s, c->b_mom, c->b_mchol, P, B);
        if (c->b_mass_kind == 3 && r > 0) {
            // u = z + U ((sqrt(1 + lam) - 1) o U^T z)  ~  N(0, I + U diag(lam) U^T)
            lowrank_proj(s, c->b_mU, c->b_mom, c->b_mc, P, r, B, mcps);
            lowrank_apply(s, c->b_mU, c->b_mw, c->b_mc, c->b_mom, c->b_mom, P, r, B, mcps);
        }
    }
    NMGP_TRY(traj_run(c, hyper, prior, eps, nsteps));
    if (c->b_mass_kind == 2) {
        TRAJ_BLAS_TRY(c, rocblas_dgemm(c->blas, rocblas_operation_none, rocblas_operation_none, (int)P, B, (int)P, &one, c->b_minv,
                                       (int)P, c->b_mom, (int)P, &zero, c->b_vel, (int)P));
        hmc_kinetic(s, c->b_mom, c->b_vel, nullptr, c->b_kin, P, B);
    } else if (c->b_mass_kind == 3) {
        lowrank_proj(s, c->b_mU, c->b_mom, c->b_mc, P, r, B, mcps);
        metric_kinetic(s, c->b_mom, c->b_mc, c->b_mw ? c->b_mw + 2 * Sr : nullptr, c->b_kin, P, r, B, mcps);
    } else {
        hmc_kinetic(s, c->b_mom, nullptr, c->b_mass_kind == 1 ? c->b_minv : nullptr, c->b_kin, P, B);
    }
    std::vector<double> h((size_t)B * 16);
    TRAJ_HIP_TRY(c, hipMemcpyAsync(q1, c->b_pars, bytes, hipMemcpyDeviceToHost, s));
    TRAJ_HIP_TRY(c, hipMemcpyAsync(kin1, c->b_kin, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, s));
    TRAJ_HIP_TRY(c, hipMemcpyAsync(h.data(), c->b_scal, h.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    TRAJ_HIP_TRY(c, hipMemcpyAsync(failed, c->b_hmc + 2 * B, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, s));
    TRAJ_HIP_TRY(c, hipStreamSynchronize(s));
    TRAJ_TRY(c, nmgp_take_launch_error(c));
    for (int k = 0; k < B; ++k) U1[k] = failed[k] ? INFINITY : h[(size_t)k * 16 + 8];
    c->b_traj_ready = true;
    return 0;
}

// accept [B]: 1 keeps a chain's end state, 0 restores the position / gradient it had before the trajectory
extern "C" int nmgp_svc_batch_traj_commit(nmgp_ctx* c, const int* accept) {
    if (!c) return NMGP_E_NULL;
    if (!accept) return nmgp_fail(c, NMGP_E_NULL, "accept must not be NULL");
    if (!c->b_traj_ready || !c->b_mom) return nmgp_fail(c, NMGP_E_STATE, "no trajectory to commit");
    HIP_TRY(c, hipSetDevice(c->device));
    const int B = c->batch;
    int* acc = c->b_hmc + 3 * B;
    HIP_TRY(c, hipMemcpyAsync(acc, accept, (size_t)B * sizeof(int), hipMemcpyHostToDevice, c->stream));
    hmc_restore(c->stream, c->b_pars, c->b_grad, c->b_q0, c->b_g0, c->b_hmc, c->b_hmc + B, acc, c->P_svc, B);
    HIP_TRY(c, hipStreamSynchronize(c->stream));        // `accept` is the caller's buffer
    return nmgp_take_launch_error(c);
}

// ---- device-resident Adam over the batch (drivers.py BatchedMAP; the MAP loop of Nonseparable_model.py:147-210 and, for all
// subjects of a rank at once, Nonseparable_model_mpisim.py:330-348) --------------------------------------------------------------
// begin: the batch's parameter vectors (nmgp_svc_batch_set_pars) are the start points; moments zero, every subject alive.
extern "C" int nmgp_svc_batch_adam_begin(nmgp_ctx* c) {
    if (!c) return NMGP_E_NULL;
    if (c->batch <= 0) return nmgp_fail(c, NMGP_E_STATE, "nmgp_svc_batch_alloc must be called first");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t B = c->batch, P = (size_t)c->P_svc;
    if (!c->b_am) {
        NMGP_TRY(nmgp_dev_alloc(c, &c->b_am, B * P));
        NMGP_TRY(nmgp_dev_alloc(c, &c->b_av, B * P));
        HIP_TRY(c, hipMalloc((void**)&c->b_alive, B * sizeof(int)));
    }
    HIP_TRY(c, hipMemsetAsync(c->b_am, 0, B * P * sizeof(double), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->b_av, 0, B * P * sizeof(double), c->stream));
    std::vector<int> ones(B, 1);
    HIP_TRY(c, hipMemcpyAsync(c->b_alive, ones.data(), B * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->b_adam_t = 0;
    return 0;
}

// One iteration for every subject: batched value+gradient at the resident parameters, then Adam's update on the device.
// out [B,5]: the verbose tuples AT the parameters the iteration started from (what the reference logs as target_value_hist);
// alive [B]: 0 from the first failed evaluation of a subject on (its parameters are frozen there).
extern "C" int nmgp_svc_batch_adam_step(nmgp_ctx* c, const double hyper[8], int prior, double lr, double beta1, double beta2,
                                        double eps, double* out, int* alive) {
    if (!c) return NMGP_E_NULL;
    if (!hyper || !out || !alive) return nmgp_fail(c, NMGP_E_NULL, "NULL argument");
    if (c->b_adam_t < 0) return nmgp_fail(c, NMGP_E_STATE, "nmgp_svc_batch_adam_begin must be called first");
    const int B = c->batch;
    NMGP_TRY(nmgp_svc_batch_eval(c, hyper, prior, 1));
    const long long t = ++c->b_adam_t;
    const double bc1 = 1.0 - std::pow(beta1, (double)t);
    const double bc2s = std::sqrt(1.0 - std::pow(beta2, (double)t));
    adam_step(c->stream, c->b_pars, c->b_grad, c->b_am, c->b_av, c->b_alive, c->b_info, c->b_scal, beta1, beta2, bc2s, eps,
              lr / bc1, c->P_svc, B);
    std::vector<double> h((size_t)B * 16);
    HIP_TRY(c, hipMemcpyAsync(h.data(), c->b_scal, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(alive, c->b_alive, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    NMGP_TRY(nmgp_take_launch_error(c));
    for (int z = 0; z < B; ++z)
        for (int k = 0; k < 5; ++k) out[(size_t)z * 5 + k] = alive[z] ? h[(size_t)z * 16 + 8 + k] : std::nan("");
    return 0;
}

// the batch's parameter vectors [B, P] as they stand in HBM
extern "C" int nmgp_svc_batch_get_pars(nmgp_ctx* c, double* pars) {
    if (!c) return NMGP_E_NULL;
    if (!pars) return nmgp_fail(c, NMGP_E_NULL, "pars must not be NULL");
    if (c->batch <= 0) return nmgp_fail(c, NMGP_E_STATE, "nmgp_svc_batch_alloc must be called first");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(pars, c->b_pars, (size_t)c->batch * c->P_svc * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int nmgp_svc_covariance(nmgp_ctx* c, const double* pars, double* out) {
    if (!c) return NMGP_E_NULL;
    if (!out) return nmgp_fail(c, NMGP_E_NULL, "out must not be NULL");
    NMGP_TRY(nmgp_svc_set_pars(c, pars));
    NMGP_TRY(nmgp_ensure_S(c));
    hipStream_t s = c->stream;
    svc_prep(s, c->d_pars, c->N, c->M, c->d_ell, c->d_Lv);
    int r = svc_cov_build(s, c->d_x, c->d_ell, c->d_Lv, c->d_pars + (c->P_svc - 1), c->d_S, c->ldS, c->N, c->M, true);
    if (r) return nmgp_fail(c, r, "unsupported number of outputs M=%d", c->M);
    HIP_TRY(c, hipMemcpy2DAsync(out, (size_t)c->n * sizeof(double), c->d_S, (size_t)c->ldS * sizeof(double),
                                (size_t)c->n * sizeof(double), (size_t)c->n, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return 0;
}

// Cholesky primitive (torch.cholesky analogue; prediction.py:974 uses it) and the developer entry for the custom
// factorisation: factors the symmetric positive definite A (lower triangle of the row-major == column-major
// symmetric matrix is read), optionally carrying a right-hand side.
extern "C" int nmgp_cholesky(nmgp_ctx* c, const double* A, int n, const double* rhs, double* out_L, double* out_z,
                             int algo) {
    if (!c) return NMGP_E_NULL;
    if (!A || !out_L) return nmgp_fail(c, NMGP_E_NULL, "A/out_L must not be NULL");
    if (n <= 0) return nmgp_fail(c, NMGP_E_SHAPE, "n must be positive");
    if (rhs && !out_z) return nmgp_fail(c, NMGP_E_NULL, "out_z must not be NULL when rhs is given");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const size_t ld = (((size_t)n + 1 + 15) / 16) * 16;
    double *dA, *dv;
    NMGP_TRY(nmgp_scratch_get(c, 2, ld * n, &dA));
    NMGP_TRY(nmgp_scratch_get(c, 3, (size_t)2 * n + 8, &dv));
    HIP_TRY(c, hipMemsetAsync(dA, 0, ld * n * sizeof(double), s));
    HIP_TRY(c, hipMemcpy2DAsync(dA, ld * sizeof(double), A, (size_t)n * sizeof(double), (size_t)n * sizeof(double),
                                (size_t)n, hipMemcpyHostToDevice, s));
    if (rhs) HIP_TRY(c, hipMemcpyAsync(dv, rhs, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemsetAsync(c->d_info + 5, 0, sizeof(int), s));
    if (algo == 1) {
        StageScope sp(c, NMGP_STAGE_CHOL);
        if (rhs) set_row(s, dA, (int)ld, n, dv, n, 1, 0, 0);
        potrf_lower(s, c->stream2, nmgp_chol_events(c, n), dA, (int)ld, n, rhs ? 1 : 0, 0, c->chol_nb1, c->d_info + 5, 1, 0, 0, nmgp_syrk_hook(c));
        if (rhs) get_row(s, dA, (int)ld, n, dv + n, n, 1, 0, 0);
    } else {
        StageScope sp(c, NMGP_STAGE_CHOL);
        BLAS_TRY(c, rocsolver_dpotrf(c->blas, rocblas_fill_lower, n, dA, (int)ld, c->d_info + 5));
        if (rhs) {
            HIP_TRY(c, hipMemcpyAsync(dv + n, dv, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
            BLAS_TRY(c, rocblas_dtrsv(c->blas, rocblas_fill_lower, rocblas_operation_none, rocblas_diagonal_non_unit, n,
                                      dA, (int)ld, dv + n, 1));
        }
    }
    HIP_TRY(c, hipMemcpy2DAsync(out_L, (size_t)n * sizeof(double), dA, ld * sizeof(double), (size_t)n * sizeof(double),
                                (size_t)n, hipMemcpyDeviceToHost, s));
    if (rhs) HIP_TRY(c, hipMemcpyAsync(out_z, dv + n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(c->h_info + 5, c->d_info + 5, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    NMGP_TRY(nmgp_take_launch_error(c));
    if (c->h_info[5] != 0)
        return nmgp_fail(c, c->h_info[5], "matrix is not positive definite (leading minor %d)", c->h_info[5]);
    return 0;
}

// ---- primitives --------------------------------------------------------------------------------
static int upload(nmgp_ctx* c, int slot, const double* h, size_t nelem, double** d) {
    NMGP_TRY(nmgp_scratch_get(c, slot, nelem, d));
    HIP_TRY(c, hipMemcpyAsync(*d, h, nelem * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return 0;
}

static int download(nmgp_ctx* c, double* h, const double* d, size_t nelem) {
    HIP_TRY(c, hipMemcpyAsync(h, d, nelem * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return nmgp_take_launch_error(c);
}

extern "C" int nmgp_pairwise_distances(nmgp_ctx* c, const double* x1, int n1, const double* x2, int n2, int d,
                                       double* out) {
    if (!c) return NMGP_E_NULL;
    if (!x1 || !out) return nmgp_fail(c, NMGP_E_NULL, "x1/out must not be NULL");
    if (!x2) n2 = n1;
    if (n1 <= 0 || n2 <= 0 || d <= 0) return nmgp_fail(c, NMGP_E_SHAPE, "bad shape n1=%d n2=%d d=%d", n1, n2, d);
    HIP_TRY(c, hipSetDevice(c->device));
    double *dx1, *dx2, *dout;
    NMGP_TRY(upload(c, 0, x1, (size_t)n1 * d, &dx1));
    if (x2) NMGP_TRY(upload(c, 1, x2, (size_t)n2 * d, &dx2)); else dx2 = dx1;
    NMGP_TRY(nmgp_scratch_get(c, 5, (size_t)n1 * n2, &dout));
    pairwise_rect(c->stream, dx1, n1, dx2, n2, d, dout);
    return download(c, out, dout, (size_t)n1 * n2);
}

extern "C" int nmgp_rbf_cov(nmgp_ctx* c, const double* x1, int n1, const double* x2, int n2, int d, double alpha,
                            double beta, double* out) {
    if (!c) return NMGP_E_NULL;
    if (!x1 || !out) return nmgp_fail(c, NMGP_E_NULL, "x1/out must not be NULL");
    if (!x2) n2 = n1;
    if (n1 <= 0 || n2 <= 0 || d <= 0) return nmgp_fail(c, NMGP_E_SHAPE, "bad shape n1=%d n2=%d d=%d", n1, n2, d);
    HIP_TRY(c, hipSetDevice(c->device));
    double *dx1, *dx2, *dout;
    NMGP_TRY(upload(c, 0, x1, (size_t)n1 * d, &dx1));
    if (x2) NMGP_TRY(upload(c, 1, x2, (size_t)n2 * d, &dx2)); else dx2 = dx1;
    NMGP_TRY(nmgp_scratch_get(c, 5, (size_t)n1 * n2, &dout));
    rbf_cov_rect(c->stream, dx1, n1, dx2, n2, d, alpha, beta, x2 == nullptr, dout);
    return download(c, out, dout, (size_t)n1 * n2);
}

extern "C" int nmgp_nonstat_rbf_cov(nmgp_ctx* c, const double* x1, const double* s1, const double* l1, int n1,
                                    const double* x2, const double* s2, const double* l2, int n2, int d, double* out) {
    if (!c) return NMGP_E_NULL;
    if (!x1 || !out) return nmgp_fail(c, NMGP_E_NULL, "x1/out must not be NULL");
    if (!x2) n2 = n1;
    if (n1 <= 0 || n2 <= 0 || d <= 0) return nmgp_fail(c, NMGP_E_SHAPE, "bad shape n1=%d n2=%d d=%d", n1, n2, d);
    HIP_TRY(c, hipSetDevice(c->device));
    double *dx1, *dx2, *ds1 = nullptr, *dl1 = nullptr, *ds2 = nullptr, *dl2 = nullptr, *dout;
    // one upload buffer per operand: slots 0..4 hold [x1 | s1 | l1], [x2 | s2 | l2]
    double* b1;
    NMGP_TRY(nmgp_scratch_get(c, 0, (size_t)n1 * (d + 2), &b1));
    dx1 = b1;
    HIP_TRY(c, hipMemcpyAsync(dx1, x1, (size_t)n1 * d * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (s1) {
        ds1 = b1 + (size_t)n1 * d;
        HIP_TRY(c, hipMemcpyAsync(ds1, s1, (size_t)n1 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    if (l1) {
        dl1 = b1 + (size_t)n1 * (d + 1);
        HIP_TRY(c, hipMemcpyAsync(dl1, l1, (size_t)n1 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    if (x2) {
        double* b2;
        NMGP_TRY(nmgp_scratch_get(c, 1, (size_t)n2 * (d + 2), &b2));
        dx2 = b2;
        HIP_TRY(c, hipMemcpyAsync(dx2, x2, (size_t)n2 * d * sizeof(double), hipMemcpyHostToDevice, c->stream));
        if (s2) {
            ds2 = b2 + (size_t)n2 * d;
            HIP_TRY(c, hipMemcpyAsync(ds2, s2, (size_t)n2 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        }
        if (l2) {
            dl2 = b2 + (size_t)n2 * (d + 1);
            HIP_TRY(c, hipMemcpyAsync(dl2, l2, (size_t)n2 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        }
    } else {
        dx2 = dx1;
        ds2 = ds1;
        dl2 = dl1;
    }
    NMGP_TRY(nmgp_scratch_get(c, 5, (size_t)n1 * n2, &dout));
    gibbs_cov_rect(c->stream, dx1, ds1, dl1, n1, dx2, ds2, dl2, n2, d, x2 == nullptr, dout);
    return download(c, out, dout, (size_t)n1 * n2);
}

extern "C" int nmgp_kron_product(nmgp_ctx* c, const double* a, int ar, int ac, const double* b, int br, int bc,
                                 double* out) {
    if (!c) return NMGP_E_NULL;
    if (!a || !b || !out) return nmgp_fail(c, NMGP_E_NULL, "a/b/out must not be NULL");
    if (ar <= 0 || ac <= 0 || br <= 0 || bc <= 0) return nmgp_fail(c, NMGP_E_SHAPE, "bad shape");
    HIP_TRY(c, hipSetDevice(c->device));
    double *da, *db, *dout;
    NMGP_TRY(upload(c, 0, a, (size_t)ar * ac, &da));
    NMGP_TRY(upload(c, 1, b, (size_t)br * bc, &db));
    const size_t tot = (size_t)ar * ac * br * bc;
    NMGP_TRY(nmgp_scratch_get(c, 5, tot, &dout));
    kron_product(c->stream, da, ar, ac, db, br, bc, dout);
    return download(c, out, dout, tot);
}

// ---- profiling / micro-benchmarks ----------------------------------------------------------------
extern "C" int nmgp_profile_enable(nmgp_ctx* c, int on) {
    if (!c) return NMGP_E_NULL;
    c->profiling = on < 0 ? 0 : on;
    return 0;
}

extern "C" int nmgp_profile_reset(nmgp_ctx* c) {
    if (!c) return NMGP_E_NULL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    profile_collect(c);
    for (auto& t : c->timers) {
        t.ms = 0;
        t.count = 0;
        t.work = 0;
        t.bytes = 0;
    }
    return 0;
}

extern "C" int nmgp_profile_read_work(nmgp_ctx* c, double ms[NMGP_STAGE_COUNT], long long count[NMGP_STAGE_COUNT],
                                      double work[NMGP_STAGE_COUNT], double bytes[NMGP_STAGE_COUNT]) {
    if (!c) return NMGP_E_NULL;
    int r = nmgp_profile_read(c, ms, count);
    if (r) return r;
    if (work)
        for (int s = 0; s < NMGP_STAGE_COUNT; ++s) work[s] = c->timers[s].work;
    if (bytes)
        for (int s = 0; s < NMGP_STAGE_COUNT; ++s) bytes[s] = c->timers[s].bytes;
    return 0;
}

extern "C" int nmgp_profile_read(nmgp_ctx* c, double ms[NMGP_STAGE_COUNT], long long count[NMGP_STAGE_COUNT]) {
    if (!c) return NMGP_E_NULL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    profile_collect(c);
    for (int s = 0; s < NMGP_STAGE_COUNT; ++s) {
        if (ms) ms[s] = c->timers[s].ms;
        if (count) count[s] = c->timers[s].count;
    }
    return 0;
}

// HBM ceilings of this chip as the library's own streaming kernels see them (flat 16-byte-per-lane kernels over buffers far
// beyond the 256 MiB Infinity Cache): gbs3 = {copy (read + write bytes), read only, write only} in GB/s.
extern "C" int nmgp_measure_hbm_rates(nmgp_ctx* c, long long bytes, int reps, double gbs3[3]) {
    if (!c) return NMGP_E_NULL;
    if (!gbs3) return nmgp_fail(c, NMGP_E_NULL, "gbs3 must not be NULL");
    if (bytes < 1024 || reps <= 0) return nmgp_fail(c, NMGP_E_SHAPE, "bytes/reps too small");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t nelem = ((size_t)bytes / 16) * 2;
    double *src, *dst;
    NMGP_TRY(nmgp_scratch_get(c, 3, nelem, &src));
    NMGP_TRY(nmgp_scratch_get(c, 4, nelem + 512, &dst));
    double* sink = dst + nelem;
    HIP_TRY(c, hipMemsetAsync(src, 0, nelem * sizeof(double), c->stream));
    hipEvent_t e0, e1;
    HIP_TRY(c, hipEventCreate(&e0));
    HIP_TRY(c, hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        stream_copy(c->stream, src, dst, nelem, mode, sink);
        HIP_TRY(c, hipEventRecord(e0, c->stream));
        for (int r = 0; r < reps; ++r) stream_copy(c->stream, src, dst, nelem, mode, sink);
        HIP_TRY(c, hipEventRecord(e1, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, e0, e1));
        gbs3[mode] = (mode == 0 ? 2.0 : 1.0) * (double)nelem * 8.0 * reps / (ms * 1e-3) / 1e9;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return nmgp_take_launch_error(c);
}

extern "C" int nmgp_measure_hbm_gbs(nmgp_ctx* c, long long bytes, int reps, double* gbs) {
    if (!gbs) return c ? nmgp_fail(c, NMGP_E_NULL, "gbs must not be NULL") : NMGP_E_NULL;
    double g3[3];
    NMGP_TRY(nmgp_measure_hbm_rates(c, bytes, reps, g3));
    *gbs = g3[0];
    return 0;
}

extern "C" int nmgp_measure_dgemm_tflops(nmgp_ctx* c, int n, int reps, double* tflops) {
    if (!c) return NMGP_E_NULL;
    if (!tflops) return nmgp_fail(c, NMGP_E_NULL, "tflops must not be NULL");
    if (n <= 0 || reps <= 0) return nmgp_fail(c, NMGP_E_SHAPE, "n/reps must be positive");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t nn = (size_t)n * n;
    double *A, *B, *C;
    NMGP_TRY(nmgp_scratch_get(c, 2, nn, &A));
    NMGP_TRY(nmgp_scratch_get(c, 3, nn, &B));
    NMGP_TRY(nmgp_scratch_get(c, 4, nn, &C));
    // pseudo-random, non-trivial operands (zero-filled operands over-state the sustainable clock)
    std::vector<double> h(nn);
    unsigned long long st = 88172645463325252ull;
    for (size_t k = 0; k < nn; ++k) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        h[k] = ((double)(st >> 11) / 9007199254740992.0) * 2.0 - 1.0;
    }
    HIP_TRY(c, hipMemcpy(A, h.data(), nn * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(B, h.data(), nn * sizeof(double), hipMemcpyHostToDevice));
    const double one = 1.0, zero = 0.0;
    hipEvent_t e0, e1;
    HIP_TRY(c, hipEventCreate(&e0));
    HIP_TRY(c, hipEventCreate(&e1));
    BLAS_TRY(c, rocblas_dgemm(c->blas, rocblas_operation_none, rocblas_operation_transpose, n, n, n, &one, A, n, B, n,
                              &zero, C, n));
    HIP_TRY(c, hipEventRecord(e0, c->stream));
    for (int r = 0; r < reps; ++r)
        BLAS_TRY(c, rocblas_dgemm(c->blas, rocblas_operation_none, rocblas_operation_transpose, n, n, n, &one, A, n, B,
                                  n, &zero, C, n));
    HIP_TRY(c, hipEventRecord(e1, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *tflops = 2.0 * (double)n * n * n * reps / (ms * 1e-3) / 1e12;
    return 0;
}
