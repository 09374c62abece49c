// Hand-written gfx950 kernels of the log-posterior path: covariance builds, Kronecker expansion,
// reductions.  All arithmetic is IEEE float64; the build uses -ffp-contract=off so that the element
// formulas round exactly like the reference's unfused torch-CPU elementwise ops (kernels.py:20,68-72).
//
// Matrices handed to rocSOLVER/rocBLAS are column-major with the LOWER triangle stored; because every
// such matrix is symmetric this is the same memory image as the reference's row-major upper triangle.
#include "nmgp_internal.h"
#include <mutex>

namespace nmgpk {

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------
// wave / block reductions (64-lane wavefronts)
// ---------------------------------------------------------------------------------------------
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// block of up to 1024 threads (16 waves); result valid in thread 0
__device__ inline double block_sum(double v, double* sh /*[16]*/) {
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    int nw = (blockDim.x + 63) >> 6;
    v = (threadIdx.x < nw) ? sh[threadIdx.x] : 0.0;
    if (w == 0) v = wave_sum(v);
    return v;
}

// ---------------------------------------------------------------------------------------------
// parameter unpacking (utils.py:10-22,38-46: exp on the tril-diagonal slots; logpos.py:342: l = exp(tilde_l))
// ---------------------------------------------------------------------------------------------
__global__ void k_svc_prep(const double* __restrict__ pars, int N, int M, int T, double* __restrict__ ell,
                           double* __restrict__ Lv) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    // blockIdx.y = chain of the batch: parameter vectors are stacked [B, P], the unpacked curves [B, N] / [B, N, T]
    pars += (size_t)blockIdx.y * ((size_t)N * (1 + T) + 1);
    ell += (size_t)blockIdx.y * N;
    Lv += (size_t)blockIdx.y * N * T;
    ell[i] = exp(pars[i]);
    const double* u = pars + N + (size_t)i * T;
    double* o = Lv + (size_t)i * T;
    int t = 0;
    for (int r = 0; r < M; ++r)
        for (int c = 0; c <= r; ++c, ++t) o[t] = (c == r) ? exp(u[t]) : u[t];
}

void svc_prep(hipStream_t s, const double* pars, int N, int M, double* ell, double* Lv, int batch) {
    int T = M * (M + 1) / 2;
    NMGP_LAUNCH(k_svc_prep, dim3(cdiv(N, 256), batch), dim3(256), 0, s, pars, N, M, T, ell, Lv);
}

// ---------------------------------------------------------------------------------------------
// kernel #1: fused nonseparable covariance
//   S[m N + i, m' N + j] = (K0(i,j) + jitter d_ij) * sum_r L_i[m,r] L_j[m',r]  + sigma2 d_(mi)(m'j)
// replaces kernels.Nonstationary_RBF_cov (kernels.py:46-73), generate_K_index_SVC (logpos.py:111-118),
// the n-major -> m-major gathers (logpos.py:347-348), kronecker_product(ones, K_x) * K_i (logpos.py:349)
// and "+ sigma2_err * eye" (logpos.py:352).  One 64 x 64 tile of locations per workgroup; the j-side
// coordinates / length-scales / factors are staged in LDS and broadcast, the i-side lives in registers;
// lanes run along i so that every wave store is 512 contiguous bytes of one column.
// Only the lower triangle is written unless FULL.
// ---------------------------------------------------------------------------------------------
template <int M, bool FULL>
__global__ __launch_bounds__(256) void k_svc_cov(const double* __restrict__ x, const double* __restrict__ ell,
                                                  const double* __restrict__ Lv, const double* __restrict__ tse,
                                                  double* __restrict__ S, int ld, int N, long long sstride,
                                                  int xstride, int cps) {
    constexpr int T = M * (M + 1) / 2;
    constexpr int TJ = 64;
    __shared__ double sx[TJ], sl[TJ], sL[TJ * T];
    const int I = blockIdx.x, J = blockIdx.y;
    if (!FULL && M == 1 && I < J) return;
    // blockIdx.z = chain of the batch: one covariance buffer per chain; xstride = 0 when all chains belong to one
    // subject (shared x), N when the batch holds several subjects: batch element z belongs to subject z / cps (cps chains per
    // subject, consecutive)
    x += (size_t)(blockIdx.z / cps) * xstride;
    ell += (size_t)blockIdx.z * N;
    Lv += (size_t)blockIdx.z * N * T;
    tse += (size_t)blockIdx.z * ((size_t)N * (1 + T) + 1);
    S += (size_t)blockIdx.z * sstride;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j0 = J * TJ;
    if (tid < TJ) {
        int j = j0 + tid;
        sx[tid] = (j < N) ? x[j] : 0.0;
        sl[tid] = (j < N) ? ell[j] : 1.0;
    }
    for (int k = tid; k < TJ * T; k += 256) {
        size_t g = (size_t)j0 * T + k;
        sL[k] = (g < (size_t)N * T) ? Lv[g] : 0.0;
    }
    __syncthreads();
    const int i = I * 64 + lane;
    if (i >= N) return;
    const double sigma2 = exp(tse[0]);
    const double xi = x[i], li = ell[i];
    const double xi2 = xi * xi, li2 = li * li;
    double Li[T];
#pragma unroll
    for (int t = 0; t < T; ++t) Li[t] = Lv[(size_t)i * T + t];
    const size_t Ns = (size_t)N;
#pragma unroll 2
    for (int jj = 0; jj < TJ / 4; ++jj) {
        const int k = w * (TJ / 4) + jj;
        const int j = j0 + k;
        if (j >= N) break;
        const double xj = sx[k], lj = sl[k];
        const double dist = (xi2 + xj * xj) - 2.0 * (xi * xj);   // kernels.py:20
        const double A = li2 + lj * lj;                          // kernels.py:69
        double kv = sqrt(2.0 * (li * lj) / A) * exp(-dist / A);  // kernels.py:70,72 (sigma == 1)
        if (i == j) kv = NMGP_JITTER + kv;                       // kernels.py:64
#pragma unroll
        for (int m = 0; m < M; ++m) {
#pragma unroll
            for (int mp = 0; mp < M; ++mp) {
                if (!FULL) {
                    if (mp > m) continue;
                    if (mp == m && i < j) continue;
                }
                double b = 0.0;
                const int rmax = (m < mp) ? m : mp;
#pragma unroll
                for (int r = 0; r <= rmax; ++r) b += Li[m * (m + 1) / 2 + r] * sL[k * T + mp * (mp + 1) / 2 + r];
                double v = kv * b;
                if (m == mp && i == j) v += sigma2;
                S[((size_t)mp * Ns + j) * ld + ((size_t)m * Ns + i)] = v;
            }
        }
    }
}

template <int M>
static void launch_svc_cov(hipStream_t s, const double* x, const double* ell, const double* Lv, const double* tse,
                           double* S, int ld, int N, bool full, int batch, long long sstride, int xstride, int cps) {
    dim3 grid(cdiv(N, 64), cdiv(N, 64), batch);
    if (full)
        NMGP_LAUNCH((k_svc_cov<M, true>), grid, dim3(256), 0, s, x, ell, Lv, tse, S, ld, N, sstride, xstride, cps);
    else
        NMGP_LAUNCH((k_svc_cov<M, false>), grid, dim3(256), 0, s, x, ell, Lv, tse, S, ld, N, sstride, xstride, cps);
}

int svc_cov_build(hipStream_t s, const double* x, const double* ell, const double* Lv, const double* tse, double* S,
                  int ld, int N, int M, bool full, int batch, long long sstride, int xstride, int cps) {
    if (cps < 1) cps = 1;
    switch (M) {
        case 1: launch_svc_cov<1>(s, x, ell, Lv, tse, S, ld, N, full, batch, sstride, xstride, cps); break;
        case 2: launch_svc_cov<2>(s, x, ell, Lv, tse, S, ld, N, full, batch, sstride, xstride, cps); break;
        case 3: launch_svc_cov<3>(s, x, ell, Lv, tse, S, ld, N, full, batch, sstride, xstride, cps); break;
        case 4: launch_svc_cov<4>(s, x, ell, Lv, tse, S, ld, N, full, batch, sstride, xstride, cps); break;
        case 5: launch_svc_cov<5>(s, x, ell, Lv, tse, S, ld, N, full, batch, sstride, xstride, cps); break;
        case 6: launch_svc_cov<6>(s, x, ell, Lv, tse, S, ld, N, full, batch, sstride, xstride, cps); break;
        case 7: launch_svc_cov<7>(s, x, ell, Lv, tse, S, ld, N, full, batch, sstride, xstride, cps); break;
        case 8: launch_svc_cov<8>(s, x, ell, Lv, tse, S, ld, N, full, batch, sstride, xstride, cps); break;
        default: return NMGP_E_UNSUPPORTED;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// symmetric N x N kernels on 1-D inputs (GP priors, separable K_x); column-major == row-major (symmetric)
// ---------------------------------------------------------------------------------------------
template <bool GIBBS, bool FULL>
__global__ __launch_bounds__(256) void k_cov_sym(const double* __restrict__ x, const double* __restrict__ sig,
                                                  const double* __restrict__ ell, int N, double alpha, double beta,
                                                  double* __restrict__ out, int ld) {
    constexpr int TJ = 64;
    __shared__ double sx[TJ], sl[TJ], ss[TJ];
    const int I = blockIdx.x, J = blockIdx.y;
    if (!FULL && I < J) return;
    x += (size_t)blockIdx.z * N;                      // blockIdx.z = subject of a multi-subject batch
    out += (size_t)blockIdx.z * ld * N;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j0 = J * TJ;
    if (tid < TJ) {
        int j = j0 + tid;
        double xv = (j < N) ? x[j] : 0.0;
        sx[tid] = GIBBS ? xv : xv / beta;                      // kernels.py:38-39 (inputs scaled by 1/beta)
        sl[tid] = (GIBBS && j < N) ? ell[j] : 1.0;
        ss[tid] = (GIBBS && sig != nullptr && j < N) ? sig[j] : 1.0;
    }
    __syncthreads();
    const int i = I * 64 + lane;
    if (i >= N) return;
    const double xi = GIBBS ? x[i] : x[i] / beta;
    const double li = GIBBS ? ell[i] : 1.0;
    const double si = (GIBBS && sig != nullptr) ? sig[i] : 1.0;
    const double xi2 = xi * xi, li2 = li * li;
    const double a2 = alpha * alpha;
    for (int jj = 0; jj < TJ / 4; ++jj) {
        const int k = w * (TJ / 4) + jj;
        const int j = j0 + k;
        if (j >= N) break;
        if (!FULL && i < j) continue;
        const double xj = sx[k];
        const double dist = (xi2 + xj * xj) - 2.0 * (xi * xj);
        double v;
        if (GIBBS) {
            const double lj = sl[k];
            const double A = li2 + lj * lj;
            v = (si * ss[k]) * sqrt(2.0 * (li * lj) / A) * exp(-dist / A);   // kernels.py:69-72
        } else {
            v = exp(-0.5 * dist) * a2;                                       // kernels.py:42
        }
        if (i == j) v = NMGP_JITTER + v;
        out[(size_t)j * ld + i] = v;
    }
}

void rbf_cov_sym(hipStream_t s, const double* x, int N, double alpha, double beta, double* out, int ld, bool full,
                 int batch) {
    dim3 grid(cdiv(N, 64), cdiv(N, 64), batch);
    if (full)
        NMGP_LAUNCH((k_cov_sym<false, true>), grid, dim3(256), 0, s, x, nullptr, nullptr, N, alpha, beta, out, ld);
    else
        NMGP_LAUNCH((k_cov_sym<false, false>), grid, dim3(256), 0, s, x, nullptr, nullptr, N, alpha, beta, out, ld);
}

void gibbs_cov_sym(hipStream_t s, const double* x, const double* sig, const double* ell, int N, double* out, int ld,
                   bool full) {
    dim3 grid(cdiv(N, 64), cdiv(N, 64));
    if (full)
        NMGP_LAUNCH((k_cov_sym<true, true>), grid, dim3(256), 0, s, x, sig, ell, N, 1.0, 1.0, out, ld);
    else
        NMGP_LAUNCH((k_cov_sym<true, false>), grid, dim3(256), 0, s, x, sig, ell, N, 1.0, 1.0, out, ld);
}

// ---------------------------------------------------------------------------------------------
// rectangular d-dimensional primitives (row-major [n1, n2] output; lanes run along n2)
//   MODE 0: pairwise_distances (kernels.py:5-21)   1: RBF_cov (24-43)   2: Nonstationary_RBF_cov (46-73)
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_rect(const double* __restrict__ x1, const double* __restrict__ s1,
                                               const double* __restrict__ l1, int n1, const double* __restrict__ x2,
                                               const double* __restrict__ s2, const double* __restrict__ l2, int n2,
                                               int d, double alpha, double beta, int sym, double* __restrict__ out) {
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= n1 || j >= n2) return;
    double xn = 0.0, yn = 0.0, dot = 0.0;
    for (int k = 0; k < d; ++k) {
        double a = x1[(size_t)i * d + k], b = x2[(size_t)j * d + k];
        if (MODE == 1) { a = a / beta; b = b / beta; }
        xn += a * a;
        yn += b * b;
        dot += a * b;
    }
    const double dist = (xn + yn) - 2.0 * dot;
    double v;
    if (MODE == 0) {
        v = dist;
    } else if (MODE == 1) {
        v = exp(-0.5 * dist) * (alpha * alpha);
        if (sym && i == j) v = NMGP_JITTER + v;
    } else {
        const double li = l1 ? l1[i] : 1.0, lj = l2 ? l2[j] : 1.0;
        const double si = s1 ? s1[i] : 1.0, sj = s2 ? s2[j] : 1.0;
        const double A = li * li + lj * lj;
        v = (si * sj) * sqrt(2.0 * (li * lj) / A) * exp(-dist / A);
        if (sym && i == j) v = NMGP_JITTER + v;
    }
    out[(size_t)i * n2 + j] = v;
}

void pairwise_rect(hipStream_t s, const double* x1, int n1, const double* x2, int n2, int d, double* out) {
    NMGP_LAUNCH((k_rect<0>), dim3(cdiv(n2, 64), cdiv(n1, 4)), dim3(256), 0, s, x1, nullptr, nullptr, n1, x2,
                       nullptr, nullptr, n2, d, 1.0, 1.0, 0, out);
}
void rbf_cov_rect(hipStream_t s, const double* x1, int n1, const double* x2, int n2, int d, double alpha, double beta,
                  bool sym, double* out) {
    NMGP_LAUNCH((k_rect<1>), dim3(cdiv(n2, 64), cdiv(n1, 4)), dim3(256), 0, s, x1, nullptr, nullptr, n1, x2,
                       nullptr, nullptr, n2, d, alpha, beta, sym ? 1 : 0, out);
}
void gibbs_cov_rect(hipStream_t s, const double* x1, const double* s1, const double* l1, int n1, const double* x2,
                    const double* s2, const double* l2, int n2, int d, bool sym, double* out) {
    NMGP_LAUNCH((k_rect<2>), dim3(cdiv(n2, 64), cdiv(n1, 4)), dim3(256), 0, s, x1, s1, l1, n1, x2, s2, l2, n2,
                       d, 1.0, 1.0, sym ? 1 : 0, out);
}

// kronecker_operation.kronecker_product (kronecker_operation.py:5-22): out[(p*br+r), (q*bc+c)] = a[p,q] b[r,c]
__global__ __launch_bounds__(256) void k_kron(const double* __restrict__ a, int ar, int ac,
                                               const double* __restrict__ b, int br, int bc,
                                               double* __restrict__ out) {
    const size_t W = (size_t)ac * bc, H = (size_t)ar * br;
    const size_t col = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= W) return;
    const int q = (int)(col / bc), c = (int)(col % bc);
    for (size_t row = blockIdx.y; row < H; row += gridDim.y) {
        const int p = (int)(row / br), r = (int)(row % br);
        out[row * W + col] = a[(size_t)p * ac + q] * b[(size_t)r * bc + c];
    }
}

void kron_product(hipStream_t s, const double* a, int ar, int ac, const double* b, int br, int bc, double* out) {
    size_t W = (size_t)ac * bc, H = (size_t)ar * br;
    int gy = (int)(H < 4096 ? H : 4096);
    NMGP_LAUNCH(k_kron, dim3(cdiv(W, 256), gy), dim3(256), 0, s, a, ar, ac, b, br, bc, out);
}

// ---------------------------------------------------------------------------------------------
// kernel #2: reductions after the factorisation: log det = 2 sum log L_rr, quad = ||L^-1 y||^2
// (replaces torch.logdet, logpos.py:353, and the mv + dot of distributions.py:22)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_logdet_quad(const double* __restrict__ L, int ld, int n,
                                                       const double* __restrict__ z, double* __restrict__ out_logdet,
                                                       double* __restrict__ out_quad, long long bstride,
                                                       int ostride) {
    __shared__ double sh[16];
    // blockIdx.x = matrix of the batch
    L += (size_t)blockIdx.x * bstride;
    if (z) z += (size_t)blockIdx.x * n;
    out_logdet += (size_t)blockIdx.x * ostride;
    if (out_quad) out_quad += (size_t)blockIdx.x * ostride;
    double a = 0.0, q = 0.0;
    for (int r = threadIdx.x; r < n; r += blockDim.x) {
        a += log(L[(size_t)r * ld + r]);
        if (z) {
            double t = z[r];
            q += t * t;
        }
    }
    a = block_sum(a, sh);
    q = block_sum(q, sh);
    if (threadIdx.x == 0) {
        out_logdet[0] = 2.0 * a;
        if (out_quad) out_quad[0] = q;
    }
}

void chol_logdet_quad(hipStream_t s, const double* L, int ld, int n, const double* z, double* out_logdet,
                      double* out_quad, int batch, long long bstride, int ostride) {
    NMGP_LAUNCH(k_logdet_quad, dim3(batch), dim3(1024), 0, s, L, ld, n, z, out_logdet, out_quad, bstride,
                       ostride);
}

void diag_logsum2(hipStream_t s, const double* L, int ld, int n, double* out) {
    NMGP_LAUNCH(k_logdet_quad, dim3(1), dim3(1024), 0, s, L, ld, n, (const double*)nullptr, out,
                       (double*)nullptr, 0LL, 0);
}

// out[c] = sum_r R[r, c]^2 (one workgroup per column)
__global__ __launch_bounds__(256) void k_col_sumsq(const double* __restrict__ R, int ld, int rows,
                                                    double* __restrict__ out) {
    __shared__ double sh[16];
    const double* col = R + (size_t)blockIdx.x * ld;
    double a = 0.0;
    for (int r = threadIdx.x; r < rows; r += blockDim.x) a += col[r] * col[r];
    a = block_sum(a, sh);
    if (threadIdx.x == 0) out[blockIdx.x] = a;
}

void col_sumsq(hipStream_t s, const double* R, int ld, int rows, int cols, double* out) {
    NMGP_LAUNCH(k_col_sumsq, dim3(cols), dim3(256), 0, s, R, ld, rows, out);
}

// alpha = W z for the UPPER-triangular W = L^-T that rides below the factor in a gradient evaluation (W[i, k] at W[i + k ld],
// exact zeros for k < i): Sigma^-1 y = L^-T (L^-1 y).  The library's dgemv streams all n^2 entries (n = 6144: 322 us, 0.94 TB/s
// on one matrix); here only the 256 x 256 blocks on and above the diagonal are read, one block per workgroup with lanes along
// i (512-byte segments per column), and the block sums are added in a fixed order by a second small kernel -- no atomics.
__global__ __launch_bounds__(256) void k_tri_gemv_part(const double* __restrict__ W, int ld, int n, const double* __restrict__ z,
                                                        double* __restrict__ part, long long wstride, long long pstride) {
    const int rb = blockIdx.x, ch = blockIdx.y;
    if (ch < rb) return;                                  // a block of structural zeros
    W += (size_t)blockIdx.z * wstride;
    z += (size_t)blockIdx.z * n;
    part += (size_t)blockIdx.z * pstride;
    __shared__ double sz[256];
    const int tid = threadIdx.x;
    const int k0 = ch * 256;
    sz[tid] = (k0 + tid < n) ? z[k0 + tid] : 0.0;
    __syncthreads();
    const int i = rb * 256 + tid;
    if (i >= n) return;
    const int kn = (n - k0 < 256) ? (n - k0) : 256;
    const double* col = W + i + (size_t)k0 * ld;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int k = 0;
    for (; k + 8 <= kn; k += 8) {                         // eight loads in flight per lane
        const double w0 = col[(size_t)(k + 0) * ld], w1 = col[(size_t)(k + 1) * ld], w2 = col[(size_t)(k + 2) * ld],
                     w3 = col[(size_t)(k + 3) * ld], w4 = col[(size_t)(k + 4) * ld], w5 = col[(size_t)(k + 5) * ld],
                     w6 = col[(size_t)(k + 6) * ld], w7 = col[(size_t)(k + 7) * ld];
        a0 = fma(w0, sz[k + 0], a0);
        a1 = fma(w1, sz[k + 1], a1);
        a2 = fma(w2, sz[k + 2], a2);
        a3 = fma(w3, sz[k + 3], a3);
        a0 = fma(w4, sz[k + 4], a0);
        a1 = fma(w5, sz[k + 5], a1);
        a2 = fma(w6, sz[k + 6], a2);
        a3 = fma(w7, sz[k + 7], a3);
    }
    for (; k < kn; ++k) a0 = fma(col[(size_t)k * ld], sz[k], a0);
    part[(size_t)ch * n + i] = (a0 + a1) + (a2 + a3);
}

__global__ __launch_bounds__(256) void k_tri_gemv_reduce(const double* __restrict__ part, int n, double* __restrict__ out,
                                                          long long pstride) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    part += (size_t)blockIdx.y * pstride;
    const int nch = (n + 255) / 256;
    double a = 0.0;
    for (int ch = i / 256; ch < nch; ++ch) a += part[(size_t)ch * n + i];
    out[(size_t)blockIdx.y * n + i] = a;
}

// (row r of a diagonal block reads columns from the block's first one on: at most NMGP_TRI_GEMV_BLOCK - 64 + r % 64 left of r's own
// 64-column block start -- inside the seeded band)
static_assert(NMGP_TRI_GEMV_BLOCK == 256, "k_tri_gemv_part / _reduce are written for 256-wide blocks");
static_assert(NMGP_TRI_GEMV_BLOCK <= 64 * (NMGP_XTRI_SEED_BLOCKS + 1), "tri_gemv_upper would read left of the band k_xtri_seed writes");
// part: at least n * ceil(n / 256) doubles per matrix, `pstride` apart
void tri_gemv_upper(hipStream_t s, const double* W, int ld, int n, const double* z, double* out, double* part, int batch,
                    long long wstride, long long pstride) {
    const int nb = cdiv(n, 256);
    NMGP_LAUNCH(k_tri_gemv_part, dim3(nb, nb, batch), dim3(256), 0, s, W, ld, n, z, part, wstride, pstride);
    NMGP_LAUNCH(k_tri_gemv_reduce, dim3(nb, batch), dim3(256), 0, s, part, n, out, pstride);
}

// ---------------------------------------------------------------------------------------------
// Device-resident leapfrog trajectory of B lock-step HMC chains (drivers.py BatchedHMC; Nonseparable_model.py:228-231 hands the
// same job to the external HMC_Sampler): positions q = the batch's parameter vectors, momenta p and gradients g stay in HBM
// between the batched value+gradient evaluations of a trajectory.  The arithmetic is the host sampler's, operation for
// operation (t = c * g; p = p - t; q = q + eps * p; no contraction), so both give the same bits.
// ---------------------------------------------------------------------------------------------
// bad[z] = the evaluation that just ran is undefined for chain z (factorisation status or a non-finite value); failed |= bad
__global__ void k_hmc_status(const int* __restrict__ info, const double* __restrict__ scal, int* __restrict__ bad,
                             int* __restrict__ failed, int B) {
    const int z = blockIdx.x * blockDim.x + threadIdx.x;
    if (z >= B) return;
    const double v0 = scal[(size_t)z * 16 + 8], v1 = scal[(size_t)z * 16 + 9];
    const int b = (info[z] != 0 || !isfinite(v0) || !isfinite(v1)) ? 1 : 0;
    bad[z] = b;
    if (failed && b) failed[z] = 1;
}

// p -= c * g for the chains whose gradient is defined (the host sampler zeroes the gradient of the others); then, if drift,
// q += eps * p
__global__ __launch_bounds__(256) void k_hmc_kick_drift(double* __restrict__ p, const double* __restrict__ g, double* __restrict__ q,
                                                         const int* __restrict__ bad, double c, double eps, int drift, long long P) {
    const int z = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const size_t o = (size_t)z * P + i;
    double pv = p[o];
    if (!bad[z]) {
        const double t = c * g[o];
        pv = pv - t;
        p[o] = pv;
    }
    if (drift) {
        const double t2 = eps * pv;
        q[o] = q[o] + t2;
    }
}

// q += eps * v with v = M^-1 p: `minv_diag` != null: v_i = minv_diag[i] * p_i (diagonal mass matrix, computed here);
// else v = `vel` [B, P] (dense mass matrix: the caller's GEMM M^-1 P).  Same operation order as the host sampler: t = eps * v; q + t.
__global__ __launch_bounds__(256) void k_hmc_drift(double* __restrict__ q, const double* __restrict__ p, const double* __restrict__ vel,
                                                    const double* __restrict__ minv_diag, double eps, long long P) {
    const int z = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const size_t o = (size_t)z * P + i;
    const double v = minv_diag ? minv_diag[i] * p[o] : vel[o];
    const double t = eps * v;
    q[o] = q[o] + t;
}

// rejected chains (accept[z] == 0) get the position, gradient and validity flag they had before the trajectory back
__global__ __launch_bounds__(256) void k_hmc_restore(double* __restrict__ q, double* __restrict__ g, const double* __restrict__ q0,
                                                      const double* __restrict__ g0, int* __restrict__ bad,
                                                      const int* __restrict__ bad0, const int* __restrict__ accept, long long P) {
    const int z = blockIdx.y;
    if (accept[z]) return;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i == 0) bad[z] = bad0[z];
    if (i >= P) return;
    const size_t o = (size_t)z * P + i;
    q[o] = q0[o];
    g[o] = g0[o];
}

// p[z, i] = mchol_diag[i] * p[z, i]: momenta p = chol(M) z of a DIAGONAL mass matrix from the standard normals the caller uploaded
__global__ __launch_bounds__(256) void k_hmc_scale(double* __restrict__ p, const double* __restrict__ d, long long P) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const size_t o = (size_t)blockIdx.y * P + i;
    p[o] = d[i] * p[o];
}

// kin[z] = 1/2 sum_i p[z, i] v[z, i], v = M^-1 p: p itself (identity), minv_diag[i] p (diagonal) or `vel` (dense: the caller's GEMM).
// One workgroup per chain, fixed summation order (strided partial sums, then a tree): the same bits on every call.
__global__ __launch_bounds__(256) void k_hmc_kinetic(const double* __restrict__ p, const double* __restrict__ vel,
                                                      const double* __restrict__ minv_diag, double* __restrict__ kin, long long P) {
    __shared__ double red[256];
    const int z = blockIdx.x, t = threadIdx.x;
    const double* pz = p + (size_t)z * P;
    const double* vz = vel ? vel + (size_t)z * P : nullptr;
    double acc = 0.0;
    for (long long i = t; i < P; i += 256) {
        const double pv = pz[i];
        const double v = vz ? vz[i] : (minv_diag ? minv_diag[i] * pv : pv);
        acc = fma(pv, v, acc);
    }
    red[t] = acc;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (t < h) red[t] += red[t + h];
        __syncthreads();
    }
    if (t == 0) kin[z] = 0.5 * red[0];
}

void hmc_scale(hipStream_t s, double* p, const double* d, long long P, int B) {
    NMGP_LAUNCH(k_hmc_scale, dim3((unsigned)((P + 255) / 256), B), dim3(256), 0, s, p, d, P);
}
void hmc_kinetic(hipStream_t s, const double* p, const double* vel, const double* minv_diag, double* kin, long long P, int B) {
    NMGP_LAUNCH(k_hmc_kinetic, dim3(B), dim3(256), 0, s, p, vel, minv_diag, kin, P);
}

void hmc_status(hipStream_t s, const int* info, const double* scal, int* bad, int* failed, int B) {
    NMGP_LAUNCH(k_hmc_status, dim3(cdiv(B, 64)), dim3(64), 0, s, info, scal, bad, failed, B);
}
void hmc_kick_drift(hipStream_t s, double* p, const double* g, double* q, const int* bad, double c, double eps, int drift,
                    long long P, int B) {
    NMGP_LAUNCH(k_hmc_kick_drift, dim3((unsigned)((P + 255) / 256), B), dim3(256), 0, s, p, g, q, bad, c, eps, drift, P);
}
void hmc_drift(hipStream_t s, double* q, const double* p, const double* vel, const double* minv_diag, double eps, long long P, int B) {
    NMGP_LAUNCH(k_hmc_drift, dim3((unsigned)((P + 255) / 256), B), dim3(256), 0, s, q, p, vel, minv_diag, eps, P);
}
void hmc_restore(hipStream_t s, double* q, double* g, const double* q0, const double* g0, int* bad, const int* bad0,
                 const int* accept, long long P, int B) {
    NMGP_LAUNCH(k_hmc_restore, dim3((unsigned)((P + 255) / 256), B), dim3(256), 0, s, q, g, q0, g0, bad, bad0, accept, P);
}

// torch.optim.Adam's update (default: no weight decay, no amsgrad) on the batch's parameter vectors, operation for operation as
// drivers.LockStepMAP spells it on the host: m = m b1 + (1 - b1) g;  v = v b2 + ((1 - b2) g) g;  denom = sqrt(v) / bc2s + eps;
// P = P - (lr / bc1) (m / denom).  Chains that are no longer alive (a failed evaluation, now or earlier) are left untouched.
__global__ __launch_bounds__(256) void k_adam_step(double* __restrict__ par, const double* __restrict__ g, double* __restrict__ m,
                                                    double* __restrict__ v, const int* __restrict__ alive, double b1, double omb1,
                                                    double b2, double omb2, double bc2s, double eps, double step, long long P) {
    const int z = blockIdx.y;
    if (!alive[z]) return;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const size_t o = (size_t)z * P + i;
    const double gv = g[o];
    const double t1 = m[o] * b1, t2 = omb1 * gv;
    const double mn = t1 + t2;
    const double t3 = v[o] * b2, t4 = omb2 * gv;
    const double t5 = t4 * gv;
    const double vn = t3 + t5;
    const double t6 = sqrt(vn) / bc2s;
    const double denom = t6 + eps;
    const double t7 = mn / denom;
    const double t8 = step * t7;
    par[o] = par[o] - t8;
    m[o] = mn;
    v[o] = vn;
}

// alive[z] &= the evaluation that just ran is defined for chain z (see k_hmc_status)
__global__ void k_alive_update(const int* __restrict__ info, const double* __restrict__ scal, int* __restrict__ alive, int B) {
    const int z = blockIdx.x * blockDim.x + threadIdx.x;
    if (z >= B) return;
    const double v0 = scal[(size_t)z * 16 + 8], v1 = scal[(size_t)z * 16 + 9];
    if (info[z] != 0 || !isfinite(v0) || !isfinite(v1)) alive[z] = 0;
}

void adam_step(hipStream_t s, double* par, const double* g, double* m, double* v, int* alive, const int* info, const double* scal,
               double b1, double b2, double bc2s, double eps, double step, long long P, int B) {
    NMGP_LAUNCH(k_alive_update, dim3(cdiv(B, 64)), dim3(64), 0, s, info, scal, alive, B);
    NMGP_LAUNCH(k_adam_step, dim3((unsigned)((P + 255) / 256), B), dim3(256), 0, s, par, g, m, v, alive, b1, 1.0 - b1, b2,
                1.0 - b2, bc2s, eps, step, P);
}

// mirror the lower triangle into the upper one (column-major n x n)
__global__ __launch_bounds__(256) void k_sym_fill(double* __restrict__ A, int ld, int n) {
    __shared__ double tile[64][65];
    A += (size_t)blockIdx.z * ld * n;
    const int bi = blockIdx.x, bj = blockIdx.y;   // tile row / col, bi >= bj handled
    if (bi < bj) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // read tile (rows bi*64.., cols bj*64..) of the lower part
    for (int c = w; c < 64; c += 4) {
        int r = bi * 64 + lane, cc = bj * 64 + c;
        tile[c][lane] = (r < n && cc < n) ? A[(size_t)cc * ld + r] : 0.0;
    }
    __syncthreads();
    // write transposed into (rows bj*64.., cols bi*64..)
    for (int c = w; c < 64; c += 4) {
        int r = bj * 64 + lane, cc = bi * 64 + c;   // element (r, cc) = lower (cc, r) = tile[lane][c]
        if (r < n && cc < n && r < cc) A[(size_t)cc * ld + r] = tile[lane][c];
    }
}

void fill_lower_to_full(hipStream_t s, double* A, int ld, int n, int batch) {
    NMGP_LAUNCH(k_sym_fill, dim3(cdiv(n, 64), cdiv(n, 64), batch), dim3(256), 0, s, A, ld, n);
}

// y[m N + i] = Y[i, m]  (logpos.py:338)
__global__ void k_transpose_y(const double* __restrict__ Y, int N, int M, double* __restrict__ y) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * M) return;
    int m = idx / N, i = idx % N;
    y[idx] = Y[(size_t)i * M + m];
}

void transpose_y(hipStream_t s, const double* Y, int N, int M, double* y) {
    NMGP_LAUNCH(k_transpose_y, dim3(cdiv((long long)N * M, 256)), dim3(256), 0, s, Y, N, M, y);
}

// prior right-hand sides: column 0 = tilde_l - mu_l, column 1+t = uL[:, t] - mu_L (logpos.py:358,363-365)
__global__ void k_svc_prior_rhs(const double* __restrict__ pars, int N, int T, double mu_l, double mu_L,
                                double* __restrict__ R, int ld) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    // blockIdx.y = chain: its 1 + T columns follow those of the previous chain (one multi-RHS solve for the batch)
    pars += (size_t)blockIdx.y * ((size_t)N * (1 + T) + 1);
    R += (size_t)blockIdx.y * (1 + T) * ld;
    R[i] = pars[i] - mu_l;
    for (int t = 0; t < T; ++t) R[(size_t)(1 + t) * ld + i] = pars[N + (size_t)i * T + t] - mu_L;
}

void svc_prior_rhs(hipStream_t s, const double* pars, int N, int T, double mu_l, double mu_L, double* R, int ld,
                   int batch) {
    NMGP_LAUNCH(k_svc_prior_rhs, dim3(cdiv(N, 256), batch), dim3(256), 0, s, pars, N, T, mu_l, mu_L, R, ld);
}

// HBM stream micro-benchmarks (16 B per lane).  Shape found with tools/lab/hbm_lab.hip on MI355X: a FLAT grid -- one chunk of
// 256 x 16 bytes per workgroup, no grid-stride loop -- reaches 6.2 TB/s for a copy, 6.7 TB/s read-only (7.0 with nontemporal
// loads) and 6.9 TB/s write-only; the same kernels as persistent grid-stride loops (the round-1/2 form, 2048 workgroups) stay
// at 4.7-5.4 TB/s.  MODE 0: copy, 1: read only, 2: write only.
template <int MODE>
__global__ __launch_bounds__(256) void k_stream(const double2* __restrict__ src, double2* __restrict__ dst, size_t n2, double* __restrict__ sink) {
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n2) return;
    if (MODE == 0) {
        dst[k] = src[k];
    } else if (MODE == 1) {
        typedef double v2d_t __attribute__((ext_vector_type(2)));
        const v2d_t v = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(src) + k);
        if (v[0] + v[1] == 1.2345e300) sink[threadIdx.x] = v[0];    // never true for the zero-filled source: keeps the load alive
    } else {
        dst[k] = make_double2(1.5, 2.5);
    }
}

void stream_copy(hipStream_t s, const double* src, double* dst, size_t nelem, int mode, double* sink) {
    const size_t n2 = nelem / 2;
    const dim3 grid((unsigned)((n2 + 255) / 256));
    if (mode == 0) NMGP_LAUNCH(k_stream<0>, grid, dim3(256), 0, s, (const double2*)src, (double2*)dst, n2, sink);
    else if (mode == 1) NMGP_LAUNCH(k_stream<1>, grid, dim3(256), 0, s, (const double2*)src, (double2*)dst, n2, sink);
    else NMGP_LAUNCH(k_stream<2>, grid, dim3(256), 0, s, (const double2*)src, (double2*)dst, n2, sink);
}


// ---------------------------------------------------------------------------------------------
// kernel #5: fused adjoint of the nonseparable likelihood.
//   G = 1/2 (alpha alpha^T - Sigma^-1),  G_ij = M x M block of the location pair (i, j)
//   Q_ij = G_ij L_j
//   d loglik / d L_i      = 2 sum_j K_x[i,j] tril(Q_ij)
//   d loglik / d tilde_l_i = sum_{j != i} 2 <Q_ij, L_i> K0[i,j] (1/2 - l_i^2/A + 2 l_i^2 d_ij/A^2),  A = l_i^2 + l_j^2
// This is what autograd re-derives through every temporary of logpos.py:339-354
// (Nonseparable_model.py:171 NegLog.backward()).  Sinv must hold the FULL symmetric inverse.
// One workgroup = 64 locations i x 64 locations j; lanes along i (coalesced column reads of Sinv),
// each wave takes 16 j; the four waves' sums are combined through LDS and written as a partial row
// part[J][i][0..T] (slot 0 = tilde_l, slots 1..T = packed tril of dL) -- no atomics, deterministic.
// ---------------------------------------------------------------------------------------------
template <int M>
__global__ __launch_bounds__(256) void k_svc_adjoint(const double* __restrict__ x, const double* __restrict__ ell,
                                                      const double* __restrict__ Lv,
                                                      const double* __restrict__ alpha,
                                                      const double* __restrict__ Sinv, int ld, int N,
                                                      double* __restrict__ part, double ssign, int xstride, int cps) {
    constexpr int T = M * (M + 1) / 2;
    constexpr int TJ = 64;
    __shared__ double sx[TJ], sl[TJ], sL[TJ * T], sa[TJ * M];
    x += (size_t)(blockIdx.z / cps) * xstride;          // (subject of batch element z: see k_svc_cov)
    __shared__ double red[2][4][64];
    const int I = blockIdx.x, J = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j0 = J * TJ;
    const size_t Ns = (size_t)N;
    {   // blockIdx.z = chain of the batch
        const size_t z = blockIdx.z;
        ell += z * Ns;
        Lv += z * Ns * T;
        alpha += z * Ns * M;
        Sinv += z * (size_t)ld * (Ns * M);
        part += z * (size_t)gridDim.y * Ns * (T + 1);
    }
    if (tid < TJ) {
        int j = j0 + tid;
        sx[tid] = (j < N) ? x[j] : 0.0;
        sl[tid] = (j < N) ? ell[j] : 1.0;
    }
    for (int k = tid; k < TJ * T; k += 256) {
        size_t g = (size_t)j0 * T + k;
        sL[k] = (g < Ns * T) ? Lv[g] : 0.0;
    }
    for (int k = tid; k < TJ * M; k += 256) {
        int jj = k / M, m = k % M;
        int j = j0 + jj;
        sa[k] = (j < N) ? alpha[(size_t)m * Ns + j] : 0.0;
    }
    __syncthreads();
    const int i = I * 64 + lane;
    const bool iv = i < N;
    const int ic = iv ? i : N - 1;
    const double xi = x[ic], li = ell[ic];
    const double xi2 = xi * xi, li2 = li * li;
    double Li[T], ai[M], acc[T + 1];
#pragma unroll
    for (int t = 0; t < T; ++t) Li[t] = Lv[(size_t)ic * T + t];
#pragma unroll
    for (int m = 0; m < M; ++m) ai[m] = alpha[(size_t)m * Ns + ic];
#pragma unroll
    for (int t = 0; t <= T; ++t) acc[t] = 0.0;
    if (iv) {
        for (int jj = 0; jj < TJ / 4; ++jj) {
            const int k = w * (TJ / 4) + jj;
            const int j = j0 + k;
            if (j >= N) break;
            const double xj = sx[k], lj = sl[k];
            const double dist = (xi2 + xj * xj) - 2.0 * (xi * xj);
            const double A = li2 + lj * lj;
            const double k0 = sqrt(2.0 * (li * lj) / A) * exp(-dist / A);
            const double kx = (i == j) ? (NMGP_JITTER + k0) : k0;
            double G[M][M];
#pragma unroll
            for (int mp = 0; mp < M; ++mp) {
                const double* col = Sinv + ((size_t)mp * Ns + j) * ld;
#pragma unroll
                for (int m = 0; m < M; ++m) G[m][mp] = 0.5 * (ai[m] * sa[k * M + mp] - ssign * col[(size_t)m * Ns + i]);
            }
            double H = 0.0;
#pragma unroll
            for (int m = 0; m < M; ++m) {
#pragma unroll
                for (int r = 0; r <= m; ++r) {
                    double q = 0.0;
#pragma unroll
                    for (int mp = r; mp < M; ++mp) q = fma(G[m][mp], sL[k * T + mp * (mp + 1) / 2 + r], q);
                    acc[1 + m * (m + 1) / 2 + r] = fma(2.0 * kx, q, acc[1 + m * (m + 1) / 2 + r]);
                    H = fma(q, Li[m * (m + 1) / 2 + r], H);
                }
            }
            if (i != j) {
                const double dlogk = 0.5 - li2 / A + 2.0 * li2 * dist / (A * A);
                acc[0] = fma(2.0 * H * k0, dlogk, acc[0]);
            }
        }
    }
    double* o = part + ((size_t)J * Ns + ic) * (T + 1);
#pragma unroll
    for (int t = 0; t <= T; ++t) {
        red[t & 1][w][lane] = acc[t];
        __syncthreads();
        if (w == 0 && iv) o[t] = (red[t & 1][0][lane] + red[t & 1][1][lane]) + (red[t & 1][2][lane] + red[t & 1][3][lane]);
    }
}

// ssign = +1 when Sinv holds Sigma^-1 (rocSOLVER potri), -1 when it holds -Sigma^-1 (C -= X X^T of the custom path)
int svc_adjoint(hipStream_t s, const double* x, const double* ell, const double* Lv, const double* alpha,
                const double* Sinv, int ld, int N, int M, double* part, double ssign, int batch, int xstride, int cps) {
    dim3 grid(cdiv(N, 64), cdiv(N, 64), batch);      // batched: Sinv matrices are ld x (N M) apart (ld == N M there)
    if (cps < 1) cps = 1;
#define NMGP_ADJ(MM) \
    NMGP_LAUNCH((k_svc_adjoint<MM>), grid, dim3(256), 0, s, x, ell, Lv, alpha, Sinv, ld, N, part, ssign, xstride, cps)
    switch (M) {
        case 1: NMGP_ADJ(1); break;
        case 2: NMGP_ADJ(2); break;
        case 3: NMGP_ADJ(3); break;
        case 4: NMGP_ADJ(4); break;
        case 5: NMGP_ADJ(5); break;
        case 6: NMGP_ADJ(6); break;
        case 7: NMGP_ADJ(7); break;
        case 8: NMGP_ADJ(8); break;
        default: return NMGP_E_UNSUPPORTED;
    }
#undef NMGP_ADJ
    return 0;
}

// trace term: out[0] = sum_r alpha_r^2, out[1] = sum_r Sinv[r, r]
__global__ __launch_bounds__(1024) void k_trace_terms(const double* __restrict__ alpha,
                                                       const double* __restrict__ Sinv, int ld, int n,
                                                       double* __restrict__ out, double ssign) {
    __shared__ double sh[16];
    alpha += (size_t)blockIdx.x * n;
    Sinv += (size_t)blockIdx.x * ld * n;
    out += (size_t)blockIdx.x * 2;
    double a = 0.0, d = 0.0;
    for (int r = threadIdx.x; r < n; r += blockDim.x) {
        a += alpha[r] * alpha[r];
        d += Sinv[(size_t)r * ld + r];
    }
    a = block_sum(a, sh);
    d = block_sum(d, sh);
    if (threadIdx.x == 0) {
        out[0] = a;
        out[1] = ssign * d;
    }
}

void trace_terms(hipStream_t s, const double* alpha, const double* Sinv, int ld, int n, double* out, double ssign,
                 int batch) {
    NMGP_LAUNCH(k_trace_terms, dim3(batch), dim3(1024), 0, s, alpha, Sinv, ld, n, out, ssign);
}

// Assemble d NegLog / d pars from the adjoint partials, the prior solves and the scalar terms.
//   R2: [N, 1+T] column-major, Sigma_prior^-1 (value - mean) per column.
//   tr:  {sum alpha^2, trace Sinv};  hyper a, b for the inverse-gamma term (distributions.py:126-134)
__global__ __launch_bounds__(256) void k_svc_grad_final(const double* __restrict__ part, int NJ, int N, int M, int T,
                                                         const double* __restrict__ Lv,
                                                         const double* __restrict__ R2, int ldR,
                                                         const double* __restrict__ pars,
                                                         const double* __restrict__ tr, double a, double b, int prior,
                                                         double* __restrict__ grad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t P = (size_t)N * (1 + T) + 1;
    {   // blockIdx.y = chain of the batch
        const size_t z = blockIdx.y;
        part += z * (size_t)NJ * N * (T + 1);
        Lv += z * (size_t)N * T;
        R2 += z * (size_t)(1 + T) * ldR;
        pars += z * P;
        tr += z * 2;
        grad += z * P;
    }
    if (i == 0) {
        const double tse = pars[P - 1];
        const double sigma2 = exp(tse);
        double g = sigma2 * (0.5 * (tr[0] - tr[1]));
        if (prior) g += (-a - 1.0) + b / sigma2 + 1.0;
        grad[P - 1] = -g;
    }
    if (i >= N) return;
    for (int t = 0; t <= T; ++t) {
        double sacc = 0.0;
        for (int J = 0; J < NJ; ++J) sacc += part[((size_t)J * N + i) * (T + 1) + t];
        if (t == 0) {
            if (prior) sacc -= R2[i];
            grad[i] = -sacc;
        } else {
            const int tt = t - 1;
            // diagonal slots carry the exp() reparametrisation (utils.py:16): d/d uL = d/d L * L
            int r = 0;
            while ((r + 1) * (r + 2) / 2 <= tt) ++r;
            const bool diag = (tt == r * (r + 1) / 2 + r);
            if (diag) sacc *= Lv[(size_t)i * T + tt];
            if (prior) sacc -= R2[(size_t)(1 + tt) * ldR + i];
            grad[N + (size_t)i * T + tt] = -sacc;
        }
    }
}

void svc_grad_final(hipStream_t s, const double* part, int NJ, int N, int M, const double* Lv, const double* R2,
                    int ldR, const double* pars, const double* tr, double a, double b, int prior, double* grad,
                    int batch) {
    int T = M * (M + 1) / 2;
    NMGP_LAUNCH(k_svc_grad_final, dim3(cdiv(N, 256), batch), dim3(256), 0, s, part, NJ, N, M, T, Lv, R2, ldR,
                       pars, tr, a, b, prior, grad);
}

// Scalar epilogue of the nonseparable objective (logpos.py:354-376 + distributions.py:22,126-134).
//   sc[SC_LOGDET], sc[SC_QUAD]: likelihood reductions;  q[0..T]: prior Mahalanobis terms;
//   hl_l / hl_L: half log-determinants of the two prior covariances.
__global__ void k_svc_finalize(const double* __restrict__ logdet, const double* __restrict__ quad,
                               const double* __restrict__ q, const double* __restrict__ hl_l,
                               const double* __restrict__ hl_L, const double* __restrict__ pars, long long P, int N,
                               int T, double a, double b, double ig_const, int prior, double* __restrict__ out5,
                               int sstride, int hstride, int cps) {
    if (threadIdx.x != 0) return;
    hl_l += (size_t)(blockIdx.x / cps) * hstride;      // per-subject prior factors in a multi-subject batch
    hl_L += (size_t)(blockIdx.x / cps) * hstride;
    // blockIdx.x = chain: scalar blocks are sstride apart, prior terms 1 + T apart, parameter vectors P apart
    logdet += (size_t)blockIdx.x * sstride;
    quad += (size_t)blockIdx.x * sstride;
    out5 += (size_t)blockIdx.x * sstride;
    q += (size_t)blockIdx.x * (1 + T);
    pars += (size_t)blockIdx.x * P;
    const double LOG2PI = 1.8378770664093453;
    const double tse = pars[P - 1];
    const double sigma2 = exp(tse);
    const double loglik = -0.5 * logdet[0] - 0.5 * quad[0];
    const double lp_l = -0.5 * (N * LOG2PI + q[0]) - hl_l[0];
    double lp_uL = 0.0;
    for (int t = 0; t < T; ++t) lp_uL += -0.5 * (N * LOG2PI + q[1 + t]) - hl_L[0];
    const double lp_s2 = (-a - 1.0) * log(sigma2) - b / sigma2 + ig_const;
    double res = 0.0;
    res += loglik;
    if (prior) {
        res += lp_l;
        res += lp_uL;
        res += lp_s2;
        res += tse;
    }
    out5[0] = -res;
    out5[1] = loglik;
    out5[2] = lp_l;
    out5[3] = lp_uL;
    out5[4] = lp_s2;
}

void svc_finalize(hipStream_t s, const double* logdet, const double* quad, const double* q, const double* hl_l,
                  const double* hl_L, const double* pars, long long P, int N, int T, double a, double b,
                  double ig_const, int prior, double* out5, int batch, int sstride, int hstride, int cps) {
    NMGP_LAUNCH(k_svc_finalize, dim3(batch), dim3(64), 0, s, logdet, quad, q, hl_l, hl_L, pars, P, N, T, a, b,
                       ig_const, prior, out5, sstride, hstride, cps < 1 ? 1 : cps);
}

// half log-determinant of a Cholesky factor: sum log L_rr
__global__ __launch_bounds__(1024) void k_half_logdet(const double* __restrict__ L, int ld, int n,
                                                       double* __restrict__ out) {
    __shared__ double sh[16];
    L += (size_t)blockIdx.x * ld * n;
    out += blockIdx.x;
    double a = 0.0;
    for (int r = threadIdx.x; r < n; r += blockDim.x) a += log(L[(size_t)r * ld + r]);
    a = block_sum(a, sh);
    if (threadIdx.x == 0) out[0] = a;
}

void half_logdet(hipStream_t s, const double* L, int ld, int n, double* out, int batch) {
    NMGP_LAUNCH(k_half_logdet, dim3(batch), dim3(1024), 0, s, L, ld, n, out);
}


// ---------------------------------------------------------------------------------------------
// Per-subject prior solves  op(L) x = r  for the few right-hand sides of the GP priors (1 + T columns per subject: the
// centred tilde_l and the T columns of uL_vecs), one workgroup per (column, subject).  With every subject bringing its
// own N x N factor the library's batched trsm (inverted diagonal blocks + many small GEMMs) took 1 ms of a 5 ms step for
// 8 subjects of N = 1024; a right-hand side is a streaming pass over its factor: 64-wide substitution steps by one wave
// on a diagonal block staged in LDS, the rest of the column block applied as a GEMV by the whole workgroup.
// TRANS = false: L x = r (forward), TRANS = true: L^T x = r (backward; the gradient of the prior terms).
// Column j of subject b: r = R + (b * nrhs + j) * N; its factor is L0 for j = 0 and L1 for j >= 1 (strides s0 / s1).
// ---------------------------------------------------------------------------------------------
template <bool TRANS>
__global__ __launch_bounds__(1024) void k_prior_trsv(const double* __restrict__ L0, int ld0, long long s0,
                                                      const double* __restrict__ L1, int ld1, long long s1,
                                                      double* __restrict__ R, int N, int nrhs, int cps) {
    const int j = blockIdx.x, b = blockIdx.y;
    const double* L = (j == 0 ? L0 + (size_t)(b / cps) * s0 : L1 + (size_t)(b / cps) * s1);      // factors of subject b / cps
    const int ld = (j == 0 ? ld0 : ld1);
    double* r = R + ((size_t)b * nrhs + j) * N;
    extern __shared__ double sh[];
    double* rv = sh;                        // [N + 1] the right-hand side / solution (one pad element for row pairs)
    double* Ld = sh + N + 2;                // [64][65] diagonal block, Ld[c * 65 + r]
    double* xs = Ld + 64 * 65;              // [64]  the block's solution
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < N; i += 1024) rv[i] = r[i];
    if (tid == 0) rv[N] = 0.0;
    const int nblk = (N + 63) / 64;
    for (int bb = 0; bb < nblk; ++bb) {
        const int kb = (TRANS ? nblk - 1 - bb : bb) * 64;
        const int nbk = N - kb < 64 ? N - kb : 64;
        __syncthreads();                    // rv up to date, Ld / xs free
        for (int e = tid; e < 64 * 64; e += 1024) {
            const int rr = e & 63, cc = e >> 6;
            Ld[cc * 65 + rr] = (rr < nbk && cc <= rr) ? L[(size_t)(kb + cc) * ld + kb + rr] : (rr == cc ? 1.0 : 0.0);
        }
        __syncthreads();
        if (w == 0) {
            // 64 substitution steps by one wave: lane = row of the block; x_k = v_k / L_kk by a true division (these are the
            // ill-conditioned prior factors: no reciprocals, no inverted blocks), broadcast by v_readlane
            double v = lane < nbk ? rv[kb + lane] : 0.0;
            const double d = Ld[lane * 65 + lane];
            if (!TRANS) {
                for (int k = 0; k < nbk; ++k) {
                    const double xk = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), k),
                                                       __builtin_amdgcn_readlane(__double2loint(v), k)) /
                                      __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(d), k),
                                                       __builtin_amdgcn_readlane(__double2loint(d), k));
                    const double lik = Ld[k * 65 + lane];              // L[kb + lane][kb + k]
                    v = (lane == k) ? xk : (lane > k ? v - lik * xk : v);
                }
            } else {
                for (int k = nbk - 1; k >= 0; --k) {
                    const double xk = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), k),
                                                       __builtin_amdgcn_readlane(__double2loint(v), k)) /
                                      __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(d), k),
                                                       __builtin_amdgcn_readlane(__double2loint(d), k));
                    const double lki = Ld[lane * 65 + k];              // L[kb + k][kb + lane]
                    v = (lane == k) ? xk : (lane < k ? v - lki * xk : v);
                }
            }
            xs[lane] = (lane < nbk) ? v : 0.0;
            if (lane < nbk) rv[kb + lane] = v;
        }
        __syncthreads();
        // The rest of the column block as a GEMV.  It is latency-bound on ONE workgroup's loads, so everything is laid out for
        // loads in flight: 16 waves, a wave takes chunks of rows, its lanes split the 64 columns of the block 8 ways and every
        // lane issues all its loads before the first use; the 8 partial sums meet by three xor-shuffles.
        if (!TRANS) {
            // rows below the block: r[i] -= sum_k L[i][kb + k] x[k].  lane = (row pair rp, k group kg): rows i0 + 2 rp, + 1;
            // columns kb + 8 kg .. + 7 (16-byte loads along i, 128-byte segments per column)
            const int rp = lane & 7, kg = lane >> 3;
            double xk[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) xk[q] = xs[8 * kg + q];
            const int first = kb + 64;
            for (int i0 = first + 16 * w; i0 < N; i0 += 16 * 16) {
                const int i = i0 + 2 * rp;
                const bool v0 = i < N, v1 = i + 1 < N;
                const int ic = v0 ? i : 0;          // (i, i + 1) with i + 1 == N reads the column's padding: ld > N for odd N
                const double* Lc = L + (size_t)(kb + 8 * kg) * ld + ic;
                double2 t[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) t[q] = *reinterpret_cast<const double2*>(Lc + (size_t)q * ld);
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    a0 = fma(t[q].x, xk[q], a0);
                    a1 = fma(t[q].y, xk[q], a1);
                }
#pragma unroll
                for (int m = 8; m < 64; m <<= 1) {
                    a0 += __shfl_xor(a0, m);
                    a1 += __shfl_xor(a1, m);
                }
                if (kg == 0 && v0) rv[i] -= a0;
                if (kg == 0 && v1) rv[i + 1] -= a1;
            }
        } else {
            // columns left of the block: r[i] -= sum_k L[kb + k][i] x[k]: 64 contiguous doubles of column i.  lane = (k group
            // kq of 8 contiguous k, column ii of 8): four 16-byte loads per lane
            const int kq = lane & 7, ii = lane >> 3;
            double xk[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) xk[q] = xs[8 * kq + q];
            for (int i0 = 8 * w; i0 < kb; i0 += 8 * 16) {
                const int i = i0 + ii;
                const bool vi = i < kb;
                const double* Lc = L + (size_t)(vi ? i : 0) * ld + kb + 8 * kq;
                double2 t[4];
                if (kb + 64 <= N) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) t[q] = *reinterpret_cast<const double2*>(Lc + 2 * q);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        t[q].x = (8 * kq + 2 * q < nbk) ? Lc[2 * q] : 0.0;
                        t[q].y = (8 * kq + 2 * q + 1 < nbk) ? Lc[2 * q + 1] : 0.0;
                    }
                }
                double a0 = 0.0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a0 = fma(t[q].x, xk[2 * q], a0);
                    a0 = fma(t[q].y, xk[2 * q + 1], a0);
                }
#pragma unroll
                for (int m = 1; m < 8; m <<= 1) a0 += __shfl_xor(a0, m);
                if (kq == 0 && vi) rv[i] -= a0;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < N; i += 1024) r[i] = rv[i];
}

void prior_trsv(hipStream_t s, bool trans, const double* L0, int ld0, long long s0, const double* L1, int ld1, long long s1,
                double* R, int N, int nrhs, int batch, int cps) {
    const size_t lds = ((size_t)N + 2 + 64 * 65 + 64) * sizeof(double);
    if (cps < 1) cps = 1;
    if (lds > 64 * 1024) {
        // beyond the default 64 KB of dynamic LDS (N > 3966): opt in to the CU's 160 KB (prediction at config 5's N = 4096).  The
        // attribute belongs to the CURRENT device: one flag per device ordinal (contexts on several GPUs of one process), set only
        // once the runtime has accepted it -- a refusal is reported through the launch that follows (NMGP_LAUNCH records it) and
        // retried on the next call.  (A context belongs to one host thread; the flags are written under a mutex all the same.)
        static std::mutex mu;
        static bool raised[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(mu);
        if (dev >= 0 && dev < 64 && !raised[dev]) {
            const hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_prior_trsv<true>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            const hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_prior_trsv<false>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            raised[dev] = (e0 == hipSuccess && e1 == hipSuccess);
            if (!raised[dev]) (void)hipGetLastError();
        }
    }
    if (trans)
        NMGP_LAUNCH(k_prior_trsv<true>, dim3(nrhs, batch), dim3(1024), lds, s, L0, ld0, s0, L1, ld1, s1, R, N, nrhs, cps);
    else
        NMGP_LAUNCH(k_prior_trsv<false>, dim3(nrhs, batch), dim3(1024), lds, s, L0, ld0, s0, L1, ld1, s1, R, N, nrhs, cps);
}

}  // namespace nmgpk
