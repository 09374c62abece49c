// gfx950 kernels of the Kronecker-structured (separable / stationary) likelihood and of deterministic
// prediction.  FP64 throughout; -ffp-contract=off.
#include "nmgp_internal.h"

namespace nmgpk {

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

__device__ inline double wave_sum_e(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ inline double block_sum_e(double v, double* sh /*[16]*/) {
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = wave_sum_e(v);
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    int nw = (blockDim.x + 63) >> 6;
    v = (threadIdx.x < nw) ? sh[threadIdx.x] : 0.0;
    if (w == 0) v = wave_sum_e(v);
    return v;
}

// ---------------------------------------------------------------------------------------------
// kernel #3: Kron-vec contraction  out = (B kron K) y = vec(K Y B^T)      (kronecker_operation.py:72-85)
//   K: [n1, n2] row-major (for the eigen path this is V_K^T, i.e. the column-major eigenvector matrix as is),
//   y: [m2 * n2] output-major,  B: [m1, m2] row-major,  out[a * n1 + r].
// One wave per output row r: lanes run along the contraction index c (coalesced 512-byte reads of K's row and
// of each y segment), wave shuffle reduction, then the m2 -> m1 mixing by B on the first m1 lanes.
// HBM-bound: K is streamed exactly once, y (m2 * n2 doubles) stays in L2.
// ---------------------------------------------------------------------------------------------
#define KMV_CH 8
template <bool VEC2>
__global__ __launch_bounds__(256) void k_kron_mv(const double* __restrict__ K, int n1, int n2,
                                                  const double* __restrict__ y, const double* __restrict__ B, int m1,
                                                  int m2, double* __restrict__ out) {
    __shared__ double sS[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + w;
    if (r >= n1) return;
    const double* row = K + (size_t)r * n2;
    for (int b0 = 0; b0 < m2; b0 += KMV_CH) {
        double acc[KMV_CH];
#pragma unroll
        for (int b = 0; b < KMV_CH; ++b) acc[b] = 0.0;
        if (VEC2) {
            // 16 bytes per lane, four independent row segments in flight per lane (1 KiB per wave instruction).
            // (Round 3 tried, at N = 4096, D = 5, where this launch takes 43-46 us = 3.0 TB/s: 8 and 32 segments in flight per lane
            // -- 43 and 93 us --, four rows per workgroup sharing their y loads -- 49 us --, and per-row start offsets staggered
            // against the power-of-two row stride -- 52 us.  Neither the bytes a wave keeps in flight, nor the L2 re-reads of y, nor
            // channel conflicts bound it; it runs once per evaluation of the eigen formulation, next to a 133 ms dsyevd.)
            const int n2h = n2 >> 1;
            const double2* row2 = reinterpret_cast<const double2*>(row);
            int c = lane;
            for (; c + 192 < n2h; c += 256) {
                double2 kv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) kv[u] = row2[c + 64 * u];
#pragma unroll
                for (int b = 0; b < KMV_CH; ++b) {
                    if (b0 + b < m2) {
                        const double2* y2 = reinterpret_cast<const double2*>(y + (size_t)(b0 + b) * n2);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const double2 yv = y2[c + 64 * u];
                            acc[b] = fma(kv[u].x, yv.x, acc[b]);
                            acc[b] = fma(kv[u].y, yv.y, acc[b]);
                        }
                    }
                }
            }
            for (; c < n2h; c += 64) {
                const double2 kv = row2[c];
#pragma unroll
                for (int b = 0; b < KMV_CH; ++b)
                    if (b0 + b < m2) {
                        const double2 yv = reinterpret_cast<const double2*>(y + (size_t)(b0 + b) * n2)[c];
                        acc[b] = fma(kv.x, yv.x, acc[b]);
                        acc[b] = fma(kv.y, yv.y, acc[b]);
                    }
            }
        } else {
            for (int c = lane; c < n2; c += 64) {
                const double kv = row[c];
#pragma unroll
                for (int b = 0; b < KMV_CH; ++b)
                    if (b0 + b < m2) acc[b] = fma(kv, y[(size_t)(b0 + b) * n2 + c], acc[b]);
            }
        }
#pragma unroll
        for (int b = 0; b < KMV_CH; ++b) {
            double s = wave_sum_e(acc[b]);
            if (lane == 0 && b0 + b < m2) sS[w][b0 + b] = s;
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the LDS stores above are visible to this wave's reads
    for (int a = lane; a < m1; a += 64) {
        double s = 0.0;
        for (int b = 0; b < m2; ++b) s = fma(B[(size_t)a * m2 + b], sS[w][b], s);
        out[(size_t)a * n1 + r] = s;
    }
}

int kron_mv(hipStream_t s, const double* K, int n1, int n2, const double* y, const double* B, int m1, int m2,
            double* out) {
    if (m2 > 64) return NMGP_E_UNSUPPORTED;
    const bool vec2 = (n2 % 2 == 0) && (((size_t)K | (size_t)y) % 16 == 0);
    if (vec2)
        NMGP_LAUNCH((k_kron_mv<true>), dim3(cdiv(n1, 4)), dim3(256), 0, s, K, n1, n2, y, B, m1, m2, out);
    else
        NMGP_LAUNCH((k_kron_mv<false>), dim3(cdiv(n1, 4)), dim3(256), 0, s, K, n1, n2, y, B, m1, m2, out);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// kernel #4: eigen-space reductions (distributions.py:44-51)
//   t = wB[p] wK[q];  w = 1/(sigma2 + t);  out[0] = sum log(t + sigma2);  out[1] = sum a^2 w;
//   out[2] = sum (a w)^2;  out[3] = sum w.   a is overwritten with a*w (alpha in the eigenbasis) when SCALE.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_eig_reduce(double* __restrict__ a, const double* __restrict__ wB, int M,
                                                      const double* __restrict__ wK, int N,
                                                      const double* __restrict__ sigma2p, int scale,
                                                      double* __restrict__ out) {
    __shared__ double sh[16];
    const double sigma2 = sigma2p[0];
    double c = 0.0, b = 0.0, aa = 0.0, sw = 0.0;
    const long long tot = (long long)M * N;
    for (long long k = threadIdx.x; k < tot; k += blockDim.x) {
        const int p = (int)(k / N), q = (int)(k % N);
        const double t = wB[p] * wK[q];
        const double w = 1.0 / (sigma2 + t);
        const double av = a[k];
        c += log(t + sigma2);
        b += (av * w) * av;
        aa += (av * w) * (av * w);
        sw += w;
        if (scale) a[k] = av * w;
    }
    c = block_sum_e(c, sh);
    b = block_sum_e(b, sh);
    aa = block_sum_e(aa, sh);
    sw = block_sum_e(sw, sh);
    if (threadIdx.x == 0) {
        out[0] = c;
        out[1] = b;
        out[2] = aa;
        out[3] = sw;
    }
}

void eig_reduce(hipStream_t s, double* a, const double* wB, int M, const double* wK, int N, const double* sigma2p,
                bool scale, double* out) {
    NMGP_LAUNCH(k_eig_reduce, dim3(1), dim3(1024), 0, s, a, wB, M, wK, N, sigma2p, scale ? 1 : 0, out);
}

// dvec[q] = sum_p wB[p] / (sigma2 + wB[p] wK[q]);  Vs[:, q] = V[:, q] * dvec[q]   (column-major N x N)
__global__ __launch_bounds__(256) void k_colscale_d(const double* __restrict__ V, const double* __restrict__ wB, int M,
                                                     const double* __restrict__ wK, int N,
                                                     const double* __restrict__ sigma2p, double* __restrict__ Vs) {
    const int q = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const double sigma2 = sigma2p[0];
    double d = 0.0;
    for (int p = 0; p < M; ++p) d += wB[p] / (sigma2 + wB[p] * wK[q]);
    Vs[(size_t)q * N + i] = V[(size_t)q * N + i] * d;
}

void colscale_d(hipStream_t s, const double* V, const double* wB, int M, const double* wK, int N, const double* sigma2p,
                double* Vs) {
    NMGP_LAUNCH(k_colscale_d, dim3(cdiv(N, 256), N), dim3(256), 0, s, V, wB, M, wK, N, sigma2p, Vs);
}

// Vs[:, q] = V[:, q] * svec[q]
__global__ __launch_bounds__(256) void k_colscale(const double* __restrict__ V, const double* __restrict__ svec,
                                                   int rows, int cols, double* __restrict__ Vs) {
    const int q = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    Vs[(size_t)q * rows + i] = V[(size_t)q * rows + i] * svec[q];
}

void colscale(hipStream_t s, const double* V, const double* svec, int rows, int cols, double* Vs) {
    NMGP_LAUNCH(k_colscale, dim3(cdiv(rows, 256), cols), dim3(256), 0, s, V, svec, rows, cols, Vs);
}

// coreB[p, p'] = sum_q wK[q] At[p,q] At[p',q]  - d_pp' sum_q wK[q] / (sigma2 + wB[p] wK[q])      (M x M, one block)
__global__ __launch_bounds__(256) void k_sep_coreB(const double* __restrict__ At, const double* __restrict__ wB, int M,
                                                    const double* __restrict__ wK, int N,
                                                    const double* __restrict__ sigma2p, double* __restrict__ coreB) {
    __shared__ double sh[16];
    const int p = blockIdx.x / M, pp = blockIdx.x % M;
    const double sigma2 = sigma2p[0];
    double acc = 0.0;
    for (int q = threadIdx.x; q < N; q += blockDim.x) {
        double v = wK[q] * At[(size_t)p * N + q] * At[(size_t)pp * N + q];
        if (p == pp) v -= wK[q] / (sigma2 + wB[p] * wK[q]);
        acc += v;
    }
    acc = block_sum_e(acc, sh);
    if (threadIdx.x == 0) coreB[p * M + pp] = acc;
}

void sep_coreB(hipStream_t s, const double* At, const double* wB, int M, const double* wK, int N,
               const double* sigma2p, double* coreB) {
    NMGP_LAUNCH(k_sep_coreB, dim3(M * M), dim3(256), 0, s, At, wB, M, wK, N, sigma2p, coreB);
}

// ---------------------------------------------------------------------------------------------
// fused adjoint of the separable likelihood w.r.t. the per-location curves:
//   dK_ij = 1/2 ( sum_p wB[p] U[i,p] U[j,p] - C[i,j] ),   Ks_ij = s_i s_j K0_ij
//   g_tl[i] = sum_{j != i} 2 dK_ij Ks_ij (1/2 - l_i^2/A + 2 l_i^2 d_ij / A^2),   g_ts[i] = 2 sum_j dK_ij Ks_ij
// C = V diag(dvec) V^T (full symmetric, column-major), U = V At^T ([N, M] column-major).
// Same tiling as the nonseparable adjoint: lanes along i, 64 x 64 location tile, deterministic partials.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sep_adjoint(const double* __restrict__ x, const double* __restrict__ ell,
                                                      const double* __restrict__ sig, const double* __restrict__ U,
                                                      const double* __restrict__ wB, int M,
                                                      const double* __restrict__ C, int N,
                                                      double* __restrict__ part) {
    constexpr int TJ = 64;
    __shared__ double sx[TJ], sl[TJ], ss[TJ], sU[TJ * NMGP_MAX_OUTPUTS];
    __shared__ double red[2][4][64];
    const int I = blockIdx.x, J = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j0 = J * TJ;
    if (tid < TJ) {
        int j = j0 + tid;
        sx[tid] = (j < N) ? x[j] : 0.0;
        sl[tid] = (j < N) ? ell[j] : 1.0;
        ss[tid] = (j < N) ? sig[j] : 1.0;
    }
    // every slot of sU is written: the contraction below runs over all NMGP_MAX_OUTPUTS slots with Ui[p] = 0 for p >= M,
    // and 0 * (whatever a previous kernel left in LDS, possibly a NaN) is not 0
    for (int k = tid; k < TJ * NMGP_MAX_OUTPUTS; k += 256) {
        int jj = k / NMGP_MAX_OUTPUTS, p = k % NMGP_MAX_OUTPUTS;
        int j = j0 + jj;
        sU[k] = (j < N && p < M) ? U[(size_t)p * N + j] : 0.0;
    }
    __syncthreads();
    const int i = I * 64 + lane;
    const bool iv = i < N;
    const int ic = iv ? i : N - 1;
    const double xi = x[ic], li = ell[ic], si = sig[ic];
    const double xi2 = xi * xi, li2 = li * li;
    double Ui[NMGP_MAX_OUTPUTS];
#pragma unroll
    for (int p = 0; p < NMGP_MAX_OUTPUTS; ++p) Ui[p] = (p < M) ? wB[p] * U[(size_t)p * N + ic] : 0.0;
    double gtl = 0.0, gts = 0.0;
    if (iv) {
        for (int jj = 0; jj < TJ / 4; ++jj) {
            const int k = w * (TJ / 4) + jj;
            const int j = j0 + k;
            if (j >= N) break;
            const double xj = sx[k], lj = sl[k];
            const double dist = (xi2 + xj * xj) - 2.0 * (xi * xj);
            const double A = li2 + lj * lj;
            const double ks = (si * ss[k]) * sqrt(2.0 * (li * lj) / A) * exp(-dist / A);
            double uu = 0.0;
#pragma unroll
            for (int p = 0; p < NMGP_MAX_OUTPUTS; ++p) uu = fma(Ui[p], sU[k * NMGP_MAX_OUTPUTS + p], uu);
            const double dk = 0.5 * (uu - C[(size_t)j * N + i]);
            const double t = dk * ks;
            gts = fma(2.0, t, gts);
            if (i != j) {
                const double dlogk = 0.5 - li2 / A + 2.0 * li2 * dist / (A * A);
                gtl = fma(2.0 * t, dlogk, gtl);
            }
        }
    }
    double* o = part + ((size_t)J * N + ic) * 2;
    double acc[2] = {gtl, gts};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        red[t & 1][w][lane] = acc[t];
        __syncthreads();
        if (w == 0 && iv) o[t] = (red[t & 1][0][lane] + red[t & 1][1][lane]) + (red[t & 1][2][lane] + red[t & 1][3][lane]);
    }
}

void sep_adjoint(hipStream_t s, const double* x, const double* ell, const double* sig, const double* U,
                 const double* wB, int M, const double* C, int N, double* part) {
    NMGP_LAUNCH(k_sep_adjoint, dim3(cdiv(N, 64), cdiv(N, 64)), dim3(256), 0, s, x, ell, sig, U, wB, M, C, N, part);
}

// sum the J partials: g[i*2 + t]
__global__ void k_sep_grad_sum(const double* __restrict__ part, int NJ, int N, double* __restrict__ g) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double a = 0.0, b = 0.0;
    for (int J = 0; J < NJ; ++J) {
        a += part[((size_t)J * N + i) * 2];
        b += part[((size_t)J * N + i) * 2 + 1];
    }
    g[i] = a;
    g[N + i] = b;
}

void sep_grad_sum(hipStream_t s, const double* part, int NJ, int N, double* g) {
    NMGP_LAUNCH(k_sep_grad_sum, dim3(cdiv(N, 256)), dim3(256), 0, s, part, NJ, N, g);
}

// elementwise helpers -----------------------------------------------------------------------------
__global__ void k_exp_vec(const double* __restrict__ in, int n, double* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = exp(in[i]);
}
void exp_vec(hipStream_t s, const double* in, int n, double* out) {
    NMGP_LAUNCH(k_exp_vec, dim3(cdiv(n, 256)), dim3(256), 0, s, in, n, out);
}

__global__ void k_fill_vec(double* __restrict__ out, int n, double v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v;
}
void fill_vec(hipStream_t s, double* out, int n, double v) {
    NMGP_LAUNCH(k_fill_vec, dim3(cdiv(n, 256)), dim3(256), 0, s, out, n, v);
}

// R[:, 0] = a - mu_a, R[:, 1] = b - mu_b
__global__ void k_two_col_rhs(const double* __restrict__ a, double mu_a, const double* __restrict__ b, double mu_b,
                              int N, double* __restrict__ R) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    R[i] = a[i] - mu_a;
    R[N + i] = b[i] - mu_b;
}
void two_col_rhs(hipStream_t s, const double* a, double mu_a, const double* b, double mu_b, int N, double* R) {
    NMGP_LAUNCH(k_two_col_rhs, dim3(cdiv(N, 256)), dim3(256), 0, s, a, mu_a, b, mu_b, N, R);
}

// y - mu (mu may be null)
__global__ void k_sub_vec(const double* __restrict__ y, const double* __restrict__ mu, int n, double* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = mu ? (y[i] - mu[i]) : y[i];
}
void sub_vec(hipStream_t s, const double* y, const double* mu, int n, double* out) {
    NMGP_LAUNCH(k_sub_vec, dim3(cdiv(n, 256)), dim3(256), 0, s, y, mu, n, out);
}

// out[0] = sum_i a[i] b[i]
__global__ __launch_bounds__(1024) void k_dot(const double* __restrict__ a, const double* __restrict__ b, int n,
                                               double* __restrict__ out) {
    __shared__ double sh[16];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += a[i] * b[i];
    acc = block_sum_e(acc, sh);
    if (threadIdx.x == 0) out[0] = acc;
}
void dot(hipStream_t s, const double* a, const double* b, int n, double* out) {
    NMGP_LAUNCH(k_dot, dim3(1), dim3(1024), 0, s, a, b, n, out);
}

// U[(m*N+i), (p*N+q)] = VB[m,p] * VK[i,q], row-major [MN, MN]; VB row-major [M,M]; VK column-major (V[i + q*N])
__global__ __launch_bounds__(256) void k_kron_eigvec(const double* __restrict__ VB, int M,
                                                      const double* __restrict__ VK, int N,
                                                      double* __restrict__ U) {
    const size_t n = (size_t)M * N;
    const size_t col = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t row = blockIdx.y;
    if (col >= n) return;
    const int m = (int)(row / N), i = (int)(row % N);
    const int p = (int)(col / N), q = (int)(col % N);
    U[row * n + col] = VB[m * M + p] * VK[(size_t)q * N + i];
}
void kron_eigvec(hipStream_t s, const double* VB, int M, const double* VK, int N, double* U) {
    size_t n = (size_t)M * N;
    NMGP_LAUNCH(k_kron_eigvec, dim3(cdiv(n, 256), (unsigned)n), dim3(256), 0, s, VB, M, VK, N, U);
}

// tvec[p*N+q] = 1 / (sigma2 + wB[p] wK[q])
__global__ void k_kron_w(const double* __restrict__ wB, int M, const double* __restrict__ wK, int N, double sigma2,
                         double* __restrict__ w) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M * N) return;
    w[k] = 1.0 / (wB[k / N] * wK[k % N] + sigma2);
}
void kron_w(hipStream_t s, const double* wB, int M, const double* wK, int N, double sigma2, double* w) {
    NMGP_LAUNCH(k_kron_w, dim3(cdiv((long long)M * N, 256)), dim3(256), 0, s, wB, M, wK, N, sigma2, w);
}

// ---------------------------------------------------------------------------------------------
// prediction kernels
// ---------------------------------------------------------------------------------------------
// Cross-covariances of the nonseparable model (prediction.py:968-972): k_e[(m N + i)] = kx_s[i] (L_i Lstar_s^T)[m, m'] for grid
// point s and output m' (e = s M + m'), kx_s[i] = Gibbs(x_i, l_i; xs_s, lstar_s), written TRANSPOSED below the covariance
// in the factorisation buffer: row R0 + (s M + m') of the column-major array A (leading dimension ld), column m N + i.  The blocked Cholesky turns every row r below the matrix into
// r L^-T (as it turns y into z = L^-1 y), so all S M cross-covariance vectors ride along ONE factorisation: no multi-right-hand-side
// solve afterwards.  Lanes run along the extra-row index (contiguous in a column).
__global__ __launch_bounds__(256) void k_svc_crosscov_rows(const double* __restrict__ x, const double* __restrict__ ell,
                                                            const double* __restrict__ Lv, int N, int M, int T,
                                                            const double* __restrict__ xs, const double* __restrict__ tl_star,
                                                            const double* __restrict__ Lstar, int S, double* __restrict__ A, int ld,
                                                            int R0) {
    const int e = blockIdx.y * 256 + threadIdx.x;          // extra row: grid point s = e / M, output m' = e % M
    const int i = blockIdx.x;
    if (e >= S * M) return;
    const int s = e / M, mp = e - s * M;
    const double xi = x[i], li = ell[i];
    const double xj = xs[s], lj = exp(tl_star[s]);
    const double dist = (xi * xi + xj * xj) - 2.0 * (xi * xj);
    const double Aij = li * li + lj * lj;
    const double kv = sqrt(2.0 * (li * lj) / Aij) * exp(-dist / Aij);
    for (int m = 0; m < M; ++m) {
        const int rmax = m < mp ? m : mp;
        double b = 0.0;
        for (int r = 0; r <= rmax; ++r)
            b += Lv[(size_t)i * T + m * (m + 1) / 2 + r] * Lstar[(size_t)s * T + mp * (mp + 1) / 2 + r];
        A[(size_t)(m * N + i) * ld + R0 + e] = kv * b;
    }
}
void svc_crosscov_rows(hipStream_t st, const double* x, const double* ell, const double* Lv, int N, int M, const double* xs,
                       const double* tl_star, const double* Lstar, int S, double* A, int ld, int R0) {
    int T = M * (M + 1) / 2;
    NMGP_LAUNCH(k_svc_crosscov_rows, dim3(N, cdiv((long long)S * M, 256)), dim3(256), 0, st, x, ell, Lv, N, M, T, xs, tl_star,
                Lstar, S, A, ld, R0);
}

// After the factorisation row R0 + e holds v_e = (L^-1 k_e)^T and row zrow z = L^-1 y:  mean[e] = v_e . z (= k_e^T Sigma^-1 y,
// prediction.py:973) and colsq[e] = |v_e|^2 (the diagonal of T T^T, :975-977).  Columns in chunks of 128 per workgroup (64 rows x 4
// column groups), partial sums in a fixed order: part[chunk][e][2].
__global__ __launch_bounds__(256) void k_pred_rows_part(const double* __restrict__ A, int ld, int n, int R0, int zrow, int E,
                                                         double* __restrict__ part) {
    __shared__ double red[2][4][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const int c0 = blockIdx.y * 128 + g * 32;
    double am = 0.0, aq = 0.0;
    if (e < E) {
        for (int k = 0; k < 32; ++k) {
            const int c = c0 + k;
            if (c < n) {
                const double v = A[(size_t)c * ld + R0 + e];
                am = fma(v, A[(size_t)c * ld + zrow], am);
                aq = fma(v, v, aq);
            }
        }
    }
    red[0][g][lane] = am;
    red[1][g][lane] = aq;
    __syncthreads();
    if (g == 0 && e < E) {
        double* o = part + ((size_t)blockIdx.y * E + e) * 2;
        o[0] = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
        o[1] = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
    }
}
__global__ __launch_bounds__(256) void k_pred_rows_sum(const double* __restrict__ part, int chunks, int E, double* __restrict__ mean,
                                                        double* __restrict__ colsq) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    double am = 0.0, aq = 0.0;
    for (int ch = 0; ch < chunks; ++ch) {
        am += part[((size_t)ch * E + e) * 2];
        aq += part[((size_t)ch * E + e) * 2 + 1];
    }
    mean[e] = am;
    colsq[e] = aq;
}
void pred_rows_reduce(hipStream_t st, const double* A, int ld, int n, int R0, int zrow, int E, double* part, double* mean,
                      double* colsq) {
    const int chunks = cdiv(n, 128);
    NMGP_LAUNCH(k_pred_rows_part, dim3(cdiv(E, 64), chunks), dim3(256), 0, st, A, ld, n, R0, zrow, E, part);
    NMGP_LAUNCH(k_pred_rows_sum, dim3(cdiv(E, 256)), dim3(256), 0, st, part, chunks, E, mean, colsq);
}

// GP regression outputs -> starred curves: tl_star[s] = mu_l + proj[s, 0]; uL_star[s, t] = mu_L + proj[s, 1+t];
// Lstar = uLvec2Lvec(uL_star) (exp on the diagonal slots).  proj: [S, 1+T] column-major (ld = S).
__global__ void k_svc_star(const double* __restrict__ proj, int S, int M, int T, double mu_l, double mu_L,
                           double* __restrict__ tl_star, double* __restrict__ Lstar) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    tl_star[s] = mu_l + proj[s];
    int t = 0;
    for (int r = 0; r < M; ++r)
        for (int c = 0; c <= r; ++c, ++t) {
            double v = mu_L + proj[(size_t)(1 + t) * S + s];
            Lstar[(size_t)s * T + t] = (c == r) ? exp(v) : v;
        }
}
void svc_star(hipStream_t st, const double* proj, int S, int M, double mu_l, double mu_L, double* tl_star,
              double* Lstar) {
    int T = M * (M + 1) / 2;
    NMGP_LAUNCH(k_svc_star, dim3(cdiv(S, 256)), dim3(256), 0, st, proj, S, M, T, mu_l, mu_L, tl_star, Lstar);
}

// var[s, m'] = kss (Lstar Lstar^T)[m', m'] - colsq[s M + m'] + sigma2, clipped (prediction.py:975-983)
__global__ void k_svc_predvar(const double* __restrict__ Lstar, const double* __restrict__ colsq, int S, int M, int T,
                              const double* __restrict__ tse, double* __restrict__ var) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= S * M) return;
    int s = k / M, mp = k % M;
    double b = 0.0;
    for (int r = 0; r <= mp; ++r) {
        double v = Lstar[(size_t)s * T + mp * (mp + 1) / 2 + r];
        b += v * v;
    }
    const double kss = NMGP_JITTER + 1.0;                // Gibbs kernel of a point with itself, X2=None (prediction.py:976)
    double v = (kss * b - colsq[k]) + exp(tse[0]);
    if (v <= 0.0) v = NMGP_PRECISION;
    var[k] = v;
}
void svc_predvar(hipStream_t st, const double* Lstar, const double* colsq, int S, int M, const double* tse,
                 double* var) {
    int T = M * (M + 1) / 2;
    NMGP_LAUNCH(k_svc_predvar, dim3(cdiv((long long)S * M, 256)), dim3(256), 0, st, Lstar, colsq, S, M, T, tse,
                       var);
}

// cross-covariance vectors of the separable / stationary models, column-major [N, S]:
//   MODE 0 (separable, prediction.py:383-384): Gibbs with (sig_i, l_i) vs (exp(ts_star), exp(tl_star))
//   MODE 1 (stationary, prediction.py:1590,1626): RBF_cov(x, xs; alpha = sig0, beta = l0)
template <int MODE>
__global__ __launch_bounds__(256) void k_sep_crossvec(const double* __restrict__ x, const double* __restrict__ sig,
                                                       const double* __restrict__ ell, int N,
                                                       const double* __restrict__ xs,
                                                       const double* __restrict__ tl_star,
                                                       const double* __restrict__ ts_star, double sig0, double l0,
                                                       int S, double* __restrict__ KX) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int s = blockIdx.y;
    if (i >= N) return;
    double v;
    if (MODE == 0) {
        const double xi = x[i], li = ell[i], xj = xs[s];
        const double lj = exp(tl_star[s]), sj = exp(ts_star[s]);
        const double dist = (xi * xi + xj * xj) - 2.0 * (xi * xj);
        const double A = li * li + lj * lj;
        v = (sig[i] * sj) * sqrt(2.0 * (li * lj) / A) * exp(-dist / A);
    } else {
        const double xi = x[i] / l0, xj = xs[s] / l0;
        const double dist = (xi * xi + xj * xj) - 2.0 * (xi * xj);
        v = exp(-0.5 * dist) * (sig0 * sig0);
    }
    KX[(size_t)s * N + i] = v;
}
void sep_crossvec(hipStream_t st, int mode, const double* x, const double* sig, const double* ell, int N,
                  const double* xs, const double* tl_star, const double* ts_star, double sig0, double l0, int S,
                  double* KX) {
    dim3 grid(cdiv(N, 256), S);
    if (mode == 0)
        NMGP_LAUNCH((k_sep_crossvec<0>), grid, dim3(256), 0, st, x, sig, ell, N, xs, tl_star, ts_star, sig0, l0, S, KX);
    else
        NMGP_LAUNCH((k_sep_crossvec<1>), grid, dim3(256), 0, st, x, sig, ell, N, xs, tl_star, ts_star, sig0, l0, S, KX);
}

// Separable/stationary predictive moments in the joint eigenbasis (prediction.py:385-401):
//   Cq[q, s] = (V_K^T kx_s)[q]  (column-major [N, S]);  A_m[p, q] = wB[p] VB[m, p] Cq[q, s]
//   mean[s, m] = sum_pq A_m[p,q] b[p,q] w[p,q],   b = a (projection of y), w = 1/(sigma2 + wB[p] wK[q])
//   var[s, m]  = a2[s, m] - sum_pq A_m[p,q]^2 w[p,q] + sigma2, a2 = B[m,m] kss[s]
// One workgroup per (s, m).
__global__ __launch_bounds__(256) void k_sep_predict(const double* __restrict__ Cq, const double* __restrict__ a,
                                                      const double* __restrict__ wB, const double* __restrict__ VB,
                                                      int M, const double* __restrict__ wK, int N, double sigma2,
                                                      const double* __restrict__ Bdiag,
                                                      const double* __restrict__ kss, int strict_clip,
                                                      double* __restrict__ mean, double* __restrict__ var) {
    __shared__ double sh[16];
    const int s = blockIdx.x / M, m = blockIdx.x % M;
    double mu = 0.0, vv = 0.0;
    for (int q = threadIdx.x; q < N; q += blockDim.x) {
        const double c = Cq[(size_t)s * N + q];
        for (int p = 0; p < M; ++p) {
            const double Am = wB[p] * VB[m * M + p] * c;
            const double w = 1.0 / (sigma2 + wB[p] * wK[q]);
            mu += Am * (a[(size_t)p * N + q] * w);
            vv += (Am * w) * Am;
        }
    }
    mu = block_sum_e(mu, sh);
    vv = block_sum_e(vv, sh);
    if (threadIdx.x == 0) {
        double v = (Bdiag[m] * kss[s] - vv) + sigma2;
        if (strict_clip ? (v < 0.0) : (v <= 0.0)) v = NMGP_PRECISION;
        mean[s * M + m] = mu;
        var[s * M + m] = v;
    }
}
void sep_predict(hipStream_t st, const double* Cq, const double* a, const double* wB, const double* VB, int M,
                 const double* wK, int N, double sigma2, const double* Bdiag, const double* kss, bool strict_clip, int S,
                 double* mean, double* var) {
    NMGP_LAUNCH(k_sep_predict, dim3(S * M), dim3(256), 0, st, Cq, a, wB, VB, M, wK, N, sigma2, Bdiag, kss,
                       strict_clip ? 1 : 0, mean, var);
}

// The Cholesky formulation of the same moments (B kron K + sigma2 I = (V_B kron I) blockdiag_p(S_p) (V_B^T kron I), S_p = wB[p] K +
// sigma2 I): the cross-covariance vectors kx_s ride every block's factorisation as extra rows, so that after it
//   dots[p, s] = (L_p^-1 kx_s) . (L_p^-1 yt_p) = kx_s^T S_p^-1 yt_p,    sqs[p, s] = |L_p^-1 kx_s|^2 = kx_s^T S_p^-1 kx_s, and
//   mean[s, m] = sum_p wB[p] VB[m, p] dots[p, s],   var[s, m] = B[m, m] kss[s] - sum_p (wB[p] VB[m, p])^2 sqs[p, s] + sigma2.
__global__ void k_sep_predict_chol(const double* __restrict__ dots, const double* __restrict__ sqs, const double* __restrict__ wB,
                                   const double* __restrict__ VB, int M, double sigma2, const double* __restrict__ Bdiag,
                                   const double* __restrict__ kss, int strict_clip, int S, double* __restrict__ mean,
                                   double* __restrict__ var) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= S * M) return;
    const int s = k / M, m = k - s * M;
    double mu = 0.0, vv = 0.0;
    for (int p = 0; p < M; ++p) {
        const double am = wB[p] * VB[m * M + p];
        mu += am * dots[(size_t)p * S + s];
        vv += (am * am) * sqs[(size_t)p * S + s];
    }
    double v = (Bdiag[m] * kss[s] - vv) + sigma2;
    if (strict_clip ? (v < 0.0) : (v <= 0.0)) v = NMGP_PRECISION;
    mean[k] = mu;
    var[k] = v;
}
void sep_predict_chol(hipStream_t st, const double* dots, const double* sqs, const double* wB, const double* VB, int M, double sigma2,
                      const double* Bdiag, const double* kss, bool strict_clip, int S, double* mean, double* var) {
    NMGP_LAUNCH(k_sep_predict_chol, dim3(cdiv((long long)S * M, 256)), dim3(256), 0, st, dots, sqs, wB, VB, M, sigma2, Bdiag, kss,
                strict_clip ? 1 : 0, S, mean, var);
}

// rows R0 .. R0 + S - 1 of every matrix of a batch := the columns of KX ([N, S] column-major): A[b][i * ld + R0 + s] = KX[s N + i]
// (the same S vectors below each of the M blocks); lanes along s
__global__ __launch_bounds__(256) void k_cols_to_rows(const double* __restrict__ KX, int N, int S, double* __restrict__ A, int ld,
                                                       int R0, long long bstride) {
    const int s = blockIdx.y * 256 + threadIdx.x;
    const int i = blockIdx.x;
    if (s >= S) return;
    A[(size_t)blockIdx.z * bstride + (size_t)i * ld + R0 + s] = KX[(size_t)s * N + i];
}
void cols_to_rows(hipStream_t st, const double* KX, int N, int S, double* A, int ld, int R0, int batch, long long bstride) {
    NMGP_LAUNCH(k_cols_to_rows, dim3(N, cdiv(S, 256), batch), dim3(256), 0, st, KX, N, S, A, ld, R0, bstride);
}

// starred scalars of the separable model: tl_star = mu_l + proj[:,0], ts_star = mu_s + proj[:,1],
// kss[s] = exp(ts_star)^2 + jitter (Gibbs kernel of the new point with itself, prediction.py:393-397)
__global__ void k_sep_star(const double* __restrict__ proj, int S, double mu_l, double mu_s,
                           double* __restrict__ tl_star, double* __restrict__ ts_star, double* __restrict__ kss) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const double tl = mu_l + proj[s], ts = mu_s + proj[S + s];
    tl_star[s] = tl;
    ts_star[s] = ts;
    const double sg = exp(ts);
    kss[s] = NMGP_JITTER + (sg * sg);
}
void sep_star(hipStream_t st, const double* proj, int S, double mu_l, double mu_s, double* tl_star, double* ts_star,
              double* kss) {
    NMGP_LAUNCH(k_sep_star, dim3(cdiv(S, 256)), dim3(256), 0, st, proj, S, mu_l, mu_s, tl_star, ts_star, kss);
}


// ---------------------------------------------------------------------------------------------
// Separable likelihood without an eigendecomposition of K_x: with B = V_B diag(wB) V_B^T (M x M, host),
//   B kron K + sigma2 I = (V_B kron I) blockdiag_p( wB[p] K + sigma2 I ) (V_B^T kron I),
// so the MN x MN problem is M independent N x N Cholesky factorisations (one batch of the blocked factorisation).
// ---------------------------------------------------------------------------------------------
// yt[p N + i] = sum_m Y[i, m] VB[m, p]      (Y row-major [N, M], VB row-major [M, M])
__global__ void k_rotate_y(const double* __restrict__ Y, const double* __restrict__ VB, int N, int M,
                           double* __restrict__ yt) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int p = blockIdx.y;
    if (i >= N) return;
    double s = 0.0;
    for (int m = 0; m < M; ++m) s += Y[(size_t)i * M + m] * VB[m * M + p];
    yt[(size_t)p * N + i] = s;
}
void rotate_y(hipStream_t s, const double* Y, const double* VB, int N, int M, double* yt) {
    NMGP_LAUNCH(k_rotate_y, dim3(cdiv(N, 256), M), dim3(256), 0, s, Y, VB, N, M, yt);
}

// out_p[i, j] = wB[p] K[i, j] + sigma2 d_ij   (lower triangle, column-major; K: ld = N, out: leading dimension ldo)
__global__ __launch_bounds__(256) void k_sep_blocks(const double* __restrict__ K, const double* __restrict__ wB,
                                                     const double* __restrict__ sigma2p, int N,
                                                     double* __restrict__ out, int ldo, long long bstride) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= N || i < j) return;
    const double w = wB[blockIdx.z];
    double v = w * K[(size_t)j * N + i];
    if (i == j) v += sigma2p[0];
    out[(size_t)blockIdx.z * bstride + (size_t)j * ldo + i] = v;
}
void sep_blocks(hipStream_t s, const double* K, const double* wB, const double* sigma2p, int N, int M, double* out,
                int ldo, long long bstride) {
    NMGP_LAUNCH(k_sep_blocks, dim3(cdiv(N, 256), N, M), dim3(256), 0, s, K, wB, sigma2p, N, out, ldo, bstride);
}

// per block p (Cneg_p = -S_p^-1, lower, ld = N) partial sums over the columns j = blockIdx.y, +gridDim.y, ...:
//   out[(p G + g) 3 + 0] += tr S_p^-1,  [.. + 1] += <S_p^-1, K> (full symmetric inner product from the lower triangles),
//   [.. + 2] += ||alpha_p||^2 (chunk g = 0 only).  The host adds the G partials (deterministic order).
#define SEP_TR_G NMGP_SEP_TR_G
__global__ __launch_bounds__(256) void k_sep_traces(const double* __restrict__ Cneg, const double* __restrict__ K,
                                                     const double* __restrict__ alpha, int N,
                                                     double* __restrict__ out) {
    __shared__ double sh[16];
    const int p = blockIdx.x, g = blockIdx.y;
    const double* C = Cneg + (size_t)p * N * N;
    double tr = 0.0, tk = 0.0, aa = 0.0;
    for (int j = g; j < N; j += SEP_TR_G) {
        for (int i = j + threadIdx.x; i < N; i += blockDim.x) {
            const double c = -C[(size_t)j * N + i];
            const double k = K[(size_t)j * N + i];
            if (i == j) {
                tr += c;
                tk += c * k;
            } else {
                tk += 2.0 * c * k;
            }
        }
    }
    if (g == 0)
        for (int i = threadIdx.x; i < N; i += blockDim.x) aa += alpha[(size_t)p * N + i] * alpha[(size_t)p * N + i];
    tr = block_sum_e(tr, sh);
    tk = block_sum_e(tk, sh);
    aa = block_sum_e(aa, sh);
    if (threadIdx.x == 0) {
        double* o = out + ((size_t)p * SEP_TR_G + g) * 3;
        o[0] = tr;
        o[1] = tk;
        o[2] = aa;
    }
}
int sep_traces(hipStream_t s, const double* Cneg, const double* K, const double* alpha, int N, int M, double* out) {
    NMGP_LAUNCH(k_sep_traces, dim3(M, SEP_TR_G), dim3(256), 0, s, Cneg, K, alpha, N, out);
    return SEP_TR_G;
}

// C[i, j] = - sum_p wB[p] Cneg_p[i, j]  (lower triangle; = sum_p wB[p] S_p^-1)
__global__ __launch_bounds__(256) void k_weighted_sum_lower(const double* __restrict__ Cneg,
                                                             const double* __restrict__ wB, int N, int M,
                                                             double* __restrict__ C) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= N || i < j) return;
    double s = 0.0;
    for (int p = 0; p < M; ++p) s += wB[p] * Cneg[(size_t)p * N * N + (size_t)j * N + i];
    C[(size_t)j * N + i] = -s;
}
void weighted_sum_lower(hipStream_t s, const double* Cneg, const double* wB, int N, int M, double* C) {
    NMGP_LAUNCH(k_weighted_sum_lower, dim3(cdiv(N, 256), N), dim3(256), 0, s, Cneg, wB, N, M, C);
}

// A[r, r] += v  (n x n, leading dimension ld)
__global__ void k_add_diag(double* __restrict__ A, int ld, int n, double v) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) A[(size_t)r * ld + r] += v;
}
void add_diag(hipStream_t s, double* A, int ld, int n, double v) {
    NMGP_LAUNCH(k_add_diag, dim3(cdiv(n, 256)), dim3(256), 0, s, A, ld, n, v);
}

}  // namespace nmgpk
