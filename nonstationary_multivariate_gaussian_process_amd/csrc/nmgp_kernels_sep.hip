// Batched pieces of the SEPARABLE objective for B chains per launch sequence (nmgp_sep_batch_eval; logpos.py:237-296,
// distributions.py:26-52; caller: Separable_model.py:160-166, :209-210).  The factorisation of the chains' B*M blocks was batched in
// round 4; these kernels take the CHAIN as a grid dimension for everything around it, so that the covariance build, the block
// assembly, the trace / weighted-sum pass over the M blocks of -S_p^-1 and the per-location adjoint are one launch each for the
// whole batch instead of 5 + 7 launches per chain:
//   k_sep_prep_b     ell = exp(tilde_l), sig = exp(tilde_sigma), yt_p = (V_B^T kron I) y        -- all chains
//   k_sep_blocks_b4  S_bp = wB_b[p] K_x,b + sigma2_b I written STRAIGHT from (x, ell_b, sig_b): K_x,b itself is stored only when the
//   (k_sep_blocks_b) gradient needs it (value path: M N^2/2 doubles per chain instead of (2M + 1) N^2/2 written + M N^2/2 read);
//                    tiles dealt to the XCDs column by column, so that each L2 hands HBM long runs of a column
//   k_sep_reduce_b   ONE pass over the M blocks -S_bp^-1 and K_x: tr S_p^-1, <S_p^-1, K_x>, |alpha_p|^2, C_b = sum_p wB[p] S_p^-1 and
//                    the M x M quadratic forms alpha_p^T K_x alpha_q (each block is read once; round 4 read the five blocks
//                    twice, in two kernels per chain, and K_x once more in a library dsymm)
//   k_sep_adjoint_b  the fused per-location adjoint of nmgp_kernels_eig.hip with the chain as blockIdx.z
// Arithmetic per element is the single-chain kernels' (k_cov_sym<GIBBS>, k_sep_blocks, k_sep_adjoint): same expressions, same order.
#include "nmgp_internal.h"

namespace nmgpk {

static inline int cdiv_s(long long a, long long b) { return (int)((a + b - 1) / b); }

__device__ inline double wave_sum_s(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}
__device__ inline double block_sum_s(double v, double* sh /*[16]*/) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = wave_sum_s(v);
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    const int nw = (blockDim.x + 63) >> 6;
    v = (threadIdx.x < nw) ? sh[threadIdx.x] : 0.0;
    if (w == 0) v = wave_sum_s(v);
    return v;
}

// pars [B, P] (P = 2N + T + 1): ell[b, i] = exp(pars[b, i]), sig[b, i] = exp(pars[b, N + i]);
// yt[(b M + p) N + i] = sum_m Y[i, m] VB_b[m, p]   (small[b]: wB [M] | VB row-major [M, M] | sigma2 | pad)
__global__ __launch_bounds__(256) void k_sep_prep_b(const double* __restrict__ pars, long long P, const double* __restrict__ Y,
                                                     const double* __restrict__ small, int small_per, int N, int M,
                                                     double* __restrict__ ell, double* __restrict__ sig, double* __restrict__ yt) {
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= N) return;
    const double* pb = pars + (size_t)b * P;
    ell[(size_t)b * N + i] = exp(pb[i]);
    sig[(size_t)b * N + i] = exp(pb[N + i]);
    const double* VB = small + (size_t)b * small_per + M;
    for (int p = 0; p < M; ++p) {
        double s = 0.0;
        for (int m = 0; m < M; ++m) s += Y[(size_t)i * M + m] * VB[m * M + p];
        yt[((size_t)b * M + p) * N + i] = s;
    }
}

void sep_prep_b(hipStream_t s, const double* pars, long long P, const double* Y, const double* small, int small_per, int N, int M,
                double* ell, double* sig, double* yt, int B) {
    NMGP_LAUNCH(k_sep_prep_b, dim3(cdiv_s(N, 256), B), dim3(256), 0, s, pars, P, Y, small, small_per, N, M, ell, sig, yt);
}

// Lower triangles of the M blocks of chain b = blockIdx.z, and of K_x,b itself when Kout != nullptr (kernels.py:46-73 with the
// jitter on the diagonal, then logpos.py:258-262's B kron K + sigma2 I in B's eigenbasis).  64 x 64 location tile per workgroup,
// j-side inputs in LDS, lanes along i: every store instruction writes 512 contiguous bytes of a column.  M is a template parameter
// so that the block weights live in registers and the M stores of an element are straight-line code.
template <int M>
__global__ __launch_bounds__(256) void k_sep_blocks_b(const double* __restrict__ x, const double* __restrict__ ell,
                                                       const double* __restrict__ sig, const double* __restrict__ small,
                                                       int small_per, int N, double* __restrict__ S, int ldo, long long bstride,
                                                       double* __restrict__ Kout, int remap) {
    constexpr int TJ = 64;
    __shared__ double sx[TJ], sl[TJ], ss[TJ];
    int I = blockIdx.x, J = blockIdx.y;
    const int b = blockIdx.z;
    if (remap) {
        // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so workgroup g of the launch
        // takes tile t = (g mod 8) * per + g / 8 of the lower-triangular tile list, enumerated column by column (J, then I = J ..):
        // the tiles ONE XCD works on at a time are vertical neighbours -- the adjacent 512-byte segments of the same 64 columns --
        // and its L2 hands HBM runs of several KB per column instead of eight L2s handing over 512 bytes each.
        const int NI = (N + 63) / 64, ntl = NI * (NI + 1) / 2, per = (ntl + 7) / 8;
        const int g = blockIdx.x, t = (g & 7) * per + (g >> 3);
        if ((g >> 3) >= per || t >= ntl) return;
        int rem = t;
        J = 0;
        while (rem >= NI - J) {
            rem -= NI - J;
            ++J;
        }
        I = J + rem;
    }
    if (I < J) return;
    const double* eb = ell + (size_t)b * N;
    const double* sb = sig + (size_t)b * N;
    const double* sm = small + (size_t)b * small_per;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j0 = J * TJ;
    if (tid < TJ) {
        const int j = j0 + tid;
        sx[tid] = (j < N) ? x[j] : 0.0;
        sl[tid] = (j < N) ? eb[j] : 1.0;
        ss[tid] = (j < N) ? sb[j] : 1.0;
    }
    __syncthreads();
    const int i = I * 64 + lane;
    if (i >= N) return;
    double wB[M];
#pragma unroll
    for (int p = 0; p < M; ++p) wB[p] = sm[p];
    const double sigma2 = sm[M + M * M];
    const double xi = x[i], li = eb[i], si = sb[i];
    const double xi2 = xi * xi, li2 = li * li;
    // Buffer stores: one uniform descriptor per block (scalar registers) + ONE 32-bit byte offset per element, so that the address
    // arithmetic of the M + 1 stores stays off the vector ALU -- which this kernel needs for the exp / sqrt / divisions (M = 5: one
    // transcendental chain per 40 bytes stored; with flat 64-bit addresses a quarter of the loop's vector instructions were adds)
    typedef int v2i_t __attribute__((ext_vector_type(2)));
    union D2 {
        double d;
        v2i_t v;
    };
    double* Sb = S + (size_t)b * M * bstride;
    __amdgpu_buffer_rsrc_t rs[M];
#pragma unroll
    for (int p = 0; p < M; ++p) rs[p] = __builtin_amdgcn_make_buffer_rsrc((void*)(Sb + (size_t)p * bstride), 0, 0x7fffffff, 0x00020000);
    const bool wantK = Kout != nullptr;
    const __amdgpu_buffer_rsrc_t rk =
        __builtin_amdgcn_make_buffer_rsrc((void*)(wantK ? Kout + (size_t)b * N * N : Sb), 0, 0x7fffffff, 0x00020000);
    for (int jj = 0; jj < TJ / 4; ++jj) {
        const int k = w * (TJ / 4) + jj;
        const int j = j0 + k;
        if (j >= N) break;
        if (i < j) continue;
        const double xj = sx[k], lj = sl[k];
        const double dist = (xi2 + xj * xj) - 2.0 * (xi * xj);
        const double A = li2 + lj * lj;
        double v = (si * ss[k]) * sqrt(2.0 * (li * lj) / A) * exp(-dist / A);   // kernels.py:69-72
        if (i == j) v = NMGP_JITTER + v;
        D2 u;
        if (wantK) {
            u.d = v;
            __builtin_amdgcn_raw_buffer_store_b64(u.v, rk, (j * N + i) * 8, 0, 0);
        }
        const int off = (j * ldo + i) * 8;                                       // bytes within a block: < 2^31 up to n = 16,383
        const double dg = (i == j) ? sigma2 : 0.0;
#pragma unroll
        for (int p = 0; p < M; ++p) {
            u.d = wB[p] * v + dg;
            __builtin_amdgcn_raw_buffer_store_b64(u.v, rs[p], off, 0, 0);
        }
    }
}

// The same blocks from 128 x 32 location tiles (the default; NMGP_SEP_BLOCKS=4): a lane owns the row pair (i, i + 1), so every store instruction
// of a wave writes ONE KILOBYTE of a column (16 bytes per lane), in the XCD-aware tile order of the remapped kernel above (a column
// of tiles per XCD at a time: its L2 hands HBM runs of many KB).  A wave takes 8 of the tile's 32 columns.  Even N only (16-byte
// alignment of K_x's columns); the launcher falls back to the 64 x 64 kernel otherwise.  Row pairs that straddle the diagonal are
// written whole (the element above the diagonal is scratch to every consumer: DESIGN section 2).
template <int M>
__global__ __launch_bounds__(256) void k_sep_blocks_b4(const double* __restrict__ x, const double* __restrict__ ell,
                                                        const double* __restrict__ sig, const double* __restrict__ small,
                                                        int small_per, int N, double* __restrict__ S, int ldo, long long bstride,
                                                        double* __restrict__ Kout) {
    constexpr int TJ = 32, TI = 128;
    __shared__ double sx[TJ], sl[TJ], ss[TJ];
    const int b = blockIdx.z;
    const int NI = (N + TI - 1) / TI, NJ = (N + TJ - 1) / TJ;
    // lower-triangular tile list, column by column: column J holds the row tiles I = J / 4 .. NI - 1
    int ntl = 0;
    for (int J = 0; J < NJ; ++J) ntl += NI - J / 4;
    const int per = (ntl + 7) / 8;
    const int g = blockIdx.x, t = (g & 7) * per + (g >> 3);
    if ((g >> 3) >= per || t >= ntl) return;
    int rem = t, J = 0;
    while (rem >= NI - J / 4) {
        rem -= NI - J / 4;
        ++J;
    }
    const int I = J / 4 + rem;
    const double* eb = ell + (size_t)b * N;
    const double* sb = sig + (size_t)b * N;
    const double* sm = small + (size_t)b * small_per;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j0 = J * TJ;
    if (tid < TJ) {
        const int j = j0 + tid;
        sx[tid] = (j < N) ? x[j] : 0.0;
        sl[tid] = (j < N) ? eb[j] : 1.0;
        ss[tid] = (j < N) ? sb[j] : 1.0;
    }
    __syncthreads();
    const int i = I * TI + 2 * lane;                   // rows i, i + 1 (N even: both valid or both out)
    if (i >= N) return;
    double wB[M];
#pragma unroll
    for (int p = 0; p < M; ++p) wB[p] = sm[p];
    const double sigma2 = sm[M + M * M];
    const double xa = x[i], la = eb[i], sa = sb[i], xb = x[i + 1], lb = eb[i + 1], sbb = sb[i + 1];
    const double xa2 = xa * xa, la2 = la * la, xb2 = xb * xb, lb2 = lb * lb;
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    union D4 {
        double d[2];
        v4i_t v;
    };
    double* Sb = S + (size_t)b * M * bstride;
    __amdgpu_buffer_rsrc_t rs[M];
#pragma unroll
    for (int p = 0; p < M; ++p) rs[p] = __builtin_amdgcn_make_buffer_rsrc((void*)(Sb + (size_t)p * bstride), 0, 0x7fffffff, 0x00020000);
    const bool wantK = Kout != nullptr;
    const __amdgpu_buffer_rsrc_t rk =
        __builtin_amdgcn_make_buffer_rsrc((void*)(wantK ? Kout + (size_t)b * N * N : Sb), 0, 0x7fffffff, 0x00020000);
    for (int jj = 0; jj < TJ / 4; ++jj) {
        const int k = w * (TJ / 4) + jj;
        const int j = j0 + k;
        if (j >= N) break;
        if (i + 1 < j) continue;                       // the whole pair lies above the diagonal
        const double xj = sx[k], lj = sl[k], sj = ss[k], xj2 = xj * xj, lj2 = lj * lj;
        const double da = (xa2 + xj2) - 2.0 * (xa * xj), Aa = la2 + lj2;
        const double db = (xb2 + xj2) - 2.0 * (xb * xj), Ab = lb2 + lj2;
        double va = (sa * sj) * sqrt(2.0 * (la * lj) / Aa) * exp(-da / Aa);       // kernels.py:69-72
        double vb = (sbb * sj) * sqrt(2.0 * (lb * lj) / Ab) * exp(-db / Ab);
        if (i == j) va = NMGP_JITTER + va;
        if (i + 1 == j) vb = NMGP_JITTER + vb;
        D4 u;
        if (wantK) {
            u.d[0] = va;
            u.d[1] = vb;
            __builtin_amdgcn_raw_buffer_store_b128(u.v, rk, (j * N + i) * 8, 0, 0);
        }
        const int off = (j * ldo + i) * 8;
        const double ga = (i == j) ? sigma2 : 0.0, gb = (i + 1 == j) ? sigma2 : 0.0;
#pragma unroll
        for (int p = 0; p < M; ++p) {
            u.d[0] = wB[p] * va + ga;
            u.d[1] = wB[p] * vb + gb;
            __builtin_amdgcn_raw_buffer_store_b128(u.v, rs[p], off, 0, 0);
        }
    }
}

template <int M>
static void launch_sep_blocks_b(hipStream_t s, const double* x, const double* ell, const double* sig, const double* small, int small_per,
                                int N, double* S, int ldo, long long bstride, double* Kout, int B) {
    // NMGP_SEP_BLOCKS: 4 (default) 128 x 32 tiles, 1 KB stores, XCD-aware order; 3: 64 x 64 tiles in the XCD-aware order (also the
    // fallback for odd N); 1: 64 x 64 tiles in grid order (round 5's first form, kept for the A/B: 2.43 / 1.69 / 1.49 ms for 16 chains
    // of N = 4096, D = 5 -- 1 / 3 / 4)
    static const int variant = [] { const char* e = std::getenv("NMGP_SEP_BLOCKS"); return e ? std::atoi(e) : 4; }();
    if (variant == 4 && (N & 1) == 0) {
        const int NI = cdiv_s(N, 128), NJ = cdiv_s(N, 32);
        int ntl = 0;
        for (int J = 0; J < NJ; ++J) ntl += NI - J / 4;
        NMGP_LAUNCH((k_sep_blocks_b4<M>), dim3(8 * ((ntl + 7) / 8), 1, B), dim3(256), 0, s, x, ell, sig, small, small_per, N, S, ldo, bstride,
                    Kout);
    } else if (variant == 3 || variant == 4) {
        const int NI = cdiv_s(N, 64), ntl = NI * (NI + 1) / 2, per = (ntl + 7) / 8;
        NMGP_LAUNCH((k_sep_blocks_b<M>), dim3(8 * per, 1, B), dim3(256), 0, s, x, ell, sig, small, small_per, N, S, ldo, bstride, Kout, 1);
    } else {
        NMGP_LAUNCH((k_sep_blocks_b<M>), dim3(cdiv_s(N, 64), cdiv_s(N, 64), B), dim3(256), 0, s, x, ell, sig, small, small_per, N, S, ldo,
                    bstride, Kout, 0);
    }
}

void sep_blocks_b(hipStream_t s, const double* x, const double* ell, const double* sig, const double* small, int small_per, int N, int M,
                  double* S, int ldo, long long bstride, double* Kout, int B) {
    switch (M) {
        case 1: launch_sep_blocks_b<1>(s, x, ell, sig, small, small_per, N, S, ldo, bstride, Kout, B); break;
        case 2: launch_sep_blocks_b<2>(s, x, ell, sig, small, small_per, N, S, ldo, bstride, Kout, B); break;
        case 3: launch_sep_blocks_b<3>(s, x, ell, sig, small, small_per, N, S, ldo, bstride, Kout, B); break;
        case 4: launch_sep_blocks_b<4>(s, x, ell, sig, small, small_per, N, S, ldo, bstride, Kout, B); break;
        case 5: launch_sep_blocks_b<5>(s, x, ell, sig, small, small_per, N, S, ldo, bstride, Kout, B); break;
        case 6: launch_sep_blocks_b<6>(s, x, ell, sig, small, small_per, N, S, ldo, bstride, Kout, B); break;
        case 7: launch_sep_blocks_b<7>(s, x, ell, sig, small, small_per, N, S, ldo, bstride, Kout, B); break;
        default: launch_sep_blocks_b<8>(s, x, ell, sig, small, small_per, N, S, ldo, bstride, Kout, B); break;      // (M <= NMGP_MAX_OUTPUTS)
    }
}

// ONE pass over the M blocks Cneg_bp = -S_bp^-1 (lower, ld = N) of chain b = blockIdx.y AND over K_x,b; workgroup g = blockIdx.x takes
// the columns j = g, g + G, ...:
//   out[((b M + p) G + g) 3 + {0, 1, 2}] = partial tr S_p^-1, <S_p^-1, K> (full symmetric inner product from the lower triangles),
//                                          |alpha_p|^2 (g = 0 only)
//   xi[(b G + g) M(M+1)/2 + idx(p, q)]   = partial alpha_p^T K alpha_q for p <= q (idx row-major over the upper triangle): the M x M
//                                          quadratic forms of d loglik / dB, which round 5's first form left to a library dsymm + dgemm
//                                          per chain (2.9 ms of a 16-chain gradient step, 256 small launches)
//   C_b[i, j] = sum_p wB[p] S_p^-1[i, j] on the lower triangle.
// The host adds the G partials in a fixed order.
template <int M>
__global__ __launch_bounds__(256) void k_sep_reduce_b(const double* __restrict__ Cneg, const double* __restrict__ K,
                                                       const double* __restrict__ alpha, const double* __restrict__ small,
                                                       int small_per, int N, int G, double* __restrict__ C,
                                                       double* __restrict__ out, double* __restrict__ xi) {
    constexpr int NX = M * (M + 1) / 2;
    __shared__ double sh[16];
    const int g = blockIdx.x, b = blockIdx.y;
    const size_t NN = (size_t)N * N;
    const double* Cb = Cneg + (size_t)b * M * NN;
    const double* Kb = K + (size_t)b * NN;
    const double* Ab = alpha + (size_t)b * M * N;
    double* Co = C + (size_t)b * NN;
    double wB[M];
#pragma unroll
    for (int p = 0; p < M; ++p) wB[p] = small[(size_t)b * small_per + p];
    double tr[M], tk[M], X[NX];
#pragma unroll
    for (int p = 0; p < M; ++p) tr[p] = tk[p] = 0.0;
#pragma unroll
    for (int e = 0; e < NX; ++e) X[e] = 0.0;
    for (int j = g; j < N; j += G) {
        double aj[M];
#pragma unroll
        for (int p = 0; p < M; ++p) aj[p] = Ab[(size_t)p * N + j];
        for (int i = j + threadIdx.x; i < N; i += 256) {
            const size_t o = (size_t)j * N + i;
            const double k = Kb[o];
            double ai[M];
#pragma unroll
            for (int p = 0; p < M; ++p) ai[p] = Ab[(size_t)p * N + i];
            double cs = 0.0;
#pragma unroll
            for (int p = 0; p < M; ++p) {
                const double c = -Cb[(size_t)p * NN + o];
                cs += wB[p] * c;
                if (i == j) {
                    tr[p] += c;
                    tk[p] += c * k;
                } else {
                    tk[p] += 2.0 * c * k;
                }
            }
            Co[o] = cs;
            // alpha_p^T K alpha_q: the element (i, j) and its mirror (j, i)
            int e = 0;
#pragma unroll
            for (int p = 0; p < M; ++p) {
#pragma unroll
                for (int q = p; q < M; ++q) {
                    const double t = (i == j) ? ai[p] * ai[q] : ai[p] * aj[q] + aj[p] * ai[q];
                    X[e] = fma(k, t, X[e]);
                    ++e;
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < M; ++p) {
        double aa = 0.0;
        if (g == 0) {
            const double* ap = Ab + (size_t)p * N;
            for (int i = threadIdx.x; i < N; i += 256) aa += ap[i] * ap[i];
        }
        const double trp = block_sum_s(tr[p], sh), tkp = block_sum_s(tk[p], sh);
        aa = block_sum_s(aa, sh);
        if (threadIdx.x == 0) {
            double* o = out + (((size_t)b * M + p) * G + g) * 3;
            o[0] = trp;
            o[1] = tkp;
            o[2] = aa;
        }
    }
#pragma unroll
    for (int e = 0; e < NX; ++e) {
        const double v = block_sum_s(X[e], sh);
        if (threadIdx.x == 0) xi[((size_t)b * G + g) * NX + e] = v;
    }
}

template <int M>
static void launch_sep_reduce_b(hipStream_t s, const double* Cneg, const double* K, const double* alpha, const double* small, int small_per,
                                int N, int G, double* C, double* out, double* xi, int B) {
    NMGP_LAUNCH((k_sep_reduce_b<M>), dim3(G, B), dim3(256), 0, s, Cneg, K, alpha, small, small_per, N, G, C, out, xi);
}

// xi: B * G * M (M + 1) / 2 doubles
void sep_reduce_b(hipStream_t s, const double* Cneg, const double* K, const double* alpha, const double* small, int small_per, int N,
                  int M, int G, double* C, double* out, double* xi, int B) {
    switch (M) {
        case 1: launch_sep_reduce_b<1>(s, Cneg, K, alpha, small, small_per, N, G, C, out, xi, B); break;
        case 2: launch_sep_reduce_b<2>(s, Cneg, K, alpha, small, small_per, N, G, C, out, xi, B); break;
        case 3: launch_sep_reduce_b<3>(s, Cneg, K, alpha, small, small_per, N, G, C, out, xi, B); break;
        case 4: launch_sep_reduce_b<4>(s, Cneg, K, alpha, small, small_per, N, G, C, out, xi, B); break;
        case 5: launch_sep_reduce_b<5>(s, Cneg, K, alpha, small, small_per, N, G, C, out, xi, B); break;
        case 6: launch_sep_reduce_b<6>(s, Cneg, K, alpha, small, small_per, N, G, C, out, xi, B); break;
        case 7: launch_sep_reduce_b<7>(s, Cneg, K, alpha, small, small_per, N, G, C, out, xi, B); break;
        default: launch_sep_reduce_b<8>(s, Cneg, K, alpha, small, small_per, N, G, C, out, xi, B); break;
    }
}

// k_sep_adjoint (nmgp_kernels_eig.hip) with the chain as blockIdx.z:
//   dK_ij = 1/2 ( sum_p wB[p] U[i,p] U[j,p] - C[i,j] ),   Ks_ij = s_i s_j K0_ij
//   g_tl[i] = sum_{j != i} 2 dK_ij Ks_ij (1/2 - l_i^2/A + 2 l_i^2 d_ij / A^2),   g_ts[i] = 2 sum_j dK_ij Ks_ij
// U = alpha of chain b ([M, N]), C of chain b full symmetric; partials part[b][J][i][2].
__global__ __launch_bounds__(256) void k_sep_adjoint_b(const double* __restrict__ x, const double* __restrict__ ell,
                                                        const double* __restrict__ sig, const double* __restrict__ U,
                                                        const double* __restrict__ small, int small_per, int M,
                                                        const double* __restrict__ C, int N, double* __restrict__ part, int NJ) {
    constexpr int TJ = 64;
    __shared__ double sx[TJ], sl[TJ], ss[TJ], sU[TJ * NMGP_MAX_OUTPUTS];
    __shared__ double red[2][4][64];
    const int I = blockIdx.x, J = blockIdx.y, b = blockIdx.z;
    ell += (size_t)b * N;
    sig += (size_t)b * N;
    U += (size_t)b * M * N;
    C += (size_t)b * N * N;
    part += (size_t)b * NJ * N * 2;
    const double* wB = small + (size_t)b * small_per;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j0 = J * TJ;
    if (tid < TJ) {
        int j = j0 + tid;
        sx[tid] = (j < N) ? x[j] : 0.0;
        sl[tid] = (j < N) ? ell[j] : 1.0;
        ss[tid] = (j < N) ? sig[j] : 1.0;
    }
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

// g[b][i] = sum_J part[b][J][i][0], g[b][N + i] = ... [1]
__global__ void k_sep_grad_sum_b(const double* __restrict__ part, int NJ, int N, double* __restrict__ g) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= N) return;
    part += (size_t)b * NJ * N * 2;
    double a = 0.0, c = 0.0;
    for (int J = 0; J < NJ; ++J) {
        a += part[((size_t)J * N + i) * 2];
        c += part[((size_t)J * N + i) * 2 + 1];
    }
    g[(size_t)b * 2 * N + i] = a;
    g[(size_t)b * 2 * N + N + i] = c;
}

void sep_adjoint_b(hipStream_t s, const double* x, const double* ell, const double* sig, const double* U, const double* small,
                   int small_per, int M, const double* C, int N, double* part, double* g, int B) {
    const int NJ = cdiv_s(N, 64);
    NMGP_LAUNCH(k_sep_adjoint_b, dim3(cdiv_s(N, 64), NJ, B), dim3(256), 0, s, x, ell, sig, U, small, small_per, M, C, N, part, NJ);
    NMGP_LAUNCH(k_sep_grad_sum_b, dim3(cdiv_s(N, 256), B), dim3(256), 0, s, part, NJ, N, g);
}

// R[(2 b + 0) N + i] = pars[b, i] - mu_a,  R[(2 b + 1) N + i] = pars[b, N + i] - mu_b: the two prior right-hand sides of every chain
__global__ void k_two_col_rhs_b(const double* __restrict__ pars, long long P, double mu_a, double mu_b, int N, double* __restrict__ R) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= N) return;
    R[((size_t)2 * b) * N + i] = pars[(size_t)b * P + i] - mu_a;
    R[((size_t)2 * b + 1) * N + i] = pars[(size_t)b * P + N + i] - mu_b;
}
void two_col_rhs_b(hipStream_t s, const double* pars, long long P, double mu_a, double mu_b, int N, double* R, int B) {
    NMGP_LAUNCH(k_two_col_rhs_b, dim3(cdiv_s(N, 256), B), dim3(256), 0, s, pars, P, mu_a, mu_b, N, R);
}

}  // namespace nmgpk
