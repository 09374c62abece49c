// Kernels of the PRIOR-FACTOR METRIC of the device-resident HMC trajectories (nmgp_svc_batch_traj_set_mass_prior).
//
// The posterior of the nonseparable model (logpos.py:299-380) is dominated by its GP priors RBF(alpha, beta) + 1e-6 I on tilde_l
// and on each of the T columns of uL_vecs (logpos.py:357-365; condition number ~1e11 at N = 2048), so no diagonal mass matrix
// preconditions it.  In the coordinates w with  pars = mu + L_blk w,  L_blk = blockdiag(chol Sigma_l, chol Sigma_L per stride-T
// column of uL_vecs, 1 for tilde_sigma2_err),  the prior is N(0, I) and the likelihood adds curvature in a few dozen smooth
// directions only: Hessian = I + U diag(lam) U^T with U [P, r] orthonormal.  The sampler's constant mass matrix is therefore
//      M^-1 = L_blk (I + U diag(lam) U^T)^-1 L_blk^T          (the reference passes M = inv(sample covariance) instead,
//                                                              Nonseparable_model_mpiKAISER.py:398-411)
// and the trajectory carries the WHITENED momentum u = L_blk^T p, for which every step is a triangular MAT-VEC with the cached
// prior factors (never a solve) plus a rank-r correction:
//      draw     u = z + U ((sqrt(1 + lam) - 1) o U^T z)              start kinetic energy 1/2 |z|^2
//      kick     u -= c L_blk^T g                                     k_prior_trmm<true>,  mode 2
//      drift    q += eps L_blk (u - U (lam / (1 + lam) o U^T u))     k_lowrank_proj / _apply, k_prior_trmm<false>, mode 1
//      kinetic  1/2 (|u|^2 - sum_k lam_k / (1 + lam_k) (U^T u)_k^2)  k_metric_kinetic
// No [P, P] matrix exists anywhere.  Every kernel computes a chain with a summation order that does not depend on the batch, so B
// chains in one launch give the bits of B single-chain launches.
#include "nmgp_internal.h"

namespace nmgpk {

static inline int cdiv_m(long long a, long long b) { return (int)((a + b - 1) / b); }

#define TRMM_CG 8      // right-hand-side columns one workgroup carries per pass over its row of factor tiles

// out[b, slot(j, i)] (op)= coef * sum_k op(L_j)[i, k] in[b, slot(j, k)]   for the 1 + T prior blocks j of chain b
//   slot(0, i) = i (tilde_l), slot(j >= 1, i) = N + i T + (j - 1) (column j - 1 of uL_vecs: stride T), slot P - 1: identity block
//   factor of block 0: L0 (+ subject * s0), of blocks j >= 1: L1 (+ subject * s1); lower, column-major; the strict upper triangle
//   of a cached factor is scratch and is never read
//   mode 0: out = acc;  mode 1: out += coef * acc;  mode 2: out -= coef * acc unless bad[b]
// grid (ceil(N / 64), B, 1 + ceil(T / TRMM_CG)): blockIdx.z = 0 is block 0, z >= 1 the column group z - 1 of blocks 1..T.
// A workgroup owns 64 rows i; it walks the factor tiles of its row (k <= i, or k >= i for the transpose), staging each 64 x 64 tile
// through LDS so that both orientations read HBM/L2 along the contiguous index and compute with lanes along i.
// LAYOUT 0: the nonseparable parameter vector (above).  LAYOUT 1: the SEPARABLE one, [tilde_l (N) | tilde_sigma (N) | uL_vec (T) |
// tilde_sigma2_err] (logpos.py:17-29): two contiguous blocks with factors L0 / L1 (grid.z = 2), the T entries of uL_vec scaled by
// `cscale` (the sd of their Normal(0, c) prior, logpos.py:283), the last entry unchanged.
template <bool TRANS, int LAYOUT>
__global__ __launch_bounds__(256) void k_prior_trmm(const double* __restrict__ L0, int ld0, long long s0,
                                                     const double* __restrict__ L1, int ld1, long long s1,
                                                     const double* __restrict__ in, double* __restrict__ out, int N, int T,
                                                     long long P, int cps, double coef, int mode, const int* __restrict__ bad,
                                                     double cscale) {
    __shared__ double Lt[64 * 65];               // Lt[k * 65 + i] = op(L)[i0 + i, kb + k]
    __shared__ double vin[TRMM_CG][64];
    __shared__ double red[4][TRMM_CG][64];
    const int ib = blockIdx.x, b = blockIdx.y, zg = blockIdx.z;
    if (mode == 2 && bad[b]) return;             // (uniform per workgroup)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int subj = b / cps;
    const double* L = zg == 0 ? L0 + (size_t)subj * s0 : L1 + (size_t)subj * s1;
    const int ld = zg == 0 ? ld0 : ld1;
    const int j0 = (LAYOUT == 1) ? zg : (zg == 0 ? 0 : 1 + (zg - 1) * TRMM_CG);      // first prior block of this workgroup
    const int nc = (LAYOUT == 1 || zg == 0) ? 1 : (T - (zg - 1) * TRMM_CG < TRMM_CG ? T - (zg - 1) * TRMM_CG : TRMM_CG);
    const double* inb = in + (size_t)b * P;
    double* outb = out + (size_t)b * P;
    const int i0 = ib * 64;
    const int nblk = (N + 63) / 64;
    double acc[TRMM_CG];
#pragma unroll
    for (int j = 0; j < TRMM_CG; ++j) acc[j] = 0.0;
    const int kb_first = TRANS ? ib : 0, kb_last = TRANS ? nblk - 1 : ib;
    for (int kbi = kb_first; kbi <= kb_last; ++kbi) {
        const int kb = kbi * 64;
        __syncthreads();                         // the previous tile has been consumed
        for (int e = tid; e < 64 * 64; e += 256) {
            const int a = e & 63, c = e >> 6;    // a runs along the contiguous index of the stored factor
            double v = 0.0;
            if (!TRANS) {
                // op(L)[i0 + a, kb + c] = L[i0 + a, kb + c]: column kb + c, rows contiguous
                const int i = i0 + a, k = kb + c;
                if (i < N && k < N && k <= i) v = L[(size_t)k * ld + i];
                Lt[c * 65 + a] = v;
            } else {
                // op(L)[i0 + c, kb + a] = L[kb + a, i0 + c]: column i0 + c, rows kb + a contiguous
                const int i = i0 + c, k = kb + a;
                if (i < N && k < N && k >= i) v = L[(size_t)i * ld + k];
                Lt[a * 65 + c] = v;
            }
        }
        for (int e = tid; e < TRMM_CG * 64; e += 256) {
            const int c = e & 63, j = e >> 6;
            const int k = kb + c;
            double v = 0.0;
            if (j < nc && k < N)
                v = (LAYOUT == 1) ? inb[(size_t)j0 * N + k] : ((j0 + j == 0) ? inb[k] : inb[(size_t)N + (size_t)k * T + (j0 + j - 1)]);
            vin[j][c] = v;
        }
        __syncthreads();
        // wave w takes 16 of the tile's 64 k; lane = row i
#pragma unroll 4
        for (int c = 16 * w; c < 16 * w + 16; ++c) {
            const double a = Lt[c * 65 + lane];
#pragma unroll
            for (int j = 0; j < TRMM_CG; ++j) acc[j] = fma(a, vin[j][c], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < TRMM_CG; ++j) red[w][j][lane] = acc[j];
    __syncthreads();
    for (int e = tid; e < TRMM_CG * 64; e += 256) {
        const int r = e & 63, j = e >> 6;
        const int i = i0 + r;
        if (j >= nc || i >= N) continue;
        const double v = ((red[0][j][r] + red[1][j][r]) + red[2][j][r]) + red[3][j][r];
        const size_t o = (LAYOUT == 1) ? (size_t)j0 * N + i : ((j0 + j == 0) ? (size_t)i : (size_t)N + (size_t)i * T + (j0 + j - 1));
        if (mode == 0) outb[o] = v;
        else if (mode == 1) outb[o] = outb[o] + coef * v;
        else outb[o] = outb[o] - coef * v;
    }
    if (ib == 0 && zg == 0 && tid == 0) {        // tilde_sigma2_err: identity block
        const double v = inb[P - 1];
        if (mode == 0) outb[P - 1] = v;
        else if (mode == 1) outb[P - 1] = outb[P - 1] + coef * v;
        else outb[P - 1] = outb[P - 1] - coef * v;
    }
    if (LAYOUT == 1 && ib == 0 && zg == 0 && tid >= 64 && tid < 64 + T) {        // uL_vec: diagonal block cscale I
        const size_t o = (size_t)2 * N + (tid - 64);
        const double v = cscale * inb[o];
        if (mode == 0) outb[o] = v;
        else if (mode == 1) outb[o] = outb[o] + coef * v;
        else outb[o] = outb[o] - coef * v;
    }
}

void prior_trmm(hipStream_t s, bool trans, const double* L0, int ld0, long long s0, const double* L1, int ld1, long long s1,
                const double* in, double* out, int N, int T, long long P, int B, int cps, double coef, int mode, const int* bad) {
    if (cps < 1) cps = 1;
    const dim3 grid(cdiv_m(N, 64), B, 1 + cdiv_m(T, TRMM_CG));
    if (trans)
        NMGP_LAUNCH((k_prior_trmm<true, 0>), grid, dim3(256), 0, s, L0, ld0, s0, L1, ld1, s1, in, out, N, T, P, cps, coef, mode, bad, 1.0);
    else
        NMGP_LAUNCH((k_prior_trmm<false, 0>), grid, dim3(256), 0, s, L0, ld0, s0, L1, ld1, s1, in, out, N, T, P, cps, coef, mode, bad, 1.0);
}

// the separable parameter layout: out = op(blockdiag(L0, L1, cscale I_T, 1)) in for B vectors of length P = 2N + T + 1 (mode 0)
void prior_trmm_sep(hipStream_t s, bool trans, const double* L0, int ld0, const double* L1, int ld1, const double* in, double* out, int N,
                    int T, long long P, int B, double cscale) {
    const dim3 grid(cdiv_m(N, 64), B, 2);
    if (trans)
        NMGP_LAUNCH((k_prior_trmm<true, 1>), grid, dim3(256), 0, s, L0, ld0, 0LL, L1, ld1, 0LL, in, out, N, T, P, B, 1.0, 0, nullptr, cscale);
    else
        NMGP_LAUNCH((k_prior_trmm<false, 1>), grid, dim3(256), 0, s, L0, ld0, 0LL, L1, ld1, 0LL, in, out, N, T, P, B, 1.0, 0, nullptr, cscale);
}

// c[b, k] = sum_i U[subject(b)][k, i] u[b, i]   (U stored as r rows of length P per subject).  One workgroup per (k, chain): strided
// partial sums, then a tree -- the order k_hmc_kinetic uses.
__global__ __launch_bounds__(256) void k_lowrank_proj(const double* __restrict__ U, const double* __restrict__ u, double* __restrict__ c,
                                                       long long P, int r, int cps) {
    __shared__ double red[256];
    const int k = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const double* Uk = U + ((size_t)(b / cps) * r + k) * P;
    const double* ub = u + (size_t)b * P;
    double acc = 0.0;
    for (long long i = t; i < P; i += 256) acc = fma(Uk[i], ub[i], acc);
    red[t] = acc;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (t < h) red[t] += red[t + h];
        __syncthreads();
    }
    if (t == 0) c[(size_t)b * r + k] = red[0];
}

// out[b, i] = in[b, i] + sum_k U[subject(b)][k, i] (wgt[subject(b)][k] c[b, k])     (out may alias in)
__global__ __launch_bounds__(256) void k_lowrank_apply(const double* __restrict__ U, const double* __restrict__ wgt,
                                                        const double* __restrict__ c, const double* in, double* out, long long P,
                                                        int r, int cps) {
    extern __shared__ double wc[];               // [r]
    const int b = blockIdx.y, t = threadIdx.x;
    const int subj = b / cps;
    for (int k = t; k < r; k += 256) wc[k] = wgt[(size_t)subj * r + k] * c[(size_t)b * r + k];
    __syncthreads();
    const long long i = (long long)blockIdx.x * 256 + t;
    if (i >= P) return;
    const double* Us = U + (size_t)subj * r * P + i;
    double acc = 0.0;
    for (int k = 0; k < r; ++k) acc = fma(Us[(size_t)k * P], wc[k], acc);
    out[(size_t)b * P + i] = in[(size_t)b * P + i] + acc;
}

// kin[b] = 1/2 (sum_i u[b, i]^2 - sum_k s[subject(b)][k] c[b, k]^2),  s = lam / (1 + lam)   (r may be 0: c, s unused)
__global__ __launch_bounds__(256) void k_metric_kinetic(const double* __restrict__ u, const double* __restrict__ c,
                                                         const double* __restrict__ sw, double* __restrict__ kin, long long P, int r,
                                                         int cps) {
    __shared__ double red[256];
    const int b = blockIdx.x, t = threadIdx.x;
    const double* ub = u + (size_t)b * P;
    double acc = 0.0;
    for (long long i = t; i < P; i += 256) {
        const double v = ub[i];
        acc = fma(v, v, acc);
    }
    red[t] = acc;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (t < h) red[t] += red[t + h];
        __syncthreads();
    }
    const double uu = red[0];
    __syncthreads();
    double a2 = 0.0;
    for (int k = t; k < r; k += 256) {
        const double ck = c[(size_t)b * r + k];
        a2 = fma(sw[(size_t)(b / cps) * r + k] * ck, ck, a2);
    }
    red[t] = a2;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (t < h) red[t] += red[t + h];
        __syncthreads();
    }
    if (t == 0) kin[b] = 0.5 * (uu - red[0]);
}

void lowrank_proj(hipStream_t s, const double* U, const double* u, double* c, long long P, int r, int B, int cps) {
    if (r <= 0) return;
    NMGP_LAUNCH(k_lowrank_proj, dim3(r, B), dim3(256), 0, s, U, u, c, P, r, cps < 1 ? 1 : cps);
}
void lowrank_apply(hipStream_t s, const double* U, const double* wgt, const double* c, const double* in, double* out, long long P,
                   int r, int B, int cps) {
    if (r <= 0) return;
    NMGP_LAUNCH(k_lowrank_apply, dim3(cdiv_m(P, 256), B), dim3(256), (size_t)r * sizeof(double), s, U, wgt, c, in, out, P, r,
                cps < 1 ? 1 : cps);
}
void metric_kinetic(hipStream_t s, const double* u, const double* c, const double* sw, double* kin, long long P, int r, int B, int cps) {
    NMGP_LAUNCH(k_metric_kinetic, dim3(B), dim3(256), 0, s, u, c, sw, kin, P, r, cps < 1 ? 1 : cps);
}

}  // namespace nmgpk
