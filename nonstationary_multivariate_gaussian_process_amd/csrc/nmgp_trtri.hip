// Blocked triangular inversion X = L^-T AFTER the factorisation (gradient evaluations; DESIGN.md "value+gradient step").
//
// SELECTABLE ALTERNATIVE (NMGP_TRTRI=1), not the default: measured equal to the riding rows at 128 chains and slower for small
// batches -- figures at trtri_post_applies() below.
// By default the n rows of X ride below the matrix through EVERY level of the factorisation's panel recursion (identity rows
// that the panel solves and updates turn into L^-T): the HBM-bound K <= 512 update classes and the leaf launches then work on
// twice the rows (72 of 288 ms of a 128-chain step).  Here the factorisation runs in its value form (the right-hand side row
// only) and X is built afterwards from products with EXPLICIT inverses of the diagonal blocks, so that all but ~2 % of the
// n^3/3 flop run in matrix-core launches with K >= 512:
//
//   leaf    X[P, P] = L_PP^-T for every 128 x 128 diagonal block (k_trtri_leaf128: substitution in LDS on 64 x 64 halves, the
//           off-diagonal 64 x 64 block by two small products);
//   levels  m = 128, 256, ... : for every pair of neighbouring m-blocks (P1, P2), ALL pairs of all matrices in two launches:
//              T' = -X[P1, P1] L[P2, P1]^T        (k_tri_gemm<0>: both operands are row panels, k-loop starts at the tile's first row)
//              X[P1, P2] = T' X[P2, P2]           (k_tri_gemm<1>: the j-side operand is K-contiguous in memory and upper triangular:
//                                                  staged transposed into LDS, k-loop ends at the tile's last column)
//           which is X12 = -X11 L21^T X22 of [[L11, 0], [L21, L22]]^-T = [[X11, X12], [0, X22]];
//   top     when n / 128 is not a power of two (6144 = 3 x 2048) the remaining blocks are combined left to right with the same
//           two launches: X[0:a, a:b] = -X[0:a, 0:a] L[a:b, 0:a]^T X[a:b, a:b].
//
// T' lives in the strictly upper triangle of the factor's own n x n region (rows < a <= columns: scratch that nobody reads
// after the factorisation), X in the rows below the factor where the riding rows used to be, with the same guarantees for
// its consumers (inverse SYRK, triangular matrix-vector product): zeros are stored left of the diagonal inside the diagonal
// 256 x 256 blocks (k_xtri_seed band + the leaf kernel's own zero block).
//
// The tile kernel is the mask-free fast path of k_syrk_lower (nmgp_chol.hip) with separate operands: 128 x 128 tile per
// 512-thread workgroup, 8 waves of 64 x 32, k-panels of 16 double-buffered in LDS, buffer loads with scalar k-offsets,
// row-paired fragment layout, no VALU work in the k-loop.
#include "nmgp_internal.h"

namespace nmgpk {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

#define TT_BM 128
#define TT_BK 16
#define TT_LD (TT_BM + 16)

// C = +-A B over `npairs` independent problems per matrix and `nbatch` matrices, all inside one buffer S (matrix b at S + b bs;
// pair p at + p pstride; operand origins offA / offB / offC in elements).  A: mrows x K, element (i, k) at A[i + k ld].
// MODE 0: B is ncols x K in ROW-PANEL form, element (j, k) at B[j + k ld]  (C = A B^T), and A is upper triangular in the sense
//         A[i, k] == 0 for k < i (zeros stored inside the diagonal 128-blocks): the k-loop of tile row bi starts at 128 bi.
// MODE 1: B is K x ncols, element (k, j) at B[k + j ld] (C = A B), upper triangular (B[k, j] == 0 for k > j, zeros stored
//         inside the diagonal 128-blocks): the k-loop of tile column bj ends at 128 (bj + 1).
// mrows, ncols multiples of 128, K a multiple of 32.  C is OVERWRITTEN (beta = 0); neg: C = -A B.
// Tile order: tiles of equal k-length get consecutive workgroup ids, longest first (workgroups are dispatched in id order and
// dealt round-robin to the XCDs: an XCD holding longer tiles than its neighbours would make the whole dispatch wait).
template <int MODE>
__global__ __launch_bounds__(512, 4) void k_tri_gemm(double* __restrict__ S, int ld, long long bs, int nbatch, long long offA,
                                                      long long offB, long long offC, long long pstride, int npairs, int mrows,
                                                      int ncols, int K, int neg, int lock) {
    __shared__ __attribute__((aligned(16))) double smem[4 * TT_BK * TT_LD];
    constexpr int SBUF = TT_BK * TT_LD;
    double* sA0 = smem;
    double* sB0 = smem + 2 * SBUF;
    const int ti = mrows >> 7, tj = ncols >> 7;
    int bi, bj, p, bz;
    if (lock == 1) {
        // XCD-lockstep order (needs npairs * nbatch % 8 == 0).  Workgroup id t runs on XCD t % 8 and ids are dispatched in order, so
        // XCD x is given the matrices x, x + 8, ... and walks, for every 8 x 8 block of tiles (longest blocks first), its matrices one
        // after the other: at any time all XCDs work on the SAME block position of different matrices (equal k-lengths: no XCD holds
        // the dispatcher up) and the ~64 tiles resident on an XCD share 8 + 8 operand panels, which its L2 then fetches once per block
        // instead of once per tile (these launches move 4.3-4.5 TB/s through the fabric otherwise).
        const int id = blockIdx.x, x = id & 7;
        int qq = id >> 3;
        const int bh = ti < 8 ? ti : 8, bw = tj < 8 ? tj : 8, tpb = bh * bw;
        const int nbr = ti / bh, nbc = tj / bw, mpx = (npairs * nbatch) >> 3;
        const int tin = qq % tpb;
        qq /= tpb;
        const int ml = qq % mpx, bp = qq / mpx;
        int Br, Bc;
        if (MODE == 0) {
            Br = bp / nbc;
            Bc = bp - Br * nbc;
        } else {
            const int u = bp / nbr;
            Bc = nbc - 1 - u;
            Br = bp - u * nbr;
        }
        bi = Br * bh + tin % bh;
        bj = Bc * bw + tin / bh;
        const int mat = ml * 8 + x;
        p = mat % npairs;
        bz = mat / npairs;
    } else if (lock == 2) {
        // matrix-major order: all tiles of one (matrix, pair) before the next -- the chip works on one matrix at a time, whose operands
        // then live in the Infinity Cache (the order of the inverse SYRK); inside, longest tile rows / columns first
        const int id = blockIdx.x, tpm = ti * tj;
        const int mat = id / tpm, tin = id - mat * tpm;
        p = mat % npairs;
        bz = mat / npairs;
        if (MODE == 0) {
            bi = tin / tj;
            bj = tin - bi * tj;
        } else {
            const int u = tin / ti;
            bj = tj - 1 - u;
            bi = tin - u * ti;
        }
    } else {
        const int id = blockIdx.x;
        if (MODE == 0) {
            const int gsz = tj * npairs * nbatch;
            bi = id / gsz;
            int rem = id - bi * gsz;
            bj = rem % tj;
            rem /= tj;
            p = rem % npairs;
            bz = rem / npairs;
        } else {
            const int gsz = ti * npairs * nbatch;
            const int u = id / gsz;
            bj = tj - 1 - u;
            int rem = id - u * gsz;
            bi = rem % ti;
            rem /= ti;
            p = rem % npairs;
            bz = rem / npairs;
        }
    }
    bi = __builtin_amdgcn_readfirstlane(bi);
    bj = __builtin_amdgcn_readfirstlane(bj);
    p = __builtin_amdgcn_readfirstlane(p);
    bz = __builtin_amdgcn_readfirstlane(bz);
    double* base = S + (size_t)bz * bs + (size_t)p * pstride;
    const double* Ap = base + offA;
    double* Cp = base + offC;
    const int row0 = bi * TT_BM, col0 = bj * TT_BM;
    const int kt0 = MODE == 0 ? 8 * bi : 0;
    const int nk = MODE == 0 ? K / TT_BK : 8 * (bj + 1);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wi = w & 1, wj = w >> 1;
    const int l15 = lane & 15, l4 = lane >> 4;
    // global -> LDS staging of the row-panel operands: thread (rp, cg) moves rows 2rp, 2rp + 1 of k-columns cg and cg + 8
    const int rp = tid & 63, cg = tid >> 6;
    const int offAl = (cg * ld + row0 + 2 * rp) * 8;
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, 0x7fffffff, 0x00020000);
    const int gstep = TT_BK * ld * 8, ghalf = 8 * ld * 8;
    int soffA = kt0 * gstep;
    // j side.  MODE 0: like the i side.  MODE 1: thread (jl, kq) moves B[k, col0 + jl] for k = 2kq, 2kq + 1 and 8 + 2kq, 9 + 2kq of
    // the panel (16 contiguous bytes each) and writes them to LDS TRANSPOSED into the same [k][j] layout (a wave = 64 consecutive
    // j of one k-pair: its four 8-byte LDS stores are 512 contiguous bytes each)
    const int jl = tid & 127, kq = tid >> 7;
    const double* Bp = base + offB + (MODE == 1 ? (size_t)col0 * ld : (size_t)0);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, 0x7fffffff, 0x00020000);
    const int offBl = MODE == 0 ? (cg * ld + col0 + 2 * rp) * 8 : (jl * ld + 2 * kq) * 8;
    int soffB = MODE == 0 ? kt0 * gstep : 0;
    v4i ra0, ra1, rb0, rb1;
    auto gload = [&]() {
        ra0 = __builtin_amdgcn_raw_buffer_load_b128(rsrcA, offAl, soffA, 0);
        ra1 = __builtin_amdgcn_raw_buffer_load_b128(rsrcA, offAl, soffA + ghalf, 0);
        if (MODE == 0) {
            rb0 = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, offBl, soffB, 0);
            rb1 = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, offBl, soffB + ghalf, 0);
            soffB += gstep;
        } else {
            rb0 = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, offBl, soffB, 0);
            rb1 = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, offBl, soffB + 64, 0);
            soffB += TT_BK * 8;
        }
        soffA += gstep;
    };
    double* wA = sA0 + cg * TT_LD + 2 * rp;
    double* wB = MODE == 0 ? sB0 + cg * TT_LD + 2 * rp : sB0 + (2 * kq) * TT_LD + jl;
    auto sstore = [&](int buf) {
        *reinterpret_cast<v4i*>(wA + buf * SBUF) = ra0;
        *reinterpret_cast<v4i*>(wA + buf * SBUF + 8 * TT_LD) = ra1;
        if (MODE == 0) {
            *reinterpret_cast<v4i*>(wB + buf * SBUF) = rb0;
            *reinterpret_cast<v4i*>(wB + buf * SBUF + 8 * TT_LD) = rb1;
        } else {
            const v2d lo = __builtin_bit_cast(v2d, rb0), hi = __builtin_bit_cast(v2d, rb1);
            wB[buf * SBUF] = lo[0];
            wB[buf * SBUF + TT_LD] = lo[1];
            wB[buf * SBUF + 8 * TT_LD] = hi[0];
            wB[buf * SBUF + 9 * TT_LD] = hi[1];
        }
    };
    gload();
    // acc[p][s][tj][r]: i = row0 + wi*64 + 32p + 2*l15 + s ; j = col0 + wj*32 + 2*(l4 + 4r) + tj
    v4d acc[2][2][2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int sx = 0; sx < 2; ++sx)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[pp][sx][t] = v4d{0.0, 0.0, 0.0, 0.0};
    sstore(0);
    __syncthreads();
    const double* rA = sA0 + wi * 64 + 2 * l15 + l4 * TT_LD;
    const double* rB = sB0 + wj * 32 + 2 * l15 + l4 * TT_LD;
    auto compute = [&](int buf) {
        const double* tA = rA + buf * SBUF;
        const double* tB = rB + buf * SBUF;
        v2d fa[2], fb, na[2], nb;
        fa[0] = *reinterpret_cast<const v2d*>(tA);
        fa[1] = *reinterpret_cast<const v2d*>(tA + 32);
        fb = *reinterpret_cast<const v2d*>(tB);
#pragma unroll
        for (int kk = 0; kk < TT_BK / 4; ++kk) {
            if (kk + 1 < TT_BK / 4) {
                na[0] = *reinterpret_cast<const v2d*>(tA + (kk + 1) * 4 * TT_LD);
                na[1] = *reinterpret_cast<const v2d*>(tA + (kk + 1) * 4 * TT_LD + 32);
                nb = *reinterpret_cast<const v2d*>(tB + (kk + 1) * 4 * TT_LD);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                    for (int sx = 0; sx < 2; ++sx)
                        acc[pp][sx][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[t], fa[pp][sx], acc[pp][sx][t], 0, 0, 0);
            if (kk + 1 < TT_BK / 4) {
                fa[0] = na[0];
                fa[1] = na[1];
                fb = nb;
            }
        }
    };
    for (int kt = kt0; kt < nk; kt += 2) {         // nk - kt0 is even
        gload();
        compute(0);
        sstore(1);
        __syncthreads();
        if (kt + 2 < nk) gload();
        compute(1);
        if (kt + 2 < nk) sstore(0);
        __syncthreads();
    }
    const double sg = neg ? -1.0 : 1.0;
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = row0 + wi * 64 + 32 * pp + 2 * l15;
                const int j = col0 + wj * 32 + 2 * (l4 + 4 * r) + t;
                double2 c;
                c.x = sg * acc[pp][0][t][r];
                c.y = sg * acc[pp][1][t][r];
                *reinterpret_cast<double2*>(&Cp[(size_t)j * ld + i]) = c;
            }
}

// X[P, P] = L_PP^-T of one 128 x 128 diagonal block P = [c0, c0 + 128) per workgroup (grid: n / 128 x batch, 256 threads).
// L at S[r + c ld] (lower), X at S[xoff + r + c ld] (upper; the block's lower part is written as zeros: consumers read it).
//   1. the two 64 x 64 diagonal halves by forward substitution in LDS, column c of W = L^-1 by thread c of a wave (true
//      division for the diagonal; W[i, c] = -(sum_{c <= k < i} L[i, k] W[k, c]) / L[i, i]); W is stored transposed (= X) in the
//      free upper triangle of the same LDS block;
//   2. X12 = -X11 L21^T X22 (64 x 64) by two register-blocked products.
__global__ __launch_bounds__(256) void k_trtri_leaf128(double* __restrict__ S, int ld, long long bs, int xoff) {
    __shared__ double buf[2][64][64];          // [half][column][row]
    S += (size_t)blockIdx.y * bs;
    const int c0 = blockIdx.x * 128;
    const int tid = threadIdx.x;
    for (int e = tid; e < 2 * 4096; e += 256) {
        const int h = e >> 12, c = (e >> 6) & 63, r = e & 63;
        buf[h][c][r] = (r >= c) ? S[(size_t)(c0 + 64 * h + c) * ld + c0 + 64 * h + r] : 0.0;
    }
    __syncthreads();
    if (tid < 128) {
        double (*B)[64] = buf[tid >> 6];
        const int c = tid & 63;
        const double dinv = 1.0 / B[c][c];
        for (int i = 1; i < 64; ++i) {                       // (uniform: the shuffle needs every lane)
            const double di = __shfl(dinv, i);
            if (c < i) {
                double s0 = B[c][i] * dinv, s1 = 0.0, s2 = 0.0, s3 = 0.0;      // L[i, c] W[c, c]
                int k = c + 1;
                for (; k + 3 < i; k += 4) {
                    s0 = fma(B[k][i], B[k][c], s0);              // L[i, k] (lower) x W[k, c] (stored at column k, row c: upper)
                    s1 = fma(B[k + 1][i], B[k + 1][c], s1);
                    s2 = fma(B[k + 2][i], B[k + 2][c], s2);
                    s3 = fma(B[k + 3][i], B[k + 3][c], s3);
                }
                for (; k < i; ++k) s0 = fma(B[k][i], B[k][c], s0);
                B[i][c] = -((s0 + s1) + (s2 + s3)) * di;
            }
        }
        B[c][c] = dinv;
    }
    __syncthreads();
    // X11, X22 (upper incl. diagonal, zeros below) and the zero block left of X22 go out; lanes along the rows of a column
    const int r = tid & 63, q = tid >> 6;
    double* Xg = S + xoff + c0;                    // X[c0 + r, c0 + c] at Xg[r + (c0 + c) ld]
    for (int u = 0; u < 16; ++u) {
        const int c = 16 * q + u;
        Xg[(size_t)(c0 + c) * ld + r] = (r <= c) ? buf[0][c][r] : 0.0;
        Xg[(size_t)(c0 + 64 + c) * ld + 64 + r] = (r <= c) ? buf[1][c][r] : 0.0;
        Xg[(size_t)(c0 + c) * ld + 64 + r] = 0.0;
    }
    // T[r, m] = sum_{k >= r} X11[r, k] L21[m, k], m = 16 q + u
    double t[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) t[u] = 0.0;
    const double* L21 = S + c0 + 64 + 16 * q;      // L21[m, k] at L21[u + (c0 + k) ld]
    for (int k = 0; k < 64; ++k) {
        const double x = (k >= r) ? buf[0][k][r] : 0.0;
        const double* lk = L21 + (size_t)(c0 + k) * ld;
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = fma(x, lk[u], t[u]);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 16; ++u) buf[0][16 * q + u][r] = t[u];
    __syncthreads();
    // X12[r, c] = -sum_{m <= c} T[r, m] X22[m, c], c = 16 q + u
#pragma unroll
    for (int u = 0; u < 16; ++u) t[u] = 0.0;
    for (int m = 0; m < 64; ++m) {
        const double tv = buf[0][m][r];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int c = 16 * q + u;
            const double x22 = (m <= c) ? buf[1][c][m] : 0.0;
            t[u] = fma(tv, x22, t[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) Xg[(size_t)(c0 + 64 + 16 * q + u) * ld + r] = -t[u];
}

static void tri_gemm_pair(hipStream_t s, double* S, int ld, long long bs, int batch, int xoff, int a0, int a1, int a2, int npairs,
                          long long pstride, const SyrkHook* hook) {
    const int m1 = a1 - a0, m2 = a2 - a1;
    // Tile order (NMGP_TRTRI_ORDER).  Default `matrix`: all tiles of one (matrix, pair) before the next, longest tile rows / columns
    // first inside -- the chip then works on one matrix at a time, whose operands live in the 256 MB Infinity Cache (the order of
    // the inverse SYRK).  `rows`: tiles of equal k-length consecutive ACROSS matrices (the working set is a tile row of all 128
    // matrices: it streams from HBM).  `lockstep`: XCD x owns the matrices x, x + 8, ... and all XCDs walk the same 8 x 8 tile block
    // (L2 reuse of the operand panels, eight matrices live).  Measured, 128 chains, factorisation + inversion: matrix 313.2 ms,
    // rows 320.0 (312.9 on a faster box), lockstep +2.4-3.9 ms on rows; the same lockstep order costs the inverse SYRK 8 %.  What
    // these launches want is Infinity-Cache locality, not L2 reuse.
    static const int order_env = [] {
        const char* e = std::getenv("NMGP_TRTRI_ORDER");
        return (e && std::strcmp(e, "lockstep") == 0) ? 1 : ((e && std::strcmp(e, "rows") == 0) ? 0 : 2);
    }();
    const int t1 = m1 / 128, t2 = m2 / 128;
    const int lock = order_env == 2 ? 2
                     : ((order_env == 1 && ((long long)npairs * batch) % 8 == 0 && (t1 < 8 || t1 % 8 == 0) && (t2 < 8 || t2 % 8 == 0)) ? 1 : 0);
    // T' = -X[a0:a1, a0:a1] L[a1:a2, a0:a1]^T  -> rows a0.., columns a1.. of the factor's region (strictly upper: scratch)
    {
        void* tok = nullptr;
        if (hook && hook->begin)
            tok = hook->begin(hook->user, s, (double)m1 * m1 * m2 * npairs * batch, 8.0 * npairs * batch * (0.5 * m1 * m1 + 2.0 * m1 * m2));
        const unsigned grid = (unsigned)((long long)(m1 / 128) * (m2 / 128) * npairs * batch);
        NMGP_LAUNCH((k_tri_gemm<0>), dim3(grid), dim3(512), 0, s, S, ld, bs, batch, (long long)(xoff + a0) + (long long)a0 * ld,
                    (long long)a1 + (long long)a0 * ld, (long long)a0 + (long long)a1 * ld, pstride, npairs, m1, m2, m1, 1, lock);
        if (tok && hook->end) hook->end(hook->user, tok);
    }
    // X[a0:a1, a1:a2] = T' X[a1:a2, a1:a2]
    {
        void* tok = nullptr;
        if (hook && hook->begin)
            tok = hook->begin(hook->user, s, (double)m1 * m2 * m2 * npairs * batch, 8.0 * npairs * batch * (0.5 * m2 * m2 + 2.0 * m1 * m2));
        const unsigned grid = (unsigned)((long long)(m1 / 128) * (m2 / 128) * npairs * batch);
        NMGP_LAUNCH((k_tri_gemm<1>), dim3(grid), dim3(512), 0, s, S, ld, bs, batch, (long long)a0 + (long long)a1 * ld,
                    (long long)(xoff + a1) + (long long)a1 * ld, (long long)(xoff + a0) + (long long)a1 * ld, pstride, npairs, m1, m2,
                    m2, 0, lock);
        if (tok && hook->end) hook->end(hook->user, tok);
    }
}

// Whether potrf_lower builds the n rows of L^-T after the factorisation (this file) instead of letting them ride through it.
// NMGP_TRTRI=1 selects the blocked inversion wherever it is defined (n a multiple of 128); the DEFAULT is the riding rows.
// Measured on MI355X (round 4, same box, alternating runs; evals/s value+gradient, riding rows -> blocked inversion): 128 chains
// of n = 6144 267.4 -> 266.9 (factorisation + inversion 312.9 -> 312.9 ms: the inversion takes 159 ms -- leaf 1.9, levels 128 /
// 256 / 512 / 1024 0.6 / 1.3 / 4.4 / 14.9, the two top combines 34.8 + 101.4 -- for n^3/3 flop per matrix = 62 TFLOP/s, 67-68
// on the flop its tiles execute, exactly what the riding rows cost); 32 / 16 chains 265.9 -> 261.7 / 256.7 -> 250.8; 4 chains
// 214.8 -> 222.1; one chain 168.2 -> 141.2; 8 subjects x N = 1024 1335 -> 1188; 64 subjects 1766 -> 1717; separable N = 4096,
// D = 5 9.09 -> 9.42 ms.  The K <= 512 classes the riding rows double are paid back one for one by the inversion's short levels
// and by the zero halves of its diagonal tiles.
bool trtri_post_applies(int n, int xtri, int lda, int batch) {
    static const int mode = [] {
        const char* e = std::getenv("NMGP_TRTRI");
        return e ? std::atoi(e) : 0;
    }();
    (void)batch;
    if (mode != 1 || xtri != n || n < 256 || (n % 128) != 0 || (lda & 1)) return false;
    if ((long long)(n + 16) * lda * 8 >= 0x7fff0000LL) return false;      // 32-bit byte offsets into a k-panel
    return true;
}

// X = L^-T into rows xoff .. xoff + n - 1 of the factorisation buffer (see the header of this file).  L must be complete.
void trtri_upper_post(hipStream_t s, double* S, int ld, int n, int xoff, int batch, long long bs, const SyrkHook* hook) {
    const int q = n / 128;
    NMGP_LAUNCH(k_trtri_leaf128, dim3(q, batch), dim3(256), 0, s, S, ld, bs, xoff);
    int top = 1;
    while ((q % (2 * top)) == 0) top *= 2;
    for (int m = 128; m < top * 128; m *= 2)
        tri_gemm_pair(s, S, ld, bs, batch, xoff, 0, m, 2 * m, n / (2 * m), (long long)2 * m * (1 + (long long)ld), hook);
    const int W = top * 128;
    for (int t = 1; t < q / top; ++t) tri_gemm_pair(s, S, ld, bs, batch, xoff, 0, t * W, (t + 1) * W, 1, 0, hook);
}

}  // namespace nmgpk
