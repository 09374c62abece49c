// Laboratory for the trailing-update kernel (NOT part of the product library): times variants of k_syrk_lower on the
// dominant shape of the batched factorisation (m = 5632 rows, K = 512, 32 matrices) over ~1 s of back-to-back launches,
// next to rocblas_dgemm_strided_batched on the same operands (full square, so twice the useful flop).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/lab/syrk_lab.hip -o tools/lab/syrk_lab -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include <algorithm>

typedef double v4d __attribute__((ext_vector_type(4)));
#define SY_BM 128
#define SY_LD (SY_BM + 16)
#define SY_SB 8

__device__ __forceinline__ int clamp_row_pair(int r, int rows) {
    int rc = r < rows - 1 ? r : rows - 2;
    return rc < 0 ? 0 : rc;
}

__device__ __forceinline__ bool decode_tile(int swz, int nbatch, int mrows, int ncols, int& bi, int& bj, int& bz) {
    const int t = blockIdx.x, q = t >> 3;
    const long long g = ((long long)(q >> 6) * 8 + (t & 7)) * 64 + (q & 63);
    bz = (int)(g / swz);
    if (bz >= nbatch) return false;
    int idx = (int)(g - (long long)bz * swz);
    const int gx = (mrows + SY_BM - 1) / SY_BM, gy = (ncols + SY_BM - 1) / SY_BM;
    int c0 = 0, W = 0, cnt = 0;
    for (;; c0 += SY_SB) {
        W = gy - c0 < SY_SB ? gy - c0 : SY_SB;
        cnt = W * (W + 1) / 2 + (gx - c0 - W) * W;
        if (idx < cnt) break;
        idx -= cnt;
    }
    const int tri = W * (W + 1) / 2;
    if (idx < tri) {
        int b = 0;
        while (idx >= W - b) { idx -= W - b; ++b; }
        bj = c0 + b;
        bi = bj + idx;
    } else {
        idx -= tri;
        const int blk = idx / (SY_SB * W), in = idx - blk * (SY_SB * W);
        const int base = c0 + W + blk * SY_SB;
        const int rb = gx - base < SY_SB ? gx - base : SY_SB;
        bj = c0 + in / rb;
        bi = base + in % rb;
    }
    return true;
}

// V = variant bits: 1 = no edge masks (full tiles assumed), 2 = no C preload (acc = 0, epilogue C -= acc... as += since B negated)
template <int NWJ, int BK, int V>
__global__ __launch_bounds__(128 * NWJ, NWJ) void k_syrk(const double* __restrict__ A, int lda, double* __restrict__ C, int ldc,
                                                        int mrows, int ncols, int K, long long bstride, long long cstride,
                                                        int swz, int nbatch, unsigned long long* stamps = nullptr) {
    const unsigned long long st_c0 = (V & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long st_r0 = (V & 16) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    constexpr int NT = 128 * NWJ, CW = 128 / NWJ, TJ = CW / 16, NQ = (64 * BK) / NT, CGS = NT / 64;
    __shared__ double sA[2][BK * SY_LD];
    __shared__ double sB[2][BK * SY_LD];
    int bi = blockIdx.x, bj = blockIdx.y, bz = blockIdx.z;
    if (swz && !decode_tile(swz, nbatch, mrows, ncols, bi, bj, bz)) return;
    if (bi < bj) return;
    A += (size_t)bz * bstride;
    C += (size_t)bz * cstride;
    const bool diag = (bi == bj);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wi = w & 1, wj = w >> 1;
    const int row0 = bi * SY_BM, col0 = bj * SY_BM;
    const int rp = tid & 63, cg = tid >> 6;
    double2 ra[NQ], rb[NQ];
    const int ri = row0 + 2 * rp, rj = col0 + 2 * rp;
    const int ric = clamp_row_pair(ri, mrows), rjc = clamp_row_pair(rj, mrows);
    const bool ix = ri < mrows, iy = ri + 1 < mrows, jx = rj < mrows, jy = rj + 1 < mrows;
    const bool ish = ix && (ric != ri), jsh = jx && (rjc != rj);
    auto gload = [&](int k0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int kc = k0 + cg + CGS * q;
            const double* colp = A + (size_t)((V & 1) ? kc : (kc < K ? kc : K - 1)) * lda;
            ra[q] = *reinterpret_cast<const double2*>(colp + ric);
            if (!diag) rb[q] = *reinterpret_cast<const double2*>(colp + rjc);
        }
    };
    auto sstore = [&](int buf, int k0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int kl = cg + CGS * q;
            const bool kin = (k0 + kl) < K;
            double2 va, vb;
            if (V & 1) {
                va = ra[q];
                vb = rb[q];
            } else {
                va.x = (kin && ix) ? (ish ? ra[q].y : ra[q].x) : 0.0;
                va.y = (kin && iy) ? ra[q].y : 0.0;
                vb.x = (kin && jx) ? (jsh ? rb[q].y : rb[q].x) : 0.0;
                vb.y = (kin && jy) ? rb[q].y : 0.0;
            }
            *reinterpret_cast<double2*>(&sA[buf][kl * SY_LD + 2 * rp]) = va;
            if (!diag) *reinterpret_cast<double2*>(&sB[buf][kl * SY_LD + 2 * rp]) = vb;
        }
    };
    const bool active = !(diag && (wi * 64 + 63 < wj * CW));
    const int nk = (K + BK - 1) / BK;
    gload(0);
    v4d acc[TJ][4];
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = row0 + wi * 64 + ti * 16 + (lane & 15);
                const int j = col0 + wj * CW + tj * 16 + (lane >> 4) + 4 * r;
                if (V & 2) acc[tj][ti][r] = 0.0;
                else acc[tj][ti][r] = (active && i < mrows && j < ncols && i >= j) ? C[(size_t)j * ldc + i] : 0.0;
            }
    sstore(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
        if (active) {
            const double* tA = sA[cur] + wi * 64 + (lane & 15) + (lane >> 4) * SY_LD;
            const double* tB = (diag ? sA[cur] : sB[cur]) + wj * CW + (lane & 15) + (lane >> 4) * SY_LD;
            double fa[4], fb[TJ], na[4], nb[TJ];
#pragma unroll
            for (int t = 0; t < 4; ++t) fa[t] = tA[t * 16];
#pragma unroll
            for (int t = 0; t < TJ; ++t) fb[t] = (V & 2) ? tB[t * 16] : -tB[t * 16];
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                if (kk + 1 < BK / 4) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) na[t] = tA[(kk + 1) * 4 * SY_LD + t * 16];
#pragma unroll
                    for (int t = 0; t < TJ; ++t) nb[t] = (V & 2) ? tB[(kk + 1) * 4 * SY_LD + t * 16] : -tB[(kk + 1) * 4 * SY_LD + t * 16];
                    if (V & 32) __builtin_amdgcn_sched_barrier(0);     // keep the next fragments' LDS reads AHEAD of this substep's MFMAs
                }
#pragma unroll
                for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
                    for (int ti = 0; ti < 4; ++ti)
                        acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[tj], fa[ti], acc[tj][ti], 0, 0, 0);
                if (kk + 1 < BK / 4) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) fa[t] = na[t];
#pragma unroll
                    for (int t = 0; t < TJ; ++t) fb[t] = nb[t];
                }
            }
        }
        if (V & 12) {
            // hypothesis test: extra integer VALU work per k-step (32 or 128 v_add_u32), result kept alive
            int junk = tid;
#pragma unroll
            for (int z = 0; z < ((V & 8) ? 128 : 32); ++z) asm volatile("v_add_u32 %0, %0, %1" : "+v"(junk) : "v"(lane));
            if (junk == 0x7fffffff) acc[0][0][0] += 1.0;
        }
        if (kt + 1 < nk) sstore(cur ^ 1, (kt + 1) * BK);
        __syncthreads();
    }
    if ((V & 16) && tid == 0 && stamps) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        stamps[2 * blockIdx.x] = c1 - st_c0;
        stamps[2 * blockIdx.x + 1] = r1 - st_r0;
    }
    if (!active) return;
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = row0 + wi * 64 + ti * 16 + (lane & 15);
                const int j = col0 + wj * CW + tj * 16 + (lane >> 4) + 4 * r;
                if (i < mrows && j < ncols && i >= j) {
                    if (V & 2) C[(size_t)j * ldc + i] -= acc[tj][ti][r];
                    else C[(size_t)j * ldc + i] = acc[tj][ti][r];
                }
            }
}


// ---- fast path: full 128x128 tiles, K a multiple of 32; 8 waves (64 x 32 per wave) ------------------------------
// Differences from k_syrk: (1) row-PAIRED fragment layout: MFMA tile (2p+s) of the i side covers rows
// 32p + 2*(lane&15) + s, so one ds_read_b128 feeds two MFMA operands and every lane owns two consecutive rows of C
// (16-byte loads/stores of C); the j side is paired the same way; (2) accumulators start as -C and the epilogue
// stores -acc: no per-step negation; (3) k-loop unrolled by two so that the LDS buffer offsets are immediates;
// (4) global addresses = uniform base (SGPR, advanced per step) + constant per-lane 32-bit offset.
typedef double v2d __attribute__((ext_vector_type(2)));
template <int V>
__global__ __launch_bounds__(512, 4) void k_syrk_fast(const double* __restrict__ A, int lda, double* __restrict__ C, int ldc,
                                                      int mrows, int ncols, int K, long long bstride, long long cstride,
                                                      int swz, int nbatch, unsigned long long* stamps = nullptr) {
    const unsigned long long st_c0 = (V & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long st_r0 = (V & 16) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned long long st_c1 = 0;
    constexpr int BK = 16;
    __shared__ double sA[2][BK * SY_LD];
    __shared__ double sB[2][BK * SY_LD];
    // V & 64/128: de-phase the two workgroups that share a CU -- the second 256 workgroups of the launch (the second resident of every
    // CU, if the dispatcher deals the first 512 round-robin) start half a tile (64) / a quarter (128) late; later workgroups inherit
    // the slot, and the phase, of the one they replace
    if ((V & 192) && blockIdx.x >= 256 && blockIdx.x < 512 && blockIdx.y == 0 && blockIdx.z == 0) {
        const long long ticks = (V & 64) ? (long long)K * 5 / 2 : (long long)K * 5 / 4;      // 100 MHz: K = 512 -> 12.8 / 6.4 us
        const long long t0 = (long long)wall_clock64();
        while ((long long)wall_clock64() - t0 < ticks) {
        }
    }
    int bi = blockIdx.x, bj = blockIdx.y, bz = blockIdx.z;
    if (swz && !decode_tile(swz, nbatch, mrows, ncols, bi, bj, bz)) return;
    if (bi < bj) return;
    A += (size_t)bz * bstride;
    C += (size_t)bz * cstride;
    const bool diag = (bi == bj);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wi = w & 1, wj = w >> 1;
    const int row0 = bi * SY_BM, col0 = bj * SY_BM;
    const int l15 = lane & 15, l4 = lane >> 4;
    // global -> LDS staging: thread (rp = tid & 63, cg = tid >> 6) moves rows 2rp, 2rp+1 of k-columns cg and cg + 8
    const int rp = tid & 63, cg = tid >> 6;
    // V & 32: every tile streams the SAME two row panels (rows 0.. and 128..): the k-loop is unchanged but all panel loads hit the
    // L2 -- the speed a perfect L2 reuse of the A panels would give (results are wrong on purpose)
    const int offA = (cg * lda + ((V & 32) ? 0 : row0) + 2 * rp) * 8, offB = (cg * lda + ((V & 32) ? 128 : col0) + 2 * rp) * 8;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
    const int gstep = BK * lda * 8, ghalf = 8 * lda * 8;
    int soff = 0;
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i ra0, ra1, rb0, rb1;
    auto gload = [&]() {
        ra0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, offA, soff, 0);
        ra1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, offA, soff + ghalf, 0);
        if (!diag) {
            rb0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, offB, soff, 0);
            rb1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, offB, soff + ghalf, 0);
        }
        soff += gstep;
    };
    double* wA = &sA[0][cg * SY_LD + 2 * rp];
    double* wB = &sB[0][cg * SY_LD + 2 * rp];
    auto sstore = [&](int buf) {
        *reinterpret_cast<v4i*>(wA + buf * (BK * SY_LD)) = ra0;
        *reinterpret_cast<v4i*>(wA + buf * (BK * SY_LD) + 8 * SY_LD) = ra1;
        if (!diag) {
            *reinterpret_cast<v4i*>(wB + buf * (BK * SY_LD)) = rb0;
            *reinterpret_cast<v4i*>(wB + buf * (BK * SY_LD) + 8 * SY_LD) = rb1;
        }
    };
    const bool active = !(diag && (wi * 64 + 63 < wj * 32));
    const int nk = K / BK;
    gload();
    // acc[p][s][tj]: i = row0 + wi*64 + 32p + 2*l15 + s ; j = col0 + wj*32 + 2*(l4 + 4r) + tj
    v4d acc[2][2][2];
    if (active) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = row0 + wi * 64 + 32 * p + 2 * l15;
                    const int j = col0 + wj * 32 + 2 * (l4 + 4 * r) + tj;
                    const double2 c = *reinterpret_cast<const double2*>(&C[(size_t)j * ldc + i]);
                    acc[p][0][tj][r] = -c.x;
                    acc[p][1][tj][r] = -c.y;
                }
    }
    sstore(0);
    __syncthreads();
    const double* rA = &sA[0][wi * 64 + 2 * l15 + l4 * SY_LD];
    const double* rB = (diag ? &sA[0][0] : &sB[0][0]) + wj * 32 + 2 * l15 + l4 * SY_LD;
    auto compute = [&](int buf, bool st, int sbuf) {
        const double* tA = rA + buf * (BK * SY_LD);
        const double* tB = rB + buf * (BK * SY_LD);
        v2d fa[2], fb, na[2], nb;
        if (active) {
            fa[0] = *reinterpret_cast<const v2d*>(tA);
            fa[1] = *reinterpret_cast<const v2d*>(tA + 32);
            fb = *reinterpret_cast<const v2d*>(tB);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            if (active) {
                if (kk + 1 < BK / 4) {
                    na[0] = *reinterpret_cast<const v2d*>(tA + (kk + 1) * 4 * SY_LD);
                    na[1] = *reinterpret_cast<const v2d*>(tA + (kk + 1) * 4 * SY_LD + 32);
                    nb = *reinterpret_cast<const v2d*>(tB + (kk + 1) * 4 * SY_LD);
                    if (V & 1) __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int sx = 0; sx < 2; ++sx)
                            acc[p][sx][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[tj], fa[p][sx], acc[p][sx][tj], 0, 0, 0);
                if (kk + 1 < BK / 4) {
                    fa[0] = na[0];
                    fa[1] = na[1];
                    fb = nb;
                }
            }
            if ((V & 2) && kk == BK / 4 - 2 && st) {
                // the next panel goes to LDS BEFORE the last substep, so the LDS-write latency hides under its MFMAs
                __builtin_amdgcn_sched_barrier(0);
                sstore(sbuf);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!(V & 2) && st) sstore(sbuf);
    };
    if (V & 16) st_c1 = __builtin_amdgcn_s_memtime();
    for (int kt = 0; kt < nk; kt += 2) {
        gload();                                   // kt + 1 < nk always (nk even)
        compute(0, true, 1);
        __syncthreads();
        if (kt + 2 < nk) gload();
        compute(1, kt + 2 < nk, 0);
        __syncthreads();
    }
    if ((V & 16) && tid == 0 && stamps) {
        const unsigned long long c2 = __builtin_amdgcn_s_memtime(), r2 = __builtin_amdgcn_s_memrealtime();
        stamps[4 * blockIdx.x] = c2 - st_c0;
        stamps[4 * blockIdx.x + 1] = r2 - st_r0;
        stamps[4 * blockIdx.x + 2] = st_c1 - st_c0;
    }
    if (!active) return;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = row0 + wi * 64 + 32 * p + 2 * l15;
                const int j = col0 + wj * 32 + 2 * (l4 + 4 * r) + tj;
                if (diag && j > i + 1) continue;
                double2 c;
                c.x = -acc[p][0][tj][r];
                c.y = -acc[p][1][tj][r];
                if (diag && j == i + 1) C[(size_t)j * ldc + i + 1] = c.y;     // pair straddles the diagonal
                else *reinterpret_cast<double2*>(&C[(size_t)j * ldc + i]) = c;
            }
}

static int tiles_of(int gx, int gy) {
    int tiles = 0;
    for (int c0 = 0; c0 < gy; c0 += SY_SB) {
        const int W = gy - c0 < SY_SB ? gy - c0 : SY_SB;
        tiles += W * (W + 1) / 2 + (gx - c0 - W) * W;
    }
    return tiles;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int n = 6144, ld = 6146, K = argc > 2 ? atoi(argv[2]) : 512, batch = argc > 3 ? atoi(argv[3]) : 32;
    const int c1 = argc > 4 ? atoi(argv[4]) : 512;        // first trailing column
    const char* which = argc > 1 ? argv[1] : "all";
    const int m = n + 1 - c1, nc = n - c1;
    const size_t per = (size_t)ld * n;
    double* d;
    CK(hipMalloc(&d, per * batch * sizeof(double)));
    {
        std::vector<double> h(per);
        unsigned long long st = 88172645463325252ull;
        for (size_t k = 0; k < per; ++k) {
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            h[k] = ((double)(st >> 11) / 9007199254740992.0) * 2.0 - 1.0;
        }
        for (int b = 0; b < batch; ++b) CK(hipMemcpy(d + b * per, h.data(), per * sizeof(double), hipMemcpyHostToDevice));
    }
    const double* A = d + (size_t)(c1 - K) * ld + c1;      // panel columns c1-K .. c1-1, rows c1 ..
    double* C = d + (size_t)c1 * ld + c1;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double elems = (double)nc * m - 0.5 * (double)nc * (nc - 1);
    const double flop = 2.0 * K * elems * batch;
    const int gx = (m + 127) / 128, gy = (nc + 127) / 128;
    const int tiles = tiles_of(gx, gy);
    const long long rounds = ((long long)tiles * batch + 511) / 512;
    auto run = [&](const char* name, auto launch) {
        if (strcmp(which, "all") && strcmp(which, name)) return;
        launch();
        CK(hipStreamSynchronize(s));
        float ms1 = 0;
        CK(hipEventRecord(e0, s));
        launch();
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms1, e0, e1));
        int reps = (int)(1200.0 / (ms1 > 0.01 ? ms1 : 0.01));
        if (reps < 3) reps = 3;
        if (reps > 2000) reps = 2000;
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s first %.3f ms  sustained %.3f ms/launch  %.1f TF/s (useful flop)\n", name, ms1, ms / reps,
               flop / (ms / reps * 1e-3) / 1e12);
        fflush(stdout);
    };
    dim3 g3(gx, gy, batch), gs((unsigned)(rounds * 512), 1, 1);
    const long long bs = (long long)per;
    if (!strcmp(which, "all") || !strcmp(which, "check")) {
        // fast path against the reference kernel on matrix 0 (full tiles: m - 1 rows)
        std::vector<double> h0(per), h1(per), h2(per);
        CK(hipMemcpy(h0.data(), d, per * sizeof(double), hipMemcpyDeviceToHost));
        const int t1 = tiles_of((m - 1 + 127) / 128, gy);
        hipLaunchKernelGGL((k_syrk<4, 16, 0>), dim3(gx, gy, 1), dim3(512), 0, s, A, ld, C, ld, m - 1, nc, K, bs, bs, 0, 1);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h1.data(), d, per * sizeof(double), hipMemcpyDeviceToHost));
        CK(hipMemcpy(d, h0.data(), per * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL((k_syrk_fast<0>), dim3((unsigned)(((t1 + 511) / 512) * 512), 1, 1), dim3(512), 0, s, A, ld, C, ld, m - 1, nc, K, bs, bs, t1, 1);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h2.data(), d, per * sizeof(double), hipMemcpyDeviceToHost));
        CK(hipMemcpy(d, h0.data(), per * sizeof(double), hipMemcpyHostToDevice));
        double maxd = 0, maxc = 0;
        size_t changed = 0;
        for (size_t k = 0; k < per; ++k) {
            const double dd = fabs(h1[k] - h2[k]);
            if (dd > maxd) maxd = dd;
            if (h1[k] != h0[k]) { ++changed; if (fabs(h1[k]) > maxc) maxc = fabs(h1[k]); }
        }
        printf("check fast vs reference kernel: max |diff| %.3e over the whole array (%zu entries updated, max |value| %.3e)\n", maxd, changed, maxc);
    }
    run("w8_plain", [&] { hipLaunchKernelGGL((k_syrk<4, 16, 0>), g3, dim3(512), 0, s, A, ld, C, ld, m, nc, K, bs, bs, 0, batch); });
    run("w8_swz", [&] { hipLaunchKernelGGL((k_syrk<4, 16, 0>), gs, dim3(512), 0, s, A, ld, C, ld, m, nc, K, bs, bs, tiles, batch); });
    run("w4_swz", [&] { hipLaunchKernelGGL((k_syrk<2, 16, 0>), gs, dim3(256), 0, s, A, ld, C, ld, m, nc, K, bs, bs, tiles, batch); });
    run("w8_swz_nomask", [&] { hipLaunchKernelGGL((k_syrk<4, 16, 1>), gs, dim3(512), 0, s, A, ld, C, ld, m - 1, nc, K, bs, bs, tiles_of((m - 1 + 127) / 128, gy), batch); });
    run("w8_swz_nopreload", [&] { hipLaunchKernelGGL((k_syrk<4, 16, 2>), gs, dim3(512), 0, s, A, ld, C, ld, m, nc, K, bs, bs, tiles, batch); });
    run("w8_swz_nomask_nopre", [&] { hipLaunchKernelGGL((k_syrk<4, 16, 3>), gs, dim3(512), 0, s, A, ld, C, ld, m - 1, nc, K, bs, bs, tiles_of((m - 1 + 127) / 128, gy), batch); });
    run("w8_clean_plus32valu", [&] { hipLaunchKernelGGL((k_syrk<4, 16, 7>), gs, dim3(512), 0, s, A, ld, C, ld, m - 1, nc, K, bs, bs, tiles_of((m - 1 + 127) / 128, gy), batch); });
    run("w8_clean_plus128valu", [&] { hipLaunchKernelGGL((k_syrk<4, 16, 11>), gs, dim3(512), 0, s, A, ld, C, ld, m - 1, nc, K, bs, bs, tiles_of((m - 1 + 127) / 128, gy), batch); });
    run("w8_nomask_sched", [&] { hipLaunchKernelGGL((k_syrk<4, 16, 33>), gs, dim3(512), 0, s, A, ld, C, ld, m - 1, nc, K, bs, bs, tiles_of((m - 1 + 127) / 128, gy), batch); });
    run("w8_nomask_nopre_sched", [&] { hipLaunchKernelGGL((k_syrk<4, 16, 35>), gs, dim3(512), 0, s, A, ld, C, ld, m - 1, nc, K, bs, bs, tiles_of((m - 1 + 127) / 128, gy), batch); });
    run("w4_nomask_sched", [&] { hipLaunchKernelGGL((k_syrk<2, 16, 33>), gs, dim3(256), 0, s, A, ld, C, ld, m - 1, nc, K, bs, bs, tiles_of((m - 1 + 127) / 128, gy), batch); });
    run("fast", [&] { hipLaunchKernelGGL((k_syrk_fast<0>), gs, dim3(512), 0, s, A, ld, C, ld, m - 1, nc, K, bs, bs, tiles_of((m - 1 + 127) / 128, gy), batch); });
    run("fast_dephase_half", [&] { hipLaunchKernelGGL((k_syrk_fast<64>), gs, dim3(512), 0, s, A, ld, C, ld, m - 1, nc, K, bs, bs, tiles_of((m - 1 + 127) / 128, gy), batch); });
    run("fast_dephase_quarter", [&] { hipLaunchKernelGGL((k_syrk_fast<128>), gs, dim3(512), 0, s, A, ld, C, ld, m - 1, nc, K, bs, bs, tiles_of((m - 1 + 127) / 128, gy), batch); });
    run("fast_again", [&] { hipLaunchKernelGGL((k_syrk_fast<0>), gs, dim3(512), 0, s, A, ld, C, ld, m - 1, nc, K, bs, bs, tiles_of((m - 1 + 127) / 128, gy), batch); });
    run("fast_samepanel", [&] { hipLaunchKernelGGL((k_syrk_fast<32>), gs, dim3(512), 0, s, A, ld, C, ld, m - 1, nc, K, bs, bs, tiles_of((m - 1 + 127) / 128, gy), batch); });
    run("fast_midstore", [&] { hipLaunchKernelGGL((k_syrk_fast<2>), gs, dim3(512), 0, s, A, ld, C, ld, m - 1, nc, K, bs, bs, tiles_of((m - 1 + 127) / 128, gy), batch); });
    run("fast_midstore_sched", [&] { hipLaunchKernelGGL((k_syrk_fast<3>), gs, dim3(512), 0, s, A, ld, C, ld, m - 1, nc, K, bs, bs, tiles_of((m - 1 + 127) / 128, gy), batch); });
    run("fast_sched", [&] { hipLaunchKernelGGL((k_syrk_fast<1>), gs, dim3(512), 0, s, A, ld, C, ld, m - 1, nc, K, bs, bs, tiles_of((m - 1 + 127) / 128, gy), batch); });
    run("w4_swz_nomask_nopre", [&] { hipLaunchKernelGGL((k_syrk<2, 16, 3>), gs, dim3(256), 0, s, A, ld, C, ld, m - 1, nc, K, bs, bs, tiles_of((m - 1 + 127) / 128, gy), batch); });
    if (!strcmp(which, "all") || !strcmp(which, "clock")) {
        // in-kernel clock of the mainloop: delta s_memtime / delta s_memrealtime (100 MHz) per workgroup, median
        unsigned long long* st;
        const size_t nst = (size_t)rounds * 512;
        CK(hipMalloc(&st, nst * 2 * sizeof(unsigned long long)));
        CK(hipMemset(st, 0, nst * 2 * sizeof(unsigned long long)));
        for (int r = 0; r < 150; ++r)
            hipLaunchKernelGGL((k_syrk<4, 16, 16>), gs, dim3(512), 0, s, A, ld, C, ld, m, nc, K, bs, bs, tiles, batch, st);
        CK(hipStreamSynchronize(s));
        std::vector<unsigned long long> h(nst * 2);
        CK(hipMemcpy(h.data(), st, nst * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::vector<double> ghz, cyc;
        for (size_t k = 0; k < nst; ++k)
            if (h[2 * k + 1] > 0) {
                ghz.push_back((double)h[2 * k] / (double)h[2 * k + 1] * 0.1);
                cyc.push_back((double)h[2 * k]);
            }
        std::sort(ghz.begin(), ghz.end());
        std::sort(cyc.begin(), cyc.end());
        if (!ghz.empty())
            printf("in-kernel clock (w8_swz, masked+preload): median %.3f GHz (p10 %.3f, p90 %.3f); tile lifetime median %.0f cycles "
                   "(ideal 2 tiles/CU sharing: %d MFMA-cycles)\n", ghz[ghz.size() / 2], ghz[ghz.size() / 10], ghz[ghz.size() * 9 / 10],
                   cyc[cyc.size() / 2], 2 * (K / 4) * 64 * 64 / 4);
        hipFree(st);
    }
    if (!strcmp(which, "all") || !strcmp(which, "clockfast")) {
        unsigned long long* st;
        const int t1 = tiles_of((m - 1 + 127) / 128, gy);
        const size_t nst = (size_t)rounds * 512;
        CK(hipMalloc(&st, nst * 4 * sizeof(unsigned long long)));
        CK(hipMemset(st, 0, nst * 4 * sizeof(unsigned long long)));
        for (int r = 0; r < 150; ++r)
            hipLaunchKernelGGL((k_syrk_fast<16>), gs, dim3(512), 0, s, A, ld, C, ld, m - 1, nc, K, bs, bs, t1, batch, st);
        CK(hipStreamSynchronize(s));
        std::vector<unsigned long long> h(nst * 4);
        CK(hipMemcpy(h.data(), st, nst * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::vector<double> ghz, cyc, pro;
        for (size_t k = 0; k < nst; ++k)
            if (h[4 * k + 1] > 0) {
                ghz.push_back((double)h[4 * k] / (double)h[4 * k + 1] * 0.1);
                cyc.push_back((double)h[4 * k]);
                pro.push_back((double)h[4 * k + 2]);
            }
        std::sort(ghz.begin(), ghz.end());
        std::sort(cyc.begin(), cyc.end());
        std::sort(pro.begin(), pro.end());
        if (!ghz.empty())
            printf("fast kernel: in-kernel clock median %.3f GHz (p10 %.3f, p90 %.3f); start->end of k-loop median %.0f cycles (p10 %.0f p90 %.0f), "
                   "prologue median %.0f (p90 %.0f); ideal shared k-loop %d cycles\n", ghz[ghz.size() / 2], ghz[ghz.size() / 10],
                   ghz[ghz.size() * 9 / 10], cyc[cyc.size() / 2], cyc[cyc.size() / 10], cyc[cyc.size() * 9 / 10], pro[pro.size() / 2],
                   pro[pro.size() * 9 / 10], 2 * (K / 4) * 64 * 64 / 4);
        hipFree(st);
    }
    {
        rocblas_handle h;
        rocblas_create_handle(&h);
        rocblas_set_stream(h, s);
        const double mone = -1.0, one = 1.0;
        const double full = 2.0 * K * (double)nc * nc * batch;
        if (!strcmp(which, "all") || !strcmp(which, "rocblas")) {
            auto launch = [&] {
                rocblas_dgemm_strided_batched(h, rocblas_operation_none, rocblas_operation_transpose, nc, nc, K, &mone, A, ld,
                                              bs, A, ld, bs, &one, C, ld, bs, batch);
            };
            launch();
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            const int reps = 60;
            for (int r = 0; r < reps; ++r) launch();
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-28s sustained %.3f ms/launch  %.1f TF/s (FULL square flop; useful-equivalent %.1f)\n", "rocblas_dgemm_sb",
                   ms / reps, full / (ms / reps * 1e-3) / 1e12, flop / (ms / reps * 1e-3) / 1e12);
        }
        rocblas_destroy_handle(h);
    }
    return 0;
}
