// HBM ceiling lab for gfx950 (developer aid, not part of libnmgp_hip.so):
//     hipcc --offload-arch=gfx950 -O3 -o tools/lab/hbm_lab tools/lab/hbm_lab.hip && tools/lab/hbm_lab [MiB per buffer, default 1024]
// Times read-only, write-only and copy kernels over buffers far larger than the 256 MiB Infinity Cache, varying
//   U   = independent 16-byte accesses in flight per lane (1, 2, 4, 8),
//   WPC = workgroups of 256 threads per CU the grid is sized for (grid-stride loop over the buffer), or "flat" (one chunk per
//         workgroup, no loop),
//   NT  = nontemporal loads / stores.
// The library's own ceiling kernel (k_stream_copy in nmgp_kernels.hip) takes the best shape found here.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef double v2d __attribute__((ext_vector_type(2)));

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            std::exit(1);                                                              \
        }                                                                              \
    } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const v2d* __restrict__ src, v2d* __restrict__ dst, size_t n2) {
    // a workgroup moves chunks of U * 256 vectors; within a chunk the U accesses of a lane are 256 vectors apart (coalesced)
    const size_t chunk = (size_t)U * 256;
    for (size_t c = blockIdx.x; c * chunk < n2; c += gridDim.x) {
        const size_t base = c * chunk + threadIdx.x;
        v2d v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t k = base + (size_t)u * 256;
            if (k < n2) v[u] = NT ? __builtin_nontemporal_load(&src[k]) : src[k];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t k = base + (size_t)u * 256;
            if (k < n2) {
                if (NT) __builtin_nontemporal_store(v[u], &dst[k]);
                else dst[k] = v[u];
            }
        }
    }
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const v2d* __restrict__ src, double* __restrict__ sink, size_t n2) {
    const size_t chunk = (size_t)U * 256;
    double acc = 0.0;
    for (size_t c = blockIdx.x; c * chunk < n2; c += gridDim.x) {
        const size_t base = c * chunk + threadIdx.x;
        v2d v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t k = base + (size_t)u * 256;
            v[u] = (k < n2) ? (NT ? __builtin_nontemporal_load(&src[k]) : src[k]) : v2d{0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u][0] + v[u][1];
    }
    if (acc == 1.2345e300) sink[threadIdx.x] = acc;       // never true for the zero-filled source; keeps the loads alive
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_write(v2d* __restrict__ dst, size_t n2, double val) {
    const size_t chunk = (size_t)U * 256;
    const v2d v = {val, val};
    for (size_t c = blockIdx.x; c * chunk < n2; c += gridDim.x) {
        const size_t base = c * chunk + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t k = base + (size_t)u * 256;
            if (k < n2) {
                if (NT) __builtin_nontemporal_store(v, &dst[k]);
                else dst[k] = v;
            }
        }
    }
}

// 8-byte stores, 512 contiguous bytes per wave-instruction (the access shape of k_svc_cov and of the C tiles)
template <int U>
__global__ __launch_bounds__(256) void k_write8(double* __restrict__ dst, size_t n, double val) {
    const size_t chunk = (size_t)U * 256;
    for (size_t c = blockIdx.x; c * chunk < n; c += gridDim.x) {
        const size_t base = c * chunk + threadIdx.x;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t k = base + (size_t)u * 256;
            if (k < n) dst[k] = val;
        }
    }
}

// ---- shaped access: a workgroup owns an R-row x CN-column block of a column-major matrix (leading dimension ld), i.e. CN
// segments of 8 R contiguous bytes, a column apart -- the shape of the panel kernels (k_trsm_64f: R = 128, CN = 64; the update
// kernel's C tile: 128 x 128; k_svc_cov: 64-row segments).  MODE 0: write, 1: read-modify-write in place, 2: read only.
template <int MODE>
__global__ __launch_bounds__(256) void k_shaped(double* __restrict__ A, int ld, int rows, int R, int CN, double* __restrict__ sink) {
    // blockIdx.x = row block (fastest, as in the library's launches), blockIdx.y = column block
    const int r0 = blockIdx.x * R, c0 = blockIdx.y * CN;
    const int pairs = R / 2;                        // 16-byte accesses per column segment
    const int per_col = pairs;                      // threads needed per column
    const int cols_par = 256 / per_col > 0 ? 256 / per_col : 1;     // columns handled concurrently
    double acc = 0.0;
    if (per_col <= 256) {
        const int cl = threadIdx.x / per_col, pr = threadIdx.x % per_col;
        if (cl < cols_par)
            for (int c = cl; c < CN; c += cols_par) {
                v2d* p = reinterpret_cast<v2d*>(A + (size_t)(c0 + c) * ld + r0 + 2 * pr);
                if (r0 + 2 * pr + 1 < rows) {
                    if (MODE == 0) *p = v2d{1.5, 2.5};
                    else if (MODE == 1) { v2d v = *p; v[0] = v[0] * 1.0000001; v[1] = v[1] + 1.0; *p = v; }
                    else { v2d v = *p; acc += v[0] + v[1]; }
                }
            }
    } else {
        for (int c = 0; c < CN; ++c)
            for (int pr = threadIdx.x; pr < pairs; pr += 256) {
                v2d* p = reinterpret_cast<v2d*>(A + (size_t)(c0 + c) * ld + r0 + 2 * pr);
                if (r0 + 2 * pr + 1 < rows) {
                    if (MODE == 0) *p = v2d{1.5, 2.5};
                    else if (MODE == 1) { v2d v = *p; v[0] = v[0] * 1.0000001; v[1] = v[1] + 1.0; *p = v; }
                    else { v2d v = *p; acc += v[0] + v[1]; }
                }
            }
    }
    if (MODE == 2 && acc == 1.2345e300) sink[threadIdx.x] = acc;
}

struct Timer {
    hipEvent_t a, b;
    Timer() {
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
    }
    template <class F>
    double run(F f, int reps) {
        f();
        CK(hipEventRecord(a, 0));
        for (int r = 0; r < reps; ++r) f();
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, a, b));
        return ms * 1e-3 / reps;
    }
};

template <int U, bool NT>
static void sweep(Timer& T, const v2d* src, v2d* dst, double* sink, size_t n2, int ncu) {
    const double bytes = (double)n2 * 16.0;
    const size_t chunks = (n2 + (size_t)U * 256 - 1) / ((size_t)U * 256);
    const int wpcs[] = {2, 4, 8, 16, 0};        // 0 = flat: one chunk per workgroup
    for (int wpc : wpcs) {
        const unsigned grid = wpc ? (unsigned)(ncu * wpc) : (unsigned)chunks;
        const double tr = T.run([&] { hipLaunchKernelGGL((k_read<U, NT>), dim3(grid), dim3(256), 0, 0, src, sink, n2); }, 5);
        const double tw = T.run([&] { hipLaunchKernelGGL((k_write<U, NT>), dim3(grid), dim3(256), 0, 0, dst, n2, 1.5); }, 5);
        const double tc = T.run([&] { hipLaunchKernelGGL((k_copy<U, NT>), dim3(grid), dim3(256), 0, 0, src, dst, n2); }, 5);
        std::printf("U=%d NT=%d wpc=%-4s grid=%-8u read %7.1f  write %7.1f  copy %7.1f GB/s\n", U, (int)NT,
                    wpc ? std::to_string(wpc).c_str() : "flat", grid, bytes / tr / 1e9, bytes / tw / 1e9, 2.0 * bytes / tc / 1e9);
        std::fflush(stdout);
    }
}

int main(int argc, char** argv) {
    const size_t mib = argc > 1 ? (size_t)std::atoll(argv[1]) : 1024;
    const size_t n2 = mib * (1u << 20) / 16;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    std::printf("%s, %d CUs, %zu MiB per buffer\n", prop.gcnArchName, ncu, mib);
    v2d *src, *dst;
    double* sink;
    CK(hipMalloc((void**)&src, n2 * 16));
    CK(hipMalloc((void**)&dst, n2 * 16));
    CK(hipMalloc((void**)&sink, 4096));
    CK(hipMemset(src, 0, n2 * 16));
    CK(hipMemset(dst, 0, n2 * 16));
    Timer T;
    const bool shaped_only = argc > 2 && std::string(argv[2]) == "shaped";
    if (!shaped_only) {
    sweep<1, false>(T, src, dst, sink, n2, ncu);
    sweep<2, false>(T, src, dst, sink, n2, ncu);
    sweep<4, false>(T, src, dst, sink, n2, ncu);
    sweep<8, false>(T, src, dst, sink, n2, ncu);
    sweep<4, true>(T, src, dst, sink, n2, ncu);
    sweep<8, true>(T, src, dst, sink, n2, ncu);
    {
        const size_t n = n2 * 2;
        const double bytes = (double)n * 8.0;
        for (int wpc : {8, 16, 0}) {
            const unsigned g4 = wpc ? (unsigned)(ncu * wpc) : (unsigned)((n + 1023) / 1024);
            const unsigned g8 = wpc ? (unsigned)(ncu * wpc) : (unsigned)((n + 2047) / 2048);
            const double t4 = T.run([&] { hipLaunchKernelGGL((k_write8<4>), dim3(g4), dim3(256), 0, 0, (double*)dst, n, 2.5); }, 5);
            const double t8 = T.run([&] { hipLaunchKernelGGL((k_write8<8>), dim3(g8), dim3(256), 0, 0, (double*)dst, n, 2.5); }, 5);
            std::printf("8-byte stores wpc=%-4s U=4 %7.1f  U=8 %7.1f GB/s\n", wpc ? std::to_string(wpc).c_str() : "flat",
                        bytes / t4 / 1e9, bytes / t8 / 1e9);
        }
    }
    }
    // shaped access over a batch of column-major matrices: 6144 rows, ld 6160, 96 matrices x 64 columns per launch (as one panel
    // solve of the 128-chain batch... scaled to fill 1 GiB: columns = as many as fit)
    {
        const int rows = 6144, ld = 6160;
        const size_t total = n2 * 2;                     // doubles in dst
        const int cols = (int)(total / ld);
        double* A = (double*)dst;
        std::printf("shaped: %d rows (ld %d) x %d columns\n", rows, ld, cols);
        const int Rs[] = {64, 128, 256, 512, 1024, 2048, 6144};
        for (int R : Rs) {
            const int CN = 8192 / R > 0 ? 8192 / R : 1;  // 64 KB per workgroup
            const dim3 grid((rows + R - 1) / R, cols / CN);
            const double bytes = (double)rows * 8.0 * (double)(grid.y * CN);
            const double tw = T.run([&] { hipLaunchKernelGGL((k_shaped<0>), grid, dim3(256), 0, 0, A, ld, rows, R, CN, sink); }, 5);
            const double tm = T.run([&] { hipLaunchKernelGGL((k_shaped<1>), grid, dim3(256), 0, 0, A, ld, rows, R, CN, sink); }, 5);
            const double tr = T.run([&] { hipLaunchKernelGGL((k_shaped<2>), grid, dim3(256), 0, 0, A, ld, rows, R, CN, sink); }, 5);
            std::printf("R=%-5d CN=%-4d segment %6d B   write %7.1f   read+write in place %7.1f   read %7.1f GB/s\n", R, CN, R * 8,
                        bytes / tw / 1e9, 2.0 * bytes / tm / 1e9, bytes / tr / 1e9);
            std::fflush(stdout);
        }
    }
    // hipMemcpyAsync device-to-device and hipMemsetAsync for reference
    {
        const double bytes = (double)n2 * 16.0;
        const double tm = T.run([&] { CK(hipMemcpyAsync(dst, src, n2 * 16, hipMemcpyDeviceToDevice, 0)); }, 5);
        const double ts = T.run([&] { CK(hipMemsetAsync(dst, 0, n2 * 16, 0)); }, 5);
        std::printf("hipMemcpyAsync D2D %7.1f GB/s (read+write)   hipMemsetAsync %7.1f GB/s\n", 2.0 * bytes / tm / 1e9, bytes / ts / 1e9);
    }
    return 0;
}
