// Laboratory (NOT part of the product library): what a captured hipGraph buys for the launch pattern of the latency-bound
// factorisation -- (A) a chain of dependent short kernels on one stream, (B) the look-ahead pattern: panel steps on the main stream,
// a long "far update" on a second stream that starts after the panel's "near update" and has to end before the next near update.
// Each pattern is timed as plain stream launches and as ONE launch of the same work captured into a graph.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/graph_lab.hip -o tools/lab/graph_lab
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e__ = (x);                                                                   \
        if (e__ != hipSuccess) {                                                                \
            std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e__)); \
            std::exit(1);                                                                       \
        }                                                                                       \
    } while (0)

// every workgroup spins for `ticks` of the 100 MHz wall clock (10 ns each), then touches memory so that it is not optimised away
__global__ void k_spin(long long ticks, int* sink) {
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < ticks) {
    }
    if (threadIdx.x == 0 && sink) sink[blockIdx.x & 1023] = (int)t0;
}

static void spin(hipStream_t s, int wgs, double us, int* sink) {
    hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(256), 0, s, (long long)(us * 100.0), sink);
}

struct Pattern {
    int panels, steps;
    double step_us, near_us, far_us;
};

// pattern A: panels * steps dependent kernels on s1
static void chain(hipStream_t s1, const Pattern& p, int* sink) {
    for (int i = 0; i < p.panels * p.steps; ++i) spin(s1, 200, p.step_us, sink);
}

// pattern B: per panel: steps on s1; near update on s1 (after the previous far update); far update on s2 behind it
static void lookahead(hipStream_t s1, hipStream_t s2, hipEvent_t* ev, const Pattern& p, int* sink) {
    for (int k = 0; k < p.panels; ++k) {
        for (int i = 0; i < p.steps; ++i) spin(s1, 200, p.step_us, sink);
        if (k > 0) CK(hipStreamWaitEvent(s1, ev[2 * k - 1], 0));      // the previous far update has to be through
        spin(s1, 600, p.near_us, sink);
        CK(hipEventRecord(ev[2 * k], s1));
        CK(hipStreamWaitEvent(s2, ev[2 * k], 0));
        spin(s2, 192, p.far_us, sink);                                // (the product masks this stream to 192 CUs)
        CK(hipEventRecord(ev[2 * k + 1], s2));
    }
    CK(hipStreamWaitEvent(s1, ev[2 * p.panels - 1], 0));
}

static unsigned int *g_flag = nullptr, *g_flag2 = nullptr;     // signal memory for the stream-memory-operation variants
static unsigned int g_seq = 0;

// pattern B without the second stream (what the launches alone cost), and with the events but an empty far update
static void lookahead_variant(hipStream_t s1, hipStream_t s2, hipEvent_t* ev, const Pattern& p, int* sink, int variant) {
    for (int k = 0; k < p.panels; ++k) {
        for (int i = 0; i < p.steps; ++i) spin(s1, 200, p.step_us, sink);
        if (variant == 5 && k > 0) CK(hipStreamWaitValue32(s1, g_flag2, g_seq, hipStreamWaitValueGte, 0xffffffffu));
        else if (variant >= 1 && variant != 3 && k > 0) CK(hipStreamWaitEvent(s1, ev[2 * k - 1], 0));
        spin(s1, 600, p.near_us, sink);
        if (variant == 4 || variant == 5) {          // main -> second stream through a value in memory instead of an event
            ++g_seq;
            CK(hipStreamWriteValue32(s1, g_flag, g_seq, 0));
            CK(hipStreamWaitValue32(s2, g_flag, g_seq, hipStreamWaitValueGte, 0xffffffffu));
            spin(s2, 1, 0.5, sink);
            if (variant == 5) CK(hipStreamWriteValue32(s2, g_flag2, g_seq, 0));      // ... and back the same way
            CK(hipEventRecord(ev[2 * k + 1], s2));
        } else if (variant >= 1) {
            CK(hipEventRecord(ev[2 * k], s1));
            CK(hipStreamWaitEvent(s2, ev[2 * k], 0));
            spin(s2, variant == 2 ? 192 : 1, variant == 2 ? p.far_us : 0.5, sink);
            CK(hipEventRecord(ev[2 * k + 1], s2));
        }
    }
    if (variant >= 1) CK(hipStreamWaitEvent(s1, ev[2 * p.panels - 1], 0));
}

template <class F>
static double time_ms(hipStream_t s1, int reps, F&& f) {
    CK(hipStreamSynchronize(s1));
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) f();
    CK(hipStreamSynchronize(s1));
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
}

int main() {
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    int* sink = nullptr;
    CK(hipMalloc((void**)&sink, 1024 * sizeof(int)));
    const Pattern p{12, 8, 20.0, 60.0, 150.0};
    hipEvent_t ev[64];
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const int reps = 20;
    const double busy_a = p.panels * p.steps * p.step_us * 1e-3;
    const double busy_b = p.panels * (p.steps * p.step_us + p.near_us) * 1e-3;      // the far updates hide under the steps

    // plain streams
    chain(s1, p, sink);
    const double a_stream = time_ms(s1, reps, [&] { chain(s1, p, sink); });
    lookahead(s1, s2, ev, p, sink);
    const double b_stream = time_ms(s1, reps, [&] { lookahead(s1, s2, ev, p, sink); });

    // the same work captured once, launched as a graph
    hipGraph_t ga, gb;
    hipGraphExec_t xa, xb;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
    chain(s1, p, sink);
    CK(hipStreamEndCapture(s1, &ga));
    CK(hipGraphInstantiate(&xa, ga, nullptr, nullptr, 0));
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
    lookahead(s1, s2, ev, p, sink);
    CK(hipStreamEndCapture(s1, &gb));
    CK(hipGraphInstantiate(&xb, gb, nullptr, nullptr, 0));
    CK(hipGraphLaunch(xa, s1));
    const double a_graph = time_ms(s1, reps, [&] { CK(hipGraphLaunch(xa, s1)); });
    CK(hipGraphLaunch(xb, s1));
    const double b_graph = time_ms(s1, reps, [&] { CK(hipGraphLaunch(xb, s1)); });

    std::printf("pattern A: %d dependent kernels of %.0f us on one stream (busy time %.3f ms)\n", p.panels * p.steps, p.step_us, busy_a);
    std::printf("  stream launches %.3f ms  (+%.2f us per kernel)\n", a_stream, (a_stream - busy_a) * 1e3 / (p.panels * p.steps));
    std::printf("  one graph       %.3f ms  (+%.2f us per kernel)\n", a_graph, (a_graph - busy_a) * 1e3 / (p.panels * p.steps));
    std::printf("pattern B: %d panels of %d steps (%.0f us) + near update (%.0f us) on the main stream, far update (%.0f us) on a second "
                "stream between two events (critical path %.3f ms)\n", p.panels, p.steps, p.step_us, p.near_us, p.far_us, busy_b);
    std::printf("  stream launches %.3f ms  (+%.2f us per panel)\n", b_stream, (b_stream - busy_b) * 1e3 / p.panels);
    std::printf("  one graph       %.3f ms  (+%.2f us per panel)\n", b_graph, (b_graph - busy_b) * 1e3 / p.panels);
    CK(hipExtMallocWithFlags((void**)&g_flag, 8, hipMallocSignalMemory));
    CK(hipMemset(g_flag, 0, 8));
    CK(hipExtMallocWithFlags((void**)&g_flag2, 8, hipMallocSignalMemory));
    CK(hipMemset(g_flag2, 0, 8));
    for (int v = 0; v < 6; ++v) {
        lookahead_variant(s1, s2, ev, p, sink, v);
        const double t = time_ms(s1, reps, [&] { lookahead_variant(s1, s2, ev, p, sink, v); });
        std::printf("  %-58s %.3f ms  (+%.2f us per panel)\n",
                    v == 0 ? "main stream alone (no second stream, no events)"
                           : (v == 1 ? "events + a 0.5 us one-workgroup kernel on the second stream"
                                     : (v == 2 ? "as pattern B (again)"
                                                : (v == 3 ? "as the second line, main stream never waits for the second"
                                                          : (v == 4 ? "as the second line, main -> second through hipStreamWriteValue32 / WaitValue32"
                                                                    : "... and second -> main through memory values as well")))),
                    t, (t - busy_b) * 1e3 / p.panels);
    }
    return 0;
}
