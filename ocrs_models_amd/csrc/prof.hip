// Per-launch kernel durations without stream events (measurement support for bench.py's roofline object).
//
// A hipEventRecord pair around a launch puts two barrier packets into the queue: ~4 us of bubble per bracketed launch, 2-3 % of the
// detection step when the ~80 launches of the dominant pass are bracketed.  hipExtLaunchKernelGGL instead takes the start / stop timestamps
// from the dispatch packet's own completion signal: the kernels stay back to back.  The launch sites of the pass families use OCRS_LAUNCH_T
// (common.h), which is a plain hipLaunchKernelGGL unless profiling is switched on here.
#include "common.h"

static OcrsProf g_prof = {0, 0, 0, nullptr};
OcrsProf& ocrs_prof() { return g_prof; }

extern "C" {

// on != 0: start recording (resets the launch counter; the event pool -- `cap` pairs -- is created on first use); 0: stop
int ocrs_prof_enable(int on) {
    constexpr int CAP = 8192;
    if (on && !g_prof.ev) {
        g_prof.ev = new hipEvent_t[2 * CAP];
        for (int i = 0; i < 2 * CAP; ++i)
            if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) return OCRS_ERR_HIP;
        g_prof.cap = CAP;
    }
    if (on) g_prof.used = 0;
    g_prof.on = on ? 1 : 0;
    return OCRS_OK;
}
// launches recorded since ocrs_prof_enable(1)
long ocrs_prof_count() { return g_prof.used; }
// durations (ms) of recorded launches [first, first + n) into the HOST array ms; synchronises the stream
int ocrs_prof_read(float* ms, long first, long n, hipStream_t st) {
    OCRS_CHECK_ARG(ms && first >= 0 && n >= 0 && first + n <= g_prof.used);
    if (hipStreamSynchronize(st) != hipSuccess) return OCRS_ERR_HIP;
    for (long i = 0; i < n; ++i)
        if (hipEventElapsedTime(&ms[i], g_prof.ev[2 * (first + i)], g_prof.ev[2 * (first + i) + 1]) != hipSuccess) return OCRS_ERR_HIP;
    return OCRS_OK;
}

}  // extern "C"
