// Per-launch kernel durations without stream events (measurement support for bench.py's roofline object).
//
// A hipEventRecord pair around a launch puts two barrier packets into the queue: ~4 us of bubble per bracketed launch, 2-3 % of the
// detection step when the ~80 launches of the dominant pass are bracketed.  hipExtLaunchKernelGGL instead takes the start / stop timestamps
// from the dispatch packet's own completion signal: the kernels stay back to back.  The launch sites of the pass families use OCRS_LAUNCH_T
// (common.h), which is a plain hipLaunchKernelGGL unless profiling is switched on here.
#include "common.h"

static OcrsProf g_prof = {0, 0, 0, nullptr};

// Stand-in for a collective's resident channel kernels: `blocks` workgroups that each hold a workgroup slot for `micros` microseconds
// (wall clock, s_memrealtime at 100 MHz) and do nothing else.  tests/test_train_loop_gpu.py runs the persistent, spin-waiting GRU launches next to
// it: a 1-rank RCCL all-reduce moves no data and may launch no kernel at all, so on a single-GPU box it cannot show what 64 resident RCCL
// channels do to a launch that needs its whole grid co-resident.
__global__ __launch_bounds__(256) void k_cu_hog(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
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

// measurement / test support: `blocks` x 256-thread workgroups resident for `micros` microseconds on stream st (see k_cu_hog)
int ocrs_cu_hog(int blocks, int micros, hipStream_t st) {
    OCRS_CHECK_ARG(blocks > 0 && blocks <= 4096 && micros >= 0 && micros <= 1000000);
    hipLaunchKernelGGL(k_cu_hog, dim3(blocks), dim3(256), 0, st, (unsigned long long)micros * 100ull);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

}  // extern "C"
