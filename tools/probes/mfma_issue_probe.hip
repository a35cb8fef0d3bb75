// How fast does ONE wave per SIMD issue v_mfma_f32_16x16x32_bf16 with LDS fragment reads interleaved?  (round 4, rec_conv3.hip design)
// MODE 0: 52 independent MFMAs per step, operands fixed.  1: + 13 ds_read_b128 per step into a second register set (used next step).
// 2: as 1, reads of a tile issued BEFORE its MFMAs.  3: 4 MFMAs per B tile ordered b-outer a-inner with s_setprio.  4: MODE 1 with 2 waves/SIMD.
// prints cycles per step (s_memtime) for wave 0 of block 0 and the kernel time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int MH, int NTW, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void k(float* out, long long* cyc, int steps, unsigned stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
    __syncthreads();
    f32x4 acc[MH][NTW];
#pragma unroll
    for (int a = 0; a < MH; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint4 af[MH], bqA[NTW], bqB[NTW];
    unsigned baddr[NTW];
#pragma unroll
    for (int b = 0; b < NTW; ++b) baddr[b] = (unsigned)((b * 64 + lane) * 16) & 0xffff;
#pragma unroll
    for (int a = 0; a < MH; ++a) af[a] = *reinterpret_cast<const uint4*>(smem + a * 1024 + lane * 16);
#pragma unroll
    for (int b = 0; b < NTW; ++b) bqA[b] = bqB[b] = *reinterpret_cast<const uint4*>(smem + baddr[b]);
    auto step = [&](const uint4 (&bq)[NTW], uint4 (&bqn)[NTW], unsigned off) {
#pragma unroll
        for (int b = 0; b < NTW; ++b) {
            if (MODE == 2) bqn[b] = *reinterpret_cast<const uint4*>(smem + ((baddr[b] + off) & 0xffff));
#pragma unroll
            for (int a = 0; a < MH; ++a)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[a]), __builtin_bit_cast(bf16x8, bq[b]), acc[a][b], 0, 0, 0);
            if (MODE == 1 || MODE == 4) bqn[b] = *reinterpret_cast<const uint4*>(smem + ((baddr[b] + off) & 0xffff));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    long long t0 = __builtin_readcyclecounter();
    unsigned off = 0;
    for (int s = 0; s < steps; s += 2) {
        step(bqA, bqB, off);
        off += stride;
        step(bqB, bqA, off);
        off += stride;
    }
    long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
#pragma unroll
    for (int a = 0; a < MH; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b) r += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    out[blockIdx.x * blockDim.x + tid] = r;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int MH, int NTW, int threads>
void run(const char* name) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    auto kern = k<MODE, MH, NTW, threads>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int steps = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 65536, 0, out, cyc, steps, 1040u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 65536, 0, out, cyc, steps, 1040u);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    const double fl = 2.0 * 16 * 16 * 32 * MH * NTW * (double)steps * (threads / 64) * 256;
    printf("%-44s %2dx%2d tiles, %d waves/CU: %8.1f us  %7.1f TF/s  s_memtime ticks/step %.1f (100 MHz ticks?)  per-MFMA ns %.2f\n", name, MH, NTW, threads / 64, ms * 1e3,
           fl / (ms * 1e-3) / 1e12, (double)h[0] / steps, ms * 1e6 / steps / (MH * NTW));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0, 4, 13, 256>("MFMA only");
    run<1, 4, 13, 256>("+13 ds_read_b128 behind the tile's MFMAs");
    run<0, 8, 8, 256>("MFMA only");
    run<0, 4, 7, 256>("MFMA only");
    run<0, 4, 7, 512>("MFMA only");
    run<1, 4, 7, 512>("+ds_read behind");
    run<2, 4, 7, 512>("+ds_read before");
    run<0, 4, 6, 512>("MFMA only");
    run<0, 4, 8, 512>("MFMA only");
    run<1, 4, 8, 512>("+ds_read behind");
    run<0, 4, 7, 1024>("MFMA only");
    run<1, 4, 7, 1024>("+ds_read behind");
    run<0, 2, 7, 1024>("MFMA only");
    return 0;
}
