// Probe of __builtin_amdgcn_global_load_lds semantics on gfx950: where does each lane's 16 bytes land?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned* __restrict__ src, unsigned* __restrict__ out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned lds[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // every lane reads a DIFFERENT (permuted) 16-byte source chunk: chunk index = (lane * 7) % 64 + 64*wave
    const unsigned* g = src + (size_t)(((lane * 7) % 64) + 64 * wave) * 4;
    unsigned* l = lds + wave * 256 + (mode == 0 ? 0 : lane * 4);  // mode 0: wave-uniform base; mode 1: per-lane pointer
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) out[i] = lds[i];
}
int main() {
    std::vector<unsigned> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;  // word i of chunk c = 4c + (i%4)
    unsigned *d, *o;
    hipMalloc(&d, 4096 * 4); hipMalloc(&o, 2048 * 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, d, o, mode);
        std::vector<unsigned> r(2048);
        hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
        printf("mode %d (%s):\n", mode, mode == 0 ? "uniform LDS base" : "per-lane LDS pointer");
        for (int w = 0; w < 2; ++w) {
            printf(" wave %d lds words [0..15]:", w);
            for (int i = 0; i < 16; ++i) printf(" %u", r[w * 256 + i]);
            printf("  ... word[252..255]: %u %u %u %u\n", r[w * 256 + 252], r[w * 256 + 253], r[w * 256 + 254], r[w * 256 + 255]);
            int ok = 1;
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) ok &= r[w * 256 + lane * 4 + j] == (unsigned)(4 * (((lane * 7) % 64) + 64 * w) + j);
            printf("   lane-linear (base + lane*16) layout matches: %d\n", ok);
        }
    }
    return 0;
}
