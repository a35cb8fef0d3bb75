// Probe: semantics of ds_read_b64_tr_b16 (gfx950 LDS transpose read).  Build: hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(int* out, int row_stride_el) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, grp = l >> 4;
    // group grp reads the 4x16 block whose rows are  (grp*4 + r), row stride = row_stride_el elements, cols 0..15
    const short* p = lds + (grp * 4 + (i >> 2)) * row_stride_el + (i & 3) * 4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    int* d; hipMalloc(&d, 64 * 4 * sizeof(int));
    for (int rs : {16, 40}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, rs);
        int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("row stride %d\n", rs);
        for (int l = 0; l < 64; l += 1) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %5d(r%d,c%d)", h[l*4+j], h[l*4+j] / rs, h[l*4+j] % rs); printf("\n"); }
    }
    return 0;
}
