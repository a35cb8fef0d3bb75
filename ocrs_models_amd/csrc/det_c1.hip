// First block of the detection net (DepthwiseConv 1 -> 8 on the greyscale image, ocrs_models/models.py:115 in_conv.seq.0), forward and backward,
// in the "wave = 64 columns x 2 rows" form (gfx950; W % 64 == 0, H % 2 == 0; other shapes keep k_dwpw_c1_fwd / k_c1_bwd).
//
// The per-pixel kernels issue 9 image loads (+ up to 6 sixteen-byte loads in the backward, four of them redundant re-reads of the generic
// gradient-source form) for every 16 / 32 useful bytes: 168 VGPRs, 3 waves per SIMD, 3.2 TB/s.  Here a wave owns 64 consecutive columns of a
// row PAIR: it loads the four image rows around the pair once (one dword per lane and row), takes the left / right neighbours from the adjacent
// lanes with DPP wave shifts (the two columns outside the segment come from ONE extra load by lanes 0..7 and are handed to lanes 0 / 63 as
// the shifts' out-of-range value), and streams the two pixels' z / gradient vectors with fully coalesced 16-byte loads: 4.5 instead of 15
// load instructions per pixel, half the registers, the same per-pixel arithmetic (same tap order, same roundings) as the per-pixel kernels.
#include "det_common.h"

namespace {
constexpr int DPP_WAVE_SHL1 = 0x130, DPP_WAVE_SHR1 = 0x138;  // result[i] = src[i + 1] / src[i - 1]; lanes without a source keep `old`

// the four image rows around a row pair at this lane's column + the 8 halo values (lane l < 8: row l >> 1, side l & 1)
struct C1Img {
    float c[4], halo;
};
struct C1Item {
    int n, h0, w0;  // image, first row of the pair, first column of the segment (wave-uniform)
    bool act;
};
__device__ __forceinline__ C1Item c1_item(int item, int items, int HP2, int WS) {
    C1Item it;
    it.act = item < items;
    const unsigned i = it.act ? (unsigned)item : 0u;
    const unsigned r = i / (unsigned)WS, seg = i - r * (unsigned)WS;
    const unsigned n = r / (unsigned)HP2;
    it.h0 = 2 * (int)(r - n * (unsigned)HP2);
    it.n = (int)n;
    it.w0 = (int)seg * 64;
    return it;
}
__device__ __forceinline__ void c1_issue_img(C1Img& im, const float* __restrict__ img, const C1Item& it, int H, int W, int lane) {
    const float* base = img + (long)it.n * H * W;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int hr = it.h0 - 1 + r;
        const bool ok = it.act && (unsigned)hr < (unsigned)H;
        im.c[r] = base[ok ? (long)hr * W + it.w0 + lane : 0];  // (unconditional load, selected address: see issue_ghat8)
    }
    const int hr = it.h0 - 1 + (lane >> 1), col = (lane & 1) ? it.w0 + 64 : it.w0 - 1;
    const bool ok = it.act && lane < 8 && (unsigned)hr < (unsigned)H && (unsigned)col < (unsigned)W;
    im.halo = base[ok ? (long)hr * W + col : 0];
}
// nb[r][0..2] = (left, centre, right) of row r, zero outside the image
__device__ __forceinline__ void c1_finish_img(const C1Img& im, const C1Item& it, int H, int W, int lane, float (&nb)[4][3]) {
    const int hh = it.h0 - 1 + (lane >> 1), col = (lane & 1) ? it.w0 + 64 : it.w0 - 1;
    const float halo = (lane < 8 && (unsigned)hh < (unsigned)H && (unsigned)col < (unsigned)W) ? im.halo : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int hr = it.h0 - 1 + r;
        const float c = (unsigned)hr < (unsigned)H ? im.c[r] : 0.f;
        const int hl = __builtin_amdgcn_readlane(__builtin_bit_cast(int, halo), 2 * r), hrr = __builtin_amdgcn_readlane(__builtin_bit_cast(int, halo), 2 * r + 1);
        nb[r][0] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(hl, __builtin_bit_cast(int, c), DPP_WAVE_SHR1, 0xF, 0xF, false));
        nb[r][1] = c;
        nb[r][2] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(hrr, __builtin_bit_cast(int, c), DPP_WAVE_SHL1, 0xF, 0xF, false));
    }
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------------------
// forward: z[p][8] = Wpw[c] * round(dw3x3(img)[p]);  gstat [2][8] += batch sums of the stored z (fixed-order block sums, fp64 across blocks)
// uplane (bf16 [N][H][W], nullable; round 5): the block output is rank one over the channels -- z[p][c] = round(Wpw[c] * u[p]) with the ROUNDED depthwise
// output u -- so 2 bytes per pixel carry all of it: consumers that take the u plane (k_rs_bwd<..., XU>, k_c1_bwd2<..., ZU>) rebuild z with one multiply
// and one rounding per channel instead of reading 16 bytes
// FS (round 5): also the forward-only sums the fused first-block backward needs (k_rs_bwd<..., C1> + k_c1_bwd_fin): fsum [20] fp64 +=
//   sum u | sum u^2 | sum u img(tap) [9] | sum img(tap) [9]   over all output pixels (u = the rounded depthwise output, img(tap) zero outside the image)
template <class T, bool FS>
__global__ __launch_bounds__(256) void k_c1_fwd2(const float* __restrict__ img, const float* __restrict__ wdw, const float* __restrict__ wpw,
                                                 T* __restrict__ z, double* __restrict__ gstat, int N, int H, int W, bf16* __restrict__ uplane,
                                                 double* __restrict__ fsum) {
    __shared__ float s_slots[4 * 20];
    float fs[FS ? 20 : 1];
    if constexpr (FS) {
#pragma unroll
        for (int i = 0; i < 20; ++i) fs[i] = 0.f;
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float wd[9], wp[8];
#pragma unroll
    for (int i = 0; i < 9; ++i) wd[i] = wdw[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) wp[i] = wpw[i];
    float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int HP2 = H >> 1, WS = W >> 6;
    const int items = N * HP2 * WS, nw = (int)gridDim.x * 4;
    int item = (int)blockIdx.x * 4 + wave;
    auto compute = [&](const C1Img& im, const C1Item& it) {
        float nb[4][3];
        c1_finish_img(im, it, H, W, lane, nb);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float u = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) u = fmaf(wd[k], nb[q + k / 3][k % 3], u);
            u = Elem<T>::round(u);
            if constexpr (FS) {
                fs[0] += u;
                fs[1] = fmaf(u, u, fs[1]);
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    fs[2 + k] = fmaf(u, nb[q + k / 3][k % 3], fs[2 + k]);
                    fs[11 + k] += nb[q + k / 3][k % 3];
                }
            }
            if (uplane) uplane[((long)it.n * H + it.h0 + q) * W + it.w0 + lane].v = f2bf(u);  // (exact: u is a bf16 value when uplane is given)
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                o[i] = wp[i] * u;
                const float r = Elem<T>::round(o[i]);
                s1[i] += r;
                s2[i] = fmaf(r, r, s2[i]);
            }
            if (z) store8(z + (((long)it.n * H + it.h0 + q) * W + it.w0 + lane) * 8, o);  // (z == null: only the u plane is kept)
        }
    };
    C1Img imA, imB;
    C1Item itA = c1_item(item, items, HP2, WS), itB;
    c1_issue_img(imA, img, itA, H, W, lane);
    while (item < items) {
        itB = c1_item(item + nw, items, HP2, WS);
        c1_issue_img(imB, img, itB, H, W, lane);
        compute(imA, itA);
        item += nw;
        if (item >= items) break;
        itA = c1_item(item + nw, items, HP2, WS);
        c1_issue_img(imA, img, itA, H, W, lane);
        compute(imB, itB);
        item += nw;
    }
    float all[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        all[i] = s1[i];
        all[8 + i] = s2[i];
    }
    const float tot = block_sum_det<16>(all, s_slots);
    if (threadIdx.x < 16) atomicAdd(&gstat[threadIdx.x], (double)tot);
    if constexpr (FS) {
        __syncthreads();
        const float ft = block_sum_det<20>(fs, s_slots);
        if (threadIdx.x < 20) atomicAdd(&fsum[threadIdx.x], (double)ft);
    }
}

// The first block's weight gradient from sums (round 5; see k_rs_bwd<..., C1>): with dz[c] = A[c] ghat1[c] + B[c] z1[c] + C[c] (coef = [A | B | C], the block's
// BatchNorm-backward coefficients once its batch sums are complete) and z1[c] = wexp[c] u,
//   dWpw[c]  = sum_p dz[c] u           = A[c] R[c] + B[c] wexp[c] sum u^2 + C[c] sum u
//   dWdw[k]  = sum_p (sum_c wexp[c] dz[c]) img(k) = T[k] + sum_c wexp[c] (B[c] wexp[c] sum u img(k) + C[c] sum img(k))
// acc64 [17] += dWpw [8] | dWdw [9] (the layout of ocrs_dwpw_c1_bwd).  z1 is taken as wexp u (its bf16 rounding, a zero-mean 2^-9 relative perturbation per
// element, is not carried through the two forward-only sums).
__global__ void k_c1_bwd_fin(const double* __restrict__ c1acc /*[8][32]*/, const double* __restrict__ fsum /*[20]*/, const float* __restrict__ coef /*[3][8]*/,
                             const float* __restrict__ wexp /*[8]*/, double* __restrict__ acc64 /*[17]*/) {
    const int i = threadIdx.x;
    if (i >= 17) return;
    double s = 0.0;
    for (int k = 0; k < 8; ++k) s += c1acc[k * 32 + i];  // (exact sums of fp32 partials: any order gives the same bits)
    if (i < 8) {
        acc64[i] += (double)coef[i] * s + (double)coef[8 + i] * (double)wexp[i] * fsum[1] + (double)coef[16 + i] * fsum[0];
    } else {
        const int k = i - 8;
        double t = s;
        for (int c = 0; c < 8; ++c) t += (double)wexp[c] * ((double)coef[8 + c] * (double)wexp[c] * fsum[2 + k] + (double)coef[16 + c] * fsum[11 + k]);
        acc64[i] += t;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// backward (single direct gradient g, no pooling: the in_conv block's second conv is the only consumer): dz = cf0 ghat + cf1 z + cf2,
// ghat = g [bn(z) > 0];  du = sum_c Wpw[c] dz[c] stays in a register;  acc64 [17] += dWpw[c] = sum u dz[c] | dWdw[k] = sum du img[p + off_k]
// ZU (round 5): z is not read -- it is rebuilt from the depthwise output this kernel recomputes anyway, z[c] = round(Wpw[c] * u): the stored values
// bit for bit, 16 bytes per pixel less
template <class T, bool ZU>
__global__ __launch_bounds__(256) void k_c1_bwd2(const float* __restrict__ img, const float* __restrict__ wdw, const float* __restrict__ wpw,
                                                 const T* __restrict__ g, const T* __restrict__ z, const float* __restrict__ bn,
                                                 const float* __restrict__ coef, double* __restrict__ acc64, int N, int H, int W) {
    __shared__ float s_slots[4 * 17];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float wd[9], wp[8], sc[8], sh[8], cf0[8], cf1[8], cf2[8], acc[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) acc[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) wd[i] = wdw[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // (kernel-uniform: scalar registers)
        wp[i] = wpw[i];
        sc[i] = bn[i];
        sh[i] = bn[8 + i];
        cf0[i] = coef[i];
        cf1[i] = coef[8 + i];
        cf2[i] = coef[16 + i];
    }
    const int HP2 = H >> 1, WS = W >> 6;
    const int items = N * HP2 * WS, nw = (int)gridDim.x * 4;
    int item = (int)blockIdx.x * 4 + wave;
    struct Buf {
        C1Img im;
        Raw8<T> g[2], z[2];
        C1Item it;
    };
    auto issue = [&](Buf& b, int i) {
        b.it = c1_item(i, items, HP2, WS);
        const long p0 = b.it.act ? (((long)b.it.n * H + b.it.h0) * W + b.it.w0 + lane) * 8 : 0;
        const long p1 = b.it.act ? p0 + (long)W * 8 : 0;
        if constexpr (!ZU) b.z[0] = load8_raw(z + p0);
        b.g[0] = load8_raw(g + p0);
        if constexpr (!ZU) b.z[1] = load8_raw(z + p1);
        b.g[1] = load8_raw(g + p1);
        c1_issue_img(b.im, img, b.it, H, W, lane);
    };
    auto compute = [&](const Buf& b) {
        float nb[4][3];
        c1_finish_img(b.im, b.it, H, W, lane, nb);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float zv[8], gv[8];
            unpack8(b.g[q], gv);
            float u = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) u = fmaf(wd[k], nb[q + k / 3][k % 3], u);
            u = Elem<T>::round(u);
            if constexpr (ZU) {
#pragma unroll
                for (int i = 0; i < 8; ++i) zv[i] = Elem<T>::round(wp[i] * u);
            } else {
                unpack8(b.z[q], zv);
            }
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float gh = fmaf(zv[i], sc[i], sh[i]) > 0.f ? gv[i] : 0.f;
                const float dz = fmaf(cf0[i], gh, fmaf(cf1[i], zv[i], cf2[i]));
                d = fmaf(wp[i], dz, d);
                acc[i] = fmaf(u, dz, acc[i]);
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[8 + k] = fmaf(d, nb[q + k / 3][k % 3], acc[8 + k]);
        }
    };
    Buf bufA, bufB;
    issue(bufA, item);
    while (item < items) {
        issue(bufB, item + nw);
        compute(bufA);
        item += nw;
        if (item >= items) break;
        issue(bufA, item + nw);
        compute(bufB);
        item += nw;
    }
    const float tot = block_sum_det<17>(acc, s_slots);
    if (threadIdx.x < 17) atomicAdd(&acc64[threadIdx.x], (double)tot);
}

extern "C" {

long det_c1v2_supported(int N, int H, int W) {
    static const int on = env_int("OCRS_C1V2", 1);
    return on && N > 0 && H >= 2 && H % 2 == 0 && W >= 64 && W % 64 == 0 && (long)N * (H / 2) * (W / 64) < (1L << 30);
}

static int c1v2_grid(int N, int H, int W, int bpc) {
    const long waves = (long)N * (H / 2) * (W / 64);
    const long g = (waves + 3) / 4, cap = (long)kNumCU * bpc;
    return (int)(g < cap ? g : cap);
}

int det_c1v2_fwd_launch(const float* img, const float* wdw, const float* wpw, void* z, double* gstat, int N, int H, int W, int dtype, hipStream_t st,
                        void* uplane, double* fsum) {
    static const int bpc = env_int("OCRS_C1V2_FWD_BPC", 8);
    const int grid = c1v2_grid(N, H, W, bpc);
    if (dtype == 1 && fsum)
        hipLaunchKernelGGL((k_c1_fwd2<bf16, true>), dim3(grid), dim3(256), 0, st, img, wdw, wpw, (bf16*)z, gstat, N, H, W, (bf16*)uplane, fsum);
    else if (dtype == 1)
        hipLaunchKernelGGL((k_c1_fwd2<bf16, false>), dim3(grid), dim3(256), 0, st, img, wdw, wpw, (bf16*)z, gstat, N, H, W, (bf16*)uplane, (double*)nullptr);
    else
        hipLaunchKernelGGL((k_c1_fwd2<float, false>), dim3(grid), dim3(256), 0, st, img, wdw, wpw, (float*)z, gstat, N, H, W, (bf16*)nullptr, (double*)nullptr);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}
int det_c1_bwd_fin_launch(const double* c1acc, const double* fsum, const float* coef, const float* wexp, double* acc64, hipStream_t st) {
    hipLaunchKernelGGL(k_c1_bwd_fin, dim3(1), dim3(64), 0, st, c1acc, fsum, coef, wexp, acc64);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

int det_c1v2_bwd_launch(const float* img, const float* wdw, const float* wpw, const void* g, const void* z, const float* bn, const float* coef,
                        double* acc64, int N, int H, int W, int dtype, hipStream_t st) {
    static const int bpc = env_int("OCRS_C1V2_BWD_BPC", 5);
    const int grid = c1v2_grid(N, H, W, bpc);
    static const int zu_env = env_int("OCRS_C1_ZU", 1);  // rebuild z from the recomputed depthwise output instead of reading it
    const bool zu = zu_env || !z;
    if (dtype == 1) {
        if (zu) hipLaunchKernelGGL((k_c1_bwd2<bf16, true>), dim3(grid), dim3(256), 0, st, img, wdw, wpw, (const bf16*)g, (const bf16*)z, bn, coef, acc64, N, H, W);
        else hipLaunchKernelGGL((k_c1_bwd2<bf16, false>), dim3(grid), dim3(256), 0, st, img, wdw, wpw, (const bf16*)g, (const bf16*)z, bn, coef, acc64, N, H, W);
    } else {
        if (zu) hipLaunchKernelGGL((k_c1_bwd2<float, true>), dim3(grid), dim3(256), 0, st, img, wdw, wpw, (const float*)g, (const float*)z, bn, coef, acc64, N, H, W);
        else hipLaunchKernelGGL((k_c1_bwd2<float, false>), dim3(grid), dim3(256), 0, st, img, wdw, wpw, (const float*)g, (const float*)z, bn, coef, acc64, N, H, W);
    }
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

}  // extern "C"
