// Shared device pieces of the detection (U-Net) kernels: activation sources with the producer's
// BatchNorm+ReLU applied on load, gradient sources (direct / through a 2x2 max-pool), pixel decode.
//
// Layout: every activation is NHWC in HBM ("pixel-major": all channels of a pixel contiguous), dtype T
// (fp32 or bf16).  A DepthwiseConv block (reference ocrs_models/models.py:7-28) stores only its
// PRE-BatchNorm output z; consumers apply  x~ = max(z*scale + shift, lo)  while loading
// (scale = gamma*rstd, shift = beta - mean*scale, lo = 0 for ReLU; identity = (1, 0, -inf)).
#pragma once
#include "common.h"

struct PixIdx {
    int n, h, w;
};
__device__ __forceinline__ PixIdx decode_pixel(long p, int H, int W) {
    PixIdx r;
    const long hw = (long)H * W;
    r.n = (int)(p / hw);
    const int rem = (int)(p - (long)r.n * hw);
    r.h = rem / W;
    r.w = rem - r.h * W;
    return r;
}

// two-source (channel-concatenated) activation: channels [0,Ca) from a, [Ca,Ca+Cb) from b.
// This is how torch.cat((upscaled, skip), 1) (models.py:89) is consumed without materialising it.
template <class T>
struct Src2 {
    const T* a;
    const T* b;
    int Ca, Cb;
};

template <class T>
__device__ __forceinline__ const T* src_ptr(const Src2<T>& s, long pix, int c0) {
    return (c0 < s.Ca) ? s.a + pix * s.Ca + c0 : s.b + pix * s.Cb + (c0 - s.Ca);
}

// transform 8 channels:  v = max(v*sc + sh, lo);  tr points at [3][C] (scale | shift | lo)
__device__ __forceinline__ void apply_tr8(float (&v)[8], const float* tr, int C, int c0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = fmaxf(fmaf(v[i], tr[c0 + i], tr[C + c0 + i]), tr[2 * C + c0 + i]);
}

// Gradient w.r.t. a block's post-activation output y, in one of two forms:
//   direct : g1[p][C] (+ g2[p][C])                         (one or two consumers of y)
//   pooled : g1[pp][C] (+ g2[pp][C]) at half resolution; y went through MaxPool2d(2) (models.py:54),
//            the gradient is routed to the FIRST maximal element of each 2x2 window (row-major).
template <class T>
struct GradSrc {
    const T* g1;
    const T* g2;
    int pooled;
};

// ghat = dL/dy * [y > 0]  (ReLU mask folded in), plus the raw z of the 8 channels.
// bn points at [3][C] (scale | shift | lo) of THIS block's BatchNorm.
template <class T>
__device__ __forceinline__ void load_ghat8(const GradSrc<T>& gs, const T* z, int C, const float* bn, long p, const PixIdx& px, int H,
                                           int W, int c0, float (&gh)[8], float (&zv)[8]) {
    load8(z + p * C + c0, zv);
    float y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = fmaf(zv[i], bn[c0 + i], bn[C + c0 + i]);
    if (!gs.pooled) {
        float g[8];
        load8(gs.g1 + p * C + c0, g);
        if (gs.g2) {
            float g2[8];
            load8(gs.g2 + p * C + c0, g2);
#pragma unroll
            for (int i = 0; i < 8; ++i) g[i] += g2[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) gh[i] = y[i] > 0.f ? g[i] : 0.f;
        return;
    }
    const int Hp = H >> 1, Wp = W >> 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) gh[i] = 0.f;
    if (px.h >= 2 * Hp || px.w >= 2 * Wp) return;  // floor mode: last odd row/col is in no window
    const int ph = px.h >> 1, pw = px.w >> 1;
    const long pp = ((long)px.n * Hp + ph) * Wp + pw;
    float g[8];
    load8(gs.g1 + pp * C + c0, g);
    if (gs.g2) {
        float g2[8];
        load8(gs.g2 + pp * C + c0, g2);
#pragma unroll
        for (int i = 0; i < 8; ++i) g[i] += g2[i];
    }
    const int own = ((px.h & 1) << 1) | (px.w & 1);
    bool win[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) win[i] = y[i] > 0.f;  // pooled value must also pass the ReLU
    const long base = ((long)px.n * H + 2 * ph) * W + 2 * pw;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k == own) continue;
        float zo[8];
        load8(z + (base + (long)(k >> 1) * W + (k & 1)) * C + c0, zo);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // compare in post-ReLU space (the pool input is relu(bn(z))): ties at 0 go to the first element
            const float yo = fmaxf(fmaf(zo[i], bn[c0 + i], bn[C + c0 + i]), 0.f);
            const float ym = fmaxf(y[i], 0.f);
            win[i] = win[i] && (k < own ? ym > yo : ym >= yo);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) gh[i] = win[i] ? g[i] : 0.f;
}
