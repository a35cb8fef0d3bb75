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
    // all tensors here have < 2^31 pixels (checked by the C entry points): 32-bit unsigned division only
    PixIdx r;
    const unsigned q = (unsigned)p, uw = (unsigned)W, uh = (unsigned)H;
    const unsigned row = q / uw;
    r.w = (int)(q - row * uw);
    r.n = (int)(row / uh);
    r.h = (int)(row - (unsigned)r.n * uh);
    return r;
}

// 2-D pixel tiling: a tile is TH x TW pixels of one image (TW = 1 << tw_shift, TH = TP >> tw_shift), so the
// 3x3 halo rows of a tile are mostly fetched by the same workgroup (L1) instead of three different ones.
struct Tiling {
    int H, W, tw_shift, tiles_x, tiles_y, ntiles;
};
static inline Tiling make_tiling(int N, int H, int W, int TP) {
    Tiling t;
    int sh = 0;
    while ((1 << sh) < W && (1 << sh) < 32 && (1 << sh) < TP) ++sh;
    t.H = H; t.W = W; t.tw_shift = sh;
    const int TW = 1 << sh, TH = TP >> sh;
    t.tiles_x = (W + TW - 1) / TW;
    t.tiles_y = (H + TH - 1) / TH;
    t.ntiles = N * t.tiles_x * t.tiles_y;
    return t;
}
struct TileOrg {
    int n, h0, w0;
};
__device__ __forceinline__ TileOrg tile_origin(const Tiling& tg, int TP, int tile) {
    TileOrg o;
    const int tpi = tg.tiles_x * tg.tiles_y;
    o.n = tile / tpi;
    const int r = tile - o.n * tpi;
    const int ty = r / tg.tiles_x;
    o.h0 = ty * (TP >> tg.tw_shift);
    o.w0 = (r - ty * tg.tiles_x) << tg.tw_shift;
    return o;
}
// pixel `pxl` (0..TP) of the tile -> image coordinates; returns validity
__device__ __forceinline__ bool tile_pixel(const Tiling& tg, const TileOrg& o, int pxl, PixIdx& px) {
    px.n = o.n;
    px.h = o.h0 + (pxl >> tg.tw_shift);
    px.w = o.w0 + (pxl & ((1 << tg.tw_shift) - 1));
    return px.h < tg.H && px.w < tg.W;
}
__device__ __forceinline__ long pix_linear(const PixIdx& px, int H, int W) { return ((long)px.n * H + px.h) * W + px.w; }

// ---- compile-time 2-D tiles for the LDS-staged stencil kernels: TH x TW pixels, halo (TH+2) x (TW+2) -----------------
template <int TW, int TH>
struct HaloTile {
    static constexpr int HW_ = TW + 2, HH_ = TH + 2, HP = HW_ * HH_;
};
struct Tiling2 {
    int H, W, tiles_x, tiles_y, ntiles;
};
static inline Tiling2 make_tiling2(int N, int H, int W, int TW, int TH) {
    Tiling2 t;
    t.H = H; t.W = W;
    t.tiles_x = (W + TW - 1) / TW;
    t.tiles_y = (H + TH - 1) / TH;
    t.ntiles = N * t.tiles_x * t.tiles_y;
    return t;
}
template <int TW, int TH>
__device__ __forceinline__ TileOrg tile_origin2(const Tiling2& tg, int tile) {
    TileOrg o;
    const int tpi = tg.tiles_x * tg.tiles_y;
    o.n = tile / tpi;
    const int r = tile - o.n * tpi;
    const int ty = r / tg.tiles_x;
    o.h0 = ty * TH;
    o.w0 = (r - ty * tg.tiles_x) * TW;
    return o;
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

// u[8] = depthwise 3x3 (zero padding 1) of the transformed input x~ at pixel px, channels c0..c0+7.
// s_tr: [3][CIN] scale|shift|lo, s_w: [9][CIN] tap-major weights (both in LDS).
template <class T>
__device__ __forceinline__ void dw_compute8(const Src2<T>& x, const float* s_tr, const float* s_w, int CIN, int c0, const PixIdx& px, int H,
                                            int W, float (&u)[8]) {
    const T* base;
    int pitch;
    if (c0 < x.Ca) {
        base = x.a + c0;
        pitch = x.Ca;
    } else {
        base = x.b + (c0 - x.Ca);
        pitch = x.Cb;
    }
    base += pix_linear(px, H, W) * pitch;
    float sc[8], sh[8], lo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        sc[i] = s_tr[c0 + i];
        sh[i] = s_tr[CIN + c0 + i];
        lo[i] = s_tr[2 * CIN + c0 + i];
        u[i] = 0.f;
    }
    // branchless taps: an out-of-image tap reads the (always valid) centre pixel and is multiplied by 0
    const bool hv[3] = {px.h > 0, true, px.h < H - 1};
    const bool wv[3] = {px.w > 0, true, px.w < W - 1};
    // all 9 tap loads are issued first (raw, 4 or 8 VGPRs each) so that they overlap; the per-tap LDS weight reads
    // are kept behind compiler barriers so they are not all hoisted (register pressure -> occupancy).
    float v9[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const bool ok = hv[t / 3] && wv[t % 3];
        load8(base + (ok ? ((t / 3 - 1) * W + (t % 3 - 1)) * pitch : 0), v9[t]);
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const bool ok = hv[t / 3] && wv[t % 3];
        float(&v)[8] = v9[t];
        if (t % 3 == 0) asm volatile("" ::: "memory");
        const float* wt = s_w + t * CIN + c0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float xt = fmaxf(fmaf(v[i], sc[i], sh[i]), lo[i]);
            u[i] = fmaf(ok ? wt[i] : 0.f, xt, u[i]);
        }
    }
}

// Stage x~ = max(x*scale+shift, lo) for channels [ch0, ch0 + CG*8) of the tile's halo region into LDS as fp32
// xs[halo pixel][CG*8]; pixels outside the image are written as 0 (that IS the conv's zero padding), so the
// tap loop needs no bounds checks.  Every input element is loaded, unpacked and transformed once (x1.3-1.6 halo).
template <class T, int CG, int TW, int TH>
__device__ __forceinline__ void stage_halo(const Src2<T>& x, const float* s_tr, int CIN, int ch0, const TileOrg& org, int H, int W,
                                           float* xs, int tid) {
    using HT = HaloTile<TW, TH>;
    for (int it = tid; it < HT::HP * CG; it += 256) {
        const int hp = it / CG, cg = it - hp * CG;
        const int hy = hp / HT::HW_, hx = hp - hy * HT::HW_;
        const int h = org.h0 + hy - 1, w = org.w0 + hx - 1;
        const int c0 = ch0 + cg * 8;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (h >= 0 && h < H && w >= 0 && w < W) {
            load8(src_ptr(x, ((long)org.n * H + h) * W + w, c0), v);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = fmaxf(fmaf(v[i], s_tr[c0 + i], s_tr[CIN + c0 + i]), s_tr[2 * CIN + c0 + i]);
        }
        store8(xs + (hp * CG + cg) * 8, v);
    }
}

// Software-pipelined form of stage_halo: halo_issue() only ISSUES the global loads of one (tile, channel-chunk) into
// registers (NIT = ceil(HP*CG/256) raw vectors per thread); halo_commit() later applies the transform and writes LDS.
// The caller issues the loads of the next tile before computing on the current one, so HBM latency hides under compute.
template <class T, int CG, int TW, int TH>
struct HaloPipe {
    using HT = HaloTile<TW, TH>;
    static constexpr int NIT = (HT::HP * CG + 255) / 256;
    Raw8<T> raw[NIT];
    unsigned okmask;

    __device__ __forceinline__ void issue(const Src2<T>& x, int ch0, const TileOrg& org, int H, int W, int tid) {
        okmask = 0;
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int it = tid + j * 256;
            const int hp = it / CG, cg = it - hp * CG;
            const int hy = hp / HT::HW_, hx = hp - hy * HT::HW_;
            const int h = org.h0 + hy - 1, w = org.w0 + hx - 1;
            const bool ok = it < HT::HP * CG && h >= 0 && h < H && w >= 0 && w < W;
            if (ok) {
                raw[j] = load8_raw(src_ptr(x, ((long)org.n * H + h) * W + w, ch0 + cg * 8));
                okmask |= 1u << j;
            }
        }
    }
    __device__ __forceinline__ void commit(const float* s_tr, int CIN, int ch0, float* xs, int tid) const {
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int it = tid + j * 256;
            if (it < HT::HP * CG) {
                const int cg = it % CG;
                const int c0 = ch0 + cg * 8;
                float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (okmask & (1u << j)) {
                    unpack8(raw[j], v);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fmaxf(fmaf(v[i], s_tr[c0 + i], s_tr[CIN + c0 + i]), s_tr[2 * CIN + c0 + i]);
                }
                store8(xs + (long)it * 8, v);
            }
        }
    }
};

// u[8] = sum over the 9 taps of w[tap][c] * xs[pixel + tap][c] for the thread's (pixel, channel group), all from LDS.
// wg (optional, CG == 1 only): the weights in the reference layout [8][9] in GLOBAL memory -- with 8 channels every thread uses the
// same 72 weights, so they are read with wave-uniform addresses (scalar loads -> SGPR operands) instead of 18 LDS reads per pixel.
template <int CG, int TW>
__device__ __forceinline__ void dw_from_lds(const float* xs, const float* s_w /*[9][CIN] tap-major*/, int CIN, int c0, int cg, int ty, int tx,
                                            float (&u)[8], const float* __restrict__ wg = nullptr) {
    constexpr int HWp = TW + 2;
    const float* xc = xs + ((ty * HWp + tx) * CG + cg) * 8;  // top-left tap of this pixel
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float v[8], wv[8];
        load8(xc + ((t / 3) * HWp + (t % 3)) * CG * 8, v);
        if (CG == 1 && wg) {
#pragma unroll
            for (int i = 0; i < 8; ++i) wv[i] = wg[i * 9 + t];
        } else
            load8(s_w + t * CIN + c0, wv);
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = fmaf(wv[i], v[i], u[i]);
    }
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
