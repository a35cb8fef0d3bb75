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

// Pixel position along a grid-stride loop without divisions in the loop: the stride is constant, so (n, h, w) advances by a fixed triple
// with two carries (~8 VALU instead of two 32-bit divisions, ~50).  Positions past the end of the tensor are never dereferenced.
struct PixIter {
    int n, h, w, sn, sh, sw, H, W;
    __device__ __forceinline__ PixIter(long first, long stride, int H_, int W_) : H(H_), W(W_) {
        const PixIdx a = decode_pixel(first, H_, W_), s = decode_pixel(stride, H_, W_);
        n = a.n; h = a.h; w = a.w;
        sn = s.n; sh = s.h; sw = s.w;
    }
    __device__ __forceinline__ PixIdx cur() const {
        PixIdx r;
        r.n = n; r.h = h; r.w = w;
        return r;
    }
    __device__ __forceinline__ void next() {
        w += sw;
        if (w >= W) { w -= W; ++h; }
        h += sh;
        if (h >= H) { h -= H; ++n; }
        n += sn;
    }
};

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

// 3x3 neighbourhood (zero outside the image) of pixel px in a single-channel fp32 image as nine plain loads in two halves (issue now, use
// one loop iteration later): the streaming first-block kernels are bound by the memory LATENCY of one iteration, not by load count (six of
// the nine loads hit lines the other three fetch).  An earlier version took the left / right columns from the neighbouring lanes with
// whole-wave DPP shifts (3 loads per pixel), but shifts and edge fix-up loads have to wait for their data on the spot, which exposed a
// full memory latency per iteration.  Loads are unconditional (out-of-image taps read element 0 and are zeroed in nb9_finish).
struct Nb9 {
    float v[9];
    unsigned ok;
};
__device__ __forceinline__ void nb9_issue(Nb9& nb, const float* __restrict__ img, const PixIdx& px, int H, int W, bool active) {
    const long rowc = ((long)px.n * H + px.h) * W + px.w;
    nb.ok = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int dy = k / 3 - 1, dx = k % 3 - 1;
        const bool ok = active && (unsigned)(px.h + dy) < (unsigned)H && (unsigned)(px.w + dx) < (unsigned)W;
        nb.v[k] = img[ok ? rowc + (long)dy * W + dx : 0];
        nb.ok |= ok ? 1u << k : 0u;
    }
}
__device__ __forceinline__ void nb9_finish(const Nb9& nb, float (&v)[9]) {
#pragma unroll
    for (int k = 0; k < 9; ++k) v[k] = (nb.ok >> k & 1u) ? nb.v[k] : 0.f;
}

// Tile origins along a TileSched without divisions in the loop: the schedule's step is constant, so the origin advances by a fixed
// (images, tile rows, tile columns) triple with two carries -- ~10 scalar instructions instead of two integer divisions (~45) per tile
// (k_dw_bwd issued more SALU than VALU instructions, most of them tile bookkeeping).
template <int TW, int TH>
struct TileIter {
    int n, ty, tx, sn, sy, sx, tiles_x, tiles_y;
    __device__ __forceinline__ TileIter(const Tiling2& tg, long first, long step) {
        const int tpi = tg.tiles_x * tg.tiles_y;
        tiles_x = tg.tiles_x;
        tiles_y = tg.tiles_y;
        n = (int)first / tpi;
        const int r = (int)first - n * tpi;
        ty = r / tiles_x;
        tx = r - ty * tiles_x;
        sn = (int)step / tpi;
        const int rs = (int)step - sn * tpi;
        sy = rs / tiles_x;
        sx = rs - sy * tiles_x;
    }
    __device__ __forceinline__ TileOrg org() const {
        TileOrg o;
        o.n = n;
        o.h0 = ty * TH;
        o.w0 = tx * TW;
        return o;
    }
    __device__ __forceinline__ void next() {
        tx += sx;
        if (tx >= tiles_x) {
            tx -= tiles_x;
            ++ty;
        }
        ty += sy;
        if (ty >= tiles_y) {
            ty -= tiles_y;
            ++n;
        }
        n += sn;
    }
};

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

// ---- HaloStager: stage x~ = max(x*scale+shift, lo) of a tile's halo region into LDS as fp32; pixels outside the image are written
// as 0 (that IS the conv's zero padding), so the tap loop needs no bounds checks.  Everything tile-invariant is hoisted out of
// the persistent tile loop.
// Bank swizzle of the planar tile (16-byte units = items): the two-pixels-per-thread readers (dw2_from_lds) read every OTHER pixel, i.e. units
// base + 2*CG*k + cg: that hits half of the 16 bank groups twice (PMC: 28-34 % of the LDS cycles of the forward / pointwise-backward
// kernels were bank conflicts).  Flipping bit log2(CG) of the unit index in every odd 16-unit block moves the second half of such a
// read onto the unused bank groups; writers (consecutive units) and one-pixel readers stay conflict-free (a permutation inside a block).
template <int CG>
__device__ __forceinline__ int xs_swz(int item) {
    return item ^ (((item >> 4) & 1) * CG);
}

// LDS layout: two PLANES of 4 channels, xs[plane][halo pixel * CG + cg][4]: consecutive lanes read consecutive 16-byte words, so the
// tap loop's ds_read_b128 are bank-conflict free (an [item][8] layout puts lanes i and i+8 on the same banks: measured 25-28 %
// of all LDS cycles were conflicts).
// A thread's items are it = tid + 256*j; since 256 % CG == 0 its channel group (and so its source tensor, pitch and the
// transform parameters) is the same for every item and every tile; the halo coordinates (hy, hx) of each item are
// tile-invariant too.  Per item this leaves: 2 adds + 2 unsigned compares (bounds), one 24-bit multiply + one 64-bit add
// (address), the load, the transform and two LDS stores -- ~40 VALU instructions instead of ~130 (the index decode, the
// 64-bit pixel arithmetic and the parameter addressing of the first, per-item version were ~2/3 of the forward kernel's VALU work).
// s_tr8: transform parameters interleaved per 8-channel group: [CIN/8][3][8] (scale | shift | lo) -> 6 aligned ds_read_b128.
template <class T>
__device__ __forceinline__ void fill_tr8(float* s_tr8, const Src2<T>& x, const float* __restrict__ tra, const float* __restrict__ trb, int CIN,
                                         int tid) {
    for (int i = tid; i < 3 * CIN; i += 256) {
        const int g = i / 24, r = (i - g * 24) >> 3, c = g * 8 + (i & 7);
        s_tr8[i] = c < x.Ca ? tra[r * x.Ca + c] : trb[r * x.Cb + (c - x.Ca)];
    }
}
// max(v, lo) as exactly one v_max_f32 (fmaxf() makes the compiler add a canonicalising v_max(x,x) per operand read from LDS)
__device__ __forceinline__ float max_lo(float v, float lo) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(lo));
    return r;
}

template <class T, int CG, int TW, int TH, int NT = 256 /* threads per block */>
struct HaloStager {
    using HT = HaloTile<TW, TH>;
    static constexpr int NITEMS = HT::HP * CG;
    static constexpr int PLANE = NITEMS * 4;  // floats
    static constexpr int NIT = (NITEMS + NT - 1) / NT;
    __device__ static __forceinline__ void put(float* xs, int it, const float (&v)[8]) {
        const int o = xs_swz<CG>(it) * 4;
        store4(xs + o, v[0], v[1], v[2], v[3]);
        store4(xs + PLANE + o, v[4], v[5], v[6], v[7]);
    }
    int hyx[NIT];   // hy | hx << 16
    int poff[NIT];  // hy * W + hx  (pixel offset from the halo's corner pixel)
    int cg;
    __device__ __forceinline__ HaloStager(int tid, int W) {
        cg = tid % CG;
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int hp = (tid + j * NT) / CG;
            const int hy = hp / HT::HW_, hx = hp - hy * HT::HW_;
            hyx[j] = hy | (hx << 16);
            poff[j] = hy * W + hx;
        }
    }
    __device__ __forceinline__ void stage(const Src2<T>& x, const float* s_tr8, int ch0, const TileOrg& org, int H, int W, float* xs,
                                          int tid) const {
        const int c0 = ch0 + cg * 8;
        const bool from_a = c0 < x.Ca;
        // corner = pixel (h0-1, w0-1): may lie before the tensor for border tiles, such items are never dereferenced.
        // corner and both products are wave-uniform (scalar ALU); only the select is per thread.
        const long corner = ((long)org.n * H + (org.h0 - 1)) * W + (org.w0 - 1);
        const long offa = corner * x.Ca, offb = corner * x.Cb;
        const T* base = (from_a ? x.a + c0 : x.b + (c0 - x.Ca)) + (from_a ? offa : offb);
        const int pitch = from_a ? x.Ca : x.Cb;
        const float* tp = s_tr8 + (c0 >> 3) * 24;
        float sc[8], sh[8], lo[8];  // same channel group for all of this thread's items
        load8(tp, sc);
        load8(tp + 8, sh);
        load8(tp + 16, lo);
        const float zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            if (NITEMS % NT != 0 && j == NIT - 1 && tid + j * NT >= NITEMS) break;
            const int h = org.h0 - 1 + (hyx[j] & 0xffff), w = org.w0 - 1 + (hyx[j] >> 16);
            if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
                float v[8];
                load8(base + __umul24(poff[j], pitch), v);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = max_lo(fmaf(v[i], sc[i], sh[i]), lo[i]);
                put(xs, tid + j * NT, v);
            } else
                put(xs, tid + j * NT, zero8);
        }
    }

    // Software-pipelined form: issue() only ISSUES the global loads of a tile into raw registers (4 VGPRs per item in bf16),
    // commit() transforms them into LDS one tile later, so the HBM latency of tile t+1 hides under the tap/MFMA/store phases of tile t.
    struct Pending {
        Raw8<T> raw[NIT];
        unsigned ok;
    };
    __device__ __forceinline__ void issue(Pending& pd, const Src2<T>& x, int ch0, const TileOrg& org, int H, int W, int tid) const {
        const int c0 = ch0 + cg * 8;
        const bool from_a = c0 < x.Ca;
        const long corner = ((long)org.n * H + (org.h0 - 1)) * W + (org.w0 - 1);
        const long offa = corner * x.Ca, offb = corner * x.Cb;
        const T* base = (from_a ? x.a + c0 : x.b + (c0 - x.Ca)) + (from_a ? offa : offb);
        const int pitch = from_a ? x.Ca : x.Cb;
        pd.ok = 0;
        // UNCONDITIONAL loads (out-of-image / surplus items read the tensor's first pixel and are zeroed in commit()): a load under a
        // divergent branch makes hipcc's waitcnt pass put vmcnt(0) in front of the NEXT load, serialising the prefetch.
        const T* dummy = from_a ? x.a : x.b;
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int h = org.h0 - 1 + (hyx[j] & 0xffff), w = org.w0 - 1 + (hyx[j] >> 16);
            const bool ok = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W && (NITEMS % NT == 0 || j < NIT - 1 || tid + j * NT < NITEMS);
            pd.raw[j] = load8_raw(ok ? base + __umul24(poff[j], pitch) : dummy);
            pd.ok |= ok ? 1u << j : 0u;
        }
    }
    __device__ __forceinline__ void commit(const Pending& pd, const float* s_tr8, int ch0, float* xs, int tid) const {
        const float* tp = s_tr8 + ((ch0 >> 3) + cg) * 24;
        float sc[8], sh[8], lo[8];
        load8(tp, sc);
        load8(tp + 8, sh);
        load8(tp + 16, lo);
        const float zero8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            if (NITEMS % NT != 0 && j == NIT - 1 && tid + j * NT >= NITEMS) break;
            if (pd.ok & (1u << j)) {
                float v[8];
                unpack8(pd.raw[j], v);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = max_lo(fmaf(v[i], sc[i], sh[i]), lo[i]);
                put(xs, tid + j * NT, v);
            } else
                put(xs, tid + j * NT, zero8);
        }
    }
};

// u[8] = sum over the 9 taps of w[tap][c] * xs[pixel + tap][c] for the thread's (pixel, channel group), all from LDS
// (xs in the HaloStager's planar layout).
template <int CG, int TW, int TH, bool SWZ = true>
__device__ __forceinline__ void dw_from_lds(const float* xs, const float* s_w /*[9][CIN] tap-major*/, int CIN, int c0, int cg, int ty, int tx,
                                            float (&u)[8]) {
    constexpr int HWp = TW + 2;
    constexpr int PLANE = HaloTile<TW, TH>::HP * CG * 4;
    const int item0 = (ty * HWp + tx) * CG + cg;  // top-left tap of this pixel
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float v[8], wv[8];
        const int it = item0 + ((t / 3) * HWp + (t % 3)) * CG;
        const float* q = xs + (SWZ ? xs_swz<CG>(it) : it) * 4;
        const float4 lo4 = *reinterpret_cast<const float4*>(q), hi4 = *reinterpret_cast<const float4*>(q + PLANE);
        v[0] = lo4.x; v[1] = lo4.y; v[2] = lo4.z; v[3] = lo4.w;
        v[4] = hi4.x; v[5] = hi4.y; v[6] = hi4.z; v[7] = hi4.w;
        load8(s_w + t * CIN + c0, wv);
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = fmaf(wv[i], v[i], u[i]);
    }
}

// u0/u1 = depthwise 3x3 of two horizontally adjacent pixels (tx, tx+1) for 8 channels, from the HaloStager's planar LDS tile:
// 3 rows x 4 columns of inputs, each row's 3 weight vectors read once.
template <int CG, int TW, int TH>
__device__ __forceinline__ void dw2_from_lds(const float* xs, const float* s_w /*[9][CIN] tap-major*/, int CIN, int c0, int cg, int ty, int tx,
                                             float (&u0)[8], float (&u1)[8]) {
    constexpr int HWp = TW + 2;
    constexpr int PLANE = HaloTile<TW, TH>::HP * CG * 4;
    const int item0 = (ty * HWp + tx) * CG + cg;
#pragma unroll
    for (int i = 0; i < 8; ++i) u0[i] = u1[i] = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float w[3][8];
#pragma unroll
        for (int c = 0; c < 3; ++c) load8(s_w + (r * 3 + c) * CIN + c0, w[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float* q = xs + xs_swz<CG>(item0 + (r * HWp + c) * CG) * 4;
            const float4 lo4 = *reinterpret_cast<const float4*>(q), hi4 = *reinterpret_cast<const float4*>(q + PLANE);
            const float v[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (c < 3) u0[i] = fmaf(w[c][i], v[i], u0[i]);
                if (c > 0) u1[i] = fmaf(w[c - 1][i], v[i], u1[i]);
            }
        }
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

// The same in two halves for software pipelining: issue_ghat8() only ISSUES the global loads (raw registers; UNCONDITIONAL, inactive
// threads read the tensors' first elements -- a load under a divergent branch would make hipcc drain vmcnt before the next load),
// finish_ghat8() does the arithmetic one phase later.  gs.pooled / gs.g2 are kernel-uniform.
template <class T>
struct GhatPend {
    Raw8<T> z, g1, g2, zo0, zo1, zo2;  // (named members, not an array: an array here ends up in scratch)
};
template <class T>
__device__ __forceinline__ void issue_ghat8(GhatPend<T>& pd, const GradSrc<T>& gs, const T* z, int C, long p, const PixIdx& px, int H, int W,
                                            int c0, bool act) {
    // Straight-line code on purpose (addresses are selected, loads are not branched over): with the loads under `if (gs.g2)` /
    // `if (gs.pooled)` hipcc routed the conditionally-defined registers through scratch with a vmcnt(0) behind each load.  Without a
    // second gradient g2 re-reads g1, without pooling the three window loads re-read one element of z: cache hits, results unused.
    const bool pooled = gs.pooled != 0;
    const int Hp = H >> 1, Wp = W >> 1;
    const int ph = px.h >> 1, pw = px.w >> 1;
    const bool inw = act && (!pooled || (px.h < 2 * Hp && px.w < 2 * Wp));
    const long pg = pooled ? ((long)px.n * Hp + ph) * Wp + pw : p;
    const T* g2b = gs.g2 ? gs.g2 : gs.g1;
    pd.z = load8_raw(act ? z + p * C + c0 : z);
    pd.g1 = load8_raw(inw ? gs.g1 + pg * C + c0 : gs.g1);
    pd.g2 = load8_raw(inw ? g2b + pg * C + c0 : g2b);
    const bool wz = inw && pooled;
    const int own = ((px.h & 1) << 1) | (px.w & 1);
    const long base = ((long)px.n * H + 2 * ph) * W + 2 * pw;
    // the three OTHER elements of the window, in window order
    const int k0 = own <= 0 ? 1 : 0, k1 = own <= 1 ? 2 : 1, k2 = own <= 2 ? 3 : 2;
    pd.zo0 = load8_raw(wz ? z + (base + (long)(k0 >> 1) * W + (k0 & 1)) * C + c0 : z);
    pd.zo1 = load8_raw(wz ? z + (base + (long)(k1 >> 1) * W + (k1 & 1)) * C + c0 : z);
    pd.zo2 = load8_raw(wz ? z + (base + (long)(k2 >> 1) * W + (k2 & 1)) * C + c0 : z);
}
template <class T>
__device__ __forceinline__ void finish_ghat8(const GhatPend<T>& pd, const GradSrc<T>& gs, int C, const float* bn, const PixIdx& px, int H, int W,
                                             int c0, float (&gh)[8], float (&zv)[8]) {
    unpack8(pd.z, zv);
    float y[8], g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = fmaf(zv[i], bn[c0 + i], bn[C + c0 + i]);
    unpack8(pd.g1, g);
    if (gs.g2) {
        float g2[8];
        unpack8(pd.g2, g2);
#pragma unroll
        for (int i = 0; i < 8; ++i) g[i] += g2[i];
    }
    if (!gs.pooled) {
#pragma unroll
        for (int i = 0; i < 8; ++i) gh[i] = y[i] > 0.f ? g[i] : 0.f;
        return;
    }
    const int Hp = H >> 1, Wp = W >> 1;
    const bool inw = px.h < 2 * Hp && px.w < 2 * Wp;  // floor mode: the last odd row / column is in no window
    const int own = ((px.h & 1) << 1) | (px.w & 1);
    bool win[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) win[i] = inw && y[i] > 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float zo[8];
        unpack8(j == 0 ? pd.zo0 : (j == 1 ? pd.zo1 : pd.zo2), zo);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float yo = fmaxf(fmaf(zo[i], bn[c0 + i], bn[C + c0 + i]), 0.f);
            const float ym = fmaxf(y[i], 0.f);
            win[i] = win[i] && (j < own ? ym > yo : ym >= yo);  // ties go to the first element of the window
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) gh[i] = win[i] ? g[i] : 0.f;
}

// BatchNorm-backward finalisation folded into a block-backward kernel's prologue (what k_bn_bwd_finalize computes, det_bwd.hip): with gsum set,
// every block derives the dz coefficients of its Cout channels from the block's complete sums instead of reading `coef`, and one block writes
// dgamma / dbeta -- one ~5 us launch less per block on the backward's critical path.
struct BnFin {
    const double* gsum;   // [2][Cout] sum ghat | sum ghat*zhat of THIS block (complete: produced by the consumers' launches), or null
    const float* gamma;   // [Cout]
    const float* saved;   // [2][Cout] mean | rstd
    float* dgamma;        // [Cout] (written)
    float* dbeta;
    long count;           // N * H * W
};
__device__ __forceinline__ void bn_fin_coef(const BnFin& fin, int COUT, float* s_cf /*[3][COUT]*/, int tid, int nt, bool writer) {
    for (int c = tid; c < COUT; c += nt) {  // (same arithmetic as k_bn_bwd_finalize)
        const double s1 = fin.gsum[c], s2 = fin.gsum[COUT + c];
        const double m1 = s1 / (double)fin.count, m2 = s2 / (double)fin.count;
        const double mean = fin.saved[c], rstd = fin.saved[COUT + c];
        const double A = (double)fin.gamma[c] * rstd;
        s_cf[c] = (float)A;
        s_cf[COUT + c] = (float)(-A * rstd * m2);
        s_cf[2 * COUT + c] = (float)(A * (-m1 + mean * rstd * m2));
        if (writer) {
            fin.dgamma[c] = (float)s2;
            fin.dbeta[c] = (float)s1;
        }
    }
}

// ---- block backward: the producers' BatchNorm-backward sums finalised by the LAST workgroup of the launch (round 5) -------------------------------
// k_mm_bwd / k_rs_bwd / k_dw_bwd hand the raw sums of the gradient flowing into their input's producers (S1 = sum ghat', S2 = sum ghat' x~) to a
// single-writer reduce kernel that also sums the weight-gradient partials -- a ~7 us launch per block ON the backward's dependency chain (the next
// block's kernel derives its dz coefficients from those sums), 27 of them per step.  With `raw` set, every workgroup instead adds its fp32 partial
// sums into fp64 device-scope atomics (an fp64 sum of fp32 values of comparable magnitude is exact, hence order-independent: still bit-reproducible),
// drains them (s_waitcnt vmcnt(0), not __threadfence: see bn_finalize_last_block) and takes a ticket; the workgroup that draws the last ticket
// converts the totals exactly as the reduce kernel does, adds them to gsum_a / gsum_b and re-zeroes raw / counter (ready for a replay).  What is
// left for the reduce kernel -- the weight gradients, which nothing in the backward reads -- is queued and runs as ONE launch at the end of the
// backward (ocrs_bwd_defer_begin / ocrs_bwd_defer_flush in det_bwd.hip).
struct BwdLast {
    double* raw;        // [BWD_LAST_SLOTS][2][CIN] fp64, zeroed (null: off -- the launcher runs the reduce kernel in line)
    unsigned* counter;  // one zeroed word
    double *gsum_a, *gsum_b;
    const float *saved_a, *saved_b;
    int Ca;             // channels of source a (the rest belong to source b)
    int mode;           // 0: S1 | S2 = sum ghat' | sum ghat' x~ (k_mm_bwd_reduce's conversion);  1: sum ghat | sum ghat (z - mean) (k_dw_partials_reduce's: x rstd)
};
constexpr int BWD_LAST_SLOTS = 8;  // replicas of the raw sums, by workgroup index: same-address atomics serialise at the memory side (~15 ns each; 768 workgroups
                                   // finishing together on one replica cost as much as the launch this replaces)
__device__ __forceinline__ void bwd_last_add(const BwdLast& bl, int CIN, int c, int which, float v) {
    atomicAdd(bl.raw + ((blockIdx.x + blockIdx.y) & (BWD_LAST_SLOTS - 1)) * 2 * CIN + which * CIN + c, (double)v);
}
// call with all threads of the block after the bwd_last_add calls; tra / trb: the sources' load transforms [3][C] (mode 0); s_flag: a free LDS int
__device__ __forceinline__ void bwd_last_finish(const BwdLast& bl, int CIN, const float* __restrict__ tra, const float* __restrict__ trb, int tid, int nt,
                                                int* s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) *s_flag = __hip_atomic_fetch_add(bl.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x * gridDim.y - 1 ? 1 : 0;
    __syncthreads();
    if (*s_flag == 0) return;
    for (int c = tid; c < CIN; c += nt) {
        double S1 = 0.0, S2 = 0.0;  // (exact sums: any order gives the same bits)
        for (int k = 0; k < BWD_LAST_SLOTS; ++k) {
            double* r = bl.raw + k * 2 * CIN;
            S1 += __hip_atomic_load(r + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            S2 += __hip_atomic_load(r + CIN + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(r + c, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(r + CIN + c, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool in_a = c < bl.Ca;
        double* gs = in_a ? bl.gsum_a : bl.gsum_b;
        if (!gs) continue;
        const int cc = in_a ? c : c - bl.Ca, Cs = in_a ? bl.Ca : CIN - bl.Ca;
        const float* sv = in_a ? bl.saved_a : bl.saved_b;
        const double mean = sv[cc], rstd = sv[Cs + cc];
        gs[cc] += S1;
        if (bl.mode == 0) {
            const float* tr = in_a ? tra : trb;
            const double sc = tr[cc], sh = tr[Cs + cc];
            if (sc != 0.0) gs[Cs + cc] += rstd * ((S2 - sh * S1) / sc - mean * S1);
        } else {
            gs[Cs + cc] += (double)((float)S2 * sv[Cs + cc]);
        }
    }
    if (tid == 0) __hip_atomic_store(bl.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// host side (det_bwd.hip): scratch for one launch (null outside ocrs_bwd_defer_begin .. _flush or when it is used up), and the queue of deferred
// weight-gradient reductions: columns [0, n0) of ws[nb][nelem] += into d0[(e / cin0) * ldw0 + e % cin0], columns [n0, n0 + n1) into d1[e - n0]
double* bwd_defer_scratch(int ndoubles);
float* rec_defer_partials(int nb, int n, float* out);  // rec_conv.hip: partial buffer [nb][n] + queued column sum into out [n] (null: not deferring)
float* bwd_defer_ws(long nfloats);  // per-block-partial workspace for launches whose entry points take none (null: not deferring / used up)
bool bwd_defer_reduce(const float* ws, int nb, int nelem, float* d0, int n0, int cin0, int ldw0, float* d1, int n1);
void bwd_reduce_or_defer(const float* ws, int nb, int nelem, float* d0, int n0, int cin0, int ldw0, float* d1, int n1, hipStream_t st);  // queue, or one-job launch now

// ---- BatchNorm2d training statistics finalised by the LAST workgroup of the forward launch that produced them (one launch less per block)
struct FwdFin {
    unsigned* counter;  // null: plain partials, the caller runs ocrs_bn_finalize_parts
    long count;
    const float *gamma, *beta;
    float eps, momentum;
    float *tr, *saved, *run_mean, *run_var;
    long long* nbt;
    float lo;
};
__device__ __forceinline__ void bn_finalize_channel(double sum, double sumsq, long count, int c, int C, const float* gamma, const float* beta, float eps, float momentum,
                                                    float* tr, float* saved, float* run_mean, float* run_var, float lo) {
    const double mean = sum / (double)count;
    double var = sumsq / (double)count - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * rstd;
    tr[c] = sc;
    tr[C + c] = beta[c] - (float)mean * sc;
    tr[2 * C + c] = lo;
    saved[c] = (float)mean;
    saved[C + c] = rstd;
    if (run_mean) {
        const double unb = count > 1 ? var * (double)count / (double)(count - 1) : var;
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mean;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
    }
}

// Forward kernels that accumulate their batch sums into gstat [2][C] (fp64 device-scope atomics): after its own adds every workgroup bumps
// fin.counter; the one that sees all the others done reads the totals (device-scope loads: the atomics were executed at the memory side) and
// does k_bn_finalize's arithmetic.  Call with all threads of the block; `s_flag`: a shared int.
__device__ __forceinline__ void bn_finalize_last_block(const FwdFin& fin, const double* gstat, int C, int tid, int nt, int* s_flag) {
    if (!fin.counter) return;
    // this thread's atomics are acknowledged (performed at the memory side: device-scope read-modify-writes bypass the XCD's L2) before the arrival
    // below is issued.  NOT __threadfence(): its release writes back every dirty line of the L2 -- the output tensor this launch has just stored --
    // and made the deep-level launches 3x slower
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) *s_flag = __hip_atomic_fetch_add(fin.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (*s_flag == 0) return;
    for (int c = tid; c < C; c += nt) {
        const double sum = __hip_atomic_load(gstat + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double sq = __hip_atomic_load(gstat + C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bn_finalize_channel(sum, sq, fin.count, c, C, fin.gamma, fin.beta, fin.eps, fin.momentum, fin.tr, fin.saved, fin.run_mean, fin.run_var, fin.lo);
    }
    if (tid == 0) {
        if (fin.nbt) *fin.nbt += 1;
        __hip_atomic_store(fin.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (ready for a replay of the same launch)
    }
}
