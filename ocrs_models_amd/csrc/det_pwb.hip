// Pointwise (1x1) conv backward of a DepthwiseConv block at the DEEP U-Net levels (gfx950, bf16, Cin in {32..256}, Cout in {64..256}), all
// channels of a tile at once.  Same contract as k_pw_bwd / k_pw_bwd8 (det_bwd.hip, det_pw8.hip):
//     dz = A*ghat + B*z + C;   du = Wpw^T dz (written);   dWpw += u^T dz,  u = dw3x3(x~) recomputed.
//
// The deep launches are latency chains, not bandwidth problems (3.7 ms of the round-1 step for 2 % of its bytes): k_pw_bwd / k_pw_bwd8 walk a
// 64-pixel tile through 32- / 64-channel chunks -- ~14 phases of (global load -> LDS -> barrier -> MFMA) per tile, 11-14 us each, and for 256
// channels the weight gradient is split over grid.y = 4 blocks that each recompute dz.  Here ONE block does a whole tile in three barriers:
//   A  every (g, z) vector of the tile's 64 pixels x Cout channels and the whole input tile + ring (100 pixels x Cin) are loaded at once
//      (one memory round trip; the next tile's loads are issued behind barrier 1 where the registers allow),
//   B  dz -> dzN [64][Cout] and x~ -> xs [100][Cin], both bf16 in LDS (up to 120 KB: one block per CU at 256 channels),
//   C  du = Wpw^T dz on MFMA (K = Cout from dzN, packed weight fragments from L2 double-buffered), u = dw3x3(x~) on the VALU -> uN [64][Cin],
//   E  dWpw += u^T dz on MFMA (K = the 64 pixels, both operands by LDS transpose reads), the full Cin x Cout gradient in registers
//      (128 per wave at 256 x 256), flushed once per block as a partial for the deterministic reducer.
#include "det_common.h"

#ifndef OCRS_PWB_TH_SMALL
#define OCRS_PWB_TH_SMALL 8  // tile rows at 64 input channels (32: always 8 -- eight waves need eight (M, N) tile pairs)
#endif
#ifndef OCRS_PWB_TH_BIG
#define OCRS_PWB_TH_BIG 4  // tile rows at >= 128 channels (32-pixel tiles: half the per-thread items)
#endif
#ifndef OCRS_PWB_PF
#define OCRS_PWB_PF 1  // prefetch the next tile's raw vectors behind barrier 1
#endif

namespace {
template <int CIN, int COUT>
struct PwbCfg {
    // (256-thread blocks -- more resident blocks to hide a tile's ~7 k-cycle latency chain -- double the per-thread items and spill 0.5-1 KB)
    static constexpr int NT = 512, NW = NT / 64;
    static constexpr int TW = 8, TH = (CIN >= 128 || COUT > 128) ? OCRS_PWB_TH_BIG : (CIN == 64 ? OCRS_PWB_TH_SMALL : 8), TP = TW * TH, NNT = TP / 16, HWp = TW + 2, HP = HWp * (TH + 2);
    static constexpr int CGI = CIN / 8, CGO = COUT / 8;
    static constexpr int PXC = CIN + 8, PZC = COUT + 8;              // bf16 pitches (16-byte pad: conflict-free fragment / transpose reads)
    static constexpr int NZI = (TP * CGO + NT - 1) / NT;             // (pixel, cout group) items per thread
    static constexpr int NXI = (HP * CGI + NT - 1) / NT;             // (staged pixel, cin group) items per thread
    static constexpr int NUI = (TP * CGI + NT - 1) / NT;             // (pixel, cin group) depthwise items per thread
    static constexpr int MTD = CIN / 16, NKD = COUT / 32;            // dgrad: M tiles, K chunks
    static constexpr int MPW = MTD >= NW ? MTD / NW : 1;             // dgrad M tiles per wave
    static constexpr int NPW = MTD >= NW ? NNT : NNT * MTD / NW;     // dgrad N tiles (16 pixels) per wave
    static constexpr int WTI = CIN / 16, WTO = COUT / 16, NTW = (WTI * WTO + NW - 1) / NW;  // wgrad output tiles: per wave
    static constexpr bool PF = OCRS_PWB_PF && !(CIN == 256 && COUT == 256);  // next tile loads in flight under the MFMA phases (registers)
    static constexpr int OFF_DZ = HP * PXC * 2, OFF_U = OFF_DZ + TP * PZC * 2, OFF_PAR = (OFF_U + TP * PXC * 2 + 15) & ~15;
    static constexpr int SMEM = OFF_PAR + (3 * CIN + 9 * CIN + 6 * COUT) * 4;
    static_assert(NT % CGO == 0 && NT % CGI == 0 && (MTD >= 2) && NPW >= 1 && TP % 32 == 0, "role mapping");
};

// resident 512-thread blocks per CU: two where the registers fit 128 (not-pooled launches up to 64 x 64 channels), else one
constexpr int pwb_bpc(int cin, int cout, bool pooled) { return ((cin <= 64 && cout <= 64) || (!pooled && cin == 128 && cout == 64)) ? 2 : 1; }

__device__ __forceinline__ void unpack4u(const uint2& r, float (&v)[4]) {
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
}
__device__ __forceinline__ void store4_opaque(bf16* p, const float (&v)[4]) {  // (see store8_opaque)
    unsigned pk[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[i]) : "v"(v[2 * i]), "v"(v[2 * i + 1]));
    *reinterpret_cast<uint2*>(p) = make_uint2(pk[0], pk[1]);
}
__device__ __forceinline__ void unpack8u(const uint4& r, float (&v)[8]) {
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
    v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}
}  // namespace

template <int CIN, int COUT, bool PPOOL, bool G2>
__global__ __launch_bounds__(512, (pwb_bpc(CIN, COUT, PPOOL) * 2)) void k_pwb(Src2<bf16> x, const float* __restrict__ tra, const float* __restrict__ trb, const float* __restrict__ wdw,
                                             const bf16* __restrict__ g1, const bf16* __restrict__ g2, const bf16* __restrict__ z,
                                             const float* __restrict__ bn, const float* __restrict__ coef, const void* __restrict__ wpk_d,
                                             bf16* __restrict__ du, float* __restrict__ ws, Tiling2 tg, BnFin fin) {
    using C = PwbCfg<CIN, COUT>;
    constexpr int NT = C::NT, TW = C::TW, TP = C::TP, HWp = C::HWp, HP = C::HP, CGI = C::CGI, CGO = C::CGO, PXC = C::PXC, PZC = C::PZC;
    constexpr int NZI = C::NZI, NXI = C::NXI, NUI = C::NUI, MTD = C::MTD, NKD = C::NKD, MPW = C::MPW, NPW = C::NPW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* xs = reinterpret_cast<bf16*>(smem);                  // [HP][PXC]   x~ on the tile + ring (0 outside the image)
    bf16* dzN = reinterpret_cast<bf16*>(smem + C::OFF_DZ);     // [TP][PZC]   dz (0 outside the image)
    bf16* uN = reinterpret_cast<bf16*>(smem + C::OFF_U);       // [TP][PXC]   depthwise output (0 outside the image)
    float* s_trx = reinterpret_cast<float*>(smem + C::OFF_PAR);  // [CIN/8][3][8]
    float* s_wdw = s_trx + 3 * CIN;                               // [9][CIN]
    float* s_bn = s_wdw + 9 * CIN;                                // [3][COUT]
    float* s_cf = s_bn + 3 * COUT;                                // [3][COUT]
    const int H = tg.H, W = tg.W;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // grid.y splits the block input's channels into ranges of CIN (256 input channels = two launches' worth of the 128-channel kernel side by
    // side: du, dWpw and u separate by input channel, dz is recomputed by both): this block owns channels c_off .. c_off + CIN of cin_total
    const int c_off = blockIdx.y * CIN, cin_total = CIN * gridDim.y;
    for (int i = tid; i < 3 * CIN; i += NT) {
        const int g = i / 24, r = (i - g * 24) >> 3, c = c_off + g * 8 + (i & 7);
        s_trx[i] = c < x.Ca ? tra[r * x.Ca + c] : trb[r * x.Cb + (c - x.Ca)];
    }
    for (int i = tid; i < 9 * CIN; i += NT) {
        const int t = i / CIN, c = i - t * CIN;
        s_wdw[i] = wdw[(c_off + c) * 9 + t];
    }
    for (int i = tid; i < 3 * COUT; i += NT) s_bn[i] = bn[i];
    if (fin.gsum) {
        bn_fin_coef(fin, COUT, s_cf, tid, NT, blockIdx.x == 0 && blockIdx.y == 0);
    } else {
        for (int i = tid; i < 3 * COUT; i += NT) s_cf[i] = coef[i];
    }
    {
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        for (int i = tid; i < C::OFF_PAR / 16; i += NT) reinterpret_cast<uint4*>(smem)[i] = z4;  // pad columns stay zero
    }
    __syncthreads();

    // ---- tile-invariant roles
    const int cgo = tid % CGO, cgi = tid % CGI;
    // dgrad: MTD >= 8: wave w owns M tiles MPW*w.., all four N tiles; else 8 / MTD waves share an M tile and split the N tiles
    const int d_m0 = MTD >= C::NW ? wave * MPW : wave % MTD, d_n0 = MTD >= C::NW ? 0 : (wave / MTD) * NPW;
    const int prow = 4 * (lane >> 4) + ((lane & 15) >> 2), pcol = (lane & 3) * 4;  // transpose-read lane geometry

    f32x4 accw[C::NTW];
#pragma unroll
    for (int j = 0; j < C::NTW; ++j) accw[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- raw vectors of one tile
    // PPOOL: the gradient arrives at half resolution and goes to the FIRST maximum of each 2x2 window.  An item is then a whole window x 4
    // channels (8-byte vectors): its four z and its g are loaded ONCE and the window's winner is found once -- with (pixel, 8 channels) items
    // every pixel re-loaded and re-compared its three neighbours (16 z loads and ~1000 VALU instructions per window and channel group
    // instead of 4 and ~210: the commit phase was 40 % of a pooled tile's time at two waves per SIMD).
    constexpr int CQO = COUT / 4, NWIN = TP / 4, NQI = PPOOL ? (NWIN * CQO + NT - 1) / NT : 1;
    struct Raw {
        uint4 z[PPOOL ? 1 : NZI], ga[PPOOL ? 1 : NZI], gb[(G2 && !PPOOL) ? NZI : 1];
        uint2 zq[PPOOL ? 4 * NQI : 1], gq1[PPOOL ? NQI : 1], gq2[(PPOOL && G2) ? NQI : 1];
        uint4 xr[NXI];
        unsigned okz, okg, okx;
    };
    auto issue = [&](Raw& r, const TileOrg& org) {
        r.okz = r.okg = r.okx = 0;
        if constexpr (PPOOL) {
            const int Hp = H >> 1, Wp = W >> 1;
#pragma unroll
            for (int j = 0; j < NQI; ++j) {
                const int it = tid + j * NT, wq = it / CQO, cq = it - wq * CQO, wy = wq / (TW / 2), wx = wq - wy * (TW / 2);
                const int h = org.h0 + 2 * wy, w = org.w0 + 2 * wx;  // (tile origins are even: the tile holds whole windows)
                const bool item = NWIN * CQO % NT == 0 || it < NWIN * CQO;
                const bool inw = item && h + 1 < H && w + 1 < W;  // floor mode: a last odd row / column is in no window
                const long p = ((long)org.n * H + h) * W + w;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool act = item && h + (e >> 1) < H && w + (e & 1) < W;
                    r.zq[4 * j + e] = *reinterpret_cast<const uint2*>(act ? z + (p + (long)(e >> 1) * W + (e & 1)) * COUT + cq * 4 : z);
                    r.okz |= act ? 1u << (4 * j + e) : 0u;
                }
                const long pg = ((long)org.n * Hp + (h >> 1)) * Wp + (w >> 1);
                r.gq1[j] = *reinterpret_cast<const uint2*>(inw ? g1 + pg * COUT + cq * 4 : g1);
                if constexpr (G2) r.gq2[j] = *reinterpret_cast<const uint2*>(inw ? g2 + pg * COUT + cq * 4 : g2);
                r.okg |= inw ? 1u << j : 0u;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NZI; ++j) {
                const int pxl = (tid + j * NT) / CGO, ty = pxl / TW, tx = pxl - ty * TW;
                const int h = org.h0 + ty, w = org.w0 + tx;
                const bool act = (TP * CGO % NT == 0 || tid + j * NT < TP * CGO) && h < H && w < W;
                const long p = ((long)org.n * H + h) * W + w;
                r.z[j] = *reinterpret_cast<const uint4*>(act ? z + p * COUT + cgo * 8 : z);
                r.ga[j] = *reinterpret_cast<const uint4*>(act ? g1 + p * COUT + cgo * 8 : g1);
                if constexpr (G2) r.gb[j] = *reinterpret_cast<const uint4*>(act ? g2 + p * COUT + cgo * 8 : g2);
                r.okz |= act ? 1u << j : 0u;
                r.okg |= act ? 1u << j : 0u;
            }
        }
        const int c0 = c_off + cgi * 8;
        const bool from_a = c0 < x.Ca;
        const bf16* xb = from_a ? x.a + c0 : x.b + (c0 - x.Ca);
        const int pitch = from_a ? x.Ca : x.Cb;
        const long corner = ((long)org.n * H + (org.h0 - 1)) * W + (org.w0 - 1);
#pragma unroll
        for (int j = 0; j < NXI; ++j) {
            const int hp = (tid + j * NT) / CGI, hy = hp / HWp, hx = hp - hy * HWp;
            const int h = org.h0 - 1 + hy, w = org.w0 - 1 + hx;
            const bool ok = (HP * CGI % NT == 0 || tid + j * NT < HP * CGI) && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
            r.xr[j] = *reinterpret_cast<const uint4*>(ok ? xb + (corner + (long)hy * W + hx) * pitch : xb);
            r.okx |= ok ? 1u << j : 0u;
        }
    };
    auto commit = [&](const Raw& r, const TileOrg& org) {
        // dz = A * ghat + B * z + C, ghat = (g1 [+ g2]) where the block's ReLU (and, pooled, the window's first maximum) lets it through
        if constexpr (PPOOL) {
#pragma unroll
            for (int j = 0; j < NQI; ++j) {
                const int it = tid + j * NT;
                if (NWIN * CQO % NT != 0 && it >= NWIN * CQO) break;
                const int wq = it / CQO, cq = it - wq * CQO, wy = wq / (TW / 2), wx = wq - wy * (TW / 2);
                const f32x4 bs = *reinterpret_cast<const f32x4*>(s_bn + cq * 4), bt = *reinterpret_cast<const f32x4*>(s_bn + COUT + cq * 4);
                const f32x4 ca = *reinterpret_cast<const f32x4*>(s_cf + cq * 4), cb = *reinterpret_cast<const f32x4*>(s_cf + COUT + cq * 4),
                            cc = *reinterpret_cast<const f32x4*>(s_cf + 2 * COUT + cq * 4);
                float g[4], zv[4][4], best[4];
                int bk[4];
                unpack4u(r.gq1[j], g);
                if constexpr (G2) {
                    float gq[4];
                    unpack4u(r.gq2[j], gq);
#pragma unroll
                    for (int i = 0; i < 4; ++i) g[i] += gq[i];
                }
                const bool inw = (r.okg >> j) & 1u;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unpack4u(r.zq[4 * j + e], zv[e]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float ym = fmaxf(fmaf(zv[e][i], bs[i], bt[i]), 0.f);  // the pool sees relu(bn(z)); a strictly larger later element wins
                        if (e == 0 || ym > best[i]) {
                            best[i] = ym;
                            bk[i] = e;
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float dz[4] = {0.f, 0.f, 0.f, 0.f};
                    if (r.okz & (1u << (4 * j + e))) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const bool win = inw && bk[i] == e && best[i] > 0.f;
                            dz[i] = fmaf(ca[i], win ? g[i] : 0.f, fmaf(cb[i], zv[e][i], cc[i]));
                        }
                    }
                    store4_opaque(dzN + ((2 * wy + (e >> 1)) * TW + 2 * wx + (e & 1)) * PZC + cq * 4, dz);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            float bs[8], bt[8], ca[8], cb[8], cc[8];
            load8(s_bn + cgo * 8, bs);
            load8(s_bn + COUT + cgo * 8, bt);
            load8(s_cf + cgo * 8, ca);
            load8(s_cf + COUT + cgo * 8, cb);
            load8(s_cf + 2 * COUT + cgo * 8, cc);
#pragma unroll
            for (int j = 0; j < NZI; ++j) {
                if (TP * CGO % NT != 0 && tid + j * NT >= TP * CGO) break;
                const int pxl = (tid + j * NT) / CGO;
                float dz[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (r.okz & (1u << j)) {
                    float zv[8], g[8];
                    unpack8u(r.z[j], zv);
                    unpack8u(r.ga[j], g);
                    if constexpr (G2) {
                        float gq[8];
                        unpack8u(r.gb[j], gq);
#pragma unroll
                        for (int i = 0; i < 8; ++i) g[i] += gq[i];
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) dz[i] = fmaf(ca[i], fmaf(zv[i], bs[i], bt[i]) > 0.f ? g[i] : 0.f, fmaf(cb[i], zv[i], cc[i]));
                }
                store8_opaque(dzN + pxl * PZC + cgo * 8, dz);
                __builtin_amdgcn_sched_barrier(0);  // (one item's unpacked vectors at a time)
            }
        }
        float sc[8], sh[8], lo[8];
        load8(s_trx + cgi * 24, sc);
        load8(s_trx + cgi * 24 + 8, sh);
        load8(s_trx + cgi * 24 + 16, lo);
#pragma unroll
        for (int j = 0; j < NXI; ++j) {
            if (HP * CGI % NT != 0 && tid + j * NT >= HP * CGI) break;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (r.okx & (1u << j)) {
                unpack8u(r.xr[j], v);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = max_lo(fmaf(v[i], sc[i], sh[i]), lo[i]);
            }
            store8_opaque(xs + ((tid + j * NT) / CGI) * PXC + cgi * 8, v);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

#ifdef OCRS_PWB_PROF
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, pc = __builtin_readcyclecounter();
#define PB_MARK(i) { const unsigned long long now_ = __builtin_readcyclecounter(); pt[i] += now_ - pc; pc = now_; }
#else
#define PB_MARK(i)
#endif
    TileSched ts(tg.ntiles);
    Raw cur;
    if (ts.first < ts.end) issue(cur, tile_origin2<TW, C::TH>(tg, (int)ts.first));
    for (long t = ts.first; t < ts.end; t += ts.step) {
        const TileOrg org = tile_origin2<TW, C::TH>(tg, (int)t);
        if (!C::PF && t != ts.first) issue(cur, org);
        PB_MARK(5)
        commit(cur, org);
        PB_MARK(0)
        __syncthreads();  // (1) dzN, xs complete
        PB_MARK(1)
        // ---- C1: du = Wpw^T dz.  ALL weight fragments of the wave are loaded before the next tile's prefetch is issued: vector loads retire in
        // order, so a fragment load behind the prefetch would wait for the whole prefetch (the first version ran 10-20 us per tile that way)
        {
            Mma<bf16>::Frag wf[NKD][MPW];
#pragma unroll
            for (int kc = 0; kc < NKD; ++kc)
#pragma unroll
                for (int a = 0; a < MPW; ++a) wf[kc][a] = Mma<bf16>::load_w(wpk_d, (long)kc * (cin_total / 16) + c_off / 16 + d_m0 + a, lane);
            __builtin_amdgcn_sched_barrier(0);
            if (C::PF && t + ts.step < ts.end) issue(cur, tile_origin2<TW, C::TH>(tg, (int)(t + ts.step)));
            __builtin_amdgcn_sched_barrier(0);
            f32x4 accd[MPW][NPW];
#pragma unroll
            for (int a = 0; a < MPW; ++a)
#pragma unroll
                for (int b = 0; b < NPW; ++b) accd[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < NKD; ++kc) {
#pragma unroll
                for (int b = 0; b < NPW; ++b) {
                    const Mma<bf16>::Frag pf = Mma<bf16>::load_p(dzN + kc * 32, PZC, (d_n0 + b) * 16, lane, 32);
#pragma unroll
                    for (int a = 0; a < MPW; ++a) accd[a][b] = Mma<bf16>::template mma<8>(wf[kc][a], pf, accd[a][b]);
                }
            }
#pragma unroll
            for (int b = 0; b < NPW; ++b) {
                const int oq = (d_n0 + b) * 16 + (lane & 15);
                const int qh = org.h0 + oq / TW, qw = org.w0 + oq % TW;
                if (qh < H && qw < W) {
                    bf16* dst = du + (((long)org.n * H + qh) * W + qw) * cin_total + c_off + d_m0 * 16 + (lane >> 4) * 4;
#pragma unroll
                    for (int a = 0; a < MPW; ++a) store4(dst + a * 16, accd[a][b][0], accd[a][b][1], accd[a][b][2], accd[a][b][3]);
                }
            }
        }
        PB_MARK(2)
        // ---- C2: u = dw3x3(x~) for this thread's (pixel, cin group) items: the nine weight vectors of the group are read once per tile
        {
            float u[NUI][8];
#pragma unroll
            for (int j = 0; j < NUI; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) u[j][i] = 0.f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                float wv[8];
                load8(s_wdw + tap * CIN + cgi * 8, wv);
#pragma unroll
                for (int j = 0; j < NUI; ++j) {
                    const int pxl = (tid + j * NT) / CGI, ty = pxl / TW, tx = pxl - ty * TW;
                    if (TP * CGI % NT == 0 || tid + j * NT < TP * CGI) {
                        float v[8];
                        unpack8u(*reinterpret_cast<const uint4*>(xs + ((ty + tap / 3) * HWp + tx + tap % 3) * PXC + cgi * 8), v);
#pragma unroll
                        for (int i = 0; i < 8; ++i) u[j][i] = fmaf(wv[i], v[i], u[j][i]);
                    }
                }
                if (tap % 3 == 2) __builtin_amdgcn_sched_barrier(0);  // (three taps' LDS reads at a time: all hoisted, the 9 x (NUI + 2) vectors spill)
            }
#pragma unroll
            for (int j = 0; j < NUI; ++j) {
                const int pxl = (tid + j * NT) / CGI, ty = pxl / TW, tx = pxl - ty * TW;
                if (TP * CGI % NT == 0 || tid + j * NT < TP * CGI) {
                    if (!(org.h0 + ty < H && org.w0 + tx < W)) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) u[j][i] = 0.f;
                    }
                    store8_opaque(uN + pxl * PXC + cgi * 8, u[j]);
                }
            }
        }
        PB_MARK(3)
        __syncthreads();  // (2) uN complete
        // ---- E: dWpw += u^T dz, K = the tile's 64 pixels
#pragma unroll
        for (int j = 0; j < C::NTW; ++j) {
            const int tt = wave + C::NW * j;
            if (C::WTI * C::WTO % C::NW == 0 || tt < C::WTI * C::WTO) {
                const int ti = tt % C::WTI, to = tt / C::WTI;
#pragma unroll
                for (int pc = 0; pc < TP / 32; ++pc) {
                    const bf16* ua = uN + (pc * 32 + prow) * PXC + ti * 16 + pcol;
                    const bf16* da = dzN + (pc * 32 + prow) * PZC + to * 16 + pcol;
                    accw[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_tr8(ua, ua + 16 * PXC), lds_tr8(da, da + 16 * PZC), accw[j], 0, 0, 0);
                }
            }
        }
        __syncthreads();  // (3) tiles free for the next commit
        PB_MARK(4)
    }
#ifdef OCRS_PWB_PROF
    if (blockIdx.x == 0 && tid == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(du);
        for (int i = 0; i < 6; ++i) o[i] = pt[i];
        o[6] = (ts.end - ts.first + ts.step - 1) / ts.step;
    }
#endif
    // ---- flush the weight-gradient partial (master layout [COUT][CIN]): D[m = ci][n = co]
#pragma unroll
    for (int j = 0; j < C::NTW; ++j) {
        const int tt = wave + C::NW * j;
        if (C::WTI * C::WTO % C::NW == 0 || tt < C::WTI * C::WTO) {
            const int ti = tt % C::WTI, to = tt / C::WTI;
            const int co = to * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                ws[(long)blockIdx.x * (cin_total * COUT) + (long)co * cin_total + c_off + ti * 16 + (lane >> 4) * 4 + r] = accw[j][r];
        }
    }
}

extern "C" {

void k_wgrad_partials_reduce_launch(const float* ws, int nb, int nelem, float* dw, int cin, int ldw, hipStream_t st);  // det_bwd.hip

// blocks of the launch (= workspace slots of Cin * Cout floats each)
static int pwb_th(int Cin, int Cout) { return (Cin >= 128 || Cout > 128) ? OCRS_PWB_TH_BIG : (Cin == 64 ? OCRS_PWB_TH_SMALL : 8); }
static int pwb_ny(int Cin) { return Cin == 256 ? 2 : 1; }  // 256 input channels: two channel ranges of the 128-channel kernel (grid.y)
int det_pwb_gx(int Cin, int Cout, int N, int H, int W, int pooled) {
    const Tiling2 tg = make_tiling2(N, H, W, 8, pwb_th(Cin, Cout));
    long g = tg.ntiles / 2;  // at least two tiles per flushing block
    const long cap = (long)kNumCU * pwb_bpc(Cin / pwb_ny(Cin), Cout, pooled != 0) / pwb_ny(Cin);  // resident blocks: a second round of blocks would double the launch
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    if (g >= 8) g &= ~7L;
    return (int)g;
}
long det_pwb_supported(int Cin, int Cout, int dtype) {
    static const int on = env_int("OCRS_PWB", 1);
    // (the instantiations with Cin >= 128 spill at 256 registers -- up to 2 KB per lane at 256 x 256 -- and stay on k_pw_bwd8)
    return on && dtype == 1 && (Cin == 32 || Cin == 64 || Cin == 128 || Cin == 256) && (Cout == 64 || Cout == 128 || Cout == 256) && Cin * 8 >= Cout;
}

int det_pwb_launch(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* g1, const void* g2,
                   int pooled, const void* z, const float* bn, const float* coef, const void* wpk_d, void* du, float* dwpw, float* ws, int Cout, int N,
                   int H, int W, const BnFin* finp, hipStream_t st) {
    const int Cin = Ca + Cb;
    const BnFin fin = finp ? *finp : BnFin{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    OCRS_CHECK_ARG(det_pwb_supported(Cin, Cout, 1) && Ca % 8 == 0 && Cb % 8 == 0 && ws);
    Src2<bf16> x{(const bf16*)xa, (const bf16*)xb, Ca, Cb};
    const Tiling2 tg = make_tiling2(N, H, W, 8, pwb_th(Cin, Cout));
    const int gx = det_pwb_gx(Cin, Cout, N, H, W, pooled), ny = pwb_ny(Cin), Cb_ = Cin / ny;  // Cb_: channels per block
    bool done = false;
#define PWB_LAUNCH(CI_, CO_, PP_, GG_)                                                                                                              \
    {                                                                                                                                               \
        using CC = PwbCfg<CI_, CO_>;                                                                                                                \
        static DevOnce attr_set;                                                                                                               \
        if (attr_set.need()) {                                                                                                                            \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pwb<CI_, CO_, PP_, GG_>), hipFuncAttributeMaxDynamicSharedMemorySize, CC::SMEM) != \
                hipSuccess)                                                                                                                         \
                return OCRS_ERR_HIP;                                                                                                                \
            attr_set.done();                                                                                                                        \
        }                                                                                                                                           \
        OCRS_LAUNCH_T((k_pwb<CI_, CO_, PP_, GG_>), dim3(gx, ny), dim3(CC::NT), CC::SMEM, st, x, tra, trb, wdw, (const bf16*)g1, (const bf16*)g2, (const bf16*)z, \
                      bn, coef, wpk_d, (bf16*)du, ws, tg, fin);                                                                                          \
        done = true;                                                                                                                                \
    }
#define PWB_CASE(CI_, CO_)                                                      \
    if (!done && Cb_ == CI_ && Cout == CO_) {                                   \
        if (pooled) {                                                           \
            if (g2) PWB_LAUNCH(CI_, CO_, true, true) else PWB_LAUNCH(CI_, CO_, true, false) \
        } else {                                                                \
            if (g2) PWB_LAUNCH(CI_, CO_, false, true) else PWB_LAUNCH(CI_, CO_, false, false) \
        }                                                                       \
    }
    PWB_CASE(32, 64) PWB_CASE(64, 64) PWB_CASE(64, 128) PWB_CASE(128, 64) PWB_CASE(128, 128) PWB_CASE(128, 256)
#undef PWB_CASE
#undef PWB_LAUNCH
    OCRS_CHECK_ARG(done);
    k_wgrad_partials_reduce_launch(ws, gx, Cin * Cout, dwpw, Cin, Cin, st);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

}  // extern "C"
