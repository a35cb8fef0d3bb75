// Pointwise (1x1) conv backward of a DepthwiseConv block at the DEEPEST U-Net levels (gfx950, bf16, Cin, Cout in {128, 256}).
// Same contract as k_pw_bwd (det_bwd.hip):  dz = A*ghat + B*z + C;  du = Wpw^T dz (written);  dWpw += u^T dz, u = dw3x3(x~) recomputed.
//
// The deepest launches are latency-bound, not bandwidth-bound: a few hundred 8x8-pixel tiles, each a chain of load -> LDS -> barrier -> MFMA phases
// over the channel chunks (k_pw_bwd: 32-channel chunks on 4 waves, ~25 phases and ~30 us per tile and grid.y block).  What shortened the GRU
// step works here too: 512 threads = EIGHT waves per block and 64-channel chunks, i.e. half the phases per tile, each wave with half of the
// accumulators (dgrad: 4 pixel tiles x 2 halves of the M tiles; wgrad: 64 output tiles over 8 waves), and the weight-gradient operands in
// NATURAL layout read through the LDS transpose read (no ds_write_b16 scatters).  The (g, z) vectors and the input halo of the next chunk
// are register-prefetched as in k_pw_bwd (weight fragments are loaded before the prefetch is issued: vector loads retire in order).
#include "det_common.h"

template <int CIN, int COUT>
struct Pw8Cfg {
    static constexpr int NT = 512, TW = 8, TH = 8, TP = 64, CG = 8, CH = CG * 8;  // 64-channel chunks
    static constexpr int NKD = COUT / CH;                                          // dz chunks (dgrad K)
    static constexpr int CIB = CIN < 128 ? CIN : 128, COB = COUT < 128 ? COUT : 128;
    static constexpr int NBI = CIN / CIB, NBO = COUT / COB;                        // grid.y = weight-gradient blocks
    static constexpr int MTD = CIN / 16, MTW = MTD / 2;                            // dgrad M tiles: total / per wave (wave >> 2 picks the half)
    static constexpr int WTI = CIB / 16, WTO = COB / 16, NTW = WTI * WTO / 8;      // wgrad output tiles: per wave
    static constexpr int PD = CH + 8, PZ = COB + 8, PU = CIB + 8;                  // bf16 pitches: dz chunk | dz slab | u slab
    static constexpr int HP = (TW + 2) * (TH + 2);
    static constexpr int OFF_DZ = TP * PD * 2, OFF_U = OFF_DZ + TP * PZ * 2, OFF_XS = (OFF_U + TP * PU * 2 + 15) & ~15;
    static constexpr int OFF_PAR = OFF_XS + HP * CH * 4;
    static constexpr int SMEM = OFF_PAR + (12 * CIN + 6 * COUT) * 4;
    static_assert(MTD % 2 == 0 && (WTI * WTO) % 8 == 0, "tile split over 8 waves");
};

template <int CIN, int COUT>
__global__ __launch_bounds__(512) void k_pw_bwd8(Src2<bf16> x, const float* __restrict__ tra, const float* __restrict__ trb,
                                                 const float* __restrict__ wdw /*master [CIN][9]*/, GradSrc<bf16> gs, const bf16* __restrict__ z,
                                                 const float* __restrict__ bn, const float* __restrict__ coef, const void* __restrict__ wpk_d,
                                                 bf16* __restrict__ du, float* __restrict__ dwpw, float* __restrict__ ws, Tiling2 tg) {
    using C = Pw8Cfg<CIN, COUT>;
    constexpr int TW = C::TW, TH = C::TH, TP = C::TP, CG = C::CG, CH = C::CH, PD = C::PD, PZ = C::PZ, PU = C::PU, MTD = C::MTD, MTW = C::MTW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* tileD = reinterpret_cast<bf16*>(smem);               // [TP][PD]  current dz chunk (dgrad pixel operand)
    bf16* dzN = reinterpret_cast<bf16*>(smem + C::OFF_DZ);     // [TP][PZ]  dz of this block's cout range (wgrad operand, natural layout)
    bf16* uN = reinterpret_cast<bf16*>(smem + C::OFF_U);       // [TP][PU]  recomputed depthwise output of this block's cin range
    float* xs = reinterpret_cast<float*>(smem + C::OFF_XS);    // HaloStager planar tile of one 64-channel chunk
    float* s_trx = reinterpret_cast<float*>(smem + C::OFF_PAR);  // [CIN/8][3][8]
    float* s_wdw = s_trx + 3 * CIN;                               // [9][CIN]
    float* s_bn = s_wdw + 9 * CIN;                                // [3][COUT]
    float* s_cf = s_bn + 3 * COUT;                                // [3][COUT]
    const int H = tg.H, W = tg.W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 3 * CIN; i += C::NT) {
        const int g = i / 24, r = (i - g * 24) >> 3, c = g * 8 + (i & 7);
        s_trx[i] = c < x.Ca ? tra[r * x.Ca + c] : trb[r * x.Cb + (c - x.Ca)];
    }
    for (int i = tid; i < 9 * CIN; i += C::NT) {
        const int t = i / CIN, c = i - t * CIN;
        s_wdw[i] = wdw[c * 9 + t];
    }
    for (int i = tid; i < 3 * COUT; i += C::NT) {
        s_bn[i] = bn[i];
        s_cf[i] = coef[i];
    }
    {   // zero the three bf16 tiles once: padding columns stay zero
        const float zero8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = tid * 8; i < TP * (PD + PZ + PU); i += C::NT * 8) store8(tileD + i, zero8);
    }
    __syncthreads();

    const int by = blockIdx.y;
    const int ci_base = (by % C::NBI) * C::CIB, co_base = (by / C::NBI) * C::COB;
    const bool do_dgrad = by == 0;
    const int pxl = tid / CG, cg = tid % CG;
    const int ty = pxl / TW, tx = pxl % TW;
    const HaloStager<bf16, CG, TW, TH, C::NT> stager(tid, W);
    const int ntile = wave & 3, mhalf = wave >> 2;  // dgrad: this wave's 16-pixel tile and half of the M tiles

    f32x4 accw[C::NTW];
#pragma unroll
    for (int j = 0; j < C::NTW; ++j) accw[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    TileSched ts(tg.ntiles);
    GhatPend<bf16> gp;
    bool have_gp = false;  // gp already holds chunk 0 of the tile that starts (issued during the previous tile's wgrad phase)
    for (long t = ts.first; t < ts.end; t += ts.step) {
        const TileOrg org = tile_origin2<TW, TH>(tg, (int)t);
        PixIdx px;
        px.n = org.n;
        px.h = org.h0 + ty;
        px.w = org.w0 + tx;
        const bool pv = px.h < H && px.w < W;
        const long p = pix_linear(px, H, W);
        f32x4 accd[MTW];
#pragma unroll
        for (int b = 0; b < MTW; ++b) accd[b] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // ---- A: dz chunks (64 channels) -> tileD (dgrad operand) + dzN (wgrad operand); dgrad MFMAs
        if (!have_gp) issue_ghat8(gp, gs, z, COUT, p, px, H, W, cg * 8, pv);
        have_gp = false;
#pragma unroll 1
        for (int kc = 0; kc < C::NKD; ++kc) {
            float dz[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const int c0 = kc * CH + cg * 8;
            if (pv) {
                float gh[8], zv[8];
                finish_ghat8(gp, gs, COUT, s_bn, px, H, W, c0, gh, zv);
#pragma unroll
                for (int i = 0; i < 8; ++i) dz[i] = fmaf(s_cf[c0 + i], gh[i], fmaf(s_cf[COUT + c0 + i], zv[i], s_cf[2 * COUT + c0 + i]));
            }
            if (kc) __syncthreads();  // previous chunk's fragment reads of tileD are done
            store8(tileD + pxl * PD + cg * 8, dz);
            if (c0 >= co_base && c0 < co_base + C::COB) store8(dzN + pxl * PZ + (c0 - co_base), dz);
            __syncthreads();
            if (do_dgrad) {
                typename Mma<bf16>::Frag wf[2][MTW];
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                    for (int b = 0; b < MTW; ++b) wf[k2][b] = Mma<bf16>::load_w(wpk_d, (long)(kc * 2 + k2) * MTD + mhalf * MTW + b, lane);
                __builtin_amdgcn_sched_barrier(0);
                if (kc + 1 < C::NKD) issue_ghat8(gp, gs, z, COUT, p, px, H, W, (kc + 1) * CH + cg * 8, pv);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const typename Mma<bf16>::Frag pf = Mma<bf16>::load_p(tileD + k2 * 32, PD, ntile * 16, lane, 32);
#pragma unroll
                    for (int b = 0; b < MTW; ++b) accd[b] = Mma<bf16>::template mma<8>(wf[k2][b], pf, accd[b]);
                }
            } else if (kc + 1 < C::NKD) {
                issue_ghat8(gp, gs, z, COUT, p, px, H, W, (kc + 1) * CH + cg * 8, pv);
            }
        }
        // first input-halo chunk of phase C: in flight during the du stores
        typename HaloStager<bf16, CG, TW, TH, C::NT>::Pending hp;
        const int kcc0 = ci_base / CH, kcc1 = (ci_base + C::CIB) / CH;
        stager.issue(hp, x, kcc0 * CH, org, H, W, tid);
        // ---- B: store du (this wave: pixels ntile*16.., M tiles of its half)
        if (do_dgrad) {
            const int oq = ntile * 16 + (lane & 15);
            PixIdx q;
            q.n = org.n;
            q.h = org.h0 + oq / TW;
            q.w = org.w0 + oq % TW;
            if (q.h < H && q.w < W) {
                bf16* dst = du + pix_linear(q, H, W) * CIN + (mhalf * MTW) * 16 + (lane >> 4) * 4;
#pragma unroll
                for (int b = 0; b < MTW; ++b) store4(dst + b * 16, accd[b][0], accd[b][1], accd[b][2], accd[b][3]);
            }
        }
        // ---- C: u = dw3x3(x~) for this block's cin range -> uN (64 channels per staged halo tile)
        for (int kc = kcc0; kc < kcc1; ++kc) {
            __syncthreads();  // xs free (previous chunk's / tile's tap reads done)
            stager.commit(hp, s_trx, kc * CH, xs, tid);
            if (kc + 1 < kcc1) stager.issue(hp, x, (kc + 1) * CH, org, H, W, tid);
            __syncthreads();
            const int c0 = kc * CH + cg * 8;
            float u[8];
            dw_from_lds<CG, TW, TH>(xs, s_wdw, CIN, c0, cg, ty, tx, u);
            if (!pv) {
#pragma unroll
                for (int i = 0; i < 8; ++i) u[i] = 0.f;
            }
            store8_opaque(uN + pxl * PU + (c0 - ci_base), u);
        }
        __syncthreads();
        if (t + ts.step < ts.end) {  // chunk 0 of the NEXT tile: in flight during the wgrad MFMAs
            const TileOrg on = tile_origin2<TW, TH>(tg, (int)(t + ts.step));
            PixIdx pn;
            pn.n = on.n;
            pn.h = on.h0 + ty;
            pn.w = on.w0 + tx;
            issue_ghat8(gp, gs, z, COUT, pix_linear(pn, H, W), pn, H, W, cg * 8, pn.h < H && pn.w < W);
            have_gp = true;
        }
        // ---- D: dWpw += u^T dz, K = the tile's 64 pixels, both operands by LDS transpose read from the natural-layout slabs
        {
            const int prow = 4 * (lane >> 4) + ((lane & 15) >> 2), pcol = (lane & 3) * 4;
#pragma unroll
            for (int j = 0; j < C::NTW; ++j) {
                const int tt = wave + 8 * j, ti = tt % C::WTI, to = tt / C::WTI;
#pragma unroll
                for (int pc = 0; pc < TP / 32; ++pc) {
                    const bf16* ua = uN + (pc * 32 + prow) * PU + ti * 16 + pcol;
                    const bf16* da = dzN + (pc * 32 + prow) * PZ + to * 16 + pcol;
                    accw[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_tr8(ua, ua + 16 * PU), lds_tr8(da, da + 16 * PZ), accw[j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // ---- flush weight-gradient partial (master layout [COUT][CIN]): D[m = ci][n = co]
#pragma unroll
    for (int j = 0; j < C::NTW; ++j) {
        const int tt = wave + 8 * j, ti = tt % C::WTI, to = tt / C::WTI;
        const int co = co_base + to * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = ci_base + ti * 16 + (lane >> 4) * 4 + r;
            const float v = accw[j][r];
            if (ws)
                ws[(long)blockIdx.x * (CIN * COUT) + (long)co * CIN + ci] = v;
            else
                atomicAdd(&dwpw[(long)co * CIN + ci], v);
        }
    }
}

extern "C" {

// Instantiated where it beats k_pw_bwd (measured, B = 32 x 1024^2: (128,128) 220 -> 195 us, (128,256) 79 -> 69, (256,128) 71 -> 68,
// (256,256) 228 -> 184).  The 64-channel configurations run thousands of tiles per launch and were 10-30 % SLOWER on eight waves
// ((64,64) 211 -> 257 us): they stay on k_pw_bwd.  OCRS_PW8=0 disables the kernel.
long det_pw8_supported(int Cin, int Cout, int dtype) {
    static const int on = env_int("OCRS_PW8", 1);
    return on && dtype == 1 && (Cin == 128 || Cin == 256) && (Cout == 128 || Cout == 256);
}
void k_wgrad_partials_reduce_launch(const float* ws, int nb, int nelem, float* dw, int cin, int ldw, hipStream_t st);  // det_bwd.hip

// gx = number of x-blocks (the caller sizes the workspace as gx * Cin * Cout floats with the same rule: see ocrs_pw_bwd_ws_floats)
int det_pw8_launch(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* g1, const void* g2,
                   int pooled, const void* z, const float* bn, const float* coef, const void* wpk_d, void* du, float* dwpw, float* ws, int Cout, int N,
                   int H, int W, int gx, hipStream_t st) {
    const int Cin = Ca + Cb;
    OCRS_CHECK_ARG(det_pw8_supported(Cin, Cout, 1) && Ca % 8 == 0 && Cb % 8 == 0 && gx > 0);
    Src2<bf16> x{(const bf16*)xa, (const bf16*)xb, Ca, Cb};
    GradSrc<bf16> gs{(const bf16*)g1, (const bf16*)g2, pooled};
    const Tiling2 tg = make_tiling2(N, H, W, 8, 8);
#define PW8_CASE(CI_, CO_)                                                                                                                 \
    if (Cin == CI_ && Cout == CO_) {                                                                                                       \
        using CC = Pw8Cfg<CI_, CO_>;                                                                                                       \
        static DevOnce attr_set;                                                                                                      \
        if (attr_set.need()) {                                                                                                                   \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pw_bwd8<CI_, CO_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != \
                hipSuccess)                                                                                                                \
                return OCRS_ERR_HIP;                                                                                                       \
            attr_set.done();                                                                                                               \
        }                                                                                                                                  \
        OCRS_LAUNCH_T((k_pw_bwd8<CI_, CO_>), dim3(gx, CC::NBI * CC::NBO), dim3(512), CC::SMEM, st, x, tra, trb, wdw, gs, (const bf16*)z, bn, coef, \
                           wpk_d, (bf16*)du, dwpw, ws, tg);                                                                                \
    }
    PW8_CASE(128, 128) PW8_CASE(128, 256) PW8_CASE(256, 128) PW8_CASE(256, 256)
#undef PW8_CASE
    if (ws) k_wgrad_partials_reduce_launch(ws, gx, Cin * Cout, dwpw, Cin, Cin, st);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

}  // extern "C"
