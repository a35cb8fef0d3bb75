// DepthwiseConv block forward at the DEEP U-Net levels (gfx950, bf16, Cin in {32..256}, Cout in {64..256}), all channels of a tile at once --
// the forward counterpart of det_pwb.hip, same contract as k_dwpw_fwd (det_fwd.hip) without the fused max-pool epilogue:
//     z = Wpw . dw3x3(x~)  (pre-BatchNorm, bf16),  gstat += [sum z | sum z^2] of the stored values.
// k_dwpw_fwd walks a tile through 32-channel chunks (stage -> barrier -> taps -> barrier -> MFMA per chunk: 8 chunks at 256 channels); the
// deep launches are a few hundred tiles, so that chain IS the launch time.  Here a tile is three phases:
//   A  the whole input tile + ring (all Cin channels) is loaded at once (the next tile's loads are issued behind barrier 1, after the wave's
//      weight fragments -- vector loads retire in order) and staged as x~ in bf16 LDS,
//   B  u = dw3x3(x~) on the VALU for all channels -> uN [pixels][Cin] (bf16),
//   C  z = Wpw u on MFMA (K = Cin from uN, packed weight fragments preloaded), store + per-channel sums in registers.
#include "det_common.h"

namespace {
template <int CIN, int COUT>
struct DwfCfg {
    static constexpr int NT = 512, NW = 8;
    static constexpr int TW = 8, TH = (CIN >= 128 || COUT > 128) ? 4 : 8, TP = TW * TH, NNT = TP / 16, HWp = TW + 2, HP = HWp * (TH + 2);
    static constexpr int CGI = CIN / 8;
    static constexpr int PXC = CIN + 8;                              // bf16 pitch
    static constexpr int NXI = (HP * CGI + NT - 1) / NT;             // (staged pixel, cin group) items per thread
    static constexpr int NUI = (TP * CGI + NT - 1) / NT;             // (pixel, cin group) depthwise items per thread
    static constexpr int MTO = COUT / 16, NKD = CIN / 32;            // M tiles (output channels), K chunks
    static constexpr int MPW = MTO >= NW ? MTO / NW : 1;             // M tiles per wave
    static constexpr int NPW = MTO >= NW ? NNT : NNT * MTO / NW;     // N tiles (16 pixels) per wave
    static constexpr int OFF_U = HP * PXC * 2, OFF_PAR = (OFF_U + TP * PXC * 2 + 15) & ~15;
    static constexpr int SMEM = OFF_PAR + (3 * CIN + 9 * CIN + NW * 2 * COUT) * 4;
    static_assert(NT % CGI == 0 && NPW >= 1 && MTO >= 4, "role mapping");
};
__device__ __forceinline__ void unpack8w(const uint4& r, float (&v)[8]) {
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
    v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}
}  // namespace

template <int CIN, int COUT>
__global__ __launch_bounds__(512) void k_dwf(Src2<bf16> x, const float* __restrict__ tra, const float* __restrict__ trb, const float* __restrict__ wdw,
                                             const void* __restrict__ wpk, bf16* __restrict__ z, double* __restrict__ gstat, Tiling2 tg, FwdFin fin) {
    using C = DwfCfg<CIN, COUT>;
    constexpr int NT = C::NT, TW = C::TW, TP = C::TP, HWp = C::HWp, HP = C::HP, CGI = C::CGI, PXC = C::PXC;
    constexpr int NXI = C::NXI, NUI = C::NUI, MTO = C::MTO, NKD = C::NKD, MPW = C::MPW, NPW = C::NPW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* xs = reinterpret_cast<bf16*>(smem);                    // [HP][PXC]  x~ on the tile + ring (0 outside the image)
    bf16* uN = reinterpret_cast<bf16*>(smem + C::OFF_U);         // [TP][PXC]  depthwise output
    float* s_trx = reinterpret_cast<float*>(smem + C::OFF_PAR);  // [CIN/8][3][8]
    float* s_wdw = s_trx + 3 * CIN;                               // [9][CIN]
    float* s_st = s_wdw + 9 * CIN;                                // [wave][2][COUT] (final reduction)
    const int H = tg.H, W = tg.W;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    fill_tr8(s_trx, x, tra, trb, CIN, tid);
    for (int i = tid; i < 9 * CIN; i += NT) {
        const int t = i / CIN, c = i - t * CIN;
        s_wdw[i] = wdw[c * 9 + t];
    }
    {
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        for (int i = tid; i < C::OFF_PAR / 16; i += NT) reinterpret_cast<uint4*>(smem)[i] = z4;  // pad columns stay zero
    }
    __syncthreads();

    const int cgi = tid % CGI;
    const int m0w = MTO >= C::NW ? wave * MPW : wave % MTO, n0w = MTO >= C::NW ? 0 : (wave / MTO) * NPW;  // this wave's output tiles
    // the wave's weight fragments: loop-invariant, in registers for the whole launch
    Mma<bf16>::Frag wf[NKD][MPW];
#pragma unroll
    for (int kc = 0; kc < NKD; ++kc)
#pragma unroll
        for (int a = 0; a < MPW; ++a) wf[kc][a] = Mma<bf16>::load_w(wpk, (long)kc * MTO + m0w + a, lane);
    float s1[MPW][4], s2[MPW][4];
#pragma unroll
    for (int a = 0; a < MPW; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) s1[a][i] = s2[a][i] = 0.f;

    struct Raw {
        uint4 xr[NXI];
        unsigned okx;
    };
    auto issue = [&](Raw& r, const TileOrg& org) {
        r.okx = 0;
        const int c0 = cgi * 8;
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
    auto commit = [&](const Raw& r) {
        float sc[8], sh[8], lo[8];
        load8(s_trx + cgi * 24, sc);
        load8(s_trx + cgi * 24 + 8, sh);
        load8(s_trx + cgi * 24 + 16, lo);
#pragma unroll
        for (int j = 0; j < NXI; ++j) {
            if (HP * CGI % NT != 0 && tid + j * NT >= HP * CGI) break;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (r.okx & (1u << j)) {
                unpack8w(r.xr[j], v);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = max_lo(fmaf(v[i], sc[i], sh[i]), lo[i]);
            }
            store8_opaque(xs + ((tid + j * NT) / CGI) * PXC + cgi * 8, v);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    TileSched ts(tg.ntiles);
    Raw cur;
    if (ts.first < ts.end) issue(cur, tile_origin2<TW, C::TH>(tg, (int)ts.first));
    for (long t = ts.first; t < ts.end; t += ts.step) {
        const TileOrg org = tile_origin2<TW, C::TH>(tg, (int)t);
        commit(cur);
        __syncthreads();  // (1) xs complete
        if (t + ts.step < ts.end) issue(cur, tile_origin2<TW, C::TH>(tg, (int)(t + ts.step)));
        // ---- B: u = dw3x3(x~)
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
                        unpack8w(*reinterpret_cast<const uint4*>(xs + ((ty + tap / 3) * HWp + tx + tap % 3) * PXC + cgi * 8), v);
#pragma unroll
                        for (int i = 0; i < 8; ++i) u[j][i] = fmaf(wv[i], v[i], u[j][i]);
                    }
                }
                if (tap % 3 == 2) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < NUI; ++j)
                if (TP * CGI % NT == 0 || tid + j * NT < TP * CGI) store8_opaque(uN + ((tid + j * NT) / CGI) * PXC + cgi * 8, u[j]);
        }
        __syncthreads();  // (2) uN complete
        // ---- C: z = Wpw u
        {
            f32x4 acc[MPW][NPW];
#pragma unroll
            for (int a = 0; a < MPW; ++a)
#pragma unroll
                for (int b = 0; b < NPW; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < NKD; ++kc)
#pragma unroll
                for (int b = 0; b < NPW; ++b) {
                    const Mma<bf16>::Frag pf = Mma<bf16>::load_p(uN + kc * 32, PXC, (n0w + b) * 16, lane, 32);
#pragma unroll
                    for (int a = 0; a < MPW; ++a) acc[a][b] = Mma<bf16>::template mma<8>(wf[kc][a], pf, acc[a][b]);
                }
#pragma unroll
            for (int b = 0; b < NPW; ++b) {
                const int oq = (n0w + b) * 16 + (lane & 15);
                const int qh = org.h0 + oq / TW, qw = org.w0 + oq % TW;
                if (qh < H && qw < W) {
                    bf16* dst = z + (((long)org.n * H + qh) * W + qw) * COUT + m0w * 16 + (lane >> 4) * 4;
#pragma unroll
                    for (int a = 0; a < MPW; ++a) {
                        const f32x4 v = acc[a][b];
                        store4(dst + a * 16, v[0], v[1], v[2], v[3]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float q = Elem<bf16>::round(v[i]);
                            s1[a][i] += q;
                            s2[a][i] = fmaf(q, q, s2[a][i]);
                        }
                    }
                }
            }
        }
        // (no third barrier: the next commit writes xs, last read in phase B before barrier 2; uN is rewritten after the next barrier 1)
    }
    // ---- statistics: lanes -> wave slots -> block sums (fixed order) -> fp64 accumulators
    for (int i = tid; i < C::NW * 2 * COUT; i += NT) s_st[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int a = 0; a < MPW; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v1 = quad16_sum(s1[a][i]), v2 = quad16_sum(s2[a][i]);
            if ((lane & 15) == 0) {
                const int m = (m0w + a) * 16 + (lane >> 4) * 4 + i;
                s_st[(wave * 2 + 0) * COUT + m] = v1;  // (one writer per (wave, channel): waves that share an M tile own different slots)
                s_st[(wave * 2 + 1) * COUT + m] = v2;
            }
        }
    __syncthreads();
    for (int i = tid; i < 2 * COUT; i += NT) {
        const int which = i / COUT, m = i - which * COUT;
        float s = 0.f;
        for (int w = 0; w < C::NW; ++w) s += s_st[(w * 2 + which) * COUT + m];
        atomicAdd(&gstat[i], (double)s);
    }
    bn_finalize_last_block(fin, gstat, COUT, tid, NT, reinterpret_cast<int*>(smem));  // (the tiles are dead; s_st lives behind them)
}

extern "C" {

long det_dwf_supported(int Cin, int Cout, int dtype) {
    static const int on = env_int("OCRS_DWF", 1);
    return on && dtype == 1 && (Cin == 32 || Cin == 64 || Cin == 128 || Cin == 256) && (Cout == 64 || Cout == 128 || Cout == 256) && Cin * 8 >= Cout;
}

int det_dwf_launch(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* wpk, void* z,
                   double* gstat, int Cout, int N, int H, int W, hipStream_t st, const FwdFin& fin) {
    const int Cin = Ca + Cb;
    OCRS_CHECK_ARG(det_dwf_supported(Cin, Cout, 1) && Ca % 8 == 0 && Cb % 8 == 0);
    Src2<bf16> x{(const bf16*)xa, (const bf16*)xb, Ca, Cb};
    bool done = false;
#define DWF_CASE(CI_, CO_)                                                                                                                   \
    if (!done && Cin == CI_ && Cout == CO_) {                                                                                                \
        using CC = DwfCfg<CI_, CO_>;                                                                                                         \
        static DevOnce attr_set;                                                                                                        \
        if (attr_set.need()) {                                                                                                                     \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dwf<CI_, CO_>), hipFuncAttributeMaxDynamicSharedMemorySize, CC::SMEM) != \
                hipSuccess)                                                                                                                  \
                return OCRS_ERR_HIP;                                                                                                         \
            attr_set.done();                                                                                                                 \
        }                                                                                                                                    \
        const Tiling2 tg = make_tiling2(N, H, W, CC::TW, CC::TH);                                                                            \
        long g = tg.ntiles;                                                                                                                  \
        const long cap = (long)kNumCU * (CI_ <= 128 ? 2 : 1); /* resident blocks (<= 128 registers up to 128 input channels) */                                                                \
        if (g > cap) g = cap;                                                                                                                \
        if (g >= 8) g &= ~7L;                                                                                                                \
        hipLaunchKernelGGL((k_dwf<CI_, CO_>), dim3((int)g), dim3(512), CC::SMEM, st, x, tra, trb, wdw, wpk, (bf16*)z, gstat, tg, fin);       \
        done = true;                                                                                                                         \
    }
    DWF_CASE(32, 64) DWF_CASE(64, 64) DWF_CASE(64, 128) DWF_CASE(128, 64) DWF_CASE(128, 128) DWF_CASE(128, 256) DWF_CASE(256, 128) DWF_CASE(256, 256)
#undef DWF_CASE
    OCRS_CHECK_ARG(done);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

}  // extern "C"
