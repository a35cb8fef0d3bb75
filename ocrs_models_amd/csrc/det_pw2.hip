// Pointwise (1x1) conv backward of a DepthwiseConv block at the top U-Net levels (gfx950, bf16, Cin, Cout <= 32: one K chunk on both
// sides), two horizontally adjacent pixels per thread.  Same contract as k_pw_bwd (det_bwd.hip):
//     dz   = A * ghat + B * z + C                       (BatchNorm/ReLU backward; ghat optionally routed through MaxPool2d(2))
//     du   = Wpw^T dz                                   (MFMA dgrad, written to HBM for k_dw_bwd)
//     dWpw += u^T dz,  u = dw3x3(x~) recomputed         (MFMA wgrad, K = pixels, operands by LDS transpose read, two-stage flush)
// What the pair buys: the depthwise recompute reads 12 + 9 LDS vectors per pair instead of 2 x 18 (k_pw_bwd was 55 % LDS-busy), tiles
// are 16 rows tall (halo re-read 1.2-1.3x instead of 1.3-1.6x), per-tile bookkeeping is amortised over twice the pixels, and in the
// max-pool-routed case the horizontal window neighbour is the thread's own second pixel (the vertical one is lane ^ 16), so the
// first-maximum test costs 16 instead of 36 VALU per (pixel, channel) and 8 instead of 12 lane shuffles per pair.
// Software pipeline as in the other tiled kernels: the next tile's raw x / z / g vectors are register-prefetched, barriers order LDS only.
#include "det_common.h"

template <int CIN, int COUT>
struct Pw2Cfg {
    static constexpr int CGI = CIN / 8, CGO = COUT / 8, CGM = CGI > CGO ? CGI : CGO;
    static constexpr int TP = 512 / CGM, TH = 16, TW = TP / TH;        // 16x32 / 16x16 / 16x8 pixel tiles
    static constexpr int PD = COUT + 8, PU = CIN + 8;                  // bf16 tile pitches (elements)
    static constexpr int PTW = TP / 64;                                // dgrad N tiles (16 pixels) per wave
    static constexpr int MTD = (CIN + 15) / 16;                        // dgrad M tiles
    static constexpr int WTI = (CIN + 15) / 16, WTO = (COUT + 15) / 16, NTL = WTI * WTO;  // wgrad output tiles: 1, 2 or 4
    static constexpr int KSTEPS = TP / 32, KSTRIDE = 4 / NTL;          // 32-pixel k-steps, spread over the 4 / NTL waves of a tile
    static_assert(KSTEPS % KSTRIDE == 0, "k-steps divide evenly over the waves");
    static constexpr int HP = (TW + 2) * (TH + 2);
    static constexpr int OFF_U = TP * PD * 2, OFF_XS = (OFF_U + TP * PU * 2 + 15) & ~15, OFF_PAR = OFF_XS + HP * CIN * 4;
    static constexpr int SMEM = OFF_PAR + (12 * CIN + 6 * COUT) * 4;
};

template <int CIN, int COUT, bool PPOOL>
__global__ __launch_bounds__(256, 3) void k_pw_bwd2(Src2<bf16> x, const float* __restrict__ tra, const float* __restrict__ trb,
                                                    const float* __restrict__ wdw /*master [CIN][9]*/, const bf16* __restrict__ g1,
                                                    const bf16* __restrict__ g2, const bf16* __restrict__ z, const float* __restrict__ bn,
                                                    const float* __restrict__ coef, const void* __restrict__ wpk_d, bf16* __restrict__ du,
                                                    float* __restrict__ dwpw, float* __restrict__ ws, Tiling2 tg, int ldu /*du row stride*/, int ldw /*dwpw row stride*/) {
    using C = Pw2Cfg<CIN, COUT>;
    constexpr int TW = C::TW, TH = C::TH, CGI = C::CGI, CGO = C::CGO, CGM = C::CGM, PD = C::PD, PU = C::PU, MTD = C::MTD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* tileD = reinterpret_cast<bf16*>(smem);                // [TP][PD] dz
    bf16* tileU = reinterpret_cast<bf16*>(smem + C::OFF_U);     // [TP][PU] recomputed depthwise output
    float* xs = reinterpret_cast<float*>(smem + C::OFF_XS);     // HaloStager planar tile
    float* s_trx = reinterpret_cast<float*>(smem + C::OFF_PAR); // [CIN/8][3][8]
    float* s_wdw = s_trx + 3 * CIN;                              // [9][CIN]
    float* s_bn = s_wdw + 9 * CIN;                               // [3][COUT]
    float* s_cf = s_bn + 3 * COUT;                               // [3][COUT]
    const int H = tg.H, W = tg.W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    fill_tr8(s_trx, x, tra, trb, CIN, tid);
    for (int i = tid; i < 9 * CIN; i += 256) {
        const int t = i / CIN, c = i - t * CIN;
        s_wdw[i] = wdw[c * 9 + t];
    }
    for (int i = tid; i < 3 * COUT; i += 256) {
        s_bn[i] = bn[i];
        s_cf[i] = coef[i];
    }
    {   // zero both tiles once: padding columns stay zero
        const float zero8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = tid * 8; i < C::TP * (PD + PU); i += 256 * 8) store8(tileD + i, zero8);
    }
    __syncthreads();
    const HaloStager<bf16, CGI, TW, TH> stager(tid, W);

    const int ppair = tid / CGM, cg = tid % CGM;             // pixel pair and channel group of this thread
    const int ty = ppair / (TW / 2), tx = (ppair % (TW / 2)) * 2;
    const int pxl = ty * TW + tx;                             // left pixel; the right one is pxl + 1
    const int gz_off = (ty * W + tx) * COUT + cg * 8;         // (z, g) element offset of the left pixel from the tile origin
    const bool has_g2 = g2 != nullptr;
    const bool dz_thread = cg < CGO, u_thread = cg < CGI;

    typename Mma<bf16>::Frag wfd[MTD];  // dgrad weights: one K chunk, MTD tiles -> registers (loop-invariant)
#pragma unroll
    for (int b = 0; b < MTD; ++b) {
        wfd[b] = Mma<bf16>::load_w(wpk_d, (long)b, lane);
        asm volatile("" : "+v"(wfd[b].q.x), "+v"(wfd[b].q.y), "+v"(wfd[b].q.z), "+v"(wfd[b].q.w));
    }
    f32x4 accw = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- software-pipeline state
    typename HaloStager<bf16, CGI, TW, TH>::Pending pend;
    Raw8<bf16> zr[2], g1r[2], g2r[2];  // PPOOL: g1r[0] / g2r[0] only (both pixels lie in the same pool window)
    unsigned okm = 0;                  // bit e: pixel e inside the image; PPOOL bit 2: the pair lies inside a pool window
    auto issue_tile = [&](const TileOrg& o) {
        stager.issue(pend, x, 0, o, H, W, tid);
        const long tb = ((long)o.n * H + o.h0) * W + o.w0;
        const int h = o.h0 + ty, w = o.w0 + tx;
        okm = 0;
        if constexpr (PPOOL) {
            // bits 0/1: pixel inside the image (dz is computed for every image pixel); bit 2: the pair lies inside a pool window
            // (floor mode: the last odd row / column is in no window and gets ghat = 0).  tx is even: both pixels share the window.
            const int Hp = H >> 1, Wp = W >> 1;
            const bool in0 = dz_thread && h < H && w < W, in1 = dz_thread && h < H && w + 1 < W;
            const bool gv = in0 && h < 2 * Hp && w < 2 * Wp;
            const bf16* zp = z + tb * COUT + gz_off;
            zr[0] = load8_raw(in0 ? zp : z);
            zr[1] = load8_raw(in1 ? zp + COUT : z);
            const long pp = ((long)o.n * Hp + (h >> 1)) * Wp + (w >> 1);
            g1r[0] = load8_raw(gv ? g1 + pp * COUT + cg * 8 : g1);
            g2r[0] = load8_raw(gv && has_g2 ? g2 + pp * COUT + cg * 8 : g1);
            okm = (in0 ? 1u : 0u) | (in1 ? 2u : 0u) | (gv ? 4u : 0u);
        } else {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const bool ld = dz_thread && h < H && w + e < W;
                const long off = ld ? tb * COUT + gz_off + e * COUT : 0;
                zr[e] = load8_raw(z + off);
                g1r[e] = load8_raw(g1 + off);
                g2r[e] = load8_raw((has_g2 ? g2 : g1) + off);
                okm |= ld ? 1u << e : 0u;
            }
        }
    };

    TileSched ts(tg.ntiles);
    TileIter<TW, TH> tit(tg, ts.first < ts.end ? ts.first : 0, ts.step);
    TileOrg org_next = tit.org();
    if (ts.first < ts.end) issue_tile(org_next);
    for (long t = ts.first; t < ts.end; t += ts.step) {
        const TileOrg org = org_next;
        // ================= phase 1: dz of the pair -> tileD; x~ halo tile -> xs =================
        {
            // PPOOL: the other window row's two z vectors come from lane ^ 16 (a tile row is 16 lanes) -- shuffled by ALL lanes
            Raw8<bf16> zo[PPOOL ? 2 : 1];
            if constexpr (PPOOL) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    zo[e].a.x = __shfl_xor((int)zr[e].a.x, 16, 64);
                    zo[e].a.y = __shfl_xor((int)zr[e].a.y, 16, 64);
                    zo[e].a.z = __shfl_xor((int)zr[e].a.z, 16, 64);
                    zo[e].a.w = __shfl_xor((int)zr[e].a.w, 16, 64);
                }
            }
            if (dz_thread) {
                const int c0 = cg * 8;
                float bs[8], bt[8];
                load8(s_bn + c0, bs);
                load8(s_bn + COUT + c0, bt);
                float zv[2][8], gh[2][8];
                unpack8(zr[0], zv[0]);
                unpack8(zr[1], zv[1]);
                if constexpr (PPOOL) {
                    // first maximum of the 2x2 window in post-ReLU space.  With m0, m1 = this row's two values, omax = max of the other
                    // row, ef = 1 if the other row comes EARLIER in row-major order (this thread is on the odd row):
                    //   left  wins <=> m0 > max(0, omax*ef)      && m0 >= max(m1, omax*(1-ef))
                    //   right wins <=> m1 > max(0, omax*ef, m0)  && m1 >= omax*(1-ef)
                    float ga[8], gb[8], zo0[8], zo1[8];
                    unpack8(g1r[0], ga);
                    unpack8(g2r[0], gb);
                    unpack8(zo[0], zo0);
                    unpack8(zo[1], zo1);
                    const float ef = (ty & 1) ? 1.f : 0.f, lf = 1.f - ef;
                    const bool gv = (okm & 4u) != 0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float m0 = max_lo(fmaf(zv[0][i], bs[i], bt[i]), 0.f), m1 = max_lo(fmaf(zv[1][i], bs[i], bt[i]), 0.f);
                        const float omax = max_lo(max_lo(fmaf(zo0[i], bs[i], bt[i]), fmaf(zo1[i], bs[i], bt[i])), 0.f);
                        const float oe = omax * ef, ol = omax * lf;
                        const float gsum = has_g2 ? ga[i] + gb[i] : ga[i];
                        gh[0][i] = (gv && m0 > oe && m0 >= max_lo(m1, ol)) ? gsum : 0.f;
                        gh[1][i] = (gv && m1 > max_lo(oe, m0) && m1 >= ol) ? gsum : 0.f;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        float ga[8], gb[8];
                        unpack8(g1r[e], ga);
                        unpack8(g2r[e], gb);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float gsum = has_g2 ? ga[i] + gb[i] : ga[i];
                            gh[e][i] = fmaf(zv[e][i], bs[i], bt[i]) > 0.f ? gsum : 0.f;
                        }
                    }
                }
                float ca[8], cb[8], cc[8];
                load8(s_cf + c0, ca);
                load8(s_cf + COUT + c0, cb);
                load8(s_cf + 2 * COUT + c0, cc);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float dz[8];
                    const bool ok = (okm >> e) & 1u;
#pragma unroll
                    for (int i = 0; i < 8; ++i) dz[i] = ok ? fmaf(ca[i], gh[e][i], fmaf(cb[i], zv[e][i], cc[i])) : 0.f;
                    store8_opaque(tileD + (pxl + e) * PD + c0, dz);
                }
            }
            stager.commit(pend, s_trx, 0, xs, tid);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + ts.step < ts.end) {
            tit.next();
            org_next = tit.org();
            issue_tile(org_next);
        }
        lds_barrier();
        // ================= phase 2: du = Wpw^T dz (MFMA) -> HBM;  u of the pair -> tileU =================
#pragma unroll
        for (int a = 0; a < C::PTW; ++a) {
            const int n0 = (wave * C::PTW + a) * 16;
            const typename Mma<bf16>::Frag pf = Mma<bf16>::load_p(tileD, PD, n0, lane, CGO * 8);
            const int oq = n0 + (lane & 15), qh = org.h0 + oq / TW, qw = org.w0 + oq % TW;
            const bool ov = qh < H && qw < W;
            bf16* dst = du + (((long)org.n * H + qh) * W + qw) * ldu + (lane >> 4) * 4;
#pragma unroll
            for (int b = 0; b < MTD; ++b) {
                const f32x4 v = Mma<bf16>::template mma<8>(wfd[b], pf, (f32x4){0.f, 0.f, 0.f, 0.f});
                if (ov && b * 16 + (lane >> 4) * 4 < CIN) store4(dst + b * 16, v[0], v[1], v[2], v[3]);
            }
        }
        if (u_thread) {
            float u0[8], u1[8], w0[8], w1[8];
            dw2_from_lds<CGI, TW, TH>(xs, s_wdw, CIN, cg * 8, cg, ty, tx, u0, u1);
            const bool rv = org.h0 + ty < H, v0 = rv && org.w0 + tx < W, v1 = rv && org.w0 + tx + 1 < W;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                w0[i] = v0 ? u0[i] : 0.f;
                w1[i] = v1 ? u1[i] : 0.f;
            }
            store8_opaque(tileU + pxl * PU + cg * 8, w0);
            store8_opaque(tileU + (pxl + 1) * PU + cg * 8, w1);
        }
        lds_barrier();
        // ================= phase 3: dWpw += u^T dz, K = the tile's pixels (LDS transpose reads), k-steps spread over the waves =================
        {
            const int ti = (wave % C::NTL) % C::WTI, to = (wave % C::NTL) / C::WTI;
            const int prow = 4 * (lane >> 4) + ((lane & 15) >> 2), pcol = (lane & 3) * 4;
#pragma unroll
            for (int m = 0; m < C::KSTEPS / C::KSTRIDE; ++m) {
                const int pc = wave / C::NTL + m * C::KSTRIDE;
                const bf16* ua = tileU + (pc * 32 + prow) * PU + ti * 16 + pcol;
                const bf16* da = tileD + (pc * 32 + prow) * PD + to * 16 + pcol;
                accw = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_tr8(ua, ua + 16 * PU), lds_tr8(da, da + 16 * PD), accw, 0, 0, 0);
            }
        }
        lds_barrier();
    }
    // ---- flush: the 4 / NTL waves of an output tile are summed through LDS, then one partial per block (workspace) or float atomics
    float* red = reinterpret_cast<float*>(smem);  // [4][256]
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * 256 + r * 64 + lane] = accw[r];
    __syncthreads();
    if (wave < C::NTL) {
        const int ti = wave % C::WTI, to = wave / C::WTI;
        const int co = to * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = accw[r];
            for (int w2 = wave + C::NTL; w2 < 4; w2 += C::NTL) v += red[w2 * 256 + r * 64 + lane];
            const int ci = ti * 16 + (lane >> 4) * 4 + r;
            if (ci < CIN && co < COUT) {
                if (ws)
                    ws[(long)blockIdx.x * (CIN * COUT) + co * CIN + ci] = v;
                else
                    atomicAdd(&dwpw[co * ldw + ci], v);
            }
        }
    }
}

extern "C" {

static int pw2_grid(int Cin, int Cout, int N, int H, int W) {
    const int cgm = (Cin > Cout ? Cin : Cout) / 8, tw = (512 / cgm) / 16;
    const long ntiles = (long)N * ((W + tw - 1) / tw) * ((H + 15) / 16);
    return persistent_grid(ntiles, 3);  // resident blocks only (3 per CU by registers and LDS): each ends with one weight-gradient partial
}
// (internal to the library: ocrs_pw_bwd / ocrs_pw_bwd_ws_floats in det_bwd.hip dispatch here)
// 1 if k_pw_bwd2 covers this configuration (bf16, one K chunk: Cin, Cout in {8, 16, 32})
long det_pw2_supported(int Cin, int Cout, int dtype) {
    return dtype == 1 && ((Cin == 8 && (Cout == 8 || Cout == 16)) || (Cin == 16 && Cout >= 8 && Cout <= 32 && (Cout & (Cout - 1)) == 0) ||
                          (Cin == 32 && (Cout == 16 || Cout == 32)));
}
long det_pw2_ws_floats(int Cin, int Cout, int N, int H, int W) { return (long)pw2_grid(Cin, Cout, N, H, W) * Cin * Cout; }

void k_wgrad_partials_reduce_launch(const float* ws, int nb, int nelem, float* dw, int cin, int ldw, hipStream_t st);  // det_bwd.hip

int det_pw2_launch(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* g1, const void* g2,
                 int pooled, const void* z, const float* bn, const float* coef, const void* wpk_d, void* du, float* dwpw, float* ws, int Cout, int N,
                 int H, int W, int ldu, int ldw, hipStream_t st) {
    // ldu / ldw: row strides (elements) of du [P][ldu] and dwpw [Cout][ldw]: Cin for a whole block, larger when this launch handles one
    // half of a concat block's input channels (the caller offsets wdw / wpk_d / du / dwpw to the half)
    const int Cin = Ca + Cb;
    OCRS_CHECK_ARG(xa && tra && wdw && g1 && z && bn && coef && wpk_d && du && dwpw && (Cb == 0 || (xb && trb)));
    OCRS_CHECK_ARG(det_pw2_supported(Cin, Cout, 1) && Ca % 8 == 0 && Cb % 8 == 0 && (long)N * H * W < (1L << 31));
    Src2<bf16> x{(const bf16*)xa, (const bf16*)xb, Ca, Cb};
    const int nb = pw2_grid(Cin, Cout, N, H, W);
#define PW2_CASE(CI_, CO_)                                                                                                                  \
    if (Cin == CI_ && Cout == CO_) {                                                                                                        \
        using CC = Pw2Cfg<CI_, CO_>;                                                                                                        \
        const Tiling2 tg = make_tiling2(N, H, W, CC::TW, CC::TH);                                                                           \
        if (pooled)                                                                                                                         \
            OCRS_LAUNCH_T((k_pw_bwd2<CI_, CO_, true>), dim3(nb), dim3(256), CC::SMEM, st, x, tra, trb, wdw, (const bf16*)g1, (const bf16*)g2, \
                               (const bf16*)z, bn, coef, wpk_d, (bf16*)du, dwpw, ws, tg, ldu, ldw);                                                   \
        else                                                                                                                                \
            OCRS_LAUNCH_T((k_pw_bwd2<CI_, CO_, false>), dim3(nb), dim3(256), CC::SMEM, st, x, tra, trb, wdw, (const bf16*)g1, (const bf16*)g2, \
                               (const bf16*)z, bn, coef, wpk_d, (bf16*)du, dwpw, ws, tg, ldu, ldw);                                                   \
    }
    PW2_CASE(8, 8) PW2_CASE(8, 16) PW2_CASE(16, 8) PW2_CASE(16, 16) PW2_CASE(16, 32) PW2_CASE(32, 16) PW2_CASE(32, 32)
#undef PW2_CASE
    if (ws) k_wgrad_partials_reduce_launch(ws, nb, Cin * Cout, dwpw, Cin, ldw, st);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

}  // extern "C"
