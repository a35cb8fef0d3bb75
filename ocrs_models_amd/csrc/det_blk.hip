// Fused backward of ONE DepthwiseConv block (reference ocrs_models/models.py:7-28) for the top U-Net levels (gfx950, bf16, Cin, Cout <= 16):
//
//     dz = BatchNorm/ReLU backward of the block's output gradient        (k_pw_bwd phase A)
//     du = Wpw^T dz                      pointwise dgrad, MFMA           (k_pw_bwd)      \  du lives ONLY in LDS here: the separate
//     dWpw += u^T dz                     pointwise wgrad, MFMA, K=pixels (k_pw_bwd)       > kernels write it to HBM (k_pw_bwd) and read
//     dx~ = dw3x3^T(du), dWdw += x~ (*) du   depthwise backward          (k_dw_bwd)      /  it back with a halo (k_dw_bwd)
//     [+ the BatchNorm-backward sums of the block(s) that produced the input, as k_dw_bwd<STATS>]
//
// One 8x16-pixel tile per iteration; everything that depends on neighbours is computed on the tile's 10x18 DOMAIN (halo ring
// included): dz and du for all 180 domain pixels (1.4x the dgrad work, but g, z, x are each read from HBM once with that halo and du
// never leaves the CU).  Per (block, pixel) bytes at Cin = Cout = 8: 83 instead of 118.  Same software pipeline as the separate
// kernels (next tile's raw vectors register-prefetched, LDS-only barriers), same two-stage flush of all weight gradients.
// Not covered (the separate kernels remain): fp32, max-pool-routed gradient sources, Cin or Cout > 16.
#include "det_common.h"

template <int CIN, int COUT>
struct BlkCfg {
    static constexpr int TH = 8, TW = 16, TP = TH * TW;                    // interior tile
    static constexpr int DW_ = TW + 2, DH_ = TH + 2, DP = DW_ * DH_;       // domain = interior + halo ring (180 pixels)
    static constexpr int DPP = (DP + 15) / 16 * 16;                         // padded to MFMA N tiles (192 = 12 tiles, 3 per wave)
    static constexpr int CGI = CIN / 8, CGO = COUT / 8, CQ = CIN / 4;
    static constexpr int PD = COUT + 8, PU = CIN + 8;                       // bf16 tile pitches (elements)
    static constexpr int NGI = (DP * CGO + 255) / 256, NXI = (DP * CGI + 255) / 256;  // prefetched (z, g) / x items per thread
    static constexpr int NQI = TP * CQ / 256;                               // depthwise-backward (pixel, channel quad) items per thread
    static constexpr int NROW = 11;                                         // per-channel partial rows: 9 taps + 2 BatchNorm-backward sums
    // LDS (bytes)
    static constexpr int OFF_TILED = 0;
    static constexpr int OFF_TILEU = OFF_TILED + DPP * PD * 2;
    static constexpr int OFF_ZRAW = OFF_TILEU + TP * PU * 2;
    static constexpr int OFF_TILEDU = (OFF_ZRAW + DP * CIN * 2 + 15) & ~15;
    static constexpr int OFF_XS = OFF_TILEDU + DPP * CIN * 4;
    static constexpr int OFF_PAR = OFF_XS + DP * CIN * 4;
    static constexpr int PAR_FLOATS = 3 * CIN + 9 * CIN + 3 * COUT + 3 * COUT + CIN;  // tr8 | wdw tap-major | bn | coef | mean
    static constexpr int TILE_BYTES = OFF_PAR + PAR_FLOATS * 4;
    static constexpr int RED_BYTES = (NROW * 4 * 256 + 4 * 256) * 4;
    static constexpr int SMEM = TILE_BYTES > RED_BYTES ? TILE_BYTES : RED_BYTES;
    static constexpr int PART = COUT * CIN + NROW * CIN;  // floats per block partial: dWpw [COUT][CIN] | dWdw [CIN][9] | sums [2][CIN]
};

#ifndef OCRS_BLK_BLOCKS
#define OCRS_BLK_BLOCKS 2  // 3 blocks/CU (168 VGPRs) spills inside the tile loop: measured 1.1-3x slower
#endif
template <int CIN, int COUT, bool STATS>
__global__ __launch_bounds__(256, OCRS_BLK_BLOCKS) void k_blk_bwd(Src2<bf16> x, const float* __restrict__ tra, const float* __restrict__ trb,
                                                    const float* __restrict__ wdw /*master [CIN][9]*/, const bf16* __restrict__ g1,
                                                    const bf16* __restrict__ g2, const bf16* __restrict__ z, const float* __restrict__ bn,
                                                    const float* __restrict__ coef, const void* __restrict__ wpk_d, bf16* __restrict__ gxa,
                                                    bf16* __restrict__ gxb, float* __restrict__ ws, const float* __restrict__ saved_a,
                                                    const float* __restrict__ saved_b, int stat_mask, Tiling2 tg) {
    using C = BlkCfg<CIN, COUT>;
    constexpr int TW = C::TW, TH = C::TH, DW_ = C::DW_, DP = C::DP, CGI = C::CGI, CGO = C::CGO, CQ = C::CQ, PD = C::PD, PU = C::PU;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* tileD = reinterpret_cast<bf16*>(smem + C::OFF_TILED);     // [DPP][PD]  dz on the domain
    bf16* tileU = reinterpret_cast<bf16*>(smem + C::OFF_TILEU);     // [TP][PU]   recomputed depthwise output (interior)
    bf16* zraw = reinterpret_cast<bf16*>(smem + C::OFF_ZRAW);       // [DP][CIN]  raw input (= the producers' z), STATS only
    float* tileDU = reinterpret_cast<float*>(smem + C::OFF_TILEDU); // [DPP][CIN] du on the domain (fp32)
    float* xs = reinterpret_cast<float*>(smem + C::OFF_XS);         // 2 planes [DP*CGI][4]: transformed input on the domain
    float* s_trx = reinterpret_cast<float*>(smem + C::OFF_PAR);     // [CIN/8][3][8]
    float* s_wdw = s_trx + 3 * CIN;                                  // [9][CIN]
    float* s_bn = s_wdw + 9 * CIN;                                   // [3][COUT]
    float* s_cf = s_bn + 3 * COUT;                                   // [3][COUT]
    float* s_mu = s_cf + 3 * COUT;                                   // [CIN] saved mean of the producer(s)
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
    if (STATS)
        for (int c = tid; c < CIN; c += 256) {
            const bool in_a = c < x.Ca, on = in_a ? (stat_mask & 1) : (stat_mask & 2);
            s_mu[c] = on ? (in_a ? saved_a[c] : saved_b[c - x.Ca]) : 0.f;
        }
    {   // zero the dz tile once (its padding rows / columns stay zero)
        const float zero8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = tid * 8; i < C::DPP * PD; i += 256 * 8) store8(tileD + i, zero8);
    }
    __syncthreads();

    // ---- tile-invariant descriptors
    const int cgo = tid % CGO, cgi = tid % CGI;
    int g_dyx[C::NGI], g_off[C::NGI];
#pragma unroll
    for (int j = 0; j < C::NGI; ++j) {
        const int d = (tid + j * 256) / CGO, dy = d / DW_, dx = d - dy * DW_;
        g_dyx[j] = dy | (dx << 16);
        g_off[j] = (dy * W + dx) * COUT + cgo * 8;
    }
    int x_dyx[C::NXI], x_poff[C::NXI];
#pragma unroll
    for (int j = 0; j < C::NXI; ++j) {
        const int d = (tid + j * 256) / CGI, dy = d / DW_, dx = d - dy * DW_;
        x_dyx[j] = dy | (dx << 16);
        x_poff[j] = dy * W + dx;
    }
    const bool xi_a = cgi * 8 < x.Ca;                       // source of this thread's input items
    const bf16* xi_base = xi_a ? x.a + cgi * 8 : x.b + (cgi * 8 - x.Ca);
    const int xi_pitch = xi_a ? x.Ca : x.Cb;
    const bool has_g2 = g2 != nullptr;

    // depthwise-backward items: (interior pixel, channel quad), the quad is the same for all items of a thread
    const int q = tid % CQ, qc0 = q * 4;
    const bool q_a = qc0 < x.Ca;
    bf16* gdst = q_a ? (gxa ? gxa + qc0 : nullptr) : (gxb ? gxb + (qc0 - x.Ca) : nullptr);
    const int gpitch = q_a ? x.Ca : x.Cb;
    float acc[9][4], st1[4], st2[4], mu4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        st1[i] = st2[i] = 0.f;
        mu4[i] = STATS ? s_mu[qc0 + i] : 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k][i] = 0.f;
    }
    f32x4 accw = (f32x4){0.f, 0.f, 0.f, 0.f};
    // dgrad weights (K = COUT <= 16 -> one chunk, M = CIN <= 16 -> one tile): one fragment, kept in registers
    typename Mma<bf16>::Frag wfd = Mma<bf16>::load_w(wpk_d, 0L, lane);
    asm volatile("" : "+v"(wfd.q.x), "+v"(wfd.q.y), "+v"(wfd.q.z), "+v"(wfd.q.w));

    // ---- software pipeline state: raw vectors of the NEXT tile
    Raw8<bf16> zr[C::NGI], g1r[C::NGI], g2r[C::NGI], xr[C::NXI];
    unsigned okg = 0, okx = 0;
    auto issue = [&](const TileOrg& o) {
        const long corner = ((long)o.n * H + (o.h0 - 1)) * W + (o.w0 - 1);  // domain corner pixel (may lie outside: never dereferenced)
        const bf16* zb = z + corner * COUT;
        const bf16* g1b = g1 + corner * COUT;
        const bf16* g2b = (has_g2 ? g2 : g1) + corner * COUT;
        okg = okx = 0;
#pragma unroll
        for (int j = 0; j < C::NGI; ++j) {
            const int h = o.h0 - 1 + (g_dyx[j] & 0xffff), w = o.w0 - 1 + (g_dyx[j] >> 16);
            const bool ok = (DP * CGO % 256 == 0 || tid + j * 256 < DP * CGO) && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
            zr[j] = load8_raw(ok ? zb + g_off[j] : z);
            g1r[j] = load8_raw(ok ? g1b + g_off[j] : g1);
            g2r[j] = load8_raw(ok ? g2b + g_off[j] : g1);
            okg |= ok ? 1u << j : 0u;
        }
        const bf16* xb = xi_base + corner * xi_pitch;
#pragma unroll
        for (int j = 0; j < C::NXI; ++j) {
            const int h = o.h0 - 1 + (x_dyx[j] & 0xffff), w = o.w0 - 1 + (x_dyx[j] >> 16);
            const bool ok = (DP * CGI % 256 == 0 || tid + j * 256 < DP * CGI) && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
            xr[j] = load8_raw(ok ? xb + __umul24(x_poff[j], xi_pitch) : xi_base);
            okx |= ok ? 1u << j : 0u;
        }
    };

    TileSched ts(tg.ntiles);
    TileIter<TW, TH> tit(tg, ts.first < ts.end ? ts.first : 0, ts.step);
    TileOrg org_next = tit.org();
    if (ts.first < ts.end) issue(org_next);
    for (long t = ts.first; t < ts.end; t += ts.step) {
        const TileOrg org = org_next;
        // ================= phase 1: commit the prefetched tile =================
        {   // dz = A * ghat + B * z + C on the domain (0 outside the image)
            float bs[8], bt[8], ca[8], cb[8], cc[8];
            load8(s_bn + cgo * 8, bs);
            load8(s_bn + COUT + cgo * 8, bt);
            load8(s_cf + cgo * 8, ca);
            load8(s_cf + COUT + cgo * 8, cb);
            load8(s_cf + 2 * COUT + cgo * 8, cc);
#pragma unroll
            for (int j = 0; j < C::NGI; ++j) {
                __builtin_amdgcn_sched_barrier(0);
                const int it = tid + j * 256;
                if (DP * CGO % 256 == 0 || it < DP * CGO) {
                    float dz[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (okg & (1u << j)) {
                        float zv[8], ga[8], gb[8];
                        unpack8(zr[j], zv);
                        unpack8(g1r[j], ga);
                        unpack8(g2r[j], gb);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float gsum = has_g2 ? ga[i] + gb[i] : ga[i];
                            const float gh = fmaf(zv[i], bs[i], bt[i]) > 0.f ? gsum : 0.f;
                            dz[i] = fmaf(ca[i], gh, fmaf(cb[i], zv[i], cc[i]));
                        }
                    }
                    store8_opaque(tileD + (it / CGO) * PD + cgo * 8, dz);
                }
            }
        }
        {   // x~ = max(x * scale + shift, lo) on the domain -> planar fp32 tile; raw x (the producers' z) -> zraw
            const float* tp = s_trx + cgi * 24;
            float sc[8], sh[8], lo[8];
            load8(tp, sc);
            load8(tp + 8, sh);
            load8(tp + 16, lo);
#pragma unroll
            for (int j = 0; j < C::NXI; ++j) {
                const int it = tid + j * 256;
                if (DP * CGI % 256 == 0 || it < DP * CGI) {
                    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (okx & (1u << j)) {
                        unpack8(xr[j], v);
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = max_lo(fmaf(v[i], sc[i], sh[i]), lo[i]);
                    }
                    store4(xs + it * 4, v[0], v[1], v[2], v[3]);
                    store4(xs + DP * CGI * 4 + it * 4, v[4], v[5], v[6], v[7]);
                    if (STATS) *reinterpret_cast<uint4*>(zraw + (it / CGI) * CIN + cgi * 8) = xr[j].a;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + ts.step < ts.end) {
            tit.next();
            org_next = tit.org();
            issue(org_next);
        }
        lds_barrier();
        // ================= phase 2: du on the domain (MFMA) -> LDS;  u on the interior -> LDS =================
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int n0 = (wave * 3 + a) * 16;
            const typename Mma<bf16>::Frag pf = Mma<bf16>::load_p(tileD, PD, n0, lane, COUT);
            const f32x4 v = Mma<bf16>::template mma<8>(wfd, pf, (f32x4){0.f, 0.f, 0.f, 0.f});
            const int m0 = (lane >> 4) * 4;
            if (m0 < CIN) *reinterpret_cast<float4*>(tileDU + (n0 + (lane & 15)) * CIN + m0) = make_float4(v[0], v[1], v[2], v[3]);
        }
        if (tid < C::TP * CGI) {
            const int p = tid / CGI, ty = p / TW, tx = p % TW;
            float u[8], uz[8];
            dw_from_lds<CGI, TW, TH, false>(xs, s_wdw, CIN, cgi * 8, cgi, ty, tx, u);  // (this kernel stores its tile unswizzled)
            const bool pv = org.h0 + ty < H && org.w0 + tx < W;
#pragma unroll
            for (int i = 0; i < 8; ++i) uz[i] = pv ? u[i] : 0.f;
            store8_opaque(tileU + p * PU + cgi * 8, uz);
        }
        lds_barrier();
        // ================= phase 3a: pointwise weight gradient (K = the 128 interior pixels, one 32-pixel step per wave) =================
        {
            const int p = wave * 32 + 4 * (lane >> 4) + ((lane & 15) >> 2), pcol = (lane & 3) * 4;
            const int d = (p / TW + 1) * DW_ + p % TW + 1;  // the same pixel in domain coordinates
            const bf16* ua = tileU + p * PU + pcol;
            const bf16* da = tileD + d * PD + pcol;
            const bf16x8 fa = lds_tr8(ua, ua + 16 * PU), fb = lds_tr8(da, da + DW_ * PD);
            accw = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, accw, 0, 0, 0);
        }
        // ================= phase 3b: depthwise backward from the LDS-resident du =================
#pragma unroll
        for (int it2 = 0; it2 < C::NQI; ++it2) {
            const int p = (tid + it2 * 256) / CQ, ty = p / TW, tx = p % TW;
            const bool valid = org.h0 + ty < H && org.w0 + tx < W;
            const int dpix = (ty + 1) * DW_ + tx + 1;
            // x~ quad of this pixel: planar tile, item (pixel * CGI + q / 2), plane q % 2
            const float4 x4 = *reinterpret_cast<const float4*>(xs + (q & 1) * (DP * CGI * 4) + (dpix * CGI + (q >> 1)) * 4);
            const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
            float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                if (k % 3 == 0) __builtin_amdgcn_sched_barrier(0);  // bound the LDS reads in flight (register pressure)
                const float4 d4 = *reinterpret_cast<const float4*>(tileDU + ((ty + 2 - k / 3) * DW_ + (tx + 2 - k % 3)) * CIN + qc0);
                const float4 w4 = *reinterpret_cast<const float4*>(s_wdw + k * CIN + qc0);
                const float dd[4] = {d4.x, d4.y, d4.z, d4.w}, wk[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    g[i] = fmaf(wk[i], dd[i], g[i]);
                    acc[k][i] = fmaf(xv[i], dd[i], acc[k][i]);  // x~ is 0 outside the image: partial tiles add nothing
                }
            }
            if (valid && gdst) store4(gdst + (((long)org.n * H + org.h0 + ty) * W + org.w0 + tx) * gpitch, g[0], g[1], g[2], g[3]);
            if constexpr (STATS) {
                const uint2 zq = *reinterpret_cast<const uint2*>(zraw + dpix * CIN + qc0);
                const float zf[4] = {__uint_as_float(zq.x << 16), __uint_as_float(zq.x & 0xffff0000u), __uint_as_float(zq.y << 16),
                                     __uint_as_float(zq.y & 0xffff0000u)};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float gh = (valid && xv[i] > 0.f) ? Elem<bf16>::round(g[i]) : 0.f;  // x~ > 0 <=> bn(z) > 0 (ReLU producers)
                    st1[i] += gh;
                    st2[i] = fmaf(gh, zf[i] - mu4[i], st2[i]);
                }
            }
        }
        lds_barrier();  // all tile readers done before the next commit
    }

    // ================= block partials -> workspace =================
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [NROW*4][256] | [4][256]
    float* part = ws + (long)blockIdx.x * C::PART;
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[(k * 4 + i) * 256 + tid] = acc[k][i];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        red[(36 + i) * 256 + tid] = st1[i];
        red[(40 + i) * 256 + tid] = st2[i];
    }
    float* redw = red + C::NROW * 4 * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) redw[wave * 256 + r * 64 + lane] = accw[r];
    __syncthreads();
    for (int j = tid; j < C::NROW * CIN; j += 256) {  // rows 0..8: dWdw taps, 9: sum ghat, 10: sum ghat * (z - mean)
        const int row = j / CIN, c = j - row * CIN;
        const float* src = red + (row * 4 + (c & 3)) * 256 + (c >> 2);
        float v = 0.f;
        for (int m = 0; m < 256 / CQ; ++m) v += src[m * CQ];
        part[COUT * CIN + (row < 9 ? c * 9 + row : 9 * CIN + (row - 9) * CIN + c)] = v;
    }
    if (wave == 0) {  // dWpw: D[ci][co] summed over the 4 k-step waves -> master layout [COUT][CIN]
        const int co = lane & 15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = (lane >> 4) * 4 + r;
            const float v = (redw[r * 64 + lane] + redw[256 + r * 64 + lane]) + (redw[512 + r * 64 + lane] + redw[768 + r * 64 + lane]);
            if (ci < CIN && co < COUT) part[co * CIN + ci] = v;
        }
    }
}

// second stage: dwpw [COUT][CIN] += , dwdw [CIN][9] += , gsum_a / gsum_b (fp64, [2][Ca] / [2][Cb]) += the fused BatchNorm-backward sums
__global__ __launch_bounds__(256) void k_blk_partials_reduce(const float* __restrict__ ws, int nb, int CIN, int COUT, int Ca, float* __restrict__ dwpw,
                                                             float* __restrict__ dwdw, double* __restrict__ gsum_a, double* __restrict__ gsum_b,
                                                             const float* __restrict__ saved_a, const float* __restrict__ saved_b) {
    const int nelem = COUT * CIN + 11 * CIN;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= nelem) return;
    const int per = (nb + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = b0 + per < nb ? b0 + per : nb;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = b0;
    for (; b + 3 < b1; b += 4) {
        s0 += ws[(long)b * nelem + e];
        s1 += ws[(long)(b + 1) * nelem + e];
        s2 += ws[(long)(b + 2) * nelem + e];
        s3 += ws[(long)(b + 3) * nelem + e];
    }
    for (; b < b1; ++b) s0 += ws[(long)b * nelem + e];
    const float v = (s0 + s1) + (s2 + s3);
    if (e < COUT * CIN) {
        atomicAdd(&dwpw[e], v);
        return;
    }
    const int r = e - COUT * CIN;
    if (r < 9 * CIN) {
        atomicAdd(&dwdw[r], v);
        return;
    }
    const int which = (r - 9 * CIN) / CIN, c = (r - 9 * CIN) % CIN, Cb = CIN - Ca;
    if (c < Ca) {
        if (gsum_a) atomicAdd(&gsum_a[which * Ca + c], (double)(which ? v * saved_a[Ca + c] : v));
    } else if (gsum_b)
        atomicAdd(&gsum_b[which * Cb + (c - Ca)], (double)(which ? v * saved_b[Cb + (c - Ca)] : v));
}

static inline int blk_chunks(int nb) { return nb >= 512 ? 64 : (nb >= 128 ? 32 : (nb >= 16 ? 8 : 1)); }
static int blk_grid(int N, int H, int W) {
    const long ntiles = (long)N * ((W + 15) / 16) * ((H + 7) / 8);
    return persistent_grid(ntiles, 3);
}

extern "C" {

// 1 if ocrs_blk_bwd covers this block shape (else use ocrs_pw_bwd + ocrs_dw_bwd)
long ocrs_blk_bwd_supported(int Cin, int Cout, int pooled, int dtype) {
    return dtype == 1 && !pooled && (Cin == 8 || Cin == 16) && (Cout == 8 || Cout == 16);
}
long ocrs_blk_bwd_ws_floats(int Cin, int Cout, int N, int H, int W) { return (long)blk_grid(N, H, W) * (Cout * Cin + 11 * Cin); }

// Fused ocrs_pw_bwd + ocrs_dw_bwd of one DepthwiseConv block (same argument meaning; du is never materialised).
//   gxa / gxb: dL/dx~ split at channel Ca (gxb null iff Cb == 0); dwpw [Cout][Cin], dwdw [Cin][9] accumulated;
//   ws: ocrs_blk_bwd_ws_floats() floats; saved_a/gsum_a, saved_b/gsum_b: as in ocrs_dw_bwd (nullable).
int ocrs_blk_bwd(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* g1, const void* g2,
                 const void* z, const float* bn, const float* coef, const void* wpk_d, void* gxa, void* gxb, float* dwpw, float* dwdw, float* ws,
                 const float* saved_a, double* gsum_a, const float* saved_b, double* gsum_b, int Cout, int N, int H, int W, int dtype,
                 hipStream_t st) {
    const int Cin = Ca + Cb;
    OCRS_CHECK_ARG(xa && tra && wdw && g1 && z && bn && coef && wpk_d && dwpw && dwdw && ws && (Cb == 0 || (xb && trb)));
    OCRS_CHECK_ARG(ocrs_blk_bwd_supported(Cin, Cout, 0, dtype) && Ca % 8 == 0 && Cb % 8 == 0 && (long)N * H * W < (1L << 31));
    OCRS_CHECK_ARG((!gsum_a || saved_a) && (!gsum_b || (saved_b && Cb > 0)));
    const int stat_mask = (gsum_a ? 1 : 0) | (gsum_b ? 2 : 0);
    Src2<bf16> x{(const bf16*)xa, (const bf16*)xb, Ca, Cb};
    const Tiling2 tg = make_tiling2(N, H, W, 16, 8);
    const int nb = blk_grid(N, H, W);
#define BLK_CASE(CI_, CO_)                                                                                                                   \
    if (Cin == CI_ && Cout == CO_) {                                                                                                         \
        using CC = BlkCfg<CI_, CO_>;                                                                                                         \
        if (stat_mask)                                                                                                                       \
            hipLaunchKernelGGL((k_blk_bwd<CI_, CO_, true>), dim3(nb), dim3(256), CC::SMEM, st, x, tra, trb, wdw, (const bf16*)g1, (const bf16*)g2, \
                               (const bf16*)z, bn, coef, wpk_d, (bf16*)gxa, (bf16*)gxb, ws, saved_a, saved_b, stat_mask, tg);                 \
        else                                                                                                                                 \
            hipLaunchKernelGGL((k_blk_bwd<CI_, CO_, false>), dim3(nb), dim3(256), CC::SMEM, st, x, tra, trb, wdw, (const bf16*)g1, (const bf16*)g2, \
                               (const bf16*)z, bn, coef, wpk_d, (bf16*)gxa, (bf16*)gxb, ws, saved_a, saved_b, 0, tg);                          \
    }
    BLK_CASE(8, 8) BLK_CASE(8, 16) BLK_CASE(16, 8) BLK_CASE(16, 16)
#undef BLK_CASE
    const int ne = Cout * Cin + 11 * Cin;
    hipLaunchKernelGGL(k_blk_partials_reduce, dim3((ne + 255) / 256, blk_chunks(nb)), dim3(256), 0, st, ws, nb, Cin, Cout, Ca, dwpw, dwdw, gsum_a, gsum_b,
                       saved_a, saved_b);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

}  // extern "C"
