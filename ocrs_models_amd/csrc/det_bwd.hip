// Detection U-Net backward kernels (gfx950).  Autograd of ocrs_models/models.py:7-143 restated as fused passes.
//
// Backward of one DepthwiseConv block  x~ -> u = dw3x3(x~) -> z = pw(u) -> y = relu(bn(z)):
//   k_bn_bwd_reduce    sum ghat, sum ghat*zhat                 (ghat = dL/dy * [y>0], through a 2x2 max-pool if needed)
//   k_bn_bwd_finalize  -> dz = A*ghat + B*z + C per channel, dgamma, dbeta
//   k_pw_bwd           dz tile + recomputed u tile in LDS; MFMA dgrad (du = Wpw^T dz) and wgrad (dWpw += u^T dz): fp32 and the deep
//                      (Cin or Cout > 32) bf16 levels; bf16 levels 0-2 run k_pw_bwd2 (det_pw2.hip)
//   k_dw_bwd           dx~ = dw3x3^T(du), dWdw
// plus ConvTranspose2d dgrad / wgrad, head backward, first-block (1->8) backward.
#include "det_common.h"
#include "loss_state.h"

// ----------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_bn_bwd_reduce(GradSrc<T> gs, const T* __restrict__ z, const float* __restrict__ bn /*[3][C]*/,
                                                       const float* __restrict__ saved /*[2][C]*/, double* __restrict__ gsum /*[2][C]*/,
                                                       int C, int H, int W, long P) {
    extern __shared__ float s_acc[];  // [2][C] (unused) | [256][16] per-thread partials
    const int CG = C / 8;
    const long gtid = (long)blockIdx.x * 256 + threadIdx.x;
    const long nthr = (long)gridDim.x * 256;
    const int c0 = (int)(gtid % CG) * 8;
    float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float mu[8], rs[8], sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        mu[i] = saved[c0 + i];
        rs[i] = saved[C + c0 + i];
        sc[i] = bn[c0 + i];
        sh[i] = bn[C + c0 + i];
    }
    if (!gs.pooled) {
        auto add = [&](const Raw8<T>& zr, const Raw8<T>& g1r, const Raw8<T>& g2r) {
            float g[8], zv[8];
            unpack8(zr, zv);
            unpack8(g1r, g);
            if (gs.g2) {
                float g2[8];
                unpack8(g2r, g2);
#pragma unroll
                for (int i = 0; i < 8; ++i) g[i] += g2[i];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float gh = fmaf(zv[i], sc[i], sh[i]) > 0.f ? g[i] : 0.f;
                s1[i] += gh;
                s2[i] = fmaf(gh, (zv[i] - mu[i]) * rs[i], s2[i]);
            }
        };
        const long step = nthr / CG;
        long p = gtid / CG;
        // two pixels' vectors in flight per thread (the loop is a pure stream: one pixel per iteration left the kernel at 3.5 TB/s)
        if (gs.g2) {
            for (; p + step < P; p += 2 * step) {
                const Raw8<T> z0 = load8_raw(z + p * C + c0), a0 = load8_raw(gs.g1 + p * C + c0), b0 = load8_raw(gs.g2 + p * C + c0);
                const Raw8<T> z1 = load8_raw(z + (p + step) * C + c0), a1 = load8_raw(gs.g1 + (p + step) * C + c0), b1 = load8_raw(gs.g2 + (p + step) * C + c0);
                add(z0, a0, b0);
                add(z1, a1, b1);
            }
            if (p < P) add(load8_raw(z + p * C + c0), load8_raw(gs.g1 + p * C + c0), load8_raw(gs.g2 + p * C + c0));
        } else {
            for (; p + step < P; p += 2 * step) {
                const Raw8<T> z0 = load8_raw(z + p * C + c0), a0 = load8_raw(gs.g1 + p * C + c0);
                const Raw8<T> z1 = load8_raw(z + (p + step) * C + c0), a1 = load8_raw(gs.g1 + (p + step) * C + c0);
                add(z0, a0, a0);
                add(z1, a1, a1);
            }
            if (p < P) {
                const Raw8<T> a0 = load8_raw(gs.g1 + p * C + c0);
                add(load8_raw(z + p * C + c0), a0, a0);
            }
        }
    } else {
        // one thread per 2x2 pooling window: the pooled gradient goes to the first maximal element (post-ReLU space)
        const int Hp = H >> 1, Wp = W >> 1;
        const long Pp = (P / ((long)H * W)) * Hp * Wp;
        for (long pp = gtid / CG; pp < Pp; pp += nthr / CG) {
            const PixIdx q = decode_pixel(pp, Hp, Wp);
            const long base = ((long)q.n * H + 2 * q.h) * W + 2 * q.w;
            float g[8];
            load8(gs.g1 + pp * C + c0, g);
            if (gs.g2) {
                float g2[8];
                load8(gs.g2 + pp * C + c0, g2);
#pragma unroll
                for (int i = 0; i < 8; ++i) g[i] += g2[i];
            }
            float best[8], bz[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float zv[8];
                load8(z + (base + (long)(k >> 1) * W + (k & 1)) * C + c0, zv);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float y = fmaxf(fmaf(zv[i], sc[i], sh[i]), 0.f);
                    if (k == 0 || y > best[i]) {
                        best[i] = y;
                        bz[i] = zv[i];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float gh = best[i] > 0.f ? g[i] : 0.f;
                s1[i] += gh;
                s2[i] = fmaf(gh, (bz[i] - mu[i]) * rs[i], s2[i]);
            }
        }
    }
    // deterministic block sum: every thread parks its 16 partials in LDS, thread j adds the 256 / CG threads of its channel group in order
    // (float LDS atomics complete in arrival order: the sums differed in the last bits from run to run); across blocks: fp64 atomics, exact
    // for fp32 partials of comparable magnitude ("Reproducibility")
    float* s_all = s_acc + 2 * C;  // [256][16]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s_all[threadIdx.x * 16 + i] = s1[i];
        s_all[threadIdx.x * 16 + 8 + i] = s2[i];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * C; j += 256) {
        const int which = j / C, c = j - which * C, cg = c >> 3, i = (c & 7) + which * 8;
        float v = 0.f;
        for (int t = cg; t < 256; t += CG) v += s_all[t * 16 + i];
        atomicAdd(&gsum[j], (double)v);
    }
}

__global__ void k_bn_bwd_finalize(const double* __restrict__ gsum, long count, int C, const float* __restrict__ gamma,
                                  const float* __restrict__ saved, float* __restrict__ coef /*[3][C]*/, float* __restrict__ dgamma,
                                  float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double s1 = gsum[c], s2 = gsum[C + c];
    const double m1 = s1 / (double)count, m2 = s2 / (double)count;
    const double mean = saved[c], rstd = saved[C + c];
    const double A = (double)gamma[c] * rstd;
    coef[c] = (float)A;
    coef[C + c] = (float)(-A * rstd * m2);
    coef[2 * C + c] = (float)(A * (-m1 + mean * rstd * m2));
    dgamma[c] = (float)s2;
    dbeta[c] = (float)s1;
}

// ----------------------------------------------------------------------------------------------
// pointwise backward, fused dgrad + wgrad on MFMA.
//   dgrad : D[cin][pixel] = sum_cout Wpw[cout][cin] * dz[pixel][cout]        (K = cout, weights packed K=COUT, M=CIN)
//   wgrad : D[cin][cout]  = sum_pixel u[pixel][cin] * dz[pixel][cout]        (K = pixel; both operands from transposed LDS tiles)
// grid.y enumerates (cin-block, cout-block) pairs of the weight gradient (blocks of <=128x128); y == 0 also does dgrad.
// ----------------------------------------------------------------------------------------------
// Weight-gradient flush of a persistent block: a plain store of its partial into the workspace slot of this block (summed by
// k_wgrad_partials_reduce: deterministic, no contention) or, without a workspace, a float atomic (same-address atomics from thousands of
// blocks were 2.5 ms of the detection step).
__device__ __forceinline__ void flush_w(float* __restrict__ dw, float* __restrict__ ws, long idx, int nelem, float v) {
    if (ws)
        ws[(long)blockIdx.x * nelem + idx] = v;
    else
        atomicAdd(&dw[idx], v);
}
// dw[(e / cin) * ldw + e % cin] += sum_b ws[b][e]  (ldw = cin: plain dw[e]);  grid ceil(nelem / 32)
__global__ __launch_bounds__(256) void k_wgrad_partials_reduce(const float* __restrict__ ws, int nb, int nelem, float* __restrict__ dw, int cin,
                                                               int ldw) {
    const long e = (long)blockIdx.x * 32 + (threadIdx.x & 31);
    float s;
    if (!det_column_sum(ws, nb, nelem, e, s)) return;
    const long eo = ldw == cin ? e : (e / cin) * ldw + e % cin;
    dw[eo] += s;
}

template <int CIN, int COUT>
struct PwBwdCfg {
    static constexpr int CGI = (CIN < 32 ? CIN : 32) / 8;
    static constexpr int CGO = (COUT < 32 ? COUT : 32) / 8;
    static constexpr int CGM = CGI > CGO ? CGI : CGO;
    static constexpr int TP = 256 / CGM;
    static constexpr int PTW = TP / 64;
    static constexpr int MTD = (CIN + 15) / 16;
    static constexpr int NKD = COUT / (CGO * 8);
    static constexpr int CIB = CIN < 128 ? CIN : 128;
    static constexpr int COB = COUT < 128 ? COUT : 128;
    static constexpr int WTI = (CIB + 15) / 16;
    static constexpr int WTO = (COB + 15) / 16;
    static constexpr int NTW = (WTI * WTO + 3) / 4;
    static constexpr int NBI = CIN / CIB;
    static constexpr int NBO = COUT / COB;
    static constexpr int TPP_BF = TP + 8;  // transposed-tile pitch (elements)
    static constexpr int TPP_F = TP + 4;
    static constexpr int TH = 8, TW = TP / 8;  // 8x32 / 8x16 / 8x8 pixel tiles
    static constexpr int HP = (TW + 2) * (TH + 2);
    // elements of the LDS region behind tileD: the transposed dz / u tiles
    __host__ __device__ static constexpr int mid_el(int tpp) { return (WTO + WTI) * 16 * tpp; }
};

template <class T, int CIN, int COUT>
__global__ __launch_bounds__(256, ((CIN > 32 || COUT > 32) && CIN * COUT <= 8192 && Elem<T>::is_bf16) ? 2 : 1) void k_pw_bwd(Src2<T> x, const float* __restrict__ tra, const float* __restrict__ trb, const float* __restrict__ wdw /*master [CIN][9]*/,
                                                GradSrc<T> gs, const T* __restrict__ z, const float* __restrict__ bn /*[3][COUT]*/,
                                                const float* __restrict__ coef /*[3][COUT]*/, const void* __restrict__ wpk_d,
                                                T* __restrict__ du /*[P][CIN]*/, float* __restrict__ dwpw /*[COUT][CIN]*/,
                                                float* __restrict__ ws /*[gridDim.x][COUT][CIN] block partials, or null: float atomics*/, Tiling2 tg) {
    using Cfg = PwBwdCfg<CIN, COUT>;
    constexpr int TW = Cfg::TW, TH = Cfg::TH;
    const int H = tg.H, W = tg.W;
    constexpr int TP = Cfg::TP, PTW = Cfg::PTW, MTD = Cfg::MTD, CGI = Cfg::CGI, CGO = Cfg::CGO, CGM = Cfg::CGM;
    constexpr int PITCH = Mma<T>::LDS_PITCH;
    constexpr int TPP = Elem<T>::is_bf16 ? Cfg::TPP_BF : Cfg::TPP_F;
    constexpr int WTI = Cfg::WTI, WTO = Cfg::WTO, NTW = Cfg::NTW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* tileD = reinterpret_cast<T*>(smem);               // [TP][PITCH]
    T* dzT = tileD + TP * PITCH;                         // [WTO*16][TPP]
    T* uT = dzT + WTO * 16 * TPP;                        // [WTI*16][TPP]
    float* xs = reinterpret_cast<float*>(smem + (((TP * PITCH + Cfg::mid_el(TPP)) * sizeof(T) + 15) & ~15));  // [HP][CGI*8]
    float* s_par = xs + Cfg::HP * CGI * 8;
    float* s_trx = s_par;                // [CIN/8][3][8] (HaloStager layout)
    float* s_wdw = s_par + 3 * CIN;      // [9][CIN]
    float* s_bn = s_par + 12 * CIN;      // [3][COUT]
    float* s_cf = s_bn + 3 * COUT;       // [3][COUT]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    fill_tr8(s_trx, x, tra, trb, CIN, tid);  // [CIN/8][3][8]
    for (int i = tid; i < 9 * CIN; i += 256) {
        const int t = i / CIN, c = i - t * CIN;
        s_wdw[i] = wdw[c * 9 + t];
    }
    for (int i = tid; i < 3 * COUT; i += 256) {
        s_bn[i] = bn[i];
        s_cf[i] = coef[i];
    }
    {
        const float zero8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = tid * 8; i < TP * PITCH + Cfg::mid_el(TPP); i += 256 * 8) store8(tileD + i, zero8);  // padding rows / columns stay zero
    }
    __syncthreads();

    const int by = blockIdx.y;
    const int ci_base = (by % Cfg::NBI) * Cfg::CIB, co_base = (by / Cfg::NBI) * Cfg::COB;
    const bool do_dgrad = by == 0;
    const int pxl = tid / CGM, cg = tid % CGM;
    const int ty = pxl / TW, tx = pxl % TW;
    const HaloStager<T, CGI, TW, TH> stager(tid, W);

    f32x4 accw[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) accw[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    TileSched ts(tg.ntiles);
    GhatPend<T> gp;        // (g, z) loads of the next dz chunk (generic path)
    bool have_gp = false;  // gp already holds chunk 0 of the tile that starts (issued during the previous tile's wgrad phase)
    for (long t = ts.first; t < ts.end; t += ts.step) {
        const TileOrg org = tile_origin2<TW, TH>(tg, (int)t);
        PixIdx px;
        px.n = org.n;
        px.h = org.h0 + ty;
        px.w = org.w0 + tx;
        const bool pv = px.h < H && px.w < W;
        const long p = pix_linear(px, H, W);
        f32x4 accd[PTW][MTD];
#pragma unroll
        for (int a = 0; a < PTW; ++a)
#pragma unroll
            for (int b = 0; b < MTD; ++b) accd[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

        if constexpr (Cfg::NKD == 1 && CIN <= 32) {
            // ---- fast path (levels 0-1: one K chunk on both sides): ALL global loads of the tile are issued before the first barrier
            // (g, z for dz and the input halo for the depthwise recompute), 3 barriers per tile instead of 5.
            {
                float dz[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                const int c0 = cg * 8;
                if (cg < CGO && pv) {
                    float gh[8], zv[8];
                    load_ghat8(gs, z, COUT, s_bn, p, px, H, W, c0, gh, zv);
#pragma unroll
                    for (int i = 0; i < 8; ++i) dz[i] = fmaf(s_cf[c0 + i], gh[i], fmaf(s_cf[COUT + c0 + i], zv[i], s_cf[2 * COUT + c0 + i]));
                }
                stager.stage(x, s_trx, 0, org, H, W, xs, tid);
                if (cg < CGO) {
                    store8(tileD + pxl * PITCH + cg * 8, dz);
#pragma unroll
                    for (int i = 0; i < 8; ++i) Elem<T>::st(dzT + (c0 + i) * TPP + pxl, dz[i]);
                }
            }
            __syncthreads();
            {
                typename Mma<T>::Frag pf[PTW];
#pragma unroll
                for (int a = 0; a < PTW; ++a) pf[a] = Mma<T>::load_p(tileD, PITCH, (wave * PTW + a) * 16, lane, CGO * 8);
#pragma unroll
                for (int b = 0; b < MTD; ++b) {
                    const typename Mma<T>::Frag wf = Mma<T>::load_w(wpk_d, (long)b, lane);
#pragma unroll
                    for (int a = 0; a < PTW; ++a) accd[a][b] = Mma<T>::template mma<CGO * 2>(wf, pf[a], accd[a][b]);
                }
#pragma unroll
                for (int a = 0; a < PTW; ++a) {
                    const int oq = (wave * PTW + a) * 16 + (lane & 15);
                    const int qh = org.h0 + oq / TW, qw = org.w0 + oq % TW;
                    const long po = ((long)org.n * H + qh) * W + qw;
#pragma unroll
                    for (int b = 0; b < MTD; ++b) {
                        const int m0 = b * 16 + (lane >> 4) * 4;
                        if (qh < H && qw < W && m0 < CIN) {
                            const f32x4 v = accd[a][b];
                            store4(du + po * CIN + m0, v[0], v[1], v[2], v[3]);
                        }
                    }
                }
                if (cg < CGI) {
                    float u[8];
                    dw_from_lds<CGI, TW, TH>(xs, s_wdw, CIN, cg * 8, cg, ty, tx, u);
#pragma unroll
                    for (int i = 0; i < 8; ++i) Elem<T>::st(uT + (cg * 8 + i) * TPP + pxl, pv ? u[i] : 0.f);
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int tt = wave + 4 * j;
                if (tt < WTI * WTO) {
                    const int ti = tt % WTI, to = tt / WTI;
#pragma unroll
                    for (int pc = 0; pc < TP / 32; ++pc) {
                        const typename Mma<T>::Frag fa = Mma<T>::load_p(uT + pc * 32, TPP, ti * 16, lane, 32);
                        const typename Mma<T>::Frag fb = Mma<T>::load_p(dzT + pc * 32, TPP, to * 16, lane, 32);
                        accw[j] = Mma<T>::template mma<8>(fa, fb, accw[j]);
                    }
                }
            }
            __syncthreads();
            continue;
        }
        // ---- A: dz chunks -> tileD (dgrad operand) + dzT (wgrad operand) ; dgrad MFMA.
        // Software-pipelined: the (g, z) loads of chunk kc+1 are in flight while chunk kc goes through LDS and the MFMAs (these
        // deep-level launches are latency-bound: exposed, the loads were 40 % of their time).  In the dgrad blocks the prefetch is issued
        // AFTER the chunk's weight-fragment loads: vector loads retire in order, so a weight load issued behind the prefetch would wait
        // for it.
        const bool gact = cg < CGO && pv;
        if (!have_gp) issue_ghat8(gp, gs, z, COUT, p, px, H, W, cg * 8, gact);
        have_gp = false;
        for (int kc = 0; kc < Cfg::NKD; ++kc) {  // (compile-time trip count: with run-time bounds -- e.g. only the cout range a wgrad-only
                                                 // block needs -- hipcc's code for this loop got 15-20 % slower)
            float dz[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const int c0 = (kc * CGO + cg) * 8;
            if (gact) {
                float gh[8], zv[8];
                finish_ghat8(gp, gs, COUT, s_bn, px, H, W, c0, gh, zv);
#pragma unroll
                for (int i = 0; i < 8; ++i) dz[i] = fmaf(s_cf[c0 + i], gh[i], fmaf(s_cf[COUT + c0 + i], zv[i], s_cf[2 * COUT + c0 + i]));
            }
            if (kc) __syncthreads();
            if (cg < CGO) {
                store8(tileD + pxl * PITCH + cg * 8, dz);
                if (c0 >= co_base && c0 < co_base + Cfg::COB) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) Elem<T>::st(dzT + (c0 - co_base + i) * TPP + pxl, dz[i]);
                }
            }
            __syncthreads();
            const int c0n = ((kc + 1) * CGO + cg) * 8;
            if (do_dgrad) {
                typename Mma<T>::Frag pf[PTW];
                if constexpr (Elem<T>::is_bf16) {
                    typename Mma<T>::Frag wf[MTD];
#pragma unroll
                    for (int b = 0; b < MTD; ++b) wf[b] = Mma<T>::load_w(wpk_d, (long)kc * MTD + b, lane);
                    __builtin_amdgcn_sched_barrier(0);
                    if (kc + 1 < Cfg::NKD) issue_ghat8(gp, gs, z, COUT, p, px, H, W, c0n, gact);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int a = 0; a < PTW; ++a) pf[a] = Mma<T>::load_p(tileD, PITCH, (wave * PTW + a) * 16, lane, CGO * 8);
#pragma unroll
                    for (int b = 0; b < MTD; ++b)
#pragma unroll
                        for (int a = 0; a < PTW; ++a) accd[a][b] = Mma<T>::template mma<CGO * 2>(wf[b], pf[a], accd[a][b]);
                } else {
                    if (kc + 1 < Cfg::NKD) issue_ghat8(gp, gs, z, COUT, p, px, H, W, c0n, gact);
#pragma unroll
                    for (int a = 0; a < PTW; ++a) pf[a] = Mma<T>::load_p(tileD, PITCH, (wave * PTW + a) * 16, lane, CGO * 8);
#pragma unroll
                    for (int b = 0; b < MTD; ++b) {
                        const typename Mma<T>::Frag wf = Mma<T>::load_w(wpk_d, (long)kc * MTD + b, lane);
#pragma unroll
                        for (int a = 0; a < PTW; ++a) accd[a][b] = Mma<T>::template mma<CGO * 2>(wf, pf[a], accd[a][b]);
                    }
                }
            } else if (kc + 1 < Cfg::NKD) {
                issue_ghat8(gp, gs, z, COUT, p, px, H, W, c0n, gact);
            }
        }
        // first input-halo chunk of phase C: in flight during the du stores
        typename HaloStager<T, CGI, TW, TH>::Pending hp;
        const int kcc0 = ci_base / (CGI * 8), kcc1 = (ci_base + Cfg::CIB) / (CGI * 8);
        stager.issue(hp, x, kcc0 * CGI * 8, org, H, W, tid);
        // ---- B: store du
        if (do_dgrad) {
#pragma unroll
            for (int a = 0; a < PTW; ++a) {
                const int oq = (wave * PTW + a) * 16 + (lane & 15);
                PixIdx q;
                q.n = org.n;
                q.h = org.h0 + oq / TW;
                q.w = org.w0 + oq % TW;
                const bool ov = q.h < H && q.w < W;
                const long po = pix_linear(q, H, W);
#pragma unroll
                for (int b = 0; b < MTD; ++b) {
                    const int m0 = b * 16 + (lane >> 4) * 4;
                    if (ov && m0 < CIN) {
                        const f32x4 v = accd[a][b];
                        store4(du + po * CIN + m0, v[0], v[1], v[2], v[3]);
                    }
                }
            }
        }
        // ---- C: recompute u = dw3x3(x~) for this block's cin range -> uT (input tile + halo staged once in LDS)
        for (int kc = kcc0; kc < kcc1; ++kc) {
            __syncthreads();  // xs free (previous chunk / previous tile readers done)
            stager.commit(hp, s_trx, kc * CGI * 8, xs, tid);
            if (kc + 1 < kcc1) stager.issue(hp, x, (kc + 1) * CGI * 8, org, H, W, tid);  // next chunk: in flight during the tap phase
            __syncthreads();
            if (cg < CGI) {
                const int c0 = (kc * CGI + cg) * 8;
                float u[8];
                dw_from_lds<CGI, TW, TH>(xs, s_wdw, CIN, c0, cg, ty, tx, u);
#pragma unroll
                for (int i = 0; i < 8; ++i) Elem<T>::st(uT + (c0 - ci_base + i) * TPP + pxl, pv ? u[i] : 0.f);
            }
        }
        __syncthreads();
        if (t + ts.step < ts.end) {  // chunk 0 of the NEXT tile: in flight during the wgrad MFMAs
            const TileOrg on = tile_origin2<TW, TH>(tg, (int)(t + ts.step));
            PixIdx pn;
            pn.n = on.n;
            pn.h = on.h0 + ty;
            pn.w = on.w0 + tx;
            const bool pvn = pn.h < H && pn.w < W;
            issue_ghat8(gp, gs, z, COUT, pix_linear(pn, H, W), pn, H, W, cg * 8, cg < CGO && pvn);
            have_gp = true;
        }
        // ---- D: wgrad MFMA over the pixel dimension
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int tt = wave + 4 * j;
            if (tt < WTI * WTO) {
                const int ti = tt % WTI, to = tt / WTI;
#pragma unroll
                for (int pc = 0; pc < TP / 32; ++pc) {
                    const typename Mma<T>::Frag fa = Mma<T>::load_p(uT + pc * 32, TPP, ti * 16, lane, 32);
                    const typename Mma<T>::Frag fb = Mma<T>::load_p(dzT + pc * 32, TPP, to * 16, lane, 32);
                    accw[j] = Mma<T>::template mma<8>(fa, fb, accw[j]);
                }
            }
        }
        __syncthreads();
    }
    // ---- flush weight gradient (master layout [COUT][CIN])
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int tt = wave + 4 * j;
        if (tt < WTI * WTO) {
            const int ti = tt % WTI, to = tt / WTI;
            const int co = co_base + to * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = ci_base + ti * 16 + (lane >> 4) * 4 + r;
                if (ci < CIN && co < COUT) flush_w(dwpw, ws, (long)co * CIN + ci, CIN * COUT, accw[j][r]);
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------
// depthwise backward: dx~[p][c] = sum_tap w[c][tap] * du[p - off(tap)][c];  dW[c][tap] = sum_p x~[p][c] * du[p - off(tap)][c]
// 4 channels per thread (weights, transform and the 36 dW partials live in registers).
// STATS: additionally accumulate, for the PRODUCER blocks of the two sources, the BatchNorm-backward reductions of the gradient this
// kernel computes (sum ghat and sum ghat*(z - mean), ghat = dx~ * [bn(z) > 0]): the input x is that producer's raw z, so its
// k_bn_bwd_reduce pass over (g, z) -- 2.1 ms of the step -- disappears.  Partials go to workspace rows 9..10 (stat_mask bit 0 / 1 =
// source a / b wants them); k_dw_partials_reduce scales by rstd and adds them to the producers' gsum [2][C] (fp64).
#ifndef OCRS_DW_BLOCKS
#define OCRS_DW_BLOCKS 3  // 168 VGPRs: the per-channel load transform lives in LDS (3 vector reads per tile) instead of 12 registers, and the
                          // final-reduction scratch aliases the tiles, so that three blocks fit a CU (registers AND LDS).  Applies to the
                          // bf16 STATS variant (every launch of a training step); the fp32 / no-stats variants would spill and keep 2
#endif
template <class T, int CG, bool STATS>
__global__ __launch_bounds__(256, (Elem<T>::is_bf16 && STATS) ? OCRS_DW_BLOCKS : 2) void k_dw_bwd(Src2<T> x, const float* __restrict__ tra, const float* __restrict__ trb,
                                                const float* __restrict__ wdw /*master [C][9]*/, const T* __restrict__ du,
                                                T* __restrict__ gxa, T* __restrict__ gxb, float* __restrict__ dwdw /*[C][9]*/,
                                                float* __restrict__ ws /*[gridDim.x][C][9 (+2)] block partials or null*/,
                                                const float* __restrict__ saved_a, const float* __restrict__ saved_b, int stat_mask, Tiling2 tg, BwdLast bl) {
    // slab of SC = CG*8 channels per block (grid.y); a thread owns 4 channels (36 dW accumulators) of TWO horizontally adjacent pixels:
    // the pair shares 6 of its 9 du taps and all weights (12 + 9 LDS vector reads instead of 36), and the tile doubles to
    // 8 x (32/CG) pixels (halo re-read 1.33-1.56x instead of 1.4-1.875x, per-tile bookkeeping amortised over twice the pixels)
    constexpr int TH = 8, TW = 32 / CG, HP = (TW + 2) * (TH + 2), SC = CG * 8, CQ = 2 * CG;
    // pixel pitch of the du tile = 1.5 x SC floats: with the pair mapping a 16-lane group reads (16 / CQ) pixel PAIRS, i.e. every other
    // pixel; at pitch SC those land on the same banks two by two (PMC: 40 % of this kernel's LDS cycles were conflicts)
    constexpr int PS = SC + SC / 2;
    constexpr int NIT = (HP * CG + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float s_mem[];
    float* ds = s_mem;           // [HP][PS] du tile + halo (fp32, 0 outside the image)
    float* s_w = ds + HP * PS;   // [9][SC] weights, tap-major
    float* s_mu = s_w + 9 * SC;  // [SC] saved mean of the producer(s) (STATS)
    float* s_trp = s_mu + SC;    // [3][SC] load transform of this block's channel slab: scale | shift | lo
    const int C = x.Ca + x.Cb;
    const int H = tg.H, W = tg.W;
    const int cb = blockIdx.y * SC;
    const int tid = threadIdx.x;
    for (int i = tid; i < 9 * SC; i += 256) {
        const int t = i / SC, c = i - t * SC;
        s_w[i] = wdw[(cb + c) * 9 + t];
    }
    for (int i = tid; i < 3 * SC; i += 256) {
        const int k = i / SC, c = cb + (i - k * SC);
        s_trp[i] = c < x.Ca ? tra[k * x.Ca + c] : trb[k * x.Cb + (c - x.Ca)];
    }
    if (STATS)
        for (int c = tid; c < SC; c += 256) {
            const int cc = cb + c;
            const bool ia = cc < x.Ca, on = ia ? (stat_mask & 1) : (stat_mask & 2);
            s_mu[c] = on ? (ia ? saved_a[cc] : saved_b[cc - x.Ca]) : 0.f;
        }
    const int ppair = tid / CQ, q = tid % CQ;
    const int ty = ppair / (TW / 2), tx = (ppair % (TW / 2)) * 2;  // left pixel of the pair
    const int c0 = cb + q * 4;
    // The 144 FMAs per tile and thread (2 pixels x 4 channels x 9 taps x {dx~, dW}) run as PACKED fp32 FMAs (v_pk_fma_f32: two channels
    // per instruction, full rate on CDNA3/4): this kernel was ~60 % VALU-busy per SIMD with scalar FMAs.  Channel pairs (0,1) / (2,3).
    f32x2 acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t][0] = acc[t][1] = (f32x2){0.f, 0.f};
    const bool in_a = c0 < x.Ca;
    const T* xsrc = in_a ? x.a + c0 : x.b + (c0 - x.Ca);
    T* gdst = in_a ? (gxa ? gxa + c0 : nullptr) : (gxb ? gxb + (c0 - x.Ca) : nullptr);
    const int xp = in_a ? x.Ca : x.Cb;
    float st1[STATS ? 4 : 1], st2[STATS ? 4 : 1];
    if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) st1[i] = st2[i] = 0.f;
    }

    // Software pipeline: the raw du halo vectors AND this thread's two x quads of the NEXT tile are in flight while the current tile is
    // computed.  Loads are unconditional (invalid items read element 0 and are zeroed at use: a load under a divergent branch makes
    // hipcc put vmcnt(0) in front of the next load) and the barriers order LDS only (__syncthreads() would drain the prefetch).
    // Tile-invariant per-thread halo coordinates / offsets are hoisted.
    int hyx[NIT], poff[NIT];
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
        const int hp = (tid + j * 256) / CG, g8 = (tid + j * 256) - hp * CG;
        const int hy = hp / (TW + 2), hx = hp - hy * (TW + 2);
        hyx[j] = hy | (hx << 16);
        poff[j] = (hy * W + hx) * C + cb + g8 * 8;  // element offset from the halo's corner pixel (fits 32 bits: <= 10 rows)
    }
    const int xoff = (ty * W + tx) * xp;  // this thread's left pixel, relative to the tile origin
    struct Pre {
        Raw8<T> du[NIT];
        Raw4<T> x[2];
        unsigned ok;  // bit j: du item j inside the image; bits 30 / 31: left / right pixel inside the image
    };
    auto issue = [&](Pre& pr, const TileOrg& o) {
        const long corner = ((long)o.n * H + (o.h0 - 1)) * W + (o.w0 - 1);
        const T* dub = du + corner * C;
        pr.ok = 0;
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int h = o.h0 - 1 + (hyx[j] & 0xffff), w = o.w0 - 1 + (hyx[j] >> 16);
            const bool ok = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W && (HP * CG % 256 == 0 || j < NIT - 1 || tid + j * 256 < HP * CG);
            pr.du[j] = load8_raw(ok ? dub + poff[j] : du);
            pr.ok |= ok ? 1u << j : 0u;
        }
        const bool v0 = o.h0 + ty < H && o.w0 + tx < W, v1 = v0 && o.w0 + tx + 1 < W;
        const T* xb = xsrc + (((long)o.n * H + o.h0) * W + o.w0) * xp + xoff;
        pr.x[0] = load4_raw(v0 ? xb : xsrc);
        pr.x[1] = load4_raw(v1 ? xb + xp : xsrc);
        pr.ok |= (v0 ? 0x40000000u : 0u) | (v1 ? 0x80000000u : 0u);
    };
    Pre cur;  // consumed at the top of an iteration (commit + x transform), then immediately refilled for the next tile
    TileSched ts(tg.ntiles);
    TileIter<TW, TH> tit(tg, ts.first < ts.end ? ts.first : 0, ts.step);  // origin of the tile being prefetched, advanced without divisions
    TileOrg org_next = tit.org();
    if (ts.first < ts.end) issue(cur, org_next);
    for (long t = ts.first; t < ts.end; t += ts.step) {
        const TileOrg org = org_next;
        lds_barrier();  // previous tile's readers of ds are done
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int it = tid + j * 256;
            if (HP * CG % 256 == 0 || j < NIT - 1 || it < HP * CG) {
                float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (cur.ok & (1u << j)) unpack8(cur.du[j], v);
                store8(ds + (it / CG) * PS + (it % CG) * 8, v);
            }
        }
        const bool valid[2] = {(cur.ok & 0x40000000u) != 0, (cur.ok & 0x80000000u) != 0};
        float xv[2][4], zr[STATS ? 2 : 1][4];  // transformed input x~ (0 outside the image) / raw input of the two pixels
        float sc[4], sh[4], lo[4];
        {
            const f32x4 a = *reinterpret_cast<const f32x4*>(s_trp + q * 4), b = *reinterpret_cast<const f32x4*>(s_trp + SC + q * 4),
                        c = *reinterpret_cast<const f32x4*>(s_trp + 2 * SC + q * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sc[i] = a[i];
                sh[i] = b[i];
                lo[i] = c[i];
            }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float r[4];
            unpack4(cur.x[e], r);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (STATS) zr[e][i] = r[i];
                xv[e][i] = valid[e] ? max_lo(fmaf(r[i], sc[i], sh[i]), lo[i]) : 0.f;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool more = t + ts.step < ts.end;
        if (more) {
            tit.next();
            org_next = tit.org();
            issue(cur, org_next);
        }
        lds_barrier();
        f32x2 g[2][2] = {{{0.f, 0.f}, {0.f, 0.f}}, {{0.f, 0.f}, {0.f, 0.f}}};
        const f32x2 xp2[2][2] = {{{xv[0][0], xv[0][1]}, {xv[0][2], xv[0][3]}}, {{xv[1][0], xv[1][1]}, {xv[1][2], xv[1][3]}}};
        // tap k = (ky, kx) pairs x~[p] with du[p - off(k)] = halo (ty + 2 - ky, tx' + 2 - kx).  Window column c (halo column tx + c) serves
        // the left pixel with kx = 2 - c (c <= 2) and the right pixel with kx = 3 - c (c >= 1).  An out-of-image pixel has x~ = 0, so it
        // adds nothing to dW (keeps the 36 accumulators out of divergent control flow).
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float* drow = ds + ((ty + 2 - ky) * (TW + 2) + tx) * PS + q * 4;
            f32x2 wk[3][2];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(s_w + (ky * 3 + kx) * SC + q * 4);
                wk[kx][0] = (f32x2){w4[0], w4[1]};
                wk[kx][1] = (f32x2){w4[2], w4[3]};
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(drow + c * PS);
                const f32x2 d[2] = {{d4[0], d4[1]}, {d4[2], d4[3]}};
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (c <= 2) {
                        g[0][i] = __builtin_elementwise_fma(wk[2 - c][i], d[i], g[0][i]);
                        acc[ky * 3 + 2 - c][i] = __builtin_elementwise_fma(xp2[0][i], d[i], acc[ky * 3 + 2 - c][i]);
                    }
                    if (c >= 1) {
                        g[1][i] = __builtin_elementwise_fma(wk[3 - c][i], d[i], g[1][i]);
                        acc[ky * 3 + 3 - c][i] = __builtin_elementwise_fma(xp2[1][i], d[i], acc[ky * 3 + 3 - c][i]);
                    }
                }
            }
        }
        if (gdst) {
            T* gp = gdst + (((long)org.n * H + org.h0) * W + org.w0) * xp + xoff;
            if (valid[0]) store4(gp, g[0][0][0], g[0][0][1], g[0][1][0], g[0][1][1]);
            if (valid[1]) store4(gp + xp, g[1][0][0], g[1][0][1], g[1][1][0], g[1][1][1]);
        }
        if constexpr (STATS) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // the producer's pw_bwd reads the STORED (rounded) gradient; x~ > 0 <=> bn(z) > 0 for the ReLU producers that ask for sums
                    const float gh = xv[e][i] > 0.f ? Elem<T>::round(g[e][i >> 1][i & 1]) : 0.f;
                    st1[i] += gh;
                    st2[i] = fmaf(gh, zr[e][i] - s_mu[q * 4 + i], st2[i]);
                }
        }
    }
    __syncthreads();
    // block reduction of the 36 per-thread partials through LDS (plain stores, then a strided sum): cheap in registers,
    // runs once per persistent block
    constexpr int NROW = STATS ? 11 : 9;  // per-channel partial rows: 9 taps (+ the two BatchNorm-backward sums)
    float* s_red = s_mem;  // [NROW*4][256]: aliases the (now dead) tile and parameters
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) s_red[(t * 4 + i) * 256 + tid] = acc[t][i >> 1][i & 1];
    if constexpr (STATS) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s_red[(36 + i) * 256 + tid] = st1[i];
            s_red[(40 + i) * 256 + tid] = st2[i];
        }
    }
    __syncthreads();
    for (int j = tid; j < NROW * SC; j += 256) {
        const int t = j / SC, c = j - t * SC;
        const float* src = s_red + (t * 4 + (c & 3)) * 256 + (c >> 2);
        float v = 0.f;
        for (int m = 0; m < 256 / CQ; ++m) v += src[m * CQ];
        if (t < 9)
            flush_w(dwdw, ws, (cb + c) * 9 + t, C * NROW, v);
        else {
            ws[(long)blockIdx.x * (C * NROW) + C * t + cb + c] = v;  // rows 9, 10: [C] each (STATS requires a workspace)
            if (bl.raw) bwd_last_add(bl, C, cb + c, t - 9, v);
        }
    }
    if constexpr (STATS) {
        if (bl.raw) {  // the producers' sums finalised here by the last workgroup (BwdLast in det_common.h): k_dw_partials_reduce leaves the dependency chain
            __syncthreads();
            bwd_last_finish(bl, C, nullptr, nullptr, tid, 256, reinterpret_cast<int*>(s_mem));
        }
    }
}
// second stage of k_dw_bwd's flush: dwdw[e] += sum_b ws[b][e] (e < 9C); with stats (nrow = 11) rows 9/10 are scaled (row 10 by rstd)
// and added to the producers' BatchNorm-backward sums gsum_a [2][Ca] / gsum_b [2][Cb] (fp64), if requested.
__global__ __launch_bounds__(256) void k_dw_partials_reduce(const float* __restrict__ ws, int nb, int C, int Ca, int nrow, float* __restrict__ dwdw,
                                                            double* __restrict__ gsum_a, double* __restrict__ gsum_b,
                                                            const float* __restrict__ saved_a, const float* __restrict__ saved_b) {
    const int nelem = C * nrow;
    const long e = (long)blockIdx.x * 32 + (threadIdx.x & 31);
    float v;
    if (!det_column_sum(ws, nb, nelem, e, v)) return;  // single writer per element: deterministic
    if (e < 9 * C) {
        dwdw[e] += v;
        return;
    }
    const int which = (int)(e - 9 * C) / C, c = (int)(e - 9 * C) % C, Cb = C - Ca;
    if (c < Ca) {
        if (gsum_a) gsum_a[which * Ca + c] += (double)(which ? v * saved_a[Ca + c] : v);
    } else if (gsum_b)
        gsum_b[which * Cb + (c - Ca)] += (double)(which ? v * saved_b[Cb + (c - Ca)] : v);
}

// ----------------------------------------------------------------------------------------------
// first block (1 -> 8) backward: ONE streaming pass.  dz -> du[p] = sum_c Wpw[c] dz[p][c] stays in a register (the image needs no gradient);
// dWpw[c] = sum_p u[p] dz[p][c] and dWdw[k] = sum_p du[p] * img[p + off_k] both use the 3x3 neighbourhood of the image that the
// recompute of u needs anyway (Nb9).  (A first version wrote du and ran a second kernel that gathered du neighbours: +167 us, +268 MB.)
template <class T>
__global__ __launch_bounds__(256) void k_c1_bwd(const float* __restrict__ img, const float* __restrict__ wdw, const float* __restrict__ wpw,
                                                GradSrc<T> gs, const T* __restrict__ z, const float* __restrict__ bn,
                                                const float* __restrict__ coef, double* __restrict__ acc64 /*[17]: dWpw[8] | dWdw[9]*/, int H, int W,
                                                long P) {
    __shared__ float s_bn[24], s_cf[24], s_slots[4 * 17];
    if (threadIdx.x < 24) {
        s_bn[threadIdx.x] = bn[threadIdx.x];
        s_cf[threadIdx.x] = coef[threadIdx.x];
    }
    __syncthreads();
    float wd[9], wp[8], acc[17];  // acc: dWpw[0..8) | dWdw[8..17)
#pragma unroll
    for (int i = 0; i < 17; ++i) acc[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) wd[i] = wdw[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) wp[i] = wpw[i];
    // software pipeline over the grid-stride loop (see k_dwpw_c1_fwd): the image neighbourhood AND the (g, z) vectors of iteration i+1 are in
    // flight while iteration i is computed; two buffers, loop unrolled by two
    const long stride = (long)gridDim.x * 256;
    PixIter pit((long)blockIdx.x * 256 + threadIdx.x, stride, H, W);
    long p = (long)blockIdx.x * 256 + threadIdx.x;
    struct Buf {
        Nb9 nb;
        GhatPend<T> gp;
        PixIdx px;
    };
    auto issue = [&](Buf& b, long pp) {
        const bool act = pp < P;
        b.px = pit.cur();
        issue_ghat8(b.gp, gs, z, 8, act ? pp : 0, b.px, H, W, 0, act);
        nb9_issue(b.nb, img, b.px, H, W, act);
    };
    auto compute = [&](const Buf& b) {
        float nb[9];
        nb9_finish(b.nb, nb);
        float gh[8], zv[8];
        finish_ghat8(b.gp, gs, 8, s_bn, b.px, H, W, 0, gh, zv);
        float u = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) u = fmaf(wd[k], nb[k], u);
        u = Elem<T>::round(u);
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float dz = fmaf(s_cf[i], gh[i], fmaf(s_cf[8 + i], zv[i], s_cf[16 + i]));
            d = fmaf(wp[i], dz, d);
            acc[i] = fmaf(u, dz, acc[i]);
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[8 + k] = fmaf(d, nb[k], acc[8 + k]);
    };
    Buf bufA, bufB;
    issue(bufA, p);
    while (p < P) {
        pit.next();
        issue(bufB, p + stride);
        compute(bufA);
        p += stride;
        if (p >= P) break;
        pit.next();
        issue(bufA, p + stride);
        compute(bufB);
        p += stride;
    }
    // deterministic block sum, then fp64 accumulation across blocks (exact for fp32 partials of comparable magnitude -> order-independent;
    // float atomics straight into the fp32 gradients were not)
    const float tot = block_sum_det<17>(acc, s_slots);
    if (threadIdx.x < 17) atomicAdd(&acc64[threadIdx.x], (double)tot);
}

// ----------------------------------------------------------------------------------------------
// ConvTranspose2d weight gradient for the top U-Net levels (bf16, Cup <= 32, Cout <= 32):
//   dW[c][o][ky][kx] = sum_{n,i,j} x~[n,i,j,c] * g[n, 2i+ky, 2j+kx, o]           GEMM: M = c, N = (tap, o), K = input positions.
// Both operands are staged in LDS in their NATURAL NHWC order (x~ tile [pos][Cup], the (2TH+1) x (2TW+1) output region [pix][Cout]) and
// read as MFMA operands with the LDS transpose read (lds_tr8): no transposing stores, every g pixel is fetched once per tile
// (the generic gather kernel re-fetches it per tap and spent its time in ds_write_b16 scatters: 1.47 ms at level 0).
// Waves split K (32 positions each) and, for Cout = 32, the N tiles; block partial -> workspace; k_convt_wgrad_reduce sums partials.
template <int CUP, int COUT>
struct CtwCfg {
    static constexpr int TW = 16, TH = COUT >= 32 ? 4 : 8, TPOS = TW * TH;
    static constexpr int GH = 2 * TH + 1, GW = 2 * TW + 1;
    static constexpr int MT = CUP / 16, NCOL = 9 * COUT, NT = (NCOL + 15) / 16;
    static constexpr int KW = TPOS / 32, NW = 4 / KW, NTW = NT / NW;  // waves along K / along N, N tiles per wave
    static_assert(NT % NW == 0, "N tiles split evenly");
    static constexpr int XI = TPOS * CUP / 8, GI = GH * GW * COUT / 8;  // 16-byte staging items
    static constexpr int NXI = (XI + 255) / 256, NGI = (GI + 255) / 256;
    static constexpr int XS_EL = TPOS * CUP, GS_EL = GH * GW * COUT;
    // fused dgrad: dx[c] = sum_{tap,o} g[2i+ky][2j+kx][o] * W[c][o][tap]: K = 9*COUT in chunks of 32, M = CUP, N = positions
    static constexpr int DKC = (9 * COUT + 31) / 32, DNT = TPOS / 16 / 4;  // K chunks, N tiles (of 16 positions) per wave
    static constexpr int WD_EL = DKC * MT * 64 * 8;                          // cached packed dgrad weight fragments (bf16 elements)
    static constexpr int STAGE_BYTES = (XS_EL + GS_EL + 8 + WD_EL) * 2 + 3 * CUP * 4 + XS_EL * 2 + CUP * 4;  // + raw x tile and means (STATS)
    static constexpr int RED_BYTES = (KW - 1) * MT * NTW * NW * 256 * 4 + 256 * 8 * 4 + 4 * 2 * CUP * 4;  // ... | dbias [256][8] | stats [wave][2][CUP]
    static constexpr int SMEM = STAGE_BYTES > RED_BYTES ? STAGE_BYTES : RED_BYTES;
    static constexpr int PART = CUP * NT * 16 + COUT + 2 * CUP;  // floats per block partial: dW [CUP][NT*16] | dbias [COUT] | BN sums [2][CUP]
};

// The same staged tiles also give (fused, levels 0-2): the input gradient dx (dgrad GEMM, pixel operand = one ds_read_b128 per
// fragment straight from the natural-layout g region, packed weights cached in LDS) and dbias = sum of g (every output pixel is owned by
// exactly one tile).  One kernel + one reduce replace k_convt_dgrad + wgrad + k_channel_sum: g is read from HBM once instead of 3 times.
// STATS: x is the raw (pre-BatchNorm) output of a block whose ONLY consumer is this ConvTranspose, so dx is that block's complete output
// gradient: the kernel also produces the block's BatchNorm-backward sums (sum ghat, sum ghat * (z - mean); ghat = dx * [bn(z) > 0], the
// raw z comes from a second, untransformed copy of the staged tile) -- no k_bn_bwd_reduce pass over (dx, z).
template <int CUP, int COUT, bool STATS>
__global__ __launch_bounds__(256) void k_convt_wgrad_tr(const bf16* __restrict__ x, const float* __restrict__ tr /*[3][CUP]*/, const bf16* __restrict__ g,
                                                        const void* __restrict__ wpk_d, bf16* __restrict__ dx, float* __restrict__ ws,
                                                        const float* __restrict__ saved /*[2][CUP] mean | rstd (STATS)*/, int h, int w, int H, int W,
                                                        Tiling2 tg) {
    using C = CtwCfg<CUP, COUT>;
    constexpr int TW = C::TW, TH = C::TH, GW = C::GW, MT = C::MT, NTW = C::NTW, KW = C::KW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* xs = reinterpret_cast<bf16*>(smem);   // [TPOS][CUP]
    bf16* gsm = xs + C::XS_EL;                  // [GH*GW][COUT]
    bf16* zero8 = gsm + C::GS_EL;               // 8 zero elements (padding columns of the last N tile)
    uint4* s_wd = reinterpret_cast<uint4*>(zero8 + 8);  // [DKC*MT][64] packed dgrad weight fragments
    float* s_tr = reinterpret_cast<float*>(s_wd + C::DKC * MT * 64);  // [CUP/8][3][8]
    float* s_mu = s_tr + 3 * CUP;                                      // [CUP] saved mean (STATS)
    bf16* xraw = reinterpret_cast<bf16*>(s_mu + CUP);                  // [TPOS][CUP] untransformed copy of the x tile (STATS)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (STATS && tid < CUP) s_mu[tid] = saved[tid];
    float st1[STATS ? MT : 1][4], st2[STATS ? MT : 1][4];
    if constexpr (STATS) {
#pragma unroll
        for (int b = 0; b < MT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) st1[b][r] = st2[b][r] = 0.f;
    }
    {
        Src2<bf16> xsrc{x, nullptr, CUP, 0};
        fill_tr8(s_tr, xsrc, tr, nullptr, CUP, tid);
        if (tid < 4) reinterpret_cast<unsigned*>(zero8)[tid] = 0u;
        for (int i = tid; i < C::DKC * MT * 64; i += 256) s_wd[i] = reinterpret_cast<const uint4*>(wpk_d)[i];
    }
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // dbias partial of this thread's channel group (tid % (COUT/8))
    // ---- tile-invariant staging descriptors
    int xoff[C::NXI], goff[C::NGI], gyx[C::NGI];
#pragma unroll
    for (int j = 0; j < C::NXI; ++j) {
        const int it = tid + j * 256, pos = it / (CUP / 8), cgx = it % (CUP / 8);
        xoff[j] = ((pos / TW) * w + (pos % TW)) * CUP + cgx * 8;
    }
#pragma unroll
    for (int j = 0; j < C::NGI; ++j) {
        const int it = tid + j * 256, gp = it / (COUT / 8), cgg = it % (COUT / 8);
        const int gy = gp / GW, gx = gp % GW;
        goff[j] = (gy * W + gx) * COUT + cgg * 8;
        gyx[j] = gy | (gx << 16);
    }
    // ---- tile-invariant operand addresses (LDS, elements)
    const int i16 = lane & 15, kg = lane >> 4, ks = wave % KW, nh = wave / KW;
    const int pr = ks * 32 + 4 * kg + (i16 >> 2);  // position supplied by this lane in the first transpose read (second: +16 = next row)
    const bf16* a_ptr = xs + pr * CUP + (i16 & 3) * 4;
    const int pii = pr / TW, pjj = pr % TW;
    const bf16* b_ptr[NTW];
#pragma unroll
    for (int q = 0; q < NTW; ++q) {
        const int n0 = (nh * NTW + q) * 16 + (i16 & 3) * 4;
        const int tap = n0 / COUT, co = n0 % COUT;
        b_ptr[q] = n0 < C::NCOL ? gsm + ((2 * pii + tap / 3) * GW + 2 * pjj + tap % 3) * COUT + co : nullptr;
    }
    f32x4 acc[MT][NTW];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int q = 0; q < NTW; ++q) acc[a][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

    Raw8<bf16> xr[C::NXI], gr[C::NGI];
    unsigned okx = 0, okg = 0;
    auto issue = [&](long tn) {
        const TileOrg o = tile_origin2<TW, TH>(tg, (int)tn);  // origin in INPUT coordinates (i0, j0)
        const bf16* xb = x + (((long)o.n * h + o.h0) * w + o.w0) * CUP;
        const bf16* gb = g + (((long)o.n * H + 2 * o.h0) * W + 2 * o.w0) * COUT;
        okx = okg = 0;
#pragma unroll
        for (int j = 0; j < C::NXI; ++j) {
            const int it = tid + j * 256, pos = it / (CUP / 8);
            const bool ok = (C::XI % 256 == 0 || it < C::XI) && o.h0 + pos / TW < h && o.w0 + pos % TW < w;
            xr[j] = load8_raw(ok ? xb + xoff[j] : x);
            okx |= ok ? 1u << j : 0u;
        }
#pragma unroll
        for (int j = 0; j < C::NGI; ++j) {
            const bool ok = (C::GI % 256 == 0 || tid + j * 256 < C::GI) && 2 * o.h0 + (gyx[j] & 0xffff) < H && 2 * o.w0 + (gyx[j] >> 16) < W;
            gr[j] = load8_raw(ok ? gb + goff[j] : g);
            okg |= ok ? 1u << j : 0u;
        }
    };
    TileSched ts(tg.ntiles);
    if (ts.first < ts.end) issue(ts.first);
    __syncthreads();  // s_tr, zero8, s_wd
    for (long t = ts.first; t < ts.end; t += ts.step) {
        const TileOrg org = tile_origin2<TW, TH>(tg, (int)t);
        // output pixels owned by this tile (for dbias): the 2TH x 2TW block, plus the extra halo row / column for the last tile of
        // each direction (H = 2h+1 / W = 2w+1 leave one more output row / column than 2 * tiles * T{H,W} can own otherwise)
        const int own_h = org.h0 + TH >= h ? C::GH : 2 * TH, own_w = org.w0 + TW >= w ? C::GW : 2 * TW;
        // commit the prefetched tile: x~ = max(x*scale+shift, lo) rounded to bf16 (what the forward MFMA consumed), g as is
#pragma unroll
        for (int j = 0; j < C::NXI; ++j) {
            const int it = tid + j * 256;
            if (C::XI % 256 == 0 || it < C::XI) {
                float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (okx & (1u << j)) {
                    const float* tp = s_tr + (it % (CUP / 8)) * 24;
                    float sc[8], sh[8], lo[8];
                    unpack8(xr[j], v);
                    load8(tp, sc);
                    load8(tp + 8, sh);
                    load8(tp + 16, lo);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = max_lo(fmaf(v[i], sc[i], sh[i]), lo[i]);
                }
                store8(xs + it * 8, v);
                if constexpr (STATS) *reinterpret_cast<uint4*>(xraw + it * 8) = (okx & (1u << j)) ? xr[j].a : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < C::NGI; ++j) {
            const int it = tid + j * 256;
            if (C::GI % 256 == 0 || it < C::GI) *reinterpret_cast<uint4*>(gsm + it * 8) = (okg & (1u << j)) ? gr[j].a : make_uint4(0, 0, 0, 0);
            if ((okg & (1u << j)) && (gyx[j] & 0xffff) < own_h && (gyx[j] >> 16) < own_w) {
                float gv[8];
                unpack8(gr[j], gv);
#pragma unroll
                for (int i = 0; i < 8; ++i) bsum[i] += gv[i];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + ts.step < ts.end) issue(t + ts.step);
        lds_barrier();
        bf16x8 af[MT];
#pragma unroll
        for (int a = 0; a < MT; ++a) af[a] = lds_tr8(a_ptr + a * 16, a_ptr + a * 16 + 16 * CUP);
#pragma unroll
        for (int q = 0; q < NTW; ++q) {
            const bf16* bp = b_ptr[q] ? b_ptr[q] : zero8;
            const bf16x8 bfr = lds_tr8(bp, b_ptr[q] ? bp + 2 * GW * COUT : zero8);
#pragma unroll
            for (int a = 0; a < MT; ++a) acc[a][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr, acc[a][q], 0, 0, 0);
        }
        // ---- dgrad: this wave's DNT tiles of 16 positions
#pragma unroll
        for (int a = 0; a < C::DNT; ++a) {
            const int n = (wave * C::DNT + a) * 16 + i16, ii = n / TW, jj = n % TW;
            f32x4 dacc[MT];
#pragma unroll
            for (int b = 0; b < MT; ++b) dacc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < C::DKC; ++kc) {
                const int k0 = kc * 32 + kg * 8, tap = k0 / COUT, o0 = k0 % COUT;
                const bf16* gp = k0 < C::NCOL ? gsm + ((2 * ii + tap / 3) * GW + 2 * jj + tap % 3) * COUT + o0 : zero8;
                const uint4 pf = *reinterpret_cast<const uint4*>(gp);
#pragma unroll
                for (int b = 0; b < MT; ++b)
                    dacc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, s_wd[(kc * MT + b) * 64 + lane]),
                                                                      __builtin_bit_cast(bf16x8, pf), dacc[b], 0, 0, 0);
            }
            const int i = org.h0 + ii, jx = org.w0 + jj;
            if (i < h && jx < w) {
#pragma unroll
                for (int b = 0; b < MT; ++b)
                    store4(dx + (((long)org.n * h + i) * w + jx) * CUP + b * 16 + kg * 4, dacc[b][0], dacc[b][1], dacc[b][2], dacc[b][3]);
                if constexpr (STATS) {
#pragma unroll
                    for (int b = 0; b < MT; ++b) {
                        const int c0 = b * 16 + kg * 4;  // this lane's 4 channels of M tile b
                        const uint2 zr = *reinterpret_cast<const uint2*>(xraw + n * CUP + c0);
                        const float zv[4] = {__uint_as_float(zr.x << 16), __uint_as_float(zr.x & 0xffff0000u), __uint_as_float(zr.y << 16),
                                             __uint_as_float(zr.y & 0xffff0000u)};
                        const float* tp = s_tr + (c0 >> 3) * 24 + (c0 & 7);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            // the producer's pw_bwd reads the STORED (rounded) gradient; bn(z) > 0 <=> the ReLU passes
                            const bool on = fmaf(zv[r], tp[r], tp[8 + r]) > 0.f;
                            const float gh = on ? Elem<bf16>::round(dacc[b][r]) : 0.f;
                            st1[b][r] += gh;
                            st2[b][r] = fmaf(gh, zv[r] - s_mu[c0 + r], st2[b][r]);
                        }
                    }
                }
            }
        }
        lds_barrier();  // all operand reads done before the next commit overwrites the tiles
    }
    // ---- block reduction over the KW k-waves (same N half), then the block partial -> ws[block][CUP][NT*16]
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [(KW-1)][NW][MT*NTW][256]
    if (ks > 0) {
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int q = 0; q < NTW; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(((ks - 1) * C::NW + nh) * MT * NTW + a * NTW + q) * 256 + r * 64 + lane] = acc[a][q][r];
    }
    __syncthreads();
    {   // dbias: threads with the same channel group (tid % (COUT/8)) are summed through LDS (behind the wgrad reduction area)
        float* bred = red + (KW - 1) * MT * NTW * C::NW * 256;  // [256][8]
        float* sst = bred + 256 * 8;                            // [wave][2][CUP] BatchNorm-backward sums (STATS): one slot per wave, no atomics
#pragma unroll
        for (int i = 0; i < 8; ++i) bred[tid * 8 + i] = bsum[i];
        if constexpr (STATS) {
            // channel c = b*16 + kg*4 + r is shared by the 16 lanes of a kg group in every wave: sum over those lanes, one plain store per
            // (wave, channel); the four waves are added in order below (float LDS atomics would complete in arrival order)
#pragma unroll
            for (int b = 0; b < MT; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a1 = quad16_sum(st1[b][r]), a2 = quad16_sum(st2[b][r]);
                    if (i16 == 0) {
                        sst[(tid >> 6) * 2 * CUP + b * 16 + kg * 4 + r] = a1;
                        sst[(tid >> 6) * 2 * CUP + CUP + b * 16 + kg * 4 + r] = a2;
                    }
                }
        }
        __syncthreads();
        if (tid < COUT) {
            const int cg8 = tid / 8, i = tid % 8;
            float v = 0.f;
            for (int t2 = cg8; t2 < 256; t2 += COUT / 8) v += bred[t2 * 8 + i];
            ws[(long)blockIdx.x * C::PART + CUP * C::NT * 16 + tid] = v;
        }
        if (STATS && tid < 2 * CUP)
            ws[(long)blockIdx.x * C::PART + CUP * C::NT * 16 + COUT + tid] = (sst[tid] + sst[2 * CUP + tid]) + (sst[4 * CUP + tid] + sst[6 * CUP + tid]);
    }
    if (ks == 0) {
        float* part = ws + (long)blockIdx.x * C::PART;
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int q = 0; q < NTW; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[a][q][r];
                    for (int k2 = 1; k2 < KW; ++k2) v += red[(((k2 - 1) * C::NW + nh) * MT * NTW + a * NTW + q) * 256 + r * 64 + lane];
                    const int m = a * 16 + kg * 4 + r, n = (nh * NTW + q) * 16 + i16;
                    part[m * (C::NT * 16) + n] = v;
                }
    }
}
// dW[(c*COUT + o)*9 + tap] += sum over block partials of ws[b][c][tap*COUT + o];  grid (ceil(CUP*NCOL/256), partial chunks)
// With gsum (nullable): the per-block BatchNorm-backward sums [2][CUP] are added to gsum (fp64; row 1 scaled by rstd = saved[CUP + c]).
__global__ __launch_bounds__(256) void k_convt_wgrad_reduce(const float* __restrict__ ws, int nblocks, int CUP, int COUT, int NT16, float* __restrict__ dW,
                                                            float* __restrict__ dbias, const float* __restrict__ saved, double* __restrict__ gsum) {
    // one writer per output element over ALL block partials (fixed order: deterministic); element e of the logical output list
    //   [CUP*9*COUT dW | COUT dbias | 2*CUP BatchNorm-backward sums]  lives at column col(e) of a partial row
    const int part = CUP * NT16 + COUT + 2 * CUP, ne = CUP * 9 * COUT + COUT;
    const long e = (long)blockIdx.x * 32 + (threadIdx.x & 31);
    const long ntot = ne + (gsum ? 2 * CUP : 0);
    long col = 0;
    if (e < (long)CUP * 9 * COUT) {
        const int c = (int)(e / (9 * COUT)), n = (int)(e - (long)c * 9 * COUT);
        col = (long)c * NT16 + n;
    } else if (e < ntot)
        col = (long)CUP * NT16 + (e - (long)CUP * 9 * COUT);
    // (det_column_sum indexes ws[b * nelem + e]: pass the row pitch as nelem and the column as e; out-of-range threads read column 0, unused)
    float s;
    const bool writer = det_column_sum(ws, nblocks, part, e < ntot ? col : 0, s);
    if (!writer || e >= ntot) return;
    if (e >= ne) {  // BatchNorm-backward sums (fp64 accumulators of the producing block)
        const int idx = (int)(e - ne), which = idx / CUP, c = idx - which * CUP;
        gsum[idx] += (double)(which ? s * saved[CUP + c] : s);
        return;
    }
    if (e >= (long)CUP * 9 * COUT) {  // dbias [COUT]
        dbias[e - (long)CUP * 9 * COUT] += s;
        return;
    }
    const int c = (int)(e / (9 * COUT)), n = (int)(e - (long)c * 9 * COUT);
    const int tap = n / COUT, o = n - tap * COUT;
    dW[((long)c * COUT + o) * 9 + tap] += s;
}

// ----------------------------------------------------------------------------------------------
// ConvTranspose2d dgrad:  dx~[n,i,j,c] = sum_{ky,kx,o} g[n, 2i+ky, 2j+kx, o] * W[c,o,ky,kx]   (cropped rows/cols get no gradient)
// GEMM: M = c (Cup), N = input pixels, K = (tap, o) = 9*Cout (zero-padded to a multiple of 32).
template <class T, int MT>
__global__ __launch_bounds__(256) void k_convt_dgrad(const T* __restrict__ g, const void* __restrict__ wpk, T* __restrict__ dx, int Cup, int Cout,
                                                     int h, int w, int H, int W, int N, int MT_total) {
    constexpr int TP = 64, PITCH = Mma<T>::LDS_PITCH;
    __shared__ __attribute__((aligned(16))) T tile[TP * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pxl = tid >> 2, cg = tid & 3;
    const long P = (long)N * h * w;
    const long ntiles = (P + TP - 1) / TP;
    const int K = 9 * Cout;
    const int nkc = (K + 31) / 32;
    const int mt0 = blockIdx.y * MT;
    TileSched ts(ntiles);
    for (long t = ts.first; t < ts.end; t += ts.step) {
        const long p = t * TP + pxl;
        const bool pv = p < P;
        const PixIdx px = decode_pixel(pv ? p : 0, h, w);
        f32x4 acc[MT];
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kc = 0; kc < nkc; ++kc) {
            const int k0 = kc * 32 + cg * 8;
            float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (pv && k0 < K) {
                const int tap = k0 / Cout, o0 = k0 - tap * Cout;
                const int Y = 2 * px.h + tap / 3, X = 2 * px.w + tap % 3;
                if (Y < H && X < W) load8(g + (((long)px.n * H + Y) * W + X) * Cout + o0, v);
            }
            if (kc) __syncthreads();
            store8(tile + pxl * PITCH + cg * 8, v);
            __syncthreads();
            const typename Mma<T>::Frag pf = Mma<T>::load_p(tile, PITCH, wave * 16, lane, 32);
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                const typename Mma<T>::Frag wf = Mma<T>::load_w(wpk, (long)kc * MT_total + mt0 + b, lane);
                acc[b] = Mma<T>::template mma<8>(wf, pf, acc[b]);
            }
        }
        const long po = t * TP + wave * 16 + (lane & 15);
        if (po < P) {
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                const int m0 = (mt0 + b) * 16 + (lane >> 4) * 4;
                if (m0 < Cup) store4(dx + po * Cup + m0, acc[b][0], acc[b][1], acc[b][2], acc[b][3]);
            }
        }
        __syncthreads();
    }
}

// ConvTranspose2d wgrad (dW[c,o,ky,kx] = sum_{n,i,j} x~[n,i,j,c] * g[n,2i+ky,2j+kx,o]) runs on the generic k_wgrad_gather (rec_conv.hip)
// with A = x~ (load transform applied), B = g gathered at stride 2.

// per-channel sum over pixels (ConvTranspose2d bias gradient)
template <class T>
__global__ __launch_bounds__(256) void k_channel_sum(const T* __restrict__ g, double* __restrict__ out, int C, long P) {
    extern __shared__ float s_acc[];
    const int CG = C / 8;
    const long gtid = (long)blockIdx.x * 256 + threadIdx.x;
    const long nthr = (long)gridDim.x * 256;
    const int c0 = (int)(gtid % CG) * 8;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long step = nthr / CG;
    long p = gtid / CG;
    for (; p + 3 * step < P; p += 4 * step) {  // 4 independent loads in flight per thread (the grid is small: see the launcher)
        float v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) load8(g + (p + u * step) * C + c0, v[u]);
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] += (v[0][i] + v[1][i]) + (v[2][i] + v[3][i]);
    }
    for (; p < P; p += step) {
        float v[8];
        load8(g + p * C + c0, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] += v[i];
    }
    // deterministic block sum (every thread parks its 8 partials, thread j adds its channel's 256 / CG threads in order), fp64 across blocks
    float* s_all = s_acc + C;  // [256][8]
#pragma unroll
    for (int i = 0; i < 8; ++i) s_all[threadIdx.x * 8 + i] = s[i];
    __syncthreads();
    for (int j = threadIdx.x; j < C; j += 256) {
        float v = 0.f;
        for (int t = j >> 3; t < 256; t += CG) v += s_all[t * 8 + (j & 7)];
        atomicAdd(&out[j], (double)v);
    }
}

// head backward: pred = sigmoid(w . x~ + b);  gl = gpred * pred * (1 - pred);  gy[p][c] = gl * w[c];  dw, db.
template <class T>
__global__ __launch_bounds__(256) void k_head_bwd(const T* __restrict__ z, const float* __restrict__ tr, const float* __restrict__ w,
                                                  const float* __restrict__ pred, const float* __restrict__ gpred, T* __restrict__ gy,
                                                  double* __restrict__ acc64 /*[9]: dw[8] | db*/, const float* __restrict__ saved /*[2][8] or null*/,
                                                  double* __restrict__ gsum /*[2][8] or null*/, long P, float* __restrict__ gl_out /*[P] or null*/) {
    // gl_out != null (round 5): instead of the 8-channel gradient gy (16 B per pixel in bf16) only gl = dL/dlogit is written (4 B); the block
    // backward that consumes it (k_rs_bwd<..., HEAD>) forms gy = round(gl * w[c]) on the fly -- the same values: 12 B per pixel less written here
    // and 12 B less read there
    // gsum != null: also the BatchNorm-backward sums (sum ghat, sum ghat*zhat) of the block that produced z -- this kernel is that
    // block's only consumer and already reads z, so its k_bn_bwd_reduce pass (0.24 ms at 32x1024^2) is not needed
    __shared__ float s_slots[4 * 25];
    float wv[8], sc[8], sh[8], lo[8], mu[8], acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, st1[8], st2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        wv[i] = w[i];
        sc[i] = tr[i];
        sh[i] = tr[8 + i];
        lo[i] = tr[16 + i];
        mu[i] = gsum ? saved[i] : 0.f;
        st1[i] = st2[i] = 0.f;
    }
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
        const float pr = pred[p];
        const float gl = gpred[p] * (1.f - pr) * pr;
        float v[8], o[8];
        load8(z + p * 8, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float pre = fmaf(v[i], sc[i], sh[i]);
            const float xv = fmaxf(pre, lo[i]);
            acc[i] = fmaf(gl, xv, acc[i]);
            o[i] = gl * wv[i];
            const float gh = pre > 0.f ? Elem<T>::round(o[i]) : 0.f;  // what the block's pw_bwd will read back
            st1[i] += gh;
            st2[i] = fmaf(gh, v[i] - mu[i], st2[i]);
        }
        acc[8] += gl;
        if (gl_out) gl_out[p] = gl;
        else store8(gy + p * 8, o);
    }
    // deterministic block sums (no LDS float atomics), fp64 accumulation across blocks
    float all[25];
#pragma unroll
    for (int i = 0; i < 9; ++i) all[i] = acc[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        all[9 + i] = st1[i];
        all[17 + i] = st2[i];
    }
    const float tot = block_sum_det<25>(all, s_slots);
    if (threadIdx.x < 9) atomicAdd(&acc64[threadIdx.x], (double)tot);
    if (gsum && threadIdx.x >= 9 && threadIdx.x < 25) {
        const int i = threadIdx.x - 9;  // 0..7: sum ghat, 8..15: sum ghat*(z - mean) -> * rstd
        atomicAdd(&gsum[i], (double)(i < 8 ? tot : tot * saved[8 + (i - 8)]));
    }
}

// Balanced-BCE backward + head backward in ONE pass (round 5): dL/dpred is formed on the fly from what the loss forward saved (class map,
// per-pixel loss, the two radix-select thresholds: the arithmetic of k_bce_bwd in loss_optim.hip, operation for operation) instead of being
// written by k_bce_bwd (4 B per pixel) and read back here (4 B) -- 33 instead of 45 B per pixel over the two launches, one launch and one
// 134 MB buffer less per step.  Four pixels per thread and iteration (16-byte loads of pred / target / lpx, one dword of classes, four 16-byte z
// vectors, one 16-byte gl store).  Only the gl form (k_rs_bwd<..., HEAD> consumes gl); P % 4 == 0.
template <class T>
__global__ __launch_bounds__(256) void k_head_bwd_loss(const T* __restrict__ z, const float* __restrict__ tr, const float* __restrict__ w,
                                                       const float* __restrict__ pred, const float* __restrict__ target,
                                                       const float* __restrict__ lpx, const unsigned char* __restrict__ cls,
                                                       const LossState* __restrict__ stt, const float* __restrict__ gout,
                                                       double* __restrict__ acc64 /*[9]: dw[8] | db*/, const float* __restrict__ saved /*[2][8]*/,
                                                       double* __restrict__ gsum /*[2][8]*/, long P, float* __restrict__ gl_out /*[P]*/) {
    __shared__ float s_slots[4 * 25];
    float wv[8], sc[8], sh[8], lo[8], mu[8], acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, st1[8], st2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        wv[i] = w[i];
        sc[i] = tr[i];
        sh[i] = tr[8 + i];
        lo[i] = tr[16 + i];
        mu[i] = saved[i];
        st1[i] = st2[i] = 0.f;
    }
    const unsigned t0 = stt->prefix[0], t1 = stt->prefix[1];
    const float f0 = stt->frac[0], f1 = stt->frac[1];
    const float s = gout[0] * stt->inv2k;
    const long nq = P >> 2;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long)gridDim.x * 256) {
        const float4 p4 = reinterpret_cast<const float4*>(pred)[q], t4 = reinterpret_cast<const float4*>(target)[q];
        const float4 l4 = reinterpret_cast<const float4*>(lpx)[q];
        const unsigned c4 = reinterpret_cast<const unsigned*>(cls)[q];
        float v[4][8];
#pragma unroll
        for (int e = 0; e < 4; ++e) load8(z + (q * 4 + e) * 8, v[e]);
        const float pe[4] = {p4.x, p4.y, p4.z, p4.w}, te[4] = {t4.x, t4.y, t4.z, t4.w}, le[4] = {l4.x, l4.y, l4.z, l4.w};
        float gle[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned c = (c4 >> (8 * e)) & 0xffu;
            const float pr = pe[e];
            float gp = 0.f;  // k_bce_bwd's grad()
            if (c) {
                const unsigned key = loss_key(le[e]);
                const unsigned thr = c == 1 ? t0 : t1;
                const float wgt = key > thr ? 1.f : (key == thr ? (c == 1 ? f0 : f1) : 0.f);
                if (wgt != 0.f) {
                    const float t = fminf(fmaxf(te[e], 0.f), 1.f);
                    gp = s * wgt * (pr - t) / fmaxf((1.f - pr) * pr, 1e-12f);
                }
            }
            const float gl = gp * (1.f - pr) * pr;
            gle[e] = gl;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float pre = fmaf(v[e][i], sc[i], sh[i]);
                const float xv = fmaxf(pre, lo[i]);
                acc[i] = fmaf(gl, xv, acc[i]);
                const float gh = pre > 0.f ? Elem<T>::round(gl * wv[i]) : 0.f;  // what the block's backward forms from gl
                st1[i] += gh;
                st2[i] = fmaf(gh, v[e][i] - mu[i], st2[i]);
            }
            acc[8] += gl;
        }
        reinterpret_cast<float4*>(gl_out)[q] = make_float4(gle[0], gle[1], gle[2], gle[3]);
    }
    float all[25];
#pragma unroll
    for (int i = 0; i < 9; ++i) all[i] = acc[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        all[9 + i] = st1[i];
        all[17 + i] = st2[i];
    }
    const float tot = block_sum_det<25>(all, s_slots);
    if (threadIdx.x < 9) atomicAdd(&acc64[threadIdx.x], (double)tot);
    if (threadIdx.x >= 9 && threadIdx.x < 25) {
        const int i = threadIdx.x - 9;
        atomicAdd(&gsum[i], (double)(i < 8 ? tot : tot * saved[8 + (i - 8)]));
    }
}

// ----------------------------------------------------------------------------------------------
// Deferred second stage of the block backward's flushes (round 5; see BwdLast in det_common.h).  Between ocrs_bwd_defer_begin and
// ocrs_bwd_defer_flush the single-writer reductions of weight-gradient partials (k_mm_bwd_reduce's weight part, k_wgrad_partials_reduce,
// k_dw_partials_reduce's tap rows) are not launched but queued; the flush runs them all as ONE launch (same column-sum order: bit-identical
// gradients).  Per-device state (one backward at a time per device; `defer_state()` below).
// ----------------------------------------------------------------------------------------------
struct RedJob {
    const float* ws;
    float *d0, *d1;
    int nb, nelem, n0, cin0, ldw0, n1, blk0;
};
constexpr int RED_MAXJOBS = 48;
struct RedJobs {
    int njobs;
    RedJob j[RED_MAXJOBS];
};
__global__ __launch_bounds__(256) void k_reduce_multi(RedJobs J) {
    int k = 0;
    while (k + 1 < J.njobs && (int)blockIdx.x >= J.j[k + 1].blk0) ++k;
    const RedJob& q = J.j[k];
    const long e = (long)((int)blockIdx.x - q.blk0) * 32 + (threadIdx.x & 31);
    float s;
    if (!det_column_sum(q.ws, q.nb, q.nelem, e < q.n0 + q.n1 ? e : (long)q.nelem, s)) return;
    if (e < q.n0)
        q.d0[(e / q.cin0) * q.ldw0 + e % q.cin0] += s;
    else
        q.d1[e - q.n0] += s;
}
// One state per DEVICE: a backward of a second model on another GPU of the same process (autograd runs one worker thread per device) has
// its own queue, scratch and partials workspace -- indexed by the calling thread's current device at every entry.
struct DeferState {
    bool on = false;
    double* scratch = nullptr;
    long cap = 0, used = 0;
    int nblocks = 0;
    RedJobs jobs;
    // workspace for per-block partials of launches whose entry points take none (the CRNN's bias / first-layer sums: rec_conv.hip, rec_conv0.hip,
    // rec_gru_seq.hip): a per-device buffer (allocated on that device at first use), bump-allocated between _begin and _flush; null outside that
    // window, when it is used up, or when the allocation is refused (e.g. inside a stream capture: callers then take their atomic path)
    float* ws = nullptr;
    long ws_used = 0;
};
constexpr int DEFER_MAXDEV = 16;
static DeferState g_defer_dev[DEFER_MAXDEV];
static DeferState& defer_state() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= DEFER_MAXDEV) d = 0;
    return g_defer_dev[d];
}
constexpr long DEFER_WS_FLOATS = 8L << 20;  // 32 MB
float* bwd_defer_ws(long nfloats) {
    DeferState& D = defer_state();
    if (!D.on || nfloats <= 0) return nullptr;
    if (!D.ws && hipMalloc(reinterpret_cast<void**>(&D.ws), DEFER_WS_FLOATS * sizeof(float)) != hipSuccess) {
        D.ws = nullptr;
        (void)hipGetLastError();
        return nullptr;
    }
    const long n = (nfloats + 63) & ~63L;
    if (D.ws_used + n > DEFER_WS_FLOATS) return nullptr;
    float* p = D.ws + D.ws_used;
    D.ws_used += n;
    return p;
}
double* bwd_defer_scratch(int ndoubles) {
    DeferState& D = defer_state();
    if (!D.on || D.used + ndoubles > D.cap) return nullptr;
    double* p = D.scratch + D.used;
    D.used += (ndoubles + 1) & ~1L;
    return p;
}
bool bwd_defer_reduce(const float* ws, int nb, int nelem, float* d0, int n0, int cin0, int ldw0, float* d1, int n1) {
    DeferState& D = defer_state();
    if (!D.on || D.jobs.njobs >= RED_MAXJOBS || n0 + n1 <= 0) return false;
    RedJob& q = D.jobs.j[D.jobs.njobs++];
    q = RedJob{ws, d0, d1, nb, nelem, n0, cin0 > 0 ? cin0 : 1, ldw0, n1, D.nblocks};
    D.nblocks += (n0 + n1 + 31) / 32;
    return true;
}
// queue the reduction (deferral window open) or run it now as a one-job launch of the same kernel (same column-sum order: bit-identical gradients)
void bwd_reduce_or_defer(const float* ws, int nb, int nelem, float* d0, int n0, int cin0, int ldw0, float* d1, int n1, hipStream_t st) {
    if (n0 + n1 <= 0 || bwd_defer_reduce(ws, nb, nelem, d0, n0, cin0, ldw0, d1, n1)) return;
    RedJobs J;
    J.njobs = 1;
    J.j[0] = RedJob{ws, d0, d1, nb, nelem, n0, cin0 > 0 ? cin0 : 1, ldw0, n1, 0};
    hipLaunchKernelGGL(k_reduce_multi, dim3((n0 + n1 + 31) / 32), dim3(256), 0, st, J);
}

// ----------------------------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------------------------
static inline int ew_grid(long items) {
    long g = (items + 255) / 256;
    const long cap = (long)kNumCU * 8;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}
// grid for kernels whose threads keep a fixed channel group: total threads must be a multiple of the group count (<= 64, divides 256)
static inline int cg_grid(long items) { return ew_grid(items); }

// wgrad-carrying persistent grids: each block flushes a weight-gradient tile with atomics, so make every block
// chew through >= 8 pixel tiles when there are enough of them.
static inline int wgrad_grid(long ntiles, int cap_blocks) {
    static const int cap_env = env_int("OCRS_WGRAD_CAP", 0);
    if (cap_env > 0) cap_blocks = cap_env;
    static const int tpb = env_int("OCRS_WGRAD_TPB", 4);  // minimum tiles per block (each block flushes one full weight-gradient partial)
    long g = ntiles / tpb;
    if (g < 1) g = 1;
    if (g > cap_blocks) g = cap_blocks;
    if (g >= 8) g &= ~7L;
    return (int)g;
}

extern "C" int ocrs_wgrad_gather(const void* A, int ldA, int CA, const float* trA, const void* B, int ldB, int CB, float* dW, float* ws, int N, int hA,
                                 int wA, int HB, int WB, int stride, int padh, int padw, int KH, int KW, int dtype, hipStream_t st);

extern "C" {

// Sum of ghat and ghat*zhat over all pixels (BatchNorm2d backward reductions).  gsum [2][C] double, ACCUMULATED (caller zeroes).
int ocrs_bn_bwd_reduce(const void* g1, const void* g2, int pooled, const void* z, const float* bn, const float* saved, double* gsum, int C,
                       int N, int H, int W, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(g1 && z && bn && saved && gsum && C % 8 == 0 && C <= 256);
    const long P = (long)N * H * W;
    // every block ends with 2C fp64 atomics onto the same addresses: at the deep levels (tens of thousands of pixels) give each thread
    // at least 8 items instead of launching 2048 nearly idle blocks (those launches were 60 us of pure flush)
    long gl = (P * (C / 8) + 256 * 8 - 1) / (256 * 8);
    static const int bpc = env_int("OCRS_BNR_BPC", 2);  // blocks per CU: every block ends in 2 C same-address fp64 atomics (~11 ns each, serial per address: 73 -> 54 us at level 1)
    const int grid = (int)(gl < 8 ? 8 : (gl > kNumCU * bpc ? kNumCU * bpc : gl));
    const size_t smem = (2 * C + 256 * 16) * sizeof(float);
    if (dtype == 1) {
        GradSrc<bf16> gs{(const bf16*)g1, (const bf16*)g2, pooled};
        OCRS_LAUNCH_T(k_bn_bwd_reduce<bf16>, dim3(grid), dim3(256), smem, st, gs, (const bf16*)z, bn, saved, gsum, C, H, W, P);
    } else {
        GradSrc<float> gs{(const float*)g1, (const float*)g2, pooled};
        OCRS_LAUNCH_T(k_bn_bwd_reduce<float>, dim3(grid), dim3(256), smem, st, gs, (const float*)z, bn, saved, gsum, C, H, W, P);
    }
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

int ocrs_bn_bwd_finalize(const double* gsum, long count, int C, const float* gamma, const float* saved, float* coef, float* dgamma,
                         float* dbeta, hipStream_t st) {
    OCRS_CHECK_ARG(gsum && gamma && saved && coef && dgamma && dbeta && C > 0 && count > 0);
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((C + 63) / 64), dim3(64), 0, st, gsum, count, C, gamma, saved, coef, dgamma, dbeta);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

}  // extern "C" (templates need C++ linkage)
static inline int wgrad_grid(long ntiles, int cap_blocks);
template <int CIN, int COUT>
static int pw_bwd_gx(int N, int H, int W) {
    using Cfg = PwBwdCfg<CIN, COUT>;
    const Tiling2 tg = make_tiling2(N, H, W, Cfg::TW, Cfg::TH);
    // one-chunk configs (fp32 parity mode at levels 0-2): as many blocks as are resident; deeper levels: the tiles-per-block rule of
    // wgrad_grid (they are latency-bound: more, shorter blocks)
    return wgrad_grid(tg.ntiles, (Cfg::NKD == 1 && CIN <= 32) ? 3 * kNumCU : 2048);
}
template <class T, int CIN, int COUT>
static int launch_pw_bwd(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* g1, const void* g2,
                         int pooled, const void* z, const float* bn, const float* coef, const void* wpk_d, void* du, float* dwpw, float* ws, int N,
                         int H, int W, hipStream_t st) {
    using Cfg = PwBwdCfg<CIN, COUT>;
    constexpr int TPP = Elem<T>::is_bf16 ? Cfg::TPP_BF : Cfg::TPP_F;
    const size_t smem = (((Cfg::TP * Mma<T>::LDS_PITCH + Cfg::mid_el(TPP)) * sizeof(T) + 15) & ~15) +
                        (Cfg::HP * Cfg::CGI * 8 + 12 * CIN + 6 * COUT) * sizeof(float);
    static DevOnce attr_set;
    if (attr_set.need()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pw_bwd<T, CIN, COUT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
            hipSuccess)
            return OCRS_ERR_HIP;
        attr_set.done();
    }
    Src2<T> x{(const T*)xa, (const T*)xb, Ca, Cb};
    GradSrc<T> gs{(const T*)g1, (const T*)g2, pooled};
    const Tiling2 tg = make_tiling2(N, H, W, Cfg::TW, Cfg::TH);
    const int gx = pw_bwd_gx<CIN, COUT>(N, H, W);
    OCRS_LAUNCH_T((k_pw_bwd<T, CIN, COUT>), dim3(gx, Cfg::NBI * Cfg::NBO), dim3(256), smem, st, x, tra, trb, wdw, gs, (const T*)z, bn, coef,
                       wpk_d, (T*)du, dwpw, ws, tg);
    if (ws) {
        const int ne = CIN * COUT;
        OCRS_LAUNCH_T(k_wgrad_partials_reduce, dim3((ne + 31) / 32), dim3(256), 0, st, ws, gx, ne, dwpw, CIN, CIN);
    }
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}
extern "C" {

#ifdef OCRS_PW_ONLY_16  // (compile-time experiments: tools/kres.py with KRES_FLAGS=-DOCRS_PW_ONLY_16)
#define PW_BWD_COMBOS(X) X(16, 16)
#else
#define PW_BWD_COMBOS(X) \
    X(8, 8) X(8, 16) X(16, 16) X(16, 32) X(32, 32) X(32, 64) X(64, 64) X(64, 128) X(128, 128) X(128, 256) X(256, 256) X(256, 128) X(128, 64) \
        X(64, 32) X(32, 16) X(16, 8)
#endif

// Pointwise-conv backward of a DepthwiseConv block: du = Wpw^T dz (written, [P][Cin]); dwpw += u^T dz (accumulated, master layout
// [Cout][Cin]); dz is formed on the fly from (g1 [+g2], z, bn, coef), u is recomputed from the block input.
// wpk_d = ocrs_pack_frags(mode 0, K=Cout, M=Cin) of W^T.
void k_wgrad_partials_reduce_launch(const float* ws, int nb, int nelem, float* dw, int cin, int ldw, hipStream_t st) {
    if (bwd_defer_reduce(ws, nb, nelem, dw, nelem, cin, ldw, nullptr, 0)) return;  // (ocrs_bwd_defer_begin .. _flush: one launch for all of them)
    OCRS_LAUNCH_T(k_wgrad_partials_reduce, dim3((nelem + 31) / 32), dim3(256), 0, st, ws, nb, nelem, dw, cin, ldw);
}
// det_pw2.hip: two-pixel-per-thread pipelined kernel for bf16, Cin, Cout <= 32 (levels 0-2)
long det_pw2_supported(int Cin, int Cout, int dtype);
long det_pw2_ws_floats(int Cin, int Cout, int N, int H, int W);
int det_pw2_launch(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* g1, const void* g2,
                   int pooled, const void* z, const float* bn, const float* coef, const void* wpk_d, void* du, float* dwpw, float* ws, int Cout, int N,
                   int H, int W, int ldu, int ldw, hipStream_t st);
// det_pw8.hip: eight-wave kernel for the deep levels (bf16, Cin, Cout in {64, 128, 256}); same tiling and workspace rule as k_pw_bwd
long det_pw8_supported(int Cin, int Cout, int dtype);
long det_pwb_supported(int Cin, int Cout, int dtype);  // det_pwb.hip
int det_pwb_gx(int Cin, int Cout, int N, int H, int W, int pooled);
int det_pwb_launch(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* g1, const void* g2,
                   int pooled, const void* z, const float* bn, const float* coef, const void* wpk_d, void* du, float* dwpw, float* ws, int Cout, int N,
                   int H, int W, const BnFin* fin, hipStream_t st);
int det_pw8_launch(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* g1, const void* g2,
                   int pooled, const void* z, const float* bn, const float* coef, const void* wpk_d, void* du, float* dwpw, float* ws, int Cout, int N,
                   int H, int W, int gx, hipStream_t st);
// a 64-channel concat input (32 | 32) is handled as two k_pw_bwd2<32, Cout> launches, one per source (see ocrs_pw_bwd)
static bool pw2_split_ok(int Ca, int Cb, int Cout, int dtype) { return Ca == 32 && Cb == 32 && det_pw2_supported(32, Cout, dtype); }

// ws: workspace of ocrs_pw_bwd_ws_floats() floats (deterministic two-stage weight-gradient reduction) or null (float atomics).
long ocrs_pw_bwd_ws_floats(int Cin, int Cout, int N, int H, int W) {
    if (det_pw2_supported(Cin, Cout, 1)) {  // (dtype is not known here: cover both kernels)
        const long a = det_pw2_ws_floats(Cin, Cout, N, H, W);
#define X(CI, CO) \
    if (Cin == CI && Cout == CO) { const long b = (long)pw_bwd_gx<CI, CO>(N, H, W) * CI * CO; return a > b ? a : b; }
        PW_BWD_COMBOS(X)
#undef X
        return a;
    }
    long half = (Cin == 64 && det_pw2_supported(32, Cout, 1)) ? det_pw2_ws_floats(32, Cout, N, H, W) : 0;  // two-launch split (32 | 32)
    if (det_pwb_supported(Cin, Cout, 1)) {
        const long c = (long)det_pwb_gx(Cin, Cout, N, H, W, 0) * Cin * Cout;  // (the not-pooled grid is the larger one)
        half = c > half ? c : half;
    }
#define X(CI, CO) \
    if (Cin == CI && Cout == CO) { const long b = (long)pw_bwd_gx<CI, CO>(N, H, W) * CI * CO; return b > half ? b : half; }
    PW_BWD_COMBOS(X)
#undef X
    return half;
}
int ocrs_pw_bwd(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* g1, const void* g2, int pooled,
                const void* z, const float* bn, const float* coef, const void* wpk_d, void* du, float* dwpw, float* ws, int Cout, int N, int H, int W,
                int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(xa && tra && wdw && g1 && z && bn && coef && wpk_d && du && dwpw && (Cb == 0 || trb));
    OCRS_CHECK_ARG((Cb == 0) == (xb == nullptr));
    const int Cin = Ca + Cb;
    OCRS_CHECK_ARG((long)N * H * W < (1L << 31));
    static const int use_pw2 = env_int("OCRS_PW2", 1);
    if (use_pw2 && det_pw2_supported(Cin, Cout, dtype))
        return det_pw2_launch(xa, xb, Ca, Cb, tra, trb, wdw, g1, g2, pooled, z, bn, coef, wpk_d, du, dwpw, ws, Cout, N, H, W, Cin, Cin, st);
    static const int use_split = env_int("OCRS_PW2_SPLIT", 1);
    if (use_pw2 && use_split && pw2_split_ok(Ca, Cb, Cout, dtype)) {
        // cat(32 | 32) -> Cout: z = Wpw[:, :32] u_a + Wpw[:, 32:] u_b, so du and dWpw separate by source; dz is the same for both.  Two launches of the
        // tuned two-pixel kernel (each re-reads g and z: 512 instead of 384 B per pixel) beat the generic 64-channel path (392 -> ~270 us at
        // level 2).  Per half: depthwise weights [32][9], packed W^T fragments (M tiles 2h, 2h+1), du / dWpw columns 32h.., row strides 64.
        const char* wp = static_cast<const char*>(wpk_d);
        const size_t half_frag_bytes = 2 * 64 * 8 * sizeof(bf16);  // two M tiles of one K chunk
        OCRS_CHECK_ARG(Cout <= 32);
        int rc = det_pw2_launch(xa, nullptr, 32, 0, tra, nullptr, wdw, g1, g2, pooled, z, bn, coef, wp, du, dwpw, ws, Cout, N, H, W, 64, 64, st);
        if (rc != OCRS_OK) return rc;
        return det_pw2_launch(xb, nullptr, 32, 0, trb, nullptr, wdw + 32 * 9, g1, g2, pooled, z, bn, coef, wp + half_frag_bytes,
                              static_cast<bf16*>(du) + 32, dwpw + 32, ws, Cout, N, H, W, 64, 64, st);
    }
    if (det_pwb_supported(Cin, Cout, dtype))  // deep levels, up to 64 input channels: a whole tile in three barriers (det_pwb.hip)
        return det_pwb_launch(xa, xb, Ca, Cb, tra, trb, wdw, g1, g2, pooled, z, bn, coef, wpk_d, du, dwpw, ws, Cout, N, H, W, nullptr, st);
    if (det_pw8_supported(Cin, Cout, dtype)) {
#define X(CI, CO)                 \
    if (Cin == CI && Cout == CO)  \
        return det_pw8_launch(xa, xb, Ca, Cb, tra, trb, wdw, g1, g2, pooled, z, bn, coef, wpk_d, du, dwpw, ws, Cout, N, H, W, pw_bwd_gx<CI, CO>(N, H, W), st);
        PW_BWD_COMBOS(X)
#undef X
    }
#define X(CI, CO)                                                                                                                         \
    if (Cin == CI && Cout == CO)                                                                                                          \
        return dtype == 1 ? launch_pw_bwd<bf16, CI, CO>(xa, xb, Ca, Cb, tra, trb, wdw, g1, g2, pooled, z, bn, coef, wpk_d, du, dwpw, ws, N, H, W, st) \
                          : launch_pw_bwd<float, CI, CO>(xa, xb, Ca, Cb, tra, trb, wdw, g1, g2, pooled, z, bn, coef, wpk_d, du, dwpw, ws, N, H, W, st);
    PW_BWD_COMBOS(X)
#undef X
    return OCRS_ERR_ARG;
}

// ocrs_bn_bwd_finalize + ocrs_pw_bwd in one call: gsum [2][Cout] (fp64, complete), gamma, saved [mean | rstd] of this block; dgamma / dbeta written;
// coef [3][Cout]: scratch (the deep-level bf16 kernel derives the coefficients in its prologue and leaves it untouched, every other path runs
// the finalize kernel into it first).
int ocrs_pw_bwd_fin(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* g1, const void* g2,
                    int pooled, const void* z, const float* bn, float* coef, const double* gsum, const float* gamma, const float* saved, float* dgamma,
                    float* dbeta, const void* wpk_d, void* du, float* dwpw, float* ws, int Cout, int N, int H, int W, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(coef && gsum && gamma && saved && dgamma && dbeta && xa && tra && wdw && g1 && z && bn && wpk_d && du && dwpw);
    const int Cin = Ca + Cb;
    if (det_pwb_supported(Cin, Cout, dtype) && ws) {
        const BnFin fin{gsum, gamma, saved, dgamma, dbeta, (long)N * H * W};
        return det_pwb_launch(xa, xb, Ca, Cb, tra, trb, wdw, g1, g2, pooled, z, bn, nullptr, wpk_d, du, dwpw, ws, Cout, N, H, W, &fin, st);
    }
    const int rc = ocrs_bn_bwd_finalize(gsum, (long)N * H * W, Cout, gamma, saved, coef, dgamma, dbeta, st);
    if (rc != OCRS_OK) return rc;
    return ocrs_pw_bwd(xa, xb, Ca, Cb, tra, trb, wdw, g1, g2, pooled, z, bn, coef, wpk_d, du, dwpw, ws, Cout, N, H, W, dtype, st);
}

// Depthwise-conv backward: gxa/gxb (either may be null) receive dL/dx~ split at channel Ca; dwdw accumulated in master layout [C][1][3][3].
static void dw_bwd_grid(int C, int N, int H, int W, int& gx, int& gy, int& cg) {
    static const int cg_max = env_int("OCRS_DW_CG", 4);  // channel groups (of 8) per block: 4 -> 8x4-pixel tiles, 2 -> 8x8, 1 -> 8x16
    cg = C / 8 < cg_max ? C / 8 : cg_max;
    gy = C / (cg * 8);
    const Tiling2 tg = make_tiling2(N, H, W, 32 / cg, 8);
    // 3 blocks per CU are resident (k_dw_bwd's launch bounds): a grid of exactly that many blocks (all y-slabs together) has no partial last
    // round -- with 8 per CU the 2048 blocks ran as 2.67 rounds of 768
    static const int bpc = env_int("OCRS_DW_BPC", 3);
    gx = persistent_grid(tg.ntiles, bpc / gy > 0 ? bpc / gy : 1);
}
// ws: ocrs_dw_bwd_ws_floats() floats (per-block partials of dwdw, summed by a second kernel) or null (float atomics).
long ocrs_dw_bwd_ws_floats(int C, int N, int H, int W) {
    int gx, gy, cg;
    dw_bwd_grid(C, N, H, W, gx, gy, cg);
    return (long)gx * C * 11;
}
// gsum_a / gsum_b (nullable; need ws): the producer blocks of source a / b get their BatchNorm-backward sums
// [sum ghat | sum ghat*zhat] ([2][Ca] / [2][Cb] fp64, ACCUMULATED: zero them before the first consumer) from this pass instead of
// ocrs_bn_bwd_reduce; saved_a / saved_b = those producers' saved [mean | rstd].  Valid when the source is that block's raw z
// with its BatchNorm+ReLU load transform (not for pooled / ConvTranspose outputs).
int ocrs_dw_bwd(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* du, void* gxa, void* gxb,
                float* dwdw, float* ws, const float* saved_a, double* gsum_a, const float* saved_b, double* gsum_b, int N, int H, int W, int dtype,
                hipStream_t st) {
    const int stat_mask = (gsum_a ? 1 : 0) | (gsum_b ? 2 : 0);
    OCRS_CHECK_ARG(!stat_mask || ws);
    OCRS_CHECK_ARG((!gsum_a || saved_a) && (!gsum_b || (saved_b && Cb > 0)));
    OCRS_CHECK_ARG(xa && tra && wdw && du && dwdw && (Ca + Cb) % 8 == 0 && Ca % 4 == 0 && (Cb == 0 || trb));
    OCRS_CHECK_ARG((Cb == 0) == (xb == nullptr));
    const int C = Ca + Cb;
    OCRS_CHECK_ARG((long)N * H * W < (1L << 31) && (C < 32 || C % 32 == 0) && Ca % 8 == 0);
    int gx, gy, cg;
    dw_bwd_grid(C, N, H, W, gx, gy, cg);
#define DWB(T_, CG_)                                                                                                                      \
    {                                                                                                                                     \
        const Tiling2 tg = make_tiling2(N, H, W, 32 / CG_, 8);                                                                            \
        const int HP = (32 / CG_ + 2) * 10;                                                                                               \
        const size_t tile_fl = HP * CG_ * 12 + 13 * CG_ * 8, red_fl = 44 * 256;                                                           \
        const size_t smem = (tile_fl > red_fl ? tile_fl : red_fl) * sizeof(float);                                                        \
        Src2<T_> x{(const T_*)xa, (const T_*)xb, Ca, Cb};                                                                                 \
        if (stat_mask)                                                                                                                    \
            OCRS_LAUNCH_T((k_dw_bwd<T_, CG_, true>), dim3(gx, gy), dim3(256), smem, st, x, tra, trb, wdw, (const T_*)du, (T_*)gxa, (T_*)gxb, \
                               dwdw, ws, saved_a, saved_b, stat_mask, tg, bl);                                                            \
        else                                                                                                                              \
            OCRS_LAUNCH_T((k_dw_bwd<T_, CG_, false>), dim3(gx, gy), dim3(256), smem, st, x, tra, trb, wdw, (const T_*)du, (T_*)gxa, (T_*)gxb, \
                               dwdw, ws, saved_a, saved_b, 0, tg, bl);                                                                    \
    }
    BwdLast bl{nullptr, nullptr, gsum_a, gsum_b, saved_a, saved_b, Ca, 1};
    // k_dw_bwd (the deep levels: 768 short workgroups per launch): the drain + ticket at the end of every workgroup costs what the ~5 us reduce launch
    // it takes off the chain saves -- A/B on one box, whole step: 12.19 / 12.18 ms with it, 12.14 / 12.10 ms without (and 12.23 / 12.20 ms with nothing
    // deferred) -- so it is off by default here; k_mm_bwd / k_rs_bwd (levels 0-2) keep it
    static const int last_on = env_int("OCRS_BWD_LAST", 1) && env_int("OCRS_BWD_LAST_DW", 0);
    if (stat_mask && last_on) {
        if (double* p = bwd_defer_scratch(BWD_LAST_SLOTS * 2 * C + 2)) {
            bl.raw = p;
            bl.counter = reinterpret_cast<unsigned*>(p + BWD_LAST_SLOTS * 2 * C);
        }
    }
    if (dtype == 1) {
        if (cg == 1) DWB(bf16, 1) else if (cg == 2) DWB(bf16, 2) else DWB(bf16, 4)
    } else {
        if (cg == 1) DWB(float, 1) else if (cg == 2) DWB(float, 2) else DWB(float, 4)
    }
#undef DWB
    if (ws) {
        const int nrow = stat_mask ? 11 : 9;
        // deferred second stage: with the sums done in the block kernel (or none asked for) only the tap rows are left, and nothing in the backward reads them
        static const int dwq = env_int("OCRS_BWD_DW_QUEUE", 1);
        if (dwq && (bl.raw || !stat_mask) && bwd_defer_reduce(ws, gx, C * nrow, nullptr, 0, 1, 1, dwdw, 9 * C)) {
        } else if (bl.raw) {
            OCRS_LAUNCH_T(k_dw_partials_reduce, dim3((C * 9 + 31) / 32), dim3(256), 0, st, ws, gx, C, Ca, nrow, dwdw, (double*)nullptr, (double*)nullptr, saved_a, saved_b);
        } else {
            OCRS_LAUNCH_T(k_dw_partials_reduce, dim3((C * nrow + 31) / 32), dim3(256), 0, st, ws, gx, C, Ca, nrow, dwdw, gsum_a,
                               gsum_b, saved_a, saved_b);
        }
    }
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

extern "C" long det_c1v2_supported(int N, int H, int W);  // det_c1.hip
extern "C" int det_c1v2_bwd_launch(const float* img, const float* wdw, const float* wpw, const void* g, const void* z, const float* bn, const float* coef,
                                   double* acc64, int N, int H, int W, int dtype, hipStream_t st);
// First block (1->8) backward.  acc64 [17] fp64 = dWpw [8] | dWdw [9], ACCUMULATED (caller-zeroed; the caller adds it to the fp32 gradients: fp64
// accumulation of the per-block fp32 partials is exact, hence order-independent -- float atomics were not).  The input image gets no gradient.
int ocrs_dwpw_c1_bwd(const float* img, const float* wdw, const float* wpw, const void* g1, const void* g2, int pooled, const void* z,
                     const float* bn, const float* coef, double* acc64, int N, int H, int W, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(img && wdw && wpw && g1 && bn && coef && acc64);
    // (z may be null on the det_c1.hip path: that kernel rebuilds z = round(wpw[c] * u) from the depthwise output it recomputes anyway)
    if (!g2 && !pooled && det_c1v2_supported(N, H, W)) return det_c1v2_bwd_launch(img, wdw, wpw, g1, z, bn, coef, acc64, N, H, W, dtype, st);  // det_c1.hip
    OCRS_CHECK_ARG(z);
    const long P = (long)N * H * W;
    int grid = ew_grid(P);
    const int resident = (dtype == 1 ? 3 : 2) * kNumCU;  // persistent grid-stride kernel: exactly the resident blocks (168 / 217 VGPRs)
    if (grid > resident) grid = resident;
    if (dtype == 1) {
        GradSrc<bf16> gs{(const bf16*)g1, (const bf16*)g2, pooled};
        hipLaunchKernelGGL(k_c1_bwd<bf16>, dim3(grid), dim3(256), 0, st, img, wdw, wpw, gs, (const bf16*)z, bn, coef, acc64, H, W, P);
    } else {
        GradSrc<float> gs{(const float*)g1, (const float*)g2, pooled};
        hipLaunchKernelGGL(k_c1_bwd<float>, dim3(grid), dim3(256), 0, st, img, wdw, wpw, gs, (const float*)z, bn, coef, acc64, H, W, P);
    }
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// ConvTranspose2d backward.  g [N][H][W][Cout] = gradient of the (cropped) output; dx [N][h][w][Cup] written;
// dW [Cup][Cout][3][3] and dbias [Cout] accumulated; ws = workspace of ocrs_convt_bwd_ws_floats() floats (or null: float atomics).  wpk_d = ocrs_pack_frags(mode 0, K=9*Cout, M=Cup, K2=Cout, s1=1, s2=9, sm=9*Cout).
extern "C" long ocrs_wgrad_gather_ws_floats(int CA, int CB, int ntaps, long P, int dtype);
static bool convt_wgrad_tr_ok(int Cup, int Cout, int dtype) {
    return dtype == 1 && ((Cup == 16 && Cout == 8) || (Cup == 32 && Cout == 16) || (Cup == 32 && Cout == 32));
}
static int convt_wgrad_tr_grid(int Cout, int N, int h, int w) {
    const int TH = Cout >= 32 ? 4 : 8;
    // resident blocks only (register-bound: 3 per CU for Cout = 8, 2 for Cout >= 16): every block ends with a full weight-gradient partial
    // (5-37 KB) that the reduce kernel reads back
    return persistent_grid((long)N * ((w + 15) / 16) * ((h + TH - 1) / TH), Cout >= 16 ? 2 : 3);
}
// det_rs32.hip: the fp32 weight / bias gradient of the wide levels as row-streaming waves (round 6)
long det_rs32_ctw_supported(int Cup, int Cout, int dtype);
long det_rs32_ctw_ws_floats(int Cup, int Cout, int N, int h, int w, int dtype);
int det_rs32_ctw_launch(const float* x, const float* tr, const float* g, float* dW, float* dbias, float* ws, int Cup, int Cout, int N, int h, int w, int H,
                        int W, hipStream_t st);
long ocrs_convt_bwd_ws_floats(int Cup, int Cout, int N, int h, int w, int dtype) {
    const long a = ocrs_wgrad_gather_ws_floats(Cup, Cout, 9, (long)N * h * w, dtype);
    const long b = convt_wgrad_tr_ok(Cup, Cout, dtype) ? (long)convt_wgrad_tr_grid(Cout, N, h, w) * (Cup * ((9 * Cout + 15) / 16 * 16) + Cout + 2 * Cup) : 0;
    const long c = det_rs32_ctw_ws_floats(Cup, Cout, N, h, w, dtype);
    return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

// 1 if ocrs_convt_bwd can also produce the BatchNorm-backward sums of the block that produced x (saved / gsum arguments)
// (not for (32, 16): with the sums that instantiation needs 208 + 80 registers -> one block per CU, 122 -> 225 us; its block keeps the
// 86-us ocrs_bn_bwd_reduce pass)
long det_ctd_supported(int Cup, int Cout, int dtype);  // det_ctd.hip
long ocrs_convt_bwd_stats_supported(int Cup, int Cout, int dtype) {
    static const int ctd_stats = env_int("OCRS_CTD_STATS", 1);  // the deep-level input-gradient kernel also produces the sums (round 5)
    if (ctd_stats && det_ctd_supported(Cup, Cout, dtype)) return 1;
    return convt_wgrad_tr_ok(Cup, Cout, dtype) && !(Cup == 32 && Cout == 16) ? 1 : 0;
}

// saved / gsum (nullable, need ws and ocrs_convt_bwd_stats_supported): x is the raw output of a block consumed ONLY by this ConvTranspose;
// its BatchNorm-backward sums [sum ghat | sum ghat*zhat] ([2][Cup] fp64, ACCUMULATED) come from this pass instead of ocrs_bn_bwd_reduce.
int ocrs_convt_bwd_parts(const void* x, const float* tr, const void* g, const void* wpk_d, void* dx, float* dW, float* dbias, double* dbias64, float* ws,
                         const float* saved, double* gsum, int Cup, int Cout, int N, int h, int w, int H, int W, int parts, int dtype, hipStream_t st);
int det_ctd_launch(const void* g, const void* wpk, void* dx, int Cup, int Cout, int N, int h, int w, int H, int W, hipStream_t st, const void* x,
                   const float* tr, const float* saved, double* gsum);
// 1 if ocrs_convt_bwd_parts can run the input gradient and the weight / bias gradients of this shape as separate calls (the generic deep-level path)
long ocrs_convt_bwd_splittable(int Cup, int Cout, int dtype) { return convt_wgrad_tr_ok(Cup, Cout, dtype) ? 0 : 1; }

int ocrs_convt_bwd(const void* x, const float* tr, const void* g, const void* wpk_d, void* dx, float* dW, float* dbias, double* dbias64, float* ws,
                   const float* saved, double* gsum, int Cup, int Cout, int N, int h, int w, int H, int W, int dtype, hipStream_t st) {
    return ocrs_convt_bwd_parts(x, tr, g, wpk_d, dx, dW, dbias, dbias64, ws, saved, gsum, Cup, Cout, N, h, w, H, W, 3, dtype, st);
}

// parts: bit 0 = input gradient dx, bit 1 = weight + bias gradients.  Separate calls only where ocrs_convt_bwd_splittable(): the weight-gradient
// half is off the backward's critical path (nothing downstream reads it), so the caller may put it on another stream, where it overlaps the
// latency-bound deep-level kernels that follow.  The tiled path (levels 0-2) computes everything from one staged tile and needs parts == 3.
int ocrs_convt_bwd_parts(const void* x, const float* tr, const void* g, const void* wpk_d, void* dx, float* dW, float* dbias, double* dbias64, float* ws,
                         const float* saved, double* gsum, int Cup, int Cout, int N, int h, int w, int H, int W, int parts, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(parts >= 1 && parts <= 3 && (parts == 3 || !convt_wgrad_tr_ok(Cup, Cout, dtype)));
    // dbias64 [Cout] fp64 (caller-zeroed): the generic (deep-level) path accumulates the bias gradient there (order-independent fp64 sums of
    // per-block fp32 partials) and the caller adds it to dbias; the tiled path (levels 0-2) adds to dbias itself from its single-writer reduce
    OCRS_CHECK_ARG(x && tr && g && wpk_d && dx && dW && dbias && dbias64 && Cup % 16 == 0 && Cout % 8 == 0 && Cup <= 256);
    OCRS_CHECK_ARG(!gsum || (saved && ws && ocrs_convt_bwd_stats_supported(Cup, Cout, dtype)));
    if (ws && convt_wgrad_tr_ok(Cup, Cout, dtype)) {
        // levels 0-2 (bf16): ONE tiled kernel produces dx, the weight-gradient partials and the bias-gradient partials; one reduce
#define CTW_CASE(CU_, CO_)                                                                                                                   \
    if (Cup == CU_ && Cout == CO_) {                                                                                                         \
        using CC = CtwCfg<CU_, CO_>;                                                                                                         \
        const Tiling2 tg = make_tiling2(N, h, w, CC::TW, CC::TH);                                                                            \
        const int nb = convt_wgrad_tr_grid(Cout, N, h, w);                                                                                   \
        if (gsum)                                                                                                                            \
            hipLaunchKernelGGL((k_convt_wgrad_tr<CU_, CO_, true>), dim3(nb), dim3(256), CC::SMEM, st, (const bf16*)x, tr, (const bf16*)g, wpk_d,   \
                               (bf16*)dx, ws, saved, h, w, H, W, tg);                                                                        \
        else                                                                                                                                 \
            hipLaunchKernelGGL((k_convt_wgrad_tr<CU_, CO_, false>), dim3(nb), dim3(256), CC::SMEM, st, (const bf16*)x, tr, (const bf16*)g, wpk_d,  \
                               (bf16*)dx, ws, saved, h, w, H, W, tg);                                                                        \
        const int ne = CU_ * 9 * CO_ + CO_ + (gsum ? 2 * CU_ : 0);                                                                           \
        hipLaunchKernelGGL(k_convt_wgrad_reduce, dim3((ne + 31) / 32), dim3(256), 0, st, ws, nb, CU_, CO_, CC::NT * 16, dW, \
                           dbias, saved, gsum);                                                                                              \
    }
        CTW_CASE(16, 8) CTW_CASE(32, 16) CTW_CASE(32, 32)
#undef CTW_CASE
        OCRS_LAUNCH_CHECK();
        return OCRS_OK;
    }
    const int MT_total = Cup / 16;
    const long P = (long)N * h * w;
    const long ntiles = (P + 63) / 64;
    const int gx = persistent_grid(ntiles, 8);
    if ((parts & 1) && det_ctd_supported(Cup, Cout, dtype)) {  // deep levels, bf16: the gradient region staged once (det_ctd.hip)
        const int rc = det_ctd_launch(g, wpk_d, dx, Cup, Cout, N, h, w, H, W, st, x, tr, saved, gsum);  // (gsum: + the producer block's BatchNorm-backward sums)
        if (rc != OCRS_OK) return rc;
    } else if (parts & 1) {
#define DG_CASE(T_, MT_)                                                                                                                     \
    hipLaunchKernelGGL((k_convt_dgrad<T_, MT_>), dim3(gx, MT_total / MT_), dim3(256), 0, st, (const T_*)g, wpk_d, (T_*)dx, Cup, Cout, h, w, H, \
                       W, N, MT_total);
#define DG_DISPATCH(T_)               \
    if (MT_total % 8 == 0) {          \
        DG_CASE(T_, 8)                \
    } else if (MT_total % 4 == 0) {   \
        DG_CASE(T_, 4)                \
    } else if (MT_total % 2 == 0) {   \
        DG_CASE(T_, 2)                \
    } else {                          \
        DG_CASE(T_, 1)                \
    }
    if (dtype == 1) {
        DG_DISPATCH(bf16)
    } else {
        DG_DISPATCH(float)
    }
#undef DG_DISPATCH
#undef DG_CASE
    OCRS_LAUNCH_CHECK();
    }
    if (!(parts & 2)) return OCRS_OK;
    if (ws && det_rs32_ctw_supported(Cup, Cout, dtype) && (long)N * H * W * Cout * 4 < (1L << 32) && (long)N * h * w * Cup * 4 < (1L << 32))
        // fp32, wide levels: weight AND bias gradient (into dbias; dbias64 stays as the caller zeroed it)
        return det_rs32_ctw_launch((const float*)x, tr, (const float*)g, dW, dbias, ws, Cup, Cout, N, h, w, H, W, st);
    {
        const int rc = ocrs_wgrad_gather(x, Cup, Cup, tr, g, Cout, Cout, dW, ws, N, h, w, H, W, 2, 0, 0, 3, 3, dtype, st);
        if (rc != OCRS_OK) return rc;
    }
    const long Pout = (long)N * H * W;
    // every block ends in Cout same-address global atomics (~15 ns each, serialised): 2048 blocks cost 30 us of atomics on a tensor that
    // streams in 10 us -> at most 2 blocks per CU, each thread keeping 4 loads in flight
    int gs = cg_grid((Pout * (Cout / 8) + 3) / 4);
    if (gs > 2 * kNumCU) gs = 2 * kNumCU;
    if (dtype == 1)
        hipLaunchKernelGGL(k_channel_sum<bf16>, dim3(gs), dim3(256), (Cout + 256 * 8) * sizeof(float), st, (const bf16*)g, dbias64, Cout, Pout);
    else
        hipLaunchKernelGGL(k_channel_sum<float>, dim3(gs), dim3(256), (Cout + 256 * 8) * sizeof(float), st, (const float*)g, dbias64, Cout, Pout);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// Head backward: gy [P][8] written (dtype T); acc64 [9] fp64 = dw [8] | db, ACCUMULATED (caller-zeroed, caller adds it to the fp32 gradients).  saved / gsum (nullable): also accumulate the BatchNorm-backward
// sums [2][8] (fp64, caller-zeroed) of the block that produced z (saved = its [mean | rstd]) -- replaces that block's ocrs_bn_bwd_reduce.
static int head_bwd_impl(const void* z, const float* tr, const float* w, const float* pred, const float* gpred, void* gy, float* gl, double* acc64,
                         const float* saved, double* gsum, long P, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(z && tr && w && pred && gpred && (gy || gl) && acc64 && P > 0 && (!gsum || saved));
    int grid = ew_grid(P);
    if (grid > 1024) grid = 1024;  // streaming kernel ending in same-address atomics: 4 blocks per CU are plenty
    if (dtype == 1)
        hipLaunchKernelGGL(k_head_bwd<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)z, tr, w, pred, gpred, (bf16*)gy, acc64, saved, gsum, P, gl);
    else
        hipLaunchKernelGGL(k_head_bwd<float>, dim3(grid), dim3(256), 0, st, (const float*)z, tr, w, pred, gpred, (float*)gy, acc64, saved, gsum, P, gl);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}
int ocrs_head_bwd(const void* z, const float* tr, const float* w, const float* pred, const float* gpred, void* gy, double* acc64,
                  const float* saved, double* gsum, long P, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(gy);
    return head_bwd_impl(z, tr, w, pred, gpred, gy, nullptr, acc64, saved, gsum, P, dtype, st);
}
// ocrs_head_bwd that writes gl [P] fp32 = dL/dlogit instead of the 8-channel gradient gy: for ocrs_mm_bwd_fin_head, which forms gy on the fly
int ocrs_head_bwd_gl(const void* z, const float* tr, const float* w, const float* pred, const float* gpred, float* gl, double* acc64,
                     const float* saved, double* gsum, long P, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(gl);
    return head_bwd_impl(z, tr, w, pred, gpred, nullptr, gl, acc64, saved, gsum, P, dtype, st);
}
// ocrs_balanced_bce_bwd + ocrs_head_bwd_gl in one pass: dL/dpred is formed on the fly from the loss forward's saved tensors (pred, target, lpx, cls,
// state: ocrs_balanced_bce_fwd) and gout [1] (the upstream gradient of the scalar loss) -- reference: the autograd of train_detection.py:225-263
// followed by models.py:143's 1x1 conv + sigmoid backward.  P % 4 == 0, saved / gsum required.
int ocrs_head_bwd_loss(const void* z, const float* tr, const float* w, const float* pred, const float* target, const float* lpx, const unsigned char* cls,
                       const void* state, const float* gout, float* gl, double* acc64, const float* saved, double* gsum, long P, int dtype,
                       hipStream_t st) {
    OCRS_CHECK_ARG(z && tr && w && pred && target && lpx && cls && state && gout && gl && acc64 && saved && gsum && P > 0 && P % 4 == 0);
    static const int bpc = env_int("OCRS_HEADL_BPC", 8);
    int grid = ew_grid(P / 4);
    if (grid > kNumCU * bpc) grid = kNumCU * bpc;
    if (dtype == 1)
        hipLaunchKernelGGL(k_head_bwd_loss<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)z, tr, w, pred, target, lpx, cls, (const LossState*)state,
                           gout, acc64, saved, gsum, P, gl);
    else
        hipLaunchKernelGGL(k_head_bwd_loss<float>, dim3(grid), dim3(256), 0, st, (const float*)z, tr, w, pred, target, lpx, cls,
                           (const LossState*)state, gout, acc64, saved, gsum, P, gl);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// Deferred second stage (see the queue above).  scratch: ndoubles zeroed fp64 values the block-backward launches carve their last-workgroup
// finalisation state from (16 Cin + 2 each; left zeroed), valid -- like every workspace `ws` passed meanwhile -- until ocrs_bwd_defer_flush, which
// launches the queued weight-gradient reductions as one kernel on `st` (the stream of the launches that produced the partials) and ends the mode.
int ocrs_bwd_defer_begin(double* scratch, long ndoubles) {
    DeferState& D = defer_state();
    OCRS_CHECK_ARG(!D.on && (scratch || ndoubles == 0) && ndoubles >= 0);
    D.on = true;
    D.scratch = scratch;
    D.cap = ndoubles;
    D.used = 0;
    D.nblocks = 0;
    D.jobs.njobs = 0;
    D.ws_used = 0;
    return OCRS_OK;
}
int ocrs_bwd_defer_flush(hipStream_t st) {
    DeferState& D = defer_state();
    const bool was = D.on;
    D.on = false;
    if (!was || D.jobs.njobs == 0) return OCRS_OK;
    hipLaunchKernelGGL(k_reduce_multi, dim3(D.nblocks), dim3(256), 0, st, D.jobs);
    D.jobs.njobs = 0;
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

}  // extern "C"
