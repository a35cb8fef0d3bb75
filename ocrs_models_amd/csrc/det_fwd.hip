// Detection U-Net forward kernels (gfx950).  Reference semantics: ocrs_models/models.py:7-143.
//
//  k_dwpw_fwd      fused [producer BN+ReLU on load] -> depthwise 3x3 -> pointwise 1x1 (MFMA) -> z + BN batch sums
//  k_dwpw_c1_fwd   the 1->8 first block (VALU)
//  k_bn_finalize   batch sums -> (scale, shift, lo) + saved mean/rstd + running-stat update
//  k_maxpool_fwd   2x2 max-pool of relu(bn(z))
//  k_convt_fwd     ConvTranspose2d k3 s2 (+bias, +crop) as a 4-pixel-gather MFMA GEMM
//  k_head_fwd      1x1 conv 8->1 + bias + sigmoid
#include "det_common.h"

// ----------------------------------------------------------------------------------------------
// generic weight-fragment packer:  W[k][m]  ->  MFMA fragments [kc][mt][lane][8]
//   mode 0: element (k, m) at src[(k / K2) * s1 + (k % K2) * s2 + m * sm]
//   mode 1: ConvTranspose2d forward "effective" weight (src = W[Cup][Cout][3][3], K2 = Cup, M = 4*Cout):
//           k = d*Cup + c (d = dy*2+dx), m = q*Cout + o (q = py*2+px) -> W[c][o][py+2dy][px+2dx] or 0
//   mode 2 (bf16 only): mode-0 addressing, SPLIT into hi = bf16(w), lo = bf16(w - hi): [kc][mt][plane = hi, lo][lane][8] (twice the bytes
//           of a plain bf16 pack) -- the A operand of the split-bf16 GEMM k_gemm_x3p (rec_gemm.hip)
// ----------------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ void pack_frags_body(const float* __restrict__ src, int mode, int K, int M, int K2, long s1, long s2, long sm, T* __restrict__ out,
                                                long idx) {
    const int MT = (M + 15) / 16;
    const int nkc = (K + 31) / 32;
    if (idx >= (long)nkc * MT * 64) return;
    const int lane = (int)(idx & 63);
    const long frag = idx >> 6;
    const int mt = (int)(frag % MT), kc = (int)(frag / MT);
    const int m = mt * 16 + (lane & 15);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = Elem<T>::is_bf16 ? kc * 32 + (lane >> 4) * 8 + j : kc * 32 + j * 4 + (lane >> 4);
        float x = 0.f;
        if (k < K && m < M) {
            if (mode == 0 || mode == 2) {
                x = src[(long)(k / K2) * s1 + (long)(k % K2) * s2 + (long)m * sm];
            } else {
                const int Cup = K2, Cout = M / 4;
                const int d = k / Cup, c = k % Cup, q = m / Cout, o = m % Cout;
                const int ky = (q >> 1) + 2 * (d >> 1), kx = (q & 1) + 2 * (d & 1);
                if (ky < 3 && kx < 3) x = src[((long)c * Cout + o) * 9 + ky * 3 + kx];
            }
        }
        v[j] = x;
    }
    if constexpr (Elem<T>::is_bf16) {
        if (mode == 2) {
            float lo[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float h = Elem<bf16>::round(v[j]);
                lo[j] = v[j] - h;
                v[j] = h;
            }
            store8(out + ((frag * 2) * 64 + lane) * 8, v);
            store8(out + ((frag * 2 + 1) * 64 + lane) * 8, lo);
            return;
        }
    }
    store8(out + idx * 8, v);
}
template <class T>
__global__ void k_pack_frags(const float* __restrict__ src, int mode, int K, int M, int K2, long s1, long s2, long sm, T* __restrict__ out) {
    pack_frags_body<T>(src, mode, K, M, K2, s1, s2, sm, out, (long)blockIdx.x * blockDim.x + threadIdx.x);
}
// all weight packs of a train step in ONE launch (they were 74 launches x 4.4 us): blockIdx.y = table row
//   table [n][9] int64 = { src, out, mode, K, M, K2, s1, s2, sm }
template <class T>
__global__ void k_pack_frags_multi(const long long* __restrict__ table) {
    const long long* r = table + (long)blockIdx.y * 9;
    pack_frags_body<T>(reinterpret_cast<const float*>(r[0]), (int)r[2], (int)r[3], (int)r[4], (int)r[5], (long)r[6], (long)r[7], (long)r[8],
                       reinterpret_cast<T*>(r[1]), (long)blockIdx.x * blockDim.x + threadIdx.x);
}

// ----------------------------------------------------------------------------------------------
// fused depthwise+pointwise forward.
//   CG  = channel groups (of 8) per K-chunk = min(CIN,32)/8   -> tile = TP = PX*256/CG pixels, PX pixels x one group per thread
//   MT  = ceil(COUT/16) output-channel tiles, every wave computes all MT tiles for its 16*PTW pixels
// D[cout][pixel] = sum_cin Wpw[cout][cin] * u[pixel][cin],  u = dw3x3(x~)   (channels = MFMA M, pixels = MFMA N)
// so each lane ends up with 4 consecutive output channels of one pixel -> 8/16-byte NHWC stores.
// ----------------------------------------------------------------------------------------------
// Tile geometry: every thread computes the depthwise output of PX horizontally adjacent pixels for one group of 8 channels.
//   PX = 2 (COUT <= 64): the two pixels share 6 of their 9 taps and all 72 weights -> 12 + 9 LDS vector reads per pair instead of
//   2 x (9 + 9): the kernel is LDS-bandwidth bound at the top levels (SQ_LDS_IDX_ACTIVE 77 % of the kernel's cycles), this removes
//   ~40 % of that traffic.  Tiles: CG=1 16x32, CG=2 16x16, CG=4 8x16 pixels (PX = 1, COUT > 64: 8x32 / 8x16 / 8x8).
template <int CG, int PX>
struct FwdTile {
    static constexpr int TP = PX * 256 / CG;
    static constexpr int TW = (PX == 2 && CG == 4) ? 16 : 32 / CG;
    static constexpr int TH = TP / TW;
};
template <int MT>
struct FwdPx {
    static constexpr int PX = MT <= 4 ? 2 : 1;
};
// LDS pitch (elements) of the depthwise-output tile [TP][CG*8 + pad]: 16-byte aligned rows, conflict-free fragment reads
template <class T, int CG>
struct FwdPitch {
    static constexpr int V = CG * 8 + (Elem<T>::is_bf16 ? 8 : 4);
};

#ifndef OCRS_PIPE_BLOCKS
#define OCRS_PIPE_BLOCKS 3
#endif
template <class T, int CG, int MT, bool ONE /* Cin == CG*8: a single K chunk (always true for CG < 4) */, bool POOL = false /* also write the 2x2 max-pool */>
__global__ __launch_bounds__(256, (ONE && Elem<T>::is_bf16) ? OCRS_PIPE_BLOCKS : (MT <= 4 ? 4 : 2)) void k_dwpw_fwd(Src2<T> x, const float* __restrict__ tra, const float* __restrict__ trb,
                                                                    const float* __restrict__ wdw /*master [CIN][1][3][3]*/,
                                                                    const void* __restrict__ wpk, T* __restrict__ z,
                                                                    double* __restrict__ gstat /*[2][COUT]*/, int CIN, int COUT, Tiling2 tg,
                                                                    const float* __restrict__ gamma /*[COUT] or null*/,
                                                                    T* __restrict__ pooled /*[N][H/2][W/2][COUT] or null*/) {
    constexpr int PX = FwdPx<MT>::PX;
    using FT = FwdTile<CG, PX>;
    constexpr int TW = FT::TW, TH = FT::TH, TP = FT::TP;
    constexpr int PTW = TP / 64;
    constexpr int KS = CG * 2;  // fp32 k-steps (of 4) per chunk
    constexpr int PITCH = FwdPitch<T, CG>::V;
    constexpr int HP = HaloTile<TW, TH>::HP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* tile = reinterpret_cast<T*>(smem);                                                   // [TP][PITCH]  dw output (MFMA operand)
    float* xs = reinterpret_cast<float*>(smem + ((TP * PITCH * sizeof(T) + 15) & ~15));     // 2 planes [HP*CG][4]: transformed input + halo
    float* s_par = xs + HP * CG * 8;                                                         // [12][CIN]: tr(3, HaloStager layout) | wdw(9, tap-major)
    float* s_stat = s_par + 12 * CIN;                                                        // [wave][2][MT*16]: one slot per wave, no atomics
    const int H = tg.H, W = tg.W;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    fill_tr8(s_par, x, tra, trb, CIN, tid);
    for (int i = tid; i < 9 * CIN; i += 256) {
        const int t = i / CIN, c = i - t * CIN;
        s_par[3 * CIN + i] = wdw[c * 9 + t];
    }
    for (int i = tid; i < 4 * 2 * MT * 16; i += 256) s_stat[i] = 0.f;
    __syncthreads();
    const HaloStager<T, CG, TW, TH> stager(tid, W);

    const int nkc = ONE ? 1 : CIN / (CG * 8);
    const int pgrp = tid / CG, cg = tid % CG;     // pixel (PX = 1) or pixel pair (PX = 2) of this thread
    const int ty = pgrp / (TW / PX), tx = (pgrp % (TW / PX)) * PX;
    const int pxl = ty * TW + tx;

    TileSched ts(tg.ntiles);
    constexpr bool PIPE = ONE && Elem<T>::is_bf16;  // single-chunk bf16 configs (Cin <= 32, levels 0-2): register-prefetch the next tile
    typename HaloStager<T, CG, TW, TH>::Pending pend;
    typename Mma<T>::Frag wfr[PIPE ? MT : 1];  // PIPE: pointwise weight fragments are tile-invariant -> registers (a global load inside
                                                // the loop would make the compiler wait vmcnt(0), i.e. for the prefetch as well)
    TileIter<TW, TH> tit(tg, ts.first < ts.end ? ts.first : 0, ts.step);  // PIPE: origin of the prefetched tile, no divisions in the loop
    TileOrg org_next = tit.org();
    if constexpr (PIPE) {
#pragma unroll
        for (int b = 0; b < MT; ++b) {
            wfr[b] = Mma<T>::load_w(wpk, (long)b, lane);
            // consume the load HERE: otherwise its first use inside the loop carries a vmcnt(0) (PIPE implies bf16: Frag = uint4)
            asm volatile("" : "+v"(wfr[b].q.x), "+v"(wfr[b].q.y), "+v"(wfr[b].q.z), "+v"(wfr[b].q.w));
        }
        if (ts.first < ts.end) stager.issue(pend, x, 0, org_next, H, W, tid);
    }
    for (long t = ts.first; t < ts.end; t += ts.step) {
        const TileOrg org = PIPE ? org_next : tile_origin2<TW, TH>(tg, (int)t);
        f32x4 acc[PTW][MT];
#pragma unroll
        for (int a = 0; a < PTW; ++a)
#pragma unroll
            for (int b = 0; b < MT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int kc = 0; kc < nkc; ++kc) {
            if constexpr (PIPE) {
                // this tile's input was issued one tile ago; transform it into LDS, then issue the next tile's loads so that
                // their HBM latency hides under the tap / MFMA / store phases below
                stager.commit(pend, s_par, 0, xs, tid);
                __builtin_amdgcn_sched_barrier(0);  // keep the next tile's loads behind ALL of this tile's commit waits
                if (t + ts.step < ts.end) {
                    tit.next();
                    org_next = tit.org();
                    stager.issue(pend, x, 0, org_next, H, W, tid);
                }
            } else
                stager.stage(x, s_par, kc * CG * 8, org, H, W, xs, tid);
            tile_barrier<PIPE>();  // xs ready; all readers of the previous xs / tile passed a barrier since
            if constexpr (PX == 2) {
                float u0[8], u1[8];
                dw2_from_lds<CG, TW, TH>(xs, s_par + 3 * CIN, CIN, (kc * CG + cg) * 8, cg, ty, tx, u0, u1);
                store8_opaque(tile + pxl * PITCH + cg * 8, u0);
                store8_opaque(tile + (pxl + 1) * PITCH + cg * 8, u1);
            } else {
                float u[8];
                dw_from_lds<CG, TW, TH>(xs, s_par + 3 * CIN, CIN, (kc * CG + cg) * 8, cg, ty, tx, u);
                store8_opaque(tile + pxl * PITCH + cg * 8, u);
            }
            tile_barrier<PIPE>();
            typename Mma<T>::Frag pf[PTW];
#pragma unroll
            for (int a = 0; a < PTW; ++a) pf[a] = Mma<T>::load_p(tile, PITCH, (wave * PTW + a) * 16, lane, CG * 8);
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                typename Mma<T>::Frag wf;
                if constexpr (PIPE)
                    wf = wfr[b];
                else
                    wf = Mma<T>::load_w(wpk, (long)kc * MT + b, lane);
#pragma unroll
                for (int a = 0; a < PTW; ++a) acc[a][b] = Mma<T>::template mma<KS>(wf, pf[a], acc[a][b]);
            }
        }
        // epilogue: store z (4 consecutive channels per lane) + per-channel sum / sum of squares
        const long tile_base = ((long)org.n * H + org.h0) * W + org.w0;
#pragma unroll
        for (int b = 0; b < MT; ++b) {
            const int m0 = b * 16 + (lane >> 4) * 4;
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < PTW; ++a) {
                const int q = (wave * PTW + a) * 16 + (lane & 15);  // output pixel of this lane (MFMA N index)
                const int oty = q / TW, otx = q % TW;
                const bool ov = org.h0 + oty < H && org.w0 + otx < W;
                if (ov && m0 < COUT) {
                    const f32x4 v = acc[a][b];
                    store4(z + (tile_base + (long)oty * W + otx) * COUT + m0, v[0], v[1], v[2], v[3]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float qv = Elem<T>::round(v[r]);  // statistics of what the consumer will read
                        s1[r] += qv;
                        s2[r] = fmaf(qv, qv, s2[r]);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a1 = quad16_sum(s1[r]), a2 = quad16_sum(s2[r]);
                if ((lane & 15) == 0) {  // the only lane of this wave that owns channel m0 + r: plain read-modify-write of the wave's slot
                    s_stat[wave * 2 * MT * 16 + m0 + r] += a1;
                    s_stat[wave * 2 * MT * 16 + MT * 16 + m0 + r] += a2;
                }
            }
            // ---- fused MaxPool2d(2) (models.py:54) in its pre-BatchNorm form: the consumer sees relu(bn(z)), monotone in z with the sign of
            // gamma (known before the batch statistics are), so the window's selected element is max z (gamma >= 0) or min z (gamma < 0); its z
            // is written to `pooled` and read through this block's load transform like any block output (see k_maxpool_fwd<T, true>, which
            // this replaces at levels 0-2: the full-size z is not read again).  A 2x2 window = this lane's pixel, the next lane's (same row:
            // an N tile is 16 consecutive pixels of a tile row) and the same two lanes of the N tile one row below (same wave: rows per
            // wave are even).  Ties between different z can only differ from k_maxpool_fwd's first-maximum rule where bn(z) rounds to the
            // same value; the selected VALUE, which is all consumers see, then differs by that rounding.
            if constexpr (POOL && PX == 2 && TW >= 16) {
                {
                    constexpr int TPR = TW / 16;
                    float sg[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) sg[r] = (m0 + r < COUT && gamma[m0 + r] < 0.f) ? -1.f : 1.f;
                    const int Hp = H >> 1, Wp = W >> 1;
#pragma unroll
                    for (int a = 0; a < PTW; ++a) {
                        const int q = (wave * PTW + a) * 16 + (lane & 15);
                        const int oty = q / TW, otx = q % TW;
                        if ((oty & 1) == 0) {  // (compile-time after unrolling: a / TPR is the row within the wave)
                            float m4[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float v0 = sg[r] * Elem<T>::round(acc[a][b][r]), v1 = sg[r] * Elem<T>::round(acc[a + TPR][b][r]);
                                const float vv = fmaxf(v0, v1);
                                m4[r] = sg[r] * fmaxf(vv, dpp_f<0xB1>(vv));  // quad_perm [1,0,3,2]: the horizontally adjacent pixel (lane ^ 1), no LDS
                            }
                            const int ph = (org.h0 + oty) >> 1, pw = (org.w0 + otx) >> 1;
                            if ((lane & 1) == 0 && ph < Hp && pw < Wp && m0 < COUT)
                                store4(pooled + (((long)org.n * Hp + ph) * Wp + pw) * COUT + m0, m4[0], m4[1], m4[2], m4[3]);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int c = tid; c < COUT; c += 256) {
        constexpr int WS = 2 * MT * 16;
        atomicAdd(&gstat[c], (double)((s_stat[c] + s_stat[WS + c]) + (s_stat[2 * WS + c] + s_stat[3 * WS + c])));
        atomicAdd(&gstat[COUT + c], (double)((s_stat[MT * 16 + c] + s_stat[WS + MT * 16 + c]) + (s_stat[2 * WS + MT * 16 + c] + s_stat[3 * WS + MT * 16 + c])));
    }
}

// ----------------------------------------------------------------------------------------------
// first block of the net: 1 -> 8 channels (models.py:115 in_conv.seq.0), input = the greyscale image itself.
template <class T>
__global__ __launch_bounds__(256) void k_dwpw_c1_fwd(const float* __restrict__ img, const float* __restrict__ wdw /*[9]*/,
                                                     const float* __restrict__ wpw /*[8]*/, T* __restrict__ z, double* __restrict__ gstat,
                                                     int H, int W, long P) {
    __shared__ float s_slots[4 * 16];
    float wd[9], wp[8];
#pragma unroll
    for (int i = 0; i < 9; ++i) wd[i] = wdw[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) wp[i] = wpw[i];
    float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // software pipeline over the grid-stride loop: the neighbourhood of iteration i+1 is in flight while iteration i is computed and
    // stored; two buffers, loop unrolled by two (a buffer that a load is still filling cannot be copied without waiting for it)
    const long stride = (long)gridDim.x * 256;
    PixIter pit((long)blockIdx.x * 256 + threadIdx.x, stride, H, W);
    long p = (long)blockIdx.x * 256 + threadIdx.x;
    auto compute = [&](const Nb9& nbr, long pp) {
        float nb[9];
        nb9_finish(nbr, nb);
        float u = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) u = fmaf(wd[k], nb[k], u);
        u = Elem<T>::round(u);
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            o[i] = wp[i] * u;
            const float q = Elem<T>::round(o[i]);
            s1[i] += q;
            s2[i] = fmaf(q, q, s2[i]);
        }
        store8(z + pp * 8, o);
    };
    Nb9 bufA, bufB;
    nb9_issue(bufA, img, pit.cur(), H, W, p < P);
    while (p < P) {
        pit.next();
        nb9_issue(bufB, img, pit.cur(), H, W, p + stride < P);
        compute(bufA, p);
        p += stride;
        if (p >= P) break;
        pit.next();
        nb9_issue(bufA, img, pit.cur(), H, W, p + stride < P);
        compute(bufB, p);
        p += stride;
    }
    float all[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        all[i] = s1[i];
        all[8 + i] = s2[i];
    }
    const float tot = block_sum_det<16>(all, s_slots);  // fixed-order block sum; fp64 across blocks (exact for comparable fp32 partials)
    if (threadIdx.x < 16) atomicAdd(&gstat[threadIdx.x], (double)tot);
}

// BatchNorm2d training-mode statistics (biased var for normalisation, unbiased into running_var, eps 1e-5,
// momentum 0.1)  ->  per-channel load transform + saved mean/rstd.
__global__ void k_bn_finalize(const double* __restrict__ gstat, long count, int C, const float* __restrict__ gamma,
                              const float* __restrict__ beta, float eps, float momentum, float* __restrict__ tr /*[3][C]*/,
                              float* __restrict__ saved /*[2][C] mean|rstd*/, float* __restrict__ run_mean, float* __restrict__ run_var,
                              long long* __restrict__ nbt, float lo) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) *nbt += 1;
    if (c >= C) return;
    const double mean = gstat[c] / (double)count;
    double var = gstat[C + c] / (double)count - mean * mean;
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

// MaxPool2d(2) (models.py:54) over relu(bn(z)); floor output size.
// RAW = false: out = max over the window of act(z).  RAW = true: out = the PRE-BatchNorm z of the element with the largest act(z) (first
// one on ties), so that the pooled tensor is consumed exactly like a block output (consumers apply the producer's load transform, which
// reproduces act(z_sel) = the max bit for bit) and the depthwise-backward passes that read it can produce the producer's
// BatchNorm-backward sums (the gradient only lands on the selected elements) -- no separate bn_bwd_reduce pass over the full-size z.
template <class T, bool RAW>
__global__ __launch_bounds__(256) void k_maxpool_fwd(const T* __restrict__ z, const float* __restrict__ tr, T* __restrict__ out, int C, int H,
                                                     int W, long Pp) {
    const int CG = C / 8;
    const int Hp = H >> 1, Wp = W >> 1;
    const long total = Pp * CG;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
        const long pp = it / CG;
        const int c0 = (int)(it - pp * CG) * 8;
        const PixIdx q = decode_pixel(pp, Hp, Wp);
        const long base = ((long)q.n * H + 2 * q.h) * W + 2 * q.w;
        float m[8], zr[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v[8], a[8];
            load8(z + (base + (long)(k >> 1) * W + (k & 1)) * C + c0, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = v[i];
            apply_tr8(a, tr, C, c0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (k == 0 || a[i] > m[i]) {
                    m[i] = a[i];
                    zr[i] = v[i];
                }
            }
        }
        if (RAW) store8(out + pp * C + c0, zr);
        else store8(out + pp * C + c0, m);
    }
}

// ConvTranspose2d(Cup -> Cout, k3, s2, bias) + crop to (H, W)  (models.py:76-78, 82-87).
// One GEMM "pixel" = an input-aligned position (i, j), i in [0, h], j in [0, w]; it produces the 2x2 output quad
// (2i+py, 2j+px) from the four inputs x~[i-dy][j-dx]:   K = (d, c) = 4*Cup,  M = (q, o) = 4*Cout.
// grid.y selects a block of MT*16 rows of M.
template <class T, int MT>
__global__ __launch_bounds__(256) void k_convt_fwd(const T* __restrict__ x, const float* __restrict__ tr, const void* __restrict__ wpk,
                                                   const float* __restrict__ bias, T* __restrict__ out, int Cup, int Cout, int h, int w,
                                                   int H, int W, int N, int MT_total) {
    constexpr int TP = 64, PITCH = Mma<T>::LDS_PITCH;
    __shared__ __attribute__((aligned(16))) T tile[TP * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pxl = tid >> 2, cg = tid & 3;
    const int hp = h + 1, wp = w + 1;
    const long P = (long)N * hp * wp;
    const long ntiles = (P + TP - 1) / TP;
    const int nkc = (4 * Cup) / 32;
    const int mt0 = blockIdx.y * MT;
    TileSched ts(ntiles);
    for (long t = ts.first; t < ts.end; t += ts.step) {
        const long p = t * TP + pxl;
        const bool pv = p < P;
        const PixIdx px = decode_pixel(pv ? p : 0, hp, wp);
        f32x4 acc[MT];
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kc = 0; kc < nkc; ++kc) {
            const int k0 = kc * 32 + cg * 8;
            const int d = k0 / Cup, c0 = k0 - d * Cup;
            const int ii = px.h - (d >> 1), jj = px.w - (d & 1);
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (pv && ii >= 0 && ii < h && jj >= 0 && jj < w) {
                load8(x + (((long)px.n * h + ii) * w + jj) * Cup + c0, v);
                apply_tr8(v, tr, Cup, c0);
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
            const PixIdx q = decode_pixel(po, hp, wp);
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                const int m0 = (mt0 + b) * 16 + (lane >> 4) * 4;
                const int par = m0 / Cout, o0 = m0 - par * Cout;
                const int Y = 2 * q.h + (par >> 1), X = 2 * q.w + (par & 1);
                if (par < 4 && Y < H && X < W) {
                    const f32x4 a = acc[b];
                    store4(out + (((long)q.n * H + Y) * W + X) * Cout + o0, a[0] + bias[o0], a[1] + bias[o0 + 1], a[2] + bias[o0 + 2],
                           a[3] + bias[o0 + 3]);
                }
            }
        }
        __syncthreads();
    }
}

// out_conv: Conv2d(8 -> 1, 1x1, bias) + Sigmoid (models.py:125-129) on relu(bn(z)).  pred is fp32 (B,1,H,W).
template <class T>
__global__ __launch_bounds__(256) void k_head_fwd(const T* __restrict__ z, const float* __restrict__ tr, const float* __restrict__ w,
                                                  const float* __restrict__ b, float* __restrict__ pred, long P) {
    float wv[8], sc[8], sh[8], lo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        wv[i] = w[i];
        sc[i] = tr[i];
        sh[i] = tr[8 + i];
        lo[i] = tr[16 + i];
    }
    const float bias = b[0];
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
        float v[8];
        load8(z + p * 8, v);
        float s = bias;
#pragma unroll
        for (int i = 0; i < 8; ++i) s = fmaf(wv[i], fmaxf(fmaf(v[i], sc[i], sh[i]), lo[i]), s);
        pred[p] = 1.f / (1.f + __expf(-s));
    }
}

// ----------------------------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------------------------
static inline int ew_grid(long items) {
    long g = (items + 255) / 256;
    const long cap = (long)kNumCU * 8;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

extern "C" {

int ocrs_pack_frags(const float* src, int mode, int K, int M, int K2, long s1, long s2, long sm, void* out, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(src && out && K > 0 && M > 0 && K2 > 0);
    const long n = (long)((K + 31) / 32) * ((M + 15) / 16) * 64;
    const int grid = (int)((n + 255) / 256);
    if (dtype == 1)
        hipLaunchKernelGGL(k_pack_frags<bf16>, dim3(grid), dim3(256), 0, st, src, mode, K, M, K2, s1, s2, sm, (bf16*)out);
    else
        hipLaunchKernelGGL(k_pack_frags<float>, dim3(grid), dim3(256), 0, st, src, mode, K, M, K2, s1, s2, sm, (float*)out);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// table: device int64 [n][9] = { src ptr, out ptr, mode, K, M, K2, s1, s2, sm } (the arguments of ocrs_pack_frags per row);
// max_frag_threads = max over rows of ceil(K/32) * ceil(M/16) * 64.
int ocrs_pack_frags_multi(const long long* table, int n, long max_frag_threads, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(table && n > 0 && max_frag_threads > 0);
    const dim3 grid((unsigned)((max_frag_threads + 255) / 256), (unsigned)n);
    if (dtype == 1)
        hipLaunchKernelGGL(k_pack_frags_multi<bf16>, grid, dim3(256), 0, st, table);
    else
        hipLaunchKernelGGL(k_pack_frags_multi<float>, grid, dim3(256), 0, st, table);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// number of bytes a packed fragment buffer needs
long ocrs_pack_frags_bytes(int K, int M, int dtype) { return (long)((K + 31) / 32) * ((M + 15) / 16) * 64 * 8 * (dtype == 1 ? 2 : 4); }

}  // extern "C" (templates need C++ linkage)
template <class T, int CG, int MT, bool ONE>
static int launch_dwpw_fwd(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* wpk, void* z,
                           double* gstat, int COUT, int N, int H, int W, const float* gamma, void* pooled, hipStream_t st) {
    using FT = FwdTile<CG, FwdPx<MT>::PX>;
    constexpr int TP = FT::TP;
    const int CIN = Ca + Cb;
    Src2<T> x{(const T*)xa, (const T*)xb, Ca, Cb};
    const Tiling2 tg = make_tiling2(N, H, W, FT::TW, FT::TH);
    const size_t smem = ((TP * FwdPitch<T, CG>::V * sizeof(T) + 15) & ~15) +
                        (HaloTile<FT::TW, FT::TH>::HP * CG * 8 + 12 * CIN + 4 * 2 * MT * 16) * sizeof(float);
    // every block ends with 2*COUT same-address fp64 atomics (~15 ns each, serialised): at the middle levels (a few thousand tiles)
    // two tiles per block halve that tail; below that parallelism matters more (measured: OCRS_FWD_TPB sweep, profiles/README.md)
    static const int fwd_tpb = env_int("OCRS_FWD_TPB", 0);
    const int tpb = fwd_tpb > 0 ? fwd_tpb : (tg.ntiles >= 2048 ? 2 : 1);
    // 4 blocks per CU are resident (registers <= 128, LDS 36 KB): a grid of exactly the resident blocks runs as ONE round (8 per CU = two rounds, each
    // block with its own prologue and 2*COUT fp64 atomics at the end: 4.66 -> 4.52 ms per step over all forward launches)
    static const int fwd_bpc = env_int("OCRS_FWD_BPC", 4);
    const int grid = persistent_grid(tg.ntiles / tpb > 0 ? tg.ntiles / tpb : 1, fwd_bpc);
    if (pooled) {
        if constexpr (FwdPx<MT>::PX == 2 && FT::TW >= 16)
            hipLaunchKernelGGL((k_dwpw_fwd<T, CG, MT, ONE, true>), dim3(grid), dim3(256), smem, st, x, tra, trb, wdw, wpk, (T*)z, gstat, CIN, COUT, tg,
                               gamma, (T*)pooled);
        else
            return OCRS_ERR_ARG;  // (ocrs_dwpw_fwd_pool_supported)
    } else
        hipLaunchKernelGGL((k_dwpw_fwd<T, CG, MT, ONE, false>), dim3(grid), dim3(256), smem, st, x, tra, trb, wdw, wpk, (T*)z, gstat, CIN, COUT, tg,
                           gamma, (T*)pooled);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}
extern "C" {

}  // extern "C" (templates need C++ linkage)

template <class T>
static int dispatch_dwpw_fwd(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* wpk, void* z,
                             double* gstat, int COUT, int N, int H, int W, const float* gamma, void* pooled, hipStream_t st) {
    const int CIN = Ca + Cb;
    const int cg = CIN >= 32 ? 4 : CIN / 8;
    const int mt = (COUT + 15) / 16;
#define DWPW_CASE(CG_, MT_, ONE_) \
    if (cg == CG_ && mt == MT_ && (CIN == CG_ * 8) == ONE_) \
        return launch_dwpw_fwd<T, CG_, MT_, ONE_>(xa, xb, Ca, Cb, tra, trb, wdw, wpk, z, gstat, COUT, N, H, W, gamma, pooled, st);
    DWPW_CASE(1, 1, true) DWPW_CASE(2, 1, true) DWPW_CASE(2, 2, true)
    DWPW_CASE(4, 1, true) DWPW_CASE(4, 2, true) DWPW_CASE(4, 4, true)  // Cin == 32: pipelined single-chunk variants
    DWPW_CASE(4, 1, false) DWPW_CASE(4, 2, false) DWPW_CASE(4, 4, false) DWPW_CASE(4, 8, false) DWPW_CASE(4, 16, false)
#undef DWPW_CASE
    return OCRS_ERR_ARG;
}
extern "C" {

// Fused DepthwiseConv block forward up to the pre-BatchNorm output (reference: ocrs_models/models.py:11-23).
//   xa/xb : NHWC inputs (xb may be null; channel concat [xa | xb] as torch.cat at models.py:89), Ca, Cb multiples of 8
//   tra/trb : [3][Ca] / [3][Cb] load transforms of the inputs (producer BN+ReLU, or identity)
//   wdw   : depthwise weights in the reference layout [Cin][1][3][3]; wpk: pointwise weights packed by ocrs_pack_frags(K=Cin, M=Cout)
//   z     : [P][Cout] pre-BN output; gstat: [2][Cout] double, sum z and sum z^2 ACCUMULATED (the caller zeroes it: one
//           memset for all layers of a step instead of one per launch)
// 1 if ocrs_dwpw_fwd can also write the 2x2-max-pooled (pre-BatchNorm) output: two-pixel tile configurations, Cout <= 64
long det_dwf_supported(int Cin, int Cout, int dtype);  // det_dwf.hip: deep-level forward, a whole tile at once (no fused max-pool)
long det_c1v2_supported(int N, int H, int W);  // det_c1.hip
int det_c1v2_fwd_launch(const float* img, const float* wdw, const float* wpw, void* z, double* gstat, int N, int H, int W, int dtype, hipStream_t st,
                        void* uplane, double* fsum);
int det_c1_bwd_fin_launch(const double* c1acc, const double* fsum, const float* coef, const float* wexp, double* acc64, hipStream_t st);
int det_dwf_launch(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* wpk, void* z,
                   double* gstat, int Cout, int N, int H, int W, hipStream_t st, const FwdFin& fin);
long ocrs_dwpw_fwd_pool_supported(int Cin, int Cout) {
    if (det_dwf_supported(Cin, Cout, 1)) return 0;  // (the deep-level kernel is faster than the fusion saves: the max-pool stays its own pass there)
    return Cout <= 64 && (Cin < 32 || Cin % 32 == 0) ? 1 : 0;
}

//   gamma / pooled (nullable, need ocrs_dwpw_fwd_pool_supported): also write MaxPool2d(2) of the block output in its pre-BatchNorm form
//           (the selected element's z, chosen by the sign of the BatchNorm weight gamma [Cout]) to pooled [N][H/2][W/2][Cout]
int ocrs_dwpw_fwd(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* wpk,
                  void* z, double* gstat, const float* gamma, void* pooled, int Cout, int N, int H, int W, int dtype, hipStream_t st) {
    const int Cin = Ca + Cb;
    OCRS_CHECK_ARG(!pooled || (gamma && ocrs_dwpw_fwd_pool_supported(Cin, Cout)));
    OCRS_CHECK_ARG(xa && tra && wdw && wpk && z && gstat && (Cb == 0 || trb));
    OCRS_CHECK_ARG(Ca % 8 == 0 && Cb % 8 == 0 && Cin >= 8 && (Cin < 32 || Cin % 32 == 0) && Cout % 8 == 0 && Cout <= 256);
    OCRS_CHECK_ARG((Cb == 0) == (xb == nullptr));
    OCRS_CHECK_ARG((long)N * H * W < (1L << 31));
    if (!pooled && det_dwf_supported(Cin, Cout, dtype))
        return det_dwf_launch(xa, xb, Ca, Cb, tra, trb, wdw, wpk, z, gstat, Cout, N, H, W, st, FwdFin{nullptr, 0, nullptr, nullptr, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f});
    return dtype == 1 ? dispatch_dwpw_fwd<bf16>(xa, xb, Ca, Cb, tra, trb, wdw, wpk, z, gstat, Cout, N, H, W, gamma, pooled, st)
                      : dispatch_dwpw_fwd<float>(xa, xb, Ca, Cb, tra, trb, wdw, wpk, z, gstat, Cout, N, H, W, gamma, pooled, st);
}

// ocrs_dwpw_fwd + ocrs_bn_finalize in one launch (the deep-level forward kernel, det_dwf.hip: ocrs_dwpw_fwd_fin_supported): the last workgroup done
// finalises the BatchNorm statistics -- counter: a zeroed device word (left zeroed); count, bn_w .. lo: as ocrs_bn_finalize.  No pooling.
long ocrs_dwpw_fwd_fin_supported(int Cin, int Cout, int dtype) { return det_dwf_supported(Cin, Cout, dtype); }
int ocrs_dwpw_fwd_fin(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const void* wpk, void* z,
                      double* gstat, unsigned* counter, long count, const float* bn_w, const float* bn_b, float eps, float momentum, float* tr, float* saved,
                      float* run_mean, float* run_var, long long* nbt, float lo, int Cout, int N, int H, int W, int dtype, hipStream_t st) {
    const int Cin = Ca + Cb;
    OCRS_CHECK_ARG(xa && tra && wdw && wpk && z && gstat && (Cb == 0 || trb) && (Cb == 0) == (xb == nullptr));
    OCRS_CHECK_ARG(counter && count > 0 && bn_w && bn_b && tr && saved && ocrs_dwpw_fwd_fin_supported(Cin, Cout, dtype) && (long)N * H * W < (1L << 31));
    return det_dwf_launch(xa, xb, Ca, Cb, tra, trb, wdw, wpk, z, gstat, Cout, N, H, W, st,
                          FwdFin{counter, count, bn_w, bn_b, eps, momentum, tr, saved, run_mean, run_var, nbt, lo});
}

// First block (1 -> 8): img fp32 (N,1,H,W); wdw [9]; wpw [8]; z [P][8]; gstat [2][8] accumulated (caller zeroes).
int ocrs_dwpw_c1_fwd(const float* img, const float* wdw, const float* wpw, void* z, double* gstat, int N, int H, int W, int dtype,
                     hipStream_t st) {
    OCRS_CHECK_ARG(img && wdw && wpw && z && gstat);
    if (det_c1v2_supported(N, H, W)) return det_c1v2_fwd_launch(img, wdw, wpw, z, gstat, N, H, W, dtype, st, nullptr, nullptr);  // det_c1.hip
    const long P = (long)N * H * W;
    const int grid = ew_grid(P);
    if (dtype == 1)
        hipLaunchKernelGGL(k_dwpw_c1_fwd<bf16>, dim3(grid), dim3(256), 0, st, img, wdw, wpw, (bf16*)z, gstat, H, W, P);
    else
        hipLaunchKernelGGL(k_dwpw_c1_fwd<float>, dim3(grid), dim3(256), 0, st, img, wdw, wpw, (float*)z, gstat, H, W, P);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// The same, additionally writing the block output's rank-one generator: uplane [N][H][W] bf16 = the rounded depthwise output u, z[p][c] = round(Wpw[c] * u[p])
// (see k_c1_fwd2).  ocrs_dwpw_c1_u_supported: 1 if this shape / dtype has the form (bf16, the 64-column x 2-row kernel).
long ocrs_dwpw_c1_u_supported(int N, int H, int W, int dtype) { return dtype == 1 && det_c1v2_supported(N, H, W) ? 1 : 0; }
int ocrs_dwpw_c1_fwd_u(const float* img, const float* wdw, const float* wpw, void* z, void* uplane, double* gstat, int N, int H, int W, int dtype,
                       hipStream_t st) {
    OCRS_CHECK_ARG(img && wdw && wpw && uplane && gstat && ocrs_dwpw_c1_u_supported(N, H, W, dtype));  // (z may be null: only the u plane is written)
    return det_c1v2_fwd_launch(img, wdw, wpw, z, gstat, N, H, W, dtype, st, uplane, nullptr);
}
// ocrs_dwpw_c1_fwd_u that also accumulates the forward-only sums of the fused first-block backward (ocrs_mm_bwd_fin_xu_c1 + ocrs_c1_bwd_fin):
// fsum [20] fp64 (zeroed by the caller) += sum u | sum u^2 | sum u img(tap) [9] | sum img(tap) [9]
int ocrs_dwpw_c1_fwd_us(const float* img, const float* wdw, const float* wpw, void* z, void* uplane, double* gstat, double* fsum, int N, int H, int W, int dtype,
                        hipStream_t st) {
    OCRS_CHECK_ARG(img && wdw && wpw && uplane && gstat && fsum && dtype == 1 && ocrs_dwpw_c1_u_supported(N, H, W, dtype));
    return det_c1v2_fwd_launch(img, wdw, wpw, z, gstat, N, H, W, dtype, st, uplane, fsum);
}
// The first block's weight gradient from the sums of ocrs_dwpw_c1_fwd_us (fsum) and ocrs_mm_bwd_fin_xu_c1 (c1acc [8][32] fp64) and the block's
// BatchNorm-backward coefficients coef [3][8] (ocrs_bn_bwd_finalize): acc64 [17] += dWpw [8] | dWdw [9], as ocrs_dwpw_c1_bwd.
int ocrs_c1_bwd_fin(const double* c1acc, const double* fsum, const float* coef, const float* wexp, double* acc64, hipStream_t st) {
    OCRS_CHECK_ARG(c1acc && fsum && coef && wexp && acc64);
    return det_c1_bwd_fin_launch(c1acc, fsum, coef, wexp, acc64, st);
}

// BatchNorm2d batch statistics -> load transform (reference: nn.BatchNorm2d at models.py:23, training mode).
int ocrs_bn_finalize(const double* gstat, long count, int C, const float* gamma, const float* beta, float eps, float momentum, float* tr,
                     float* saved, float* run_mean, float* run_var, long long* nbt, float lo, hipStream_t st) {
    OCRS_CHECK_ARG(gstat && gamma && beta && tr && saved && C > 0 && count > 0);
    hipLaunchKernelGGL(k_bn_finalize, dim3((C + 63) / 64), dim3(64), 0, st, gstat, count, C, gamma, beta, eps, momentum, tr, saved, run_mean,
                       run_var, nbt, lo);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// MaxPool2d(2) over relu(bn(z))  (models.py:54).  out: [N][H/2][W/2][C]; raw = 1: the selected elements' pre-BatchNorm z instead of the max
int ocrs_maxpool_fwd(const void* z, const float* tr, void* out, int C, int N, int H, int W, int raw, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(z && tr && out && C % 8 == 0 && H >= 2 && W >= 2 && (raw == 0 || raw == 1));
    const long Pp = (long)N * (H / 2) * (W / 2);
    const int grid = ew_grid(Pp * (C / 8));
#define OCRS_POOL(T_, R_) hipLaunchKernelGGL((k_maxpool_fwd<T_, R_>), dim3(grid), dim3(256), 0, st, (const T_*)z, tr, (T_*)out, C, H, W, Pp)
    if (dtype == 1) {
        if (raw) OCRS_POOL(bf16, true);
        else OCRS_POOL(bf16, false);
    } else {
        if (raw) OCRS_POOL(float, true);
        else OCRS_POOL(float, false);
    }
#undef OCRS_POOL
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

}  // extern "C" (templates need C++ linkage)
// ConvTranspose2d forward for the top levels (bf16, Cup = 16 / 32): 2-D tiles of 8 x 16 quad positions.  The transformed input tile
// (+1 halo row / column: a quad position (qy, qx) reads inputs (qy - dy, qx - dx), dy, dx in {0,1}) is staged ONCE in LDS in bf16
// NHWC order and every MFMA pixel-operand fragment is a single ds_read_b128 straight from it (k = d*Cup + c: 8 consecutive channels of one
// neighbour); packed weight fragments are cached in LDS; the next tile's input is register-prefetched.  Replaces per-chunk
// global gathers (each input pixel fetched 4x) + two barriers per 32-wide K chunk: level 0 0.52 -> 0.25 ms.
template <int CUP, int COUT>
struct CtfCfg {
    static constexpr int TH = 8, TW = 16, TPOS = TH * TW;                 // quad positions per tile
    static constexpr int XH = TH + 1, XW = TW + 1, XP = XH * XW;           // input tile with the top / left halo
    static constexpr int NKC = 4 * CUP / 32, MT = 4 * COUT / 16;           // K chunks of 32, M tiles of 16 (M = (parity, cout))
    static constexpr int XI = XP * CUP / 8, NXI = (XI + 255) / 256;       // 16-byte staging items
    static constexpr int XPITCH = CUP + 8;                                 // elements: rows of 48 / 80 B keep the b128 reads spread over banks
    static constexpr int SMEM = (XP * XPITCH + NKC * MT * 64 * 8) * 2 + 3 * CUP * 4 + 4 * COUT * 4;
};
template <int CUP, int COUT>
__global__ __launch_bounds__(256) void k_convt_fwd_tile(const bf16* __restrict__ x, const float* __restrict__ tr, const void* __restrict__ wpk,
                                                        const float* __restrict__ bias, bf16* __restrict__ out, int h, int w, int H, int W,
                                                        Tiling2 tg) {
    using C = CtfCfg<CUP, COUT>;
    constexpr int TW = C::TW, TH = C::TH, XW = C::XW, MT = C::MT, NKC = C::NKC, XPITCH = C::XPITCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* xs = reinterpret_cast<bf16*>(smem);                              // [XP][XPITCH]
    uint4* s_wf = reinterpret_cast<uint4*>(xs + C::XP * XPITCH);           // [NKC*MT][64] packed weight fragments
    float* s_tr = reinterpret_cast<float*>(s_wf + NKC * MT * 64);          // [CUP/8][3][8]
    float* s_bias = s_tr + 3 * CUP;                                        // [4*COUT] bias per M row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        Src2<bf16> xsrc{x, nullptr, CUP, 0};
        fill_tr8(s_tr, xsrc, tr, nullptr, CUP, tid);
        for (int i = tid; i < NKC * MT * 64; i += 256) s_wf[i] = reinterpret_cast<const uint4*>(wpk)[i];
        for (int i = tid; i < 4 * COUT; i += 256) s_bias[i] = bias[i % COUT];
    }
    // tile-invariant staging descriptors: item it -> halo pixel (hy, hx), channel group
    int xoff[C::NXI], xyx[C::NXI];
#pragma unroll
    for (int j = 0; j < C::NXI; ++j) {
        const int it = tid + j * 256, hp = it / (CUP / 8), cg8 = it % (CUP / 8);
        const int hy = hp / XW, hx = hp % XW;
        xoff[j] = (hy * w + hx) * CUP + cg8 * 8;
        xyx[j] = hy | (hx << 16);
    }
    Raw8<bf16> xr[C::NXI];
    unsigned okx = 0;
    auto issue = [&](const TileOrg& o) {  // o = (n, qy0, qx0); halo corner = input pixel (qy0 - 1, qx0 - 1)
        const bf16* xb = x + (((long)o.n * h + (o.h0 - 1)) * w + (o.w0 - 1)) * CUP;
        okx = 0;
#pragma unroll
        for (int j = 0; j < C::NXI; ++j) {
            const int iy = o.h0 - 1 + (xyx[j] & 0xffff), ix = o.w0 - 1 + (xyx[j] >> 16);
            const bool ok = (C::XI % 256 == 0 || tid + j * 256 < C::XI) && (unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w;
            xr[j] = load8_raw(ok ? xb + xoff[j] : x);
            okx |= ok ? 1u << j : 0u;
        }
    };
    // this lane's two N tiles (pixels) per wave: quad position q = (wave*2 + a)*16 + (lane & 15); operand k-group kg = lane >> 4
    const int kg = lane >> 4;
    TileSched ts(tg.ntiles);
    TileOrg org_next = tile_origin2<TW, TH>(tg, (int)(ts.first < ts.end ? ts.first : 0));
    if (ts.first < ts.end) issue(org_next);
    __syncthreads();
    for (long t = ts.first; t < ts.end; t += ts.step) {
        const TileOrg org = org_next;
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
                store8_opaque(xs + (it / (CUP / 8)) * XPITCH + (it % (CUP / 8)) * 8, v);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + ts.step < ts.end) {
            org_next = tile_origin2<TW, TH>(tg, (int)(t + ts.step));
            issue(org_next);
        }
        lds_barrier();
        f32x4 acc[2][MT];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < MT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            const int k0 = kc * 32 + kg * 8, d = k0 / CUP, c0 = k0 % CUP;  // neighbour d = dy*2 + dx, channels c0..c0+7
            uint4 pf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int q = (wave * 2 + a) * 16 + (lane & 15), qy = q / TW, qx = q % TW;
                // input (qy - dy, qx - dx) = halo pixel (qy + 1 - dy, qx + 1 - dx)
                pf[a] = *reinterpret_cast<const uint4*>(xs + ((qy + 1 - (d >> 1)) * XW + (qx + 1 - (d & 1))) * XPITCH + c0);
            }
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                const uint4 wf = s_wf[(kc * MT + b) * 64 + lane];
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, pf[a]), acc[a][b], 0, 0, 0);
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int q = (wave * 2 + a) * 16 + (lane & 15), qy = org.h0 + q / TW, qx = org.w0 + q % TW;
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                const int m0 = b * 16 + kg * 4, par = m0 / COUT, o0 = m0 % COUT;
                const int Y = 2 * qy + (par >> 1), X = 2 * qx + (par & 1);
                if (qy <= h && qx <= w && Y < H && X < W) {
                    const f32x4 v = acc[a][b];
                    store4(out + (((long)org.n * H + Y) * W + X) * COUT + o0, v[0] + s_bias[m0], v[1] + s_bias[m0 + 1], v[2] + s_bias[m0 + 2],
                           v[3] + s_bias[m0 + 3]);
                }
            }
        }
        lds_barrier();  // all fragment reads done before the next commit overwrites xs
    }
}

template <class T>
static int dispatch_convt_fwd(const void* x, const float* tr, const void* wpk, const float* bias, void* out, int Cup, int Cout, int N, int h,
                              int w, int H, int W, hipStream_t st) {
    const int MT_total = (4 * Cout) / 16;
    const long P = (long)N * (h + 1) * (w + 1);
    const long ntiles = (P + 63) / 64;
#define CONVT_CASE(MT_)                                                                                                                  \
    {                                                                                                                                    \
        const int gy = MT_total / MT_;                                                                                                   \
        int gx = persistent_grid(ntiles, 8);                                                                                             \
        hipLaunchKernelGGL((k_convt_fwd<T, MT_>), dim3(gx, gy), dim3(256), 0, st, (const T*)x, tr, wpk, bias, (T*)out, Cup, Cout, h, w, \
                           H, W, N, MT_total);                                                                                           \
    }
    if (MT_total % 8 == 0)
        CONVT_CASE(8)
    else if (MT_total % 4 == 0)
        CONVT_CASE(4)
    else
        CONVT_CASE(2)
#undef CONVT_CASE
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}
extern "C" {

long det_ctf_supported(int Cup, int Cout, int dtype);  // det_ctf.hip
int det_ctf_launch(const void* x, const float* tr, const void* wpk, const float* bias, void* out, int Cup, int Cout, int N, int h, int w, int H, int W,
                   hipStream_t st);
// ConvTranspose2d(Cup->Cout, k=3, s=2, bias) cropped to (H, W)  (models.py:76-78,82-87).
//   x [N][h][w][Cup] with load transform tr [3][Cup]; wpk = ocrs_pack_frags(mode 1, K=4*Cup, M=4*Cout); out [N][H][W][Cout]
int ocrs_convt_fwd(const void* x, const float* tr, const void* wpk, const float* bias, void* out, int Cup, int Cout, int N, int h, int w,
                   int H, int W, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(x && tr && wpk && bias && out);
    OCRS_CHECK_ARG(Cup % 8 == 0 && Cout % 8 == 0 && (H == 2 * h || H == 2 * h + 1) && (W == 2 * w || W == 2 * w + 1));
#define CTF_CASE(CU_, CO_)                                                                                                                \
    if (dtype == 1 && Cup == CU_ && Cout == CO_) {                                                                                       \
        using CC = CtfCfg<CU_, CO_>;                                                                                                      \
        const Tiling2 tg = make_tiling2(N, h + 1, w + 1, CC::TW, CC::TH); /* quad positions: (h+1) x (w+1) */                             \
        hipLaunchKernelGGL((k_convt_fwd_tile<CU_, CO_>), dim3(persistent_grid(tg.ntiles, 4)), dim3(256), CC::SMEM, st, (const bf16*)x, tr, wpk, bias, \
                           (bf16*)out, h, w, H, W, tg);                                                                                   \
        OCRS_LAUNCH_CHECK();                                                                                                              \
        return OCRS_OK;                                                                                                                   \
    }
    CTF_CASE(16, 8) CTF_CASE(32, 16) CTF_CASE(32, 32)
#undef CTF_CASE
    if (det_ctf_supported(Cup, Cout, dtype)) return det_ctf_launch(x, tr, wpk, bias, out, Cup, Cout, N, h, w, H, W, st);  // deep levels (det_ctf.hip)
    return dtype == 1 ? dispatch_convt_fwd<bf16>(x, tr, wpk, bias, out, Cup, Cout, N, h, w, H, W, st)
                      : dispatch_convt_fwd<float>(x, tr, wpk, bias, out, Cup, Cout, N, h, w, H, W, st);
}

// out_conv 1x1 (8->1) + bias + sigmoid (models.py:125-129).  pred fp32 [N][1][H][W].
int ocrs_head_fwd(const void* z, const float* tr, const float* w, const float* b, float* pred, long P, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(z && tr && w && b && pred && P > 0);
    const int grid = ew_grid(P);
    if (dtype == 1)
        hipLaunchKernelGGL(k_head_fwd<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)z, tr, w, b, pred, P);
    else
        hipLaunchKernelGGL(k_head_fwd<float>, dim3(grid), dim3(256), 0, st, (const float*)z, tr, w, b, pred, P);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

}  // extern "C"
