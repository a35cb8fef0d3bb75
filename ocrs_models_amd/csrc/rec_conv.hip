// Dense convolution / GEMM kernels of the CRNN recogniser (gfx950).  Reference: ocrs_models/models.py:179-251.
//
//  k_conv_igemm    implicit-GEMM KHxKW convolution on MFMA (also plain GEMM with KH=KW=1): forward, dgrad (flipped weights),
//                  GRU input projections, Linear.  Input tile (+halo) of 32 channels staged once in LDS; the nine taps read
//                  MFMA B-fragments straight from that tile at shifted pixel offsets (no im2col buffer).
//  k_wgrad_gather  weight gradient D[rowch][(tap, colch)] = sum_pos A[pos][rowch] * B[pos*stride + tap - pad][colch] (K = positions)
//  k_conv0_*       first layer 1->32 3x3 + bias + ReLU + MaxPool2 fused (fwd, and bwd with window recompute)
//  k_act_pool_fwd / k_rec_bn_reduce / k_dz_apply    BN+ReLU(+MaxPool (2,2)|(2,1)) forward / backward pieces
//  k_avgpool_*     BN (no ReLU) + AvgPool2d((4,1)) on H=5, written directly as the (T, N, C) GRU input
#include "det_common.h"

// ---------------------------------------------------------------------------------------------------------------------
// WM = waves along M: the four waves tile the block's output WM x (4 / WM) -- WM = 1: every wave computes all MT tiles for a quarter of the
// pixels; 2: half of the M tiles x half of the pixel tiles; 4: a quarter of the M tiles x all pixel tiles.  Each weight fragment is fetched
// by 4 / WM waves (the fragments come from L2 -- 72 KB per 32-channel chunk for MT = 8 -- and their re-fetch by every wave was the kernel's
// largest data stream), each pixel fragment is read from LDS by WM waves.
template <class T, int MT, int TH, int TW, int WM = 1>
__global__ __launch_bounds__(256) void k_conv_igemm(const T* __restrict__ x, int ldx, const void* __restrict__ wpk, T* __restrict__ out, int ldo,
                                                    const float* __restrict__ bias, int relu, double* __restrict__ gstat, int Cin, int M,
                                                    int MT_total, int N, int Hi, int Wi, int Ho, int Wo, int KH, int KW, int padh, int padw) {
    constexpr int PITCH = Mma<T>::LDS_PITCH;
    constexpr int NTILES = TH * TW / 16, PTW = NTILES * WM / 4, TPR = TW / 16;  // N-tiles per block / per wave / per tile row
    constexpr int MTW = MT / WM;                                                // M-tiles per wave
    static_assert(NTILES % 4 == 0 && TW % 16 == 0 && MT % WM == 0 && (WM == 1 || WM == 2 || WM == 4), "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* xs = reinterpret_cast<T*>(smem);  // [(TH+KH-1)*(TW+KW-1)][PITCH]
    const int HWp = TW + KW - 1, HHp = TH + KH - 1, HP = HWp * HHp;
    float* s_stat = reinterpret_cast<float*>(smem + ((HP * PITCH * sizeof(T) + 15) & ~15));  // [4 / WM][2][MT*16]: one slot per pixel-wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mt0 = blockIdx.y * MT;
    const int mw0 = (wave % WM) * MTW;  // first M tile (within the block's MT) of this wave
    if (gstat) {
        for (int i = tid; i < (4 / WM) * 2 * MT * 16; i += 256) s_stat[i] = 0.f;
        __syncthreads();
    }
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
    const int ntiles = N * tiles_x * tiles_y;
    const int ncc = Cin / 32;
    int oty[PTW], otx[PTW];
#pragma unroll
    for (int a = 0; a < PTW; ++a) {
        const int q = (wave / WM) * PTW + a;
        oty[a] = q / TPR;
        otx[a] = (q % TPR) * 16;
    }
    TileSched ts(ntiles);
    for (long t = ts.first; t < ts.end; t += ts.step) {
        const int tpi = tiles_x * tiles_y;
        const int n = (int)t / tpi, r = (int)t - n * tpi;
        const int h0 = (r / tiles_x) * TH, w0 = (r % tiles_x) * TW;
        f32x4 acc[PTW][MTW];
#pragma unroll
        for (int a = 0; a < PTW; ++a)
#pragma unroll
            for (int b = 0; b < MTW; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int cc = 0; cc < ncc; ++cc) {
            __syncthreads();  // previous chunk's fragment reads done
            for (int it = tid; it < HP * 4; it += 256) {
                const int hp = it >> 2, g8 = it & 3;
                const int hy = hp / HWp, hx = hp - hy * HWp;
                const int h = h0 + hy - padh, w = w0 + hx - padw;
                float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (h >= 0 && h < Hi && w >= 0 && w < Wi) load8(x + (((long)n * Hi + h) * Wi + w) * ldx + cc * 32 + g8 * 8, v);
                store8(xs + hp * PITCH + g8 * 8, v);
            }
            __syncthreads();
            for (int tap = 0; tap < KH * KW; ++tap) {
                const int ky = tap / KW, kx = tap - ky * KW;
                typename Mma<T>::Frag pf[PTW];
#pragma unroll
                for (int a = 0; a < PTW; ++a) pf[a] = Mma<T>::load_p(xs, PITCH, (oty[a] + ky) * HWp + otx[a] + kx, lane, 32);
                const long kc = (long)tap * ncc + cc;
#pragma unroll
                for (int b = 0; b < MTW; ++b) {
                    // M tiles past the packed weight (M = 97 -> 7 tiles, this block covers 8): read tile 0 instead and use a zero fragment.  The
                    // unguarded read ran 2 KB past the end of the packed buffer in the last K chunk -- a memory access fault whenever that
                    // buffer was the last thing in its allocator segment -- and put the next chunk's weights into the pad columns before
                    const int mtile = mt0 + mw0 + b;
                    typename Mma<T>::Frag wf = Mma<T>::load_w(wpk, kc * MT_total + (mtile < MT_total ? mtile : 0), lane);
                    if (mtile >= MT_total) wf = typename Mma<T>::Frag{};
#pragma unroll
                    for (int a = 0; a < PTW; ++a) acc[a][b] = Mma<T>::template mma<8>(wf, pf[a], acc[a][b]);
                }
            }
        }
        // epilogue
#pragma unroll
        for (int b = 0; b < MTW; ++b) {
            const int m0 = (mt0 + mw0 + b) * 16 + (lane >> 4) * 4;
            float bs[4] = {0.f, 0.f, 0.f, 0.f};
            if (bias) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) bs[r4] = m0 + r4 < M ? bias[m0 + r4] : 0.f;
            }
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < PTW; ++a) {
                const int h = h0 + oty[a], w = w0 + otx[a] + (lane & 15);
                if (h < Ho && w < Wo && m0 < ldo) {
                    float v[4];
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        v[r4] = acc[a][b][r4] + bs[r4];
                        if (relu) v[r4] = fmaxf(v[r4], 0.f);
                    }
                    store4(out + (((long)n * Ho + h) * Wo + w) * ldo + m0, v[0], v[1], v[2], v[3]);
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const float q = Elem<T>::round(v[r4]);
                        s1[r4] += q;
                        s2[r4] = fmaf(q, q, s2[r4]);
                    }
                }
            }
            if (gstat) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float a1 = quad16_sum(s1[r4]), a2 = quad16_sum(s2[r4]);
                    if ((lane & 15) == 0) {  // (wave, channel) has exactly one owner lane: plain adds in program order -> run-to-run bit-stable
                        float* slot = s_stat + (wave / WM) * 2 * MT * 16 + (mw0 + b) * 16 + (lane >> 4) * 4 + r4;
                        slot[0] += a1;
                        slot[MT * 16] += a2;
                    }
                }
            }
        }
    }
    if (gstat) {
        __syncthreads();
        for (int i = tid; i < MT * 16; i += 256) {
            const int m = mt0 * 16 + i;
            if (m < M) {
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int wn = 0; wn < 4 / WM; ++wn) {
                    a1 += s_stat[wn * 2 * MT * 16 + i];
                    a2 += s_stat[wn * 2 * MT * 16 + MT * 16 + i];
                }
                atomicAdd(&gstat[m], (double)a1);  // fp64 sums of fp32 partials are exact, hence order-independent (DESIGN.md "Reproducibility")
                atomicAdd(&gstat[M + m], (double)a2);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// D[ra][(tap, cb)] += sum_pos A~[pos][ra] * B[bpos(pos, tap)][cb];  A grid (N, hA, wA), B grid (N, HB, WB),
// bpos = (i*stride + ky - padh, j*stride + kx - padw).  Block = (<=128 A channels) x (128 columns (tap, cb)); K = positions.
// dW index = (ra * CB + cb) * ntaps + tap  (covers Conv2d [Cout][Cin][kh][kw] with A = dz, and ConvTranspose2d / Linear / GRU).
// LDS tiles are stored transposed ([channel][pixel], pixel = MFMA K) with the 8-pixel column blocks XOR-swizzled by
// (row >> 3) & 7: with 16-byte-aligned rows every 8th row would otherwise start on the same bank (16-way conflicts on the
// scalar transposed stores); lanes are mapped pixel-fastest so 16 lanes write 16 consecutive pixels of one channel.
template <class T>
__device__ __forceinline__ typename Mma<T>::Frag load_swz(const T* tile, int tpp, int row0, int pc, int lane);
template <>
__device__ __forceinline__ Mma<bf16>::Frag load_swz<bf16>(const bf16* tile, int tpp, int row0, int pc, int lane) {
    const int row = row0 + (lane & 15);
    const int blk = (pc * 4 + (lane >> 4)) ^ ((row >> 3) & 7);
    Mma<bf16>::Frag f;
    f.q = *reinterpret_cast<const uint4*>(tile + row * tpp + blk * 8);
    return f;
}
template <>
__device__ __forceinline__ Mma<float>::Frag load_swz<float>(const float* tile, int tpp, int row0, int pc, int lane) {
    const int row = row0 + (lane & 15);
    const int sw = (row >> 3) & 7;
    Mma<float>::Frag f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const int px = pc * 32 + ks * 4 + (lane >> 4);
        f.v[ks] = tile[row * tpp + (((px >> 3) ^ sw) << 3) + (px & 7)];
    }
    return f;
}

// store the same channel of two adjacent pixels (px even) as one LDS word / pair
__device__ __forceinline__ void st_pair(bf16* dst, float a, float b) { *reinterpret_cast<unsigned*>(dst) = pack2bf(a, b); }
__device__ __forceinline__ void st_pair(float* dst, float a, float b) { *reinterpret_cast<float2*>(dst) = make_float2(a, b); }

// TP pixels per tile (128 for bf16, 64 for fp32: LDS).  Software-pipelined: the raw global loads of tile t+1 are issued before the
// MFMA phase of tile t (the kernel runs at one block per CU, so nothing else would hide the HBM latency).
template <class T, int TP>
__global__ __launch_bounds__(256) void k_wgrad_gather(const T* __restrict__ A, int ldA, int CA, const float* __restrict__ trA, const T* __restrict__ B,
                                                      int ldB, int CB, float* __restrict__ dW, int N, int hA, int wA, int HB, int WB, int stride,
                                                      int padh, int padw, int KH, int KW, float* __restrict__ ws) {
    constexpr int TPP = Elem<T>::is_bf16 ? TP + 8 : TP + 4;
    constexpr int NPB = TP / 64;       // 64-pixel blocks per tile (swizzle works inside a 64-pixel block)
    constexpr int NIT = TP / 32;       // (pixel pair, group) items per thread, A side (max) and B side
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* aT = reinterpret_cast<T*>(smem);  // [128][TPP]
    T* bT = aT + 128 * TPP;              // [128][TPP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int CA8 = (CA + 7) & ~7;
    const int nbi = (CA8 + 127) / 128;
    const int ci_base = (blockIdx.y % nbi) * 128;
    const int j_base = (blockIdx.y / nbi) * 128;
    const int ntaps = KH * KW;
    const int J = ntaps * CB;
    const int rows_here = (CA8 - ci_base) < 128 ? (CA8 - ci_base) : 128;
    const int NGA = rows_here / 8;
    const int NG4 = (NGA + 3) / 4;
    const int WTI = (rows_here + 15) / 16;
    const long P = (long)N * hA * wA;
    const long ntiles = (P + TP - 1) / TP;
    {
        const float zero8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = tid * 8; i < 256 * TPP; i += 256 * 8) store8(aT + i, zero8);
    }
    // work items: (pixel PAIR, 8-channel group); lanes = 4 groups x 16 pairs per wave -> 64-byte global segments per pixel,
    // 32-bit LDS stores (two pixels of one channel).  A: item j of this thread = (group gA[j], pair ppA[j]); B: fixed column group.
    const int g4 = tid & 3, pp16 = (tid >> 2) & 15, blk = tid >> 6;
    const int jjB = (blk * 4 + g4) * 8;
    const int j0B = j_base + jjB;
    const bool colB = j0B < J;
    const int tapB = colB ? j0B / CB : 0, c0B = j0B - tapB * CB;
    const int dyB = tapB / KW - padh, dxB = tapB % KW - padw;

    Raw8<T> ra[NIT][2], rb[NIT][2];
    unsigned oka = 0, okb = 0;
    auto issue = [&](long t) {
        const long p0 = t * TP;
        oka = okb = 0;
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int it = tid + 256 * j;
            const int grp = g4 + 4 * ((it >> 6) % NG4), pp = pp16 + 16 * ((it >> 6) / NG4);
            if (grp < NGA && pp < TP / 2) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const long p = p0 + 2 * pp + e;
                    if (p < P) {
                        ra[j][e] = load8_raw(A + p * ldA + ci_base + grp * 8);
                        oka |= 1u << (2 * j + e);
                    }
                }
            }
        }
        const PixIdx base = decode_pixel(p0 < P ? p0 : 0, hA, wA);
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int pp = pp16 + 16 * j;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int off = 2 * pp + e;
                if (colB && p0 + off < P) {
                    int w = base.w + off, h = base.h, n = base.n;
                    while (w >= wA) {
                        w -= wA;
                        if (++h >= hA) {
                            h = 0;
                            ++n;
                        }
                    }
                    const int Y = h * stride + dyB, X = w * stride + dxB;
                    if (Y >= 0 && Y < HB && X >= 0 && X < WB) {
                        rb[j][e] = load8_raw(B + (((long)n * HB + Y) * WB + X) * ldB + c0B);
                        okb |= 1u << (2 * j + e);
                    }
                }
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int it = tid + 256 * j;
            const int grp = g4 + 4 * ((it >> 6) % NG4), pp = pp16 + 16 * ((it >> 6) / NG4);
            if (grp < NGA && pp < TP / 2) {
                const int c0 = grp * 8;
                float v[2][8];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[e][i] = 0.f;
                    if (oka & (1u << (2 * j + e))) {
                        unpack8(ra[j][e], v[e]);
                        if (trA) apply_tr8(v[e], trA, CA, ci_base + c0);
                    }
                }
                const int px = 2 * pp, pb = px >> 6, pl = px & 63;
                T* dst = aT + c0 * TPP + pb * 64 + ((((pl >> 3) ^ ((c0 >> 3) & 7)) << 3) | (pl & 7));
#pragma unroll
                for (int i = 0; i < 8; ++i) st_pair(dst + i * TPP, v[0][i], v[1][i]);
            }
        }
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            const int pp = pp16 + 16 * j;
            float v[2][8];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[e][i] = 0.f;
                if (okb & (1u << (2 * j + e))) unpack8(rb[j][e], v[e]);
            }
            const int px = 2 * pp, pb = px >> 6, pl = px & 63;
            T* dst = bT + jjB * TPP + pb * 64 + ((((pl >> 3) ^ ((jjB >> 3) & 7)) << 3) | (pl & 7));
#pragma unroll
            for (int i = 0; i < 8; ++i) st_pair(dst + i * TPP, v[0][i], v[1][i]);
        }
    };

    f32x4 acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    TileSched ts(ntiles);
    if (ts.first < ts.end) issue(ts.first);
    __syncthreads();
    for (long t = ts.first; t < ts.end; t += ts.step) {
        commit();
        __syncthreads();
        if (t + ts.step < ts.end) issue(t + ts.step);  // in flight during the MFMA phase
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int tt = wave + 4 * j;
            if (tt < WTI * 8) {
                const int ti = tt % WTI, tj = tt / WTI;
#pragma unroll
                for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
                        const typename Mma<T>::Frag fa = load_swz<T>(aT + pb * 64, TPP, ti * 16, pc, lane);
                        const typename Mma<T>::Frag fb = load_swz<T>(bT + pb * 64, TPP, tj * 16, pc, lane);
                        acc[j] = Mma<T>::template mma<8>(fa, fb, acc[j]);
                    }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int tt = wave + 4 * j;
        if (tt < WTI * 8) {
            const int ti = tt % WTI, tj = tt / WTI;
            const int jc = j_base + tj * 16 + (lane & 15);
            if (jc < J) {
                const int ra0 = ci_base + ti * 16 + (lane >> 4) * 4;
                if (ws) {
                    // two-stage atomic-free flush: ws[blockIdx.x][jc][ra] (4 consecutive rows per lane = one 16-byte store)
                    if (ra0 < CA8)
                        *reinterpret_cast<float4*>(ws + ((long)blockIdx.x * J + jc) * CA8 + ra0) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
                } else {
                    const int tap = jc / CB, cb = jc - tap * CB;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (ra0 + r < CA) atomicAdd(&dW[((long)(ra0 + r) * CB + cb) * ntaps + tap], acc[j][r]);
                }
            }
        }
    }
}

// dW[(ra*CB + cb)*ntaps + tap] += sum_bx ws[bx][jc = tap*CB + cb][ra];  grid ceil(n / 32), n = ntaps * CB * CA8 (det_column_sum: 32 elements x
// 8 interleaved chains per block, eight loads in flight per chain -- one thread per element walking its gx partials in turn was a pure
// latency chain: 15-60 us per launch for a few MB)
__global__ __launch_bounds__(256) void k_wgrad_gather_reduce(const float* __restrict__ ws, int gx, int CA, int CA8, int CB, int ntaps,
                                                             float* __restrict__ dW) {
    const long n = (long)ntaps * CB * CA8;
    const long i = (long)blockIdx.x * 32 + (threadIdx.x & 31);
    const int ra = (int)(i % CA8);
    float s;
    if (!det_column_sum(ws, gx, n, (i < n && ra < CA) ? i : n, s)) return;
    const long jc = i / CA8;
    const int tap = (int)(jc / CB), cb = (int)(jc - (long)tap * CB);
    dW[((long)ra * CB + cb) * ntaps + tap] += s;
}

// ---------------------------------------------------------------------------------------------------------------------
// Conv2d 3x3 / stride 1 / pad 1 weight gradient with full operand reuse:
//   dW[co][ci][ky][kx] += sum_{n,h,w} dz[n][h][w][co] * x[n][h+ky-1][w+kx-1][ci]
// Block = 8x16 pixel tile x 128 output channels x one 32-channel slice of Cin (grid.y) x ALL 9 taps:
//   dzT [128 co][128 px]                         (swizzled, staged once per tile)
//   xT  [3 kx][32 ci][10 halo rows x 16 px]      three column-shifted copies of the transposed input halo, so the B fragment of
//                                                 tap (ky,kx) is an ALIGNED 8-pixel read at row offset ky -- no per-tap re-staging.
// D[co][(tap, ci32)] = 8 x 18 MFMA tiles = 36 per wave (144 accumulator registers), K = 128 pixels per tile.
template <class T>
__global__ __launch_bounds__(256) void k_conv3x3_wgrad(const T* __restrict__ dz, int Cout, const T* __restrict__ x, int Cin, float* __restrict__ dW, int N,
                                                       int H, int W, float* __restrict__ ws) {
    constexpr int TH = 8, TW = 16, TP = 128;
    constexpr int DPP = Elem<T>::is_bf16 ? TP + 8 : TP + 4;  // dzT pitch
    constexpr int XR = 10 * 16;                                // pixels of one shifted halo image
    constexpr int XPP = Elem<T>::is_bf16 ? XR + 8 : XR + 4;   // xT pitch
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* dzT = reinterpret_cast<T*>(smem);   // [128][DPP]
    T* xT = dzT + 128 * DPP;               // [3][32][XPP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ci_base = blockIdx.y * 32;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int ntiles = N * tiles_x * tiles_y;
    const int g4 = tid & 3, pp16 = (tid >> 2) & 15, blk = tid >> 6;
    f32x4 acc[36];
#pragma unroll
    for (int j = 0; j < 36; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    TileSched ts(ntiles);
    for (long t = ts.first; t < ts.end; t += ts.step) {
        const int tpi = tiles_x * tiles_y;
        const int n = (int)t / tpi, r = (int)t - n * tpi;
        const int h0 = (r / tiles_x) * TH, w0 = (r % tiles_x) * TW;
        __syncthreads();  // previous tile's fragment reads are done
        // ---- dzT: (pixel pair along w, 8-channel group) items: 64 pairs x 16 groups = 1024 -> 4 per thread
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int grp = g4 + 4 * blk, pp = pp16 + 16 * j;   // grp 0..15, pp 0..63 (pixel pair index: ty = pp / 8, tx = 2*(pp % 8))
            const int ty = pp >> 3, tx = (pp & 7) * 2;
            float v[2][8];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[e][i] = 0.f;
                const int h = h0 + ty, w = w0 + tx + e;
                if (h < H && w < W && grp * 8 < Cout) load8(dz + (((long)n * H + h) * W + w) * Cout + grp * 8, v[e]);
            }
            const int px = ty * 16 + tx, pb = px >> 6, pl = px & 63;
            T* dst = dzT + grp * 8 * DPP + pb * 64 + ((((pl >> 3) ^ (grp & 7)) << 3) | (pl & 7));
#pragma unroll
            for (int i = 0; i < 8; ++i) st_pair(dst + i * DPP, v[0][i], v[1][i]);
        }
        // ---- xT: wave = 8-channel group, lanes walk the 10x18 halo pixels; each pixel goes into the (up to) three shifted copies
        for (int hp = lane; hp < 10 * 18; hp += 64) {
            const int hy = hp / 18, hx = hp - hy * 18;
            const int h = h0 + hy - 1, w = w0 + hx - 1;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (h >= 0 && h < H && w >= 0 && w < W) load8(x + (((long)n * H + h) * W + w) * Cin + ci_base + wave * 8, v);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = hx - kx;  // column of this pixel in the copy shifted by kx
                if (tx >= 0 && tx < 16) {
                    T* dst = xT + (kx * 32 + wave * 8) * XPP + hy * 16 + tx;
#pragma unroll
                    for (int i = 0; i < 8; ++i) Elem<T>::st(dst + i * XPP, v[i]);
                }
            }
        }
        __syncthreads();
        // ---- MFMA: 4 K-chunks of 32 pixels (= 2 tile rows); wave tiles tt = wave + 4j: ti = tt % 8, tj = tt / 8 (tap = tj/2, ci16 = tj%2)
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            const int kg = lane >> 4;
            const int trow = 2 * pc + (kg >> 1), tcol = (kg & 1) * 8;  // tile row / column start of this lane's 8 pixels
            typename Mma<T>::Frag fa[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) fa[a] = load_swz<T>(dzT + (pc >> 1) * 64, DPP, (wave + 4 * a) * 16, pc & 1, lane);
#pragma unroll
            for (int tj = 0; tj < 18; ++tj) {
                const int tap = tj >> 1, ky = tap / 3, kx = tap % 3;
                const T* src = xT + (kx * 32 + (tj & 1) * 16 + (lane & 15)) * XPP + (trow + ky) * 16 + tcol;
                typename Mma<T>::Frag fb;
                if constexpr (Elem<T>::is_bf16) {
                    fb.q = *reinterpret_cast<const uint4*>(src);
                } else {
                    // fp32 fragment: k-step ks holds pixel ks*4 + kg of the 32-pixel chunk (see Mma<float>)
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        const int pk = ks * 4 + kg;  // pixel within the chunk
                        fb.v[ks] = xT[(kx * 32 + (tj & 1) * 16 + (lane & 15)) * XPP + (2 * pc + (pk >> 4) + ky) * 16 + (pk & 15)];
                    }
                }
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[tj * 2 + a] = Mma<T>::template mma<8>(fa[a], fb, acc[tj * 2 + a]);
            }
        }
    }
    // ---- flush: acc[tj*2 + a]: rows co = (wave + 4a)*16 + (lane>>4)*4 + r, column ci = ci_base + (tj&1)*16 + (lane&15), tap = tj>>1
    if (ws) {
        // two-stage, atomic-free and deterministic: this block's partial goes to ws[blockIdx.x][tap][ci][co] with 16-byte stores
        // (4 consecutive co per lane); k_wgrad3x3_reduce sums over blockIdx.x.  (37k float atomics per block were the bottleneck.)
        float* wb = ws + (long)blockIdx.x * 9 * Cin * Cout;
#pragma unroll
        for (int tj = 0; tj < 18; ++tj)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int ci = ci_base + (tj & 1) * 16 + (lane & 15), tap = tj >> 1;
                const int co0 = (wave + 4 * a) * 16 + (lane >> 4) * 4;
                if (co0 < Cout) {
                    const f32x4 v = acc[tj * 2 + a];
                    *reinterpret_cast<float4*>(wb + ((long)tap * Cin + ci) * Cout + co0) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        return;
    }
#pragma unroll
    for (int tj = 0; tj < 18; ++tj)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int ci = ci_base + (tj & 1) * 16 + (lane & 15), tap = tj >> 1;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int co = (wave + 4 * a) * 16 + (lane >> 4) * 4 + r4;
                if (co < Cout && ci < Cin) atomicAdd(&dW[((long)co * Cin + ci) * 9 + tap], acc[tj * 2 + a][r4]);
            }
        }
}

// bf16 variant on the LDS transpose read (see lds_tr8 in common.h): both tiles are staged in their NATURAL NHWC order with plain 16-byte
// stores (dz tile [128 px][Cout], input halo [10 x 18 px][32 ci]) and the K = pixel operands are produced by ds_read_b64_tr_b16 --
// no transposing ds_write_b16 scatters, no shifted copies -- and the next tile's loads are register-prefetched under the 144 MFMAs.
// Same decomposition and accumulator layout as k_conv3x3_wgrad (so the flush is shared).
template <int COUT_MAX>
__global__ __launch_bounds__(256) void k_conv3x3_wgrad_tr(const bf16* __restrict__ dz, int Cout, const bf16* __restrict__ x, int Cin,
                                                          float* __restrict__ dW, int N, int H, int W, float* __restrict__ ws) {
    constexpr int TH = 8, TW = 16;
    constexpr int DP = COUT_MAX + 8;   // dz tile pitch (elements)
    constexpr int XP = 40;             // halo pitch: 32 ci + 8
    constexpr int NDI = 128 * (COUT_MAX / 8) / 256, NXI = 3;  // 16-byte staging items per thread (dz; x halo: 180 px * 4 groups = 720)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* dzs = reinterpret_cast<bf16*>(smem);   // [128][DP]
    bf16* xh = dzs + 128 * DP;                   // [180][XP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ci_base = blockIdx.y * 32;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int ntiles = N * tiles_x * tiles_y;
    const int CG8 = Cout / 8;  // 8-channel groups of dz actually present (<= COUT_MAX / 8)
    f32x4 acc[36];
#pragma unroll
    for (int j = 0; j < 36; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // zero the tiles once: channel columns >= Cout of the dz tile stay zero
    for (int i = tid * 8; i < 128 * DP + 180 * XP; i += 256 * 8) *reinterpret_cast<uint4*>(dzs + i) = make_uint4(0, 0, 0, 0);
    Raw8<bf16> dr[NDI], xr[NXI];
    unsigned okd = 0, okx = 0;
    auto issue = [&](long t) {
        const int tpi = tiles_x * tiles_y;
        const int n = (int)t / tpi, r = (int)t - n * tpi;
        const int h0 = (r / tiles_x) * TH, w0 = (r % tiles_x) * TW;
        okd = okx = 0;
#pragma unroll
        for (int j = 0; j < NDI; ++j) {
            const int it = tid + j * 256, px = it / (COUT_MAX / 8), g8 = it % (COUT_MAX / 8);
            const int h = h0 + px / TW, w = w0 + px % TW;
            const bool ok = h < H && w < W && g8 < CG8;
            dr[j] = load8_raw(ok ? dz + (((long)n * H + h) * W + w) * Cout + g8 * 8 : dz);
            okd |= ok ? 1u << j : 0u;
        }
#pragma unroll
        for (int j = 0; j < NXI; ++j) {
            const int it = tid + j * 256, hp = it >> 2, g8 = it & 3;
            const int h = h0 + hp / 18 - 1, w = w0 + hp % 18 - 1;
            const bool ok = it < 720 && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
            xr[j] = load8_raw(ok ? x + (((long)n * H + h) * W + w) * Cin + ci_base + g8 * 8 : x);
            okx |= ok ? 1u << j : 0u;
        }
    };
    // tile-invariant operand addresses
    const int i16 = lane & 15, kg = lane >> 4;
    const int prow = 4 * kg + (i16 >> 2), pcol = (i16 & 3) * 4;
    TileSched ts(ntiles);
    if (ts.first < ts.end) issue(ts.first);
    __syncthreads();
    for (long t = ts.first; t < ts.end; t += ts.step) {
#pragma unroll
        for (int j = 0; j < NDI; ++j) {
            const int it = tid + j * 256, px = it / (COUT_MAX / 8), g8 = it % (COUT_MAX / 8);
            *reinterpret_cast<uint4*>(dzs + px * DP + g8 * 8) = (okd & (1u << j)) ? dr[j].a : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NXI; ++j) {
            const int it = tid + j * 256;
            if (it < 720) *reinterpret_cast<uint4*>(xh + (it >> 2) * XP + (it & 3) * 8) = (okx & (1u << j)) ? xr[j].a : make_uint4(0, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + ts.step < ts.end) issue(t + ts.step);
        lds_barrier();
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            const int p = pc * 32 + prow, ty = p >> 4, tx = p & 15;  // this lane's supplied position (second read: next tile row)
            bf16x8 fa[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const bf16* ap = dzs + p * DP + (wave + 4 * a) * 16 + pcol;
                fa[a] = lds_tr8(ap, ap + 16 * DP);
            }
#pragma unroll
            for (int tj = 0; tj < 18; ++tj) {
                const int tap = tj >> 1, ky = tap / 3, kx = tap % 3;
                const bf16* bp = xh + ((ty + ky) * 18 + tx + kx) * XP + (tj & 1) * 16 + pcol;
                const bf16x8 fb = lds_tr8(bp, bp + 18 * XP);
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[tj * 2 + a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb, acc[tj * 2 + a], 0, 0, 0);
            }
        }
        lds_barrier();
    }
    // ---- flush (same accumulator layout as k_conv3x3_wgrad)
    if (ws) {
        float* wb = ws + (long)blockIdx.x * 9 * Cin * Cout;
#pragma unroll
        for (int tj = 0; tj < 18; ++tj)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int ci = ci_base + (tj & 1) * 16 + (lane & 15), tap = tj >> 1;
                const int co0 = (wave + 4 * a) * 16 + (lane >> 4) * 4;
                if (co0 < Cout) {
                    const f32x4 v = acc[tj * 2 + a];
                    *reinterpret_cast<float4*>(wb + ((long)tap * Cin + ci) * Cout + co0) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        return;
    }
#pragma unroll
    for (int tj = 0; tj < 18; ++tj)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int ci = ci_base + (tj & 1) * 16 + (lane & 15), tap = tj >> 1;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int co = (wave + 4 * a) * 16 + (lane >> 4) * 4 + r4;
                if (co < Cout && ci < Cin) atomicAdd(&dW[((long)co * Cin + ci) * 9 + tap], acc[tj * 2 + a][r4]);
            }
        }
}


// ---------------------------------------------------------------------------------------------------------------------
// first layer: Conv2d(1,32,3,pad 1,bias) + ReLU + MaxPool2d(2) (models.py:180-187).  One thread per (pooled pixel, 8 out channels).
template <class T>
__global__ __launch_bounds__(256) void k_conv0_fwd(const float* __restrict__ img, const float* __restrict__ w /*[32][9]*/,
                                                   const float* __restrict__ bias, T* __restrict__ out /*[N][H/2][W/2][32]*/, int N, int H, int W) {
    const int Hp = H >> 1, Wp = W >> 1;
    // a thread keeps its group of 8 output channels (the grid's thread count is a multiple of 4): their 72 weights + 8 biases live in registers
    // (re-loading them per pixel made 20 of the kernel's 37 load instructions per store)
    const long gtid = (long)blockIdx.x * 256 + threadIdx.x, nthr = (long)gridDim.x * 256;
    const int c0 = (int)(gtid & 3) * 8;
    float wk[8][9], bs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        bs[i] = bias[c0 + i];
#pragma unroll
        for (int k = 0; k < 9; ++k) wk[i][k] = w[(c0 + i) * 9 + k];
    }
    const long Pp = (long)N * Hp * Wp;
    for (long pp = gtid >> 2; pp < Pp; pp += nthr >> 2) {
        const PixIdx q = decode_pixel(pp, Hp, Wp);
        float patch[4][4];
#pragma unroll
        for (int dy = 0; dy < 4; ++dy)
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                const int h = 2 * q.h + dy - 1, ww = 2 * q.w + dx - 1;
                const bool ok = h >= 0 && h < H && ww >= 0 && ww < W;
                const float v = img[ok ? ((long)q.n * H + h) * W + ww : 0];  // (unconditional load, selected address)
                patch[dy][dx] = ok ? v : 0.f;
            }
        float m[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float best = 0.f;  // ReLU floor: max(relu(a), relu(b), ...) = max(0, a, b, ...)
#pragma unroll
            for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                for (int ox = 0; ox < 2; ++ox) {
                    float s = bs[i];
#pragma unroll
                    for (int k = 0; k < 9; ++k) s = fmaf(wk[i][k], patch[oy + k / 3][ox + k % 3], s);
                    best = fmaxf(best, s);
                }
            m[i] = best;
        }
        store8(out + pp * 32 + c0, m);
    }
}

// backward of the fused first layer: dW [32][9], db [32] accumulated.  g = gradient w.r.t. the pooled output [N][H/2][W/2][32].
template <class T>
__global__ __launch_bounds__(256) void k_conv0_bwd(const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ bias,
                                                   const T* __restrict__ g, float* __restrict__ dW, float* __restrict__ db, int N, int H, int W) {
    __shared__ float s_acc[32 * 10];
    for (int i = threadIdx.x; i < 320; i += 256) s_acc[i] = 0.f;
    __syncthreads();
    const int Hp = H >> 1, Wp = W >> 1;
    const long gtid = (long)blockIdx.x * 256 + threadIdx.x, nthr = (long)gridDim.x * 256;
    const int c0 = (int)(gtid & 3) * 8;
    float acc[8][10];
    float wk[8][9], bs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        bs[i] = bias[c0 + i];
#pragma unroll
        for (int k = 0; k < 9; ++k) wk[i][k] = w[(c0 + i) * 9 + k];
#pragma unroll
        for (int k = 0; k < 10; ++k) acc[i][k] = 0.f;
    }
    const long Pp = (long)N * Hp * Wp;
    for (long pp = gtid >> 2; pp < Pp; pp += nthr >> 2) {
        const PixIdx q = decode_pixel(pp, Hp, Wp);
        float patch[4][4];
#pragma unroll
        for (int dy = 0; dy < 4; ++dy)
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                const int h = 2 * q.h + dy - 1, ww = 2 * q.w + dx - 1;
                patch[dy][dx] = (h >= 0 && h < H && ww >= 0 && ww < W) ? img[((long)q.n * H + h) * W + ww] : 0.f;
            }
        float gv[8];
        load8(g + pp * 32 + c0, gv);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float best = 0.f;
            int bo = -1;  // -1: every candidate <= 0 -> ReLU kills the gradient
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float s = bs[i];
#pragma unroll
                for (int k = 0; k < 9; ++k) s = fmaf(wk[i][k], patch[(o >> 1) + k / 3][(o & 1) + k % 3], s);
                if (s > best) {
                    best = s;
                    bo = o;
                }
            }
            const float gi = bo >= 0 ? gv[i] : 0.f;
            const int oy = bo >= 0 ? (bo >> 1) : 0, ox = bo >= 0 ? (bo & 1) : 0;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                // select the patch element of the winning position without dynamic register indexing
                float pv = patch[k / 3][k % 3];
                pv = (oy == 0 && ox == 1) ? patch[k / 3][1 + k % 3] : pv;
                pv = (oy == 1 && ox == 0) ? patch[1 + k / 3][k % 3] : pv;
                pv = (oy == 1 && ox == 1) ? patch[1 + k / 3][1 + k % 3] : pv;
                acc[i][k] = fmaf(gi, pv, acc[i][k]);
            }
            acc[i][9] += gi;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const float v = lane_class_sum<4>(acc[i][k]);
            if ((threadIdx.x & 63) < 4) atomicAdd(&s_acc[(c0 + i) * 10 + k], v);
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 320; i += 256) {
        const int c = i / 10, k = i - c * 10;
        if (k < 9)
            atomicAdd(&dW[c * 9 + k], s_acc[i]);
        else
            atomicAdd(&db[c], s_acc[i]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// out = maxpool_{PHxPW}( max(z*scale+shift, lo) )  (BN+ReLU+MaxPool, or ReLU+MaxPool with identity scale/shift)
template <class T>
__global__ __launch_bounds__(256) void k_act_pool_fwd(const T* __restrict__ z, const float* __restrict__ tr, T* __restrict__ out, int C, int N, int H,
                                                      int W, int PH, int PW) {
    const int CG = C / 8, Hp = H / PH, Wp = W / PW;
    const long total = (long)N * Hp * Wp * CG;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
        const long pp = it / CG;
        const int c0 = (int)(it - pp * CG) * 8;
        const PixIdx q = decode_pixel(pp, Hp, Wp);
        float m[8];
        for (int k = 0; k < PH * PW; ++k) {
            float v[8];
            load8(z + (((long)q.n * H + q.h * PH + k / PW) * W + q.w * PW + k % PW) * C + c0, v);
            apply_tr8(v, tr, C, c0);
#pragma unroll
            for (int i = 0; i < 8; ++i) m[i] = k == 0 ? v[i] : fmaxf(m[i], v[i]);
        }
        store8(out + pp * C + c0, m);
    }
}

// Block-level sums of the per-thread partials s1 / s2 (thread t holds the 8 channels of group t % (C / 8)) in a FIXED order -- every thread
// parks its values in LDS, thread o < 2 C then adds the 256 / CG contributions of its channel in thread order (float LDS atomics complete in
// arrival order: the statistics differed in their last bits from run to run) -- then ONE fp64 atomic per block and channel (an fp64 sum of
// fp32 partials is exact, hence order-independent; DESIGN.md "Reproducibility").
__device__ __forceinline__ void group_sums_to_gsum(const float (&s1)[8], const float (&s2)[8], int C, double* __restrict__ gsum) {
    __shared__ float s_all[256][17];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s_all[threadIdx.x][i] = s1[i];
        s_all[threadIdx.x][8 + i] = s2[i];
    }
    __syncthreads();
    const int CG = C / 8;
    for (int o = threadIdx.x; o < 2 * C; o += 256) {
        const int which = o >= C ? 1 : 0, c = o - which * C, col = which * 8 + (c & 7);
        float a = 0.f;
        for (int t = c >> 3; t < 256; t += CG) a += s_all[t][col];
        atomicAdd(&gsum[o], (double)a);
    }
}

// BN-backward reductions through ReLU (+ max-pool PHxPW, gradient to the FIRST maximum of each window):
// gsum[0][c] = sum ghat, gsum[1][c] = sum ghat * zhat.  g is at pooled resolution.
template <class T>
__global__ __launch_bounds__(256) void k_rec_bn_reduce(const T* __restrict__ g, const T* __restrict__ z, const float* __restrict__ bn,
                                                       const float* __restrict__ saved, double* __restrict__ gsum, int C, int N, int H, int W,
                                                       int PH, int PW) {
    const int CG = C / 8, Hp = H / PH, Wp = W / PW;
    const long gtid = (long)blockIdx.x * 256 + threadIdx.x, nthr = (long)gridDim.x * 256;
    const int c0 = (int)(gtid % CG) * 8;
    float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long Pp = (long)N * Hp * Wp;
    for (long pp = gtid / CG; pp < Pp; pp += nthr / CG) {
        const PixIdx q = decode_pixel(pp, Hp, Wp);
        float gv[8], best[8], bz[8];
        load8(g + pp * C + c0, gv);
        for (int k = 0; k < PH * PW; ++k) {
            float zv[8];
            load8(z + (((long)q.n * H + q.h * PH + k / PW) * W + q.w * PW + k % PW) * C + c0, zv);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float y = fmaxf(fmaf(zv[i], bn[c0 + i], bn[C + c0 + i]), 0.f);
                if (k == 0 || y > best[i]) {
                    best[i] = y;
                    bz[i] = zv[i];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float gh = best[i] > 0.f ? gv[i] : 0.f;
            s1[i] += gh;
            s2[i] = fmaf(gh, (bz[i] - saved[c0 + i]) * saved[C + c0 + i], s2[i]);
        }
    }
    group_sums_to_gsum(s1, s2, C, gsum);
}

// dz = coefA * ghat + coefB * z + coefC for every pixel (ghat routed through ReLU and the PHxPW max-pool); pixels outside
// any window (floor mode) get coefB*z + coefC.  With bn = identity and coef = (1,0,0) this is the plain ReLU(+pool) backward.
template <class T>
__global__ __launch_bounds__(256) void k_dz_apply(const T* __restrict__ g, const T* __restrict__ z, const float* __restrict__ bn,
                                                  const float* __restrict__ coef, T* __restrict__ dz, int C, int N, int H, int W, int PH, int PW) {
    const int CG = C / 8;
    // one thread per (window-grid cell incl. the partial last row/col, 8 channels)
    const int Hc = (H + PH - 1) / PH, Wc = (W + PW - 1) / PW, Hp = H / PH, Wp = W / PW;
    const long total = (long)N * Hc * Wc * CG;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
        const long cell = it / CG;
        const int c0 = (int)(it - cell * CG) * 8;
        const PixIdx q = decode_pixel(cell, Hc, Wc);
        const bool full = q.h < Hp && q.w < Wp;
        float gv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (full) load8(g + (((long)q.n * Hp + q.h) * Wp + q.w) * C + c0, gv);
        float zs[4][8], best[8];
        int bk[8];
        for (int k = 0; k < PH * PW; ++k) {
            const int h = q.h * PH + k / PW, w = q.w * PW + k % PW;
            if (h < H && w < W) {
                load8(z + (((long)q.n * H + h) * W + w) * C + c0, zs[k]);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float y = fmaxf(fmaf(zs[k][i], bn[c0 + i], bn[C + c0 + i]), 0.f);
                    if (k == 0 || y > best[i]) {
                        best[i] = y;
                        bk[i] = k;
                    }
                }
            }
        }
        for (int k = 0; k < PH * PW; ++k) {
            const int h = q.h * PH + k / PW, w = q.w * PW + k % PW;
            if (h < H && w < W) {
                float o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float gh = (full && bk[i] == k && best[i] > 0.f) ? gv[i] : 0.f;
                    o[i] = fmaf(coef[c0 + i], gh, fmaf(coef[C + c0 + i], zs[k][i], coef[2 * C + c0 + i]));
                }
                store8(dz + (((long)q.n * H + h) * W + w) * C + c0, o);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The three window kernels above for the window shapes the model has -- (2,2), (2,1), (1,1) -- on tensors the windows tile exactly: window
// shape known at compile time, every load of a cell issued before the first use (the generic kernels wait for each window element in turn,
// 3.3-3.9 TB/s), same arithmetic and same first-maximum rule.
template <class T, int PH, int PW>
__global__ __launch_bounds__(256) void k_act_pool_fwd_t(const T* __restrict__ z, const float* __restrict__ tr, T* __restrict__ out, int C, int N, int H,
                                                        int W) {
    const int CG = C / 8, Hp = H / PH, Wp = W / PW;
    // the grid's thread count is a multiple of CG (256 % CG == 0): a thread keeps its channel group, whose parameters live in registers
    const long gtid = (long)blockIdx.x * 256 + threadIdx.x, nthr = (long)gridDim.x * 256;
    const int c0 = (int)(gtid % CG) * 8;
    float sc[8], sh[8], lo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        sc[i] = tr[c0 + i];
        sh[i] = tr[C + c0 + i];
        lo[i] = tr[2 * C + c0 + i];
    }
    const long Pp = (long)N * Hp * Wp;
    for (long pp = gtid / CG; pp < Pp; pp += nthr / CG) {
        const PixIdx q = decode_pixel(pp, Hp, Wp);
        const T* zb = z + (((long)q.n * H + q.h * PH) * W + q.w * PW) * C + c0;
        Raw8<T> raw[PH * PW];
#pragma unroll
        for (int k = 0; k < PH * PW; ++k) raw[k] = load8_raw(zb + ((long)(k / PW) * W + k % PW) * C);
        float m[8];
#pragma unroll
        for (int k = 0; k < PH * PW; ++k) {
            float v[8];
            unpack8(raw[k], v);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float y = fmaxf(fmaf(v[i], sc[i], sh[i]), lo[i]);
                m[i] = k == 0 ? y : fmaxf(m[i], y);
            }
        }
        store8(out + pp * C + c0, m);
    }
}

template <class T, int PH, int PW>
__global__ __launch_bounds__(256) void k_rec_bn_reduce_t(const T* __restrict__ g, const T* __restrict__ z, const float* __restrict__ bn,
                                                         const float* __restrict__ saved, double* __restrict__ gsum, int C, int N, int H, int W) {
    const int CG = C / 8, Hp = H / PH, Wp = W / PW;
    const long gtid = (long)blockIdx.x * 256 + threadIdx.x, nthr = (long)gridDim.x * 256;
    const int c0 = (int)(gtid % CG) * 8;
    float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float sc[8], sh[8], mu[8], rs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        sc[i] = bn[c0 + i];
        sh[i] = bn[C + c0 + i];
        mu[i] = saved[c0 + i];
        rs[i] = saved[C + c0 + i];
    }
    const long Pp = (long)N * Hp * Wp;
    for (long pp = gtid / CG; pp < Pp; pp += nthr / CG) {
        const PixIdx q = decode_pixel(pp, Hp, Wp);
        const T* zb = z + (((long)q.n * H + q.h * PH) * W + q.w * PW) * C + c0;
        const Raw8<T> graw = load8_raw(g + pp * C + c0);
        Raw8<T> raw[PH * PW];
#pragma unroll
        for (int k = 0; k < PH * PW; ++k) raw[k] = load8_raw(zb + ((long)(k / PW) * W + k % PW) * C);
        float gv[8], best[8], bz[8];
        unpack8(graw, gv);
#pragma unroll
        for (int k = 0; k < PH * PW; ++k) {
            float zv[8];
            unpack8(raw[k], zv);
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
            const float gh = best[i] > 0.f ? gv[i] : 0.f;
            s1[i] += gh;
            s2[i] = fmaf(gh, (bz[i] - mu[i]) * rs[i], s2[i]);
        }
    }
    group_sums_to_gsum(s1, s2, C, gsum);
}

// NTB threads per block: 256, or 1024 when the column sums are asked for (a quarter of the trailing same-address atomics at the same occupancy:
// with 256-thread blocks the atomic tail made the fused pass slower than dz_apply + a separate column-sum kernel)
template <class T, int PH, int PW, int NTB>
__global__ __launch_bounds__(NTB) void k_dz_apply_t(const T* __restrict__ g, const T* __restrict__ z, const float* __restrict__ bn,
                                                    const float* __restrict__ coef, T* __restrict__ dz, int C, int N, int H, int W,
                                                    float* __restrict__ dsum /*nullable [C]: += column sums of the stored dz (a bias gradient)*/,
                                                    float* __restrict__ dsum_ws /*nullable: per-block partials [gridDim.x][C] instead of the atomics (rec_defer_partials)*/) {
    const int CG = C / 8, Hp = H / PH, Wp = W / PW;
    const long gtid = (long)blockIdx.x * NTB + threadIdx.x, nthr = (long)gridDim.x * NTB;
    const int c0 = (int)(gtid % CG) * 8;  // (fixed per thread, see k_act_pool_fwd_t)
    float sc[8], sh[8], ca[8], cb[8], cc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        sc[i] = bn[c0 + i];
        sh[i] = bn[C + c0 + i];
        ca[i] = coef[c0 + i];
        cb[i] = coef[C + c0 + i];
        cc[i] = coef[2 * C + c0 + i];
    }
    float ds[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const long Pp = (long)N * Hp * Wp;
    for (long pp = gtid / CG; pp < Pp; pp += nthr / CG) {
        const PixIdx q = decode_pixel(pp, Hp, Wp);
        const long zoff = (((long)q.n * H + q.h * PH) * W + q.w * PW) * C + c0;
        const Raw8<T> graw = load8_raw(g + pp * C + c0);
        Raw8<T> raw[PH * PW];
#pragma unroll
        for (int k = 0; k < PH * PW; ++k) raw[k] = load8_raw(z + zoff + ((long)(k / PW) * W + k % PW) * C);
        float gv[8], zs[PH * PW][8], best[8];
        int bk[8];
        unpack8(graw, gv);
#pragma unroll
        for (int k = 0; k < PH * PW; ++k) {
            unpack8(raw[k], zs[k]);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float y = fmaxf(fmaf(zs[k][i], sc[i], sh[i]), 0.f);
                if (k == 0 || y > best[i]) {
                    best[i] = y;
                    bk[i] = k;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < PH * PW; ++k) {
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float gh = (bk[i] == k && best[i] > 0.f) ? gv[i] : 0.f;
                o[i] = fmaf(ca[i], gh, fmaf(cb[i], zs[k][i], cc[i]));
                if (dsum) ds[i] += Elem<T>::round(o[i]);
            }
            store8(dz + zoff + ((long)(k / PW) * W + k % PW) * C, o);
        }
    }
    if (dsum) {  // (kernel-uniform) fixed-order block sums, then one fp32 atomic per block and channel, like k_col_sum4's
        __shared__ float s_all[NTB][9];
#pragma unroll
        for (int i = 0; i < 8; ++i) s_all[threadIdx.x][i] = ds[i];
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += NTB) {
            float a = 0.f;
            for (int t = c >> 3; t < NTB; t += CG) a += s_all[t][c & 7];
            if (dsum_ws) dsum_ws[(long)blockIdx.x * C + c] = a;
            else atomicAdd(&dsum[c], a);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// BatchNorm2d (no ReLU) + AvgPool2d((4,1)) on H=5 (models.py:234-242) written as the GRU input seq[t=w][n][c] (fp32):
//   seq = mean_{h<4}(z[n][h][w][c]) * scale + shift
template <class T>
__global__ __launch_bounds__(256) void k_avgpool_fwd(const T* __restrict__ z, const float* __restrict__ tr, float* __restrict__ seq, int C, int N, int H,
                                                     int W) {
    const int CG = C / 8;
    const long total = (long)N * W * CG;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
        const long nw = it / CG;
        const int c0 = (int)(it - nw * CG) * 8;
        const int n = (int)(nw / W), w = (int)(nw - (long)n * W);
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int h = 0; h < 4; ++h) {
            float v[8];
            load8(z + (((long)n * H + h) * W + w) * C + c0, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] += v[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = fmaf(s[i] * 0.25f, tr[c0 + i], tr[C + c0 + i]);
        store8(seq + ((long)w * N + n) * C + c0, s);
    }
}

// reductions for its backward: ghat[n][h<4][w][c] = gseq[w][n][c] / 4, row 4: 0.  BN count = N*H*W.
template <class T>
__global__ __launch_bounds__(256) void k_avgpool_bn_reduce(const float* __restrict__ gseq, const T* __restrict__ z, const float* __restrict__ saved,
                                                           double* __restrict__ gsum, int C, int N, int H, int W) {
    const int CG = C / 8;
    const long gtid = (long)blockIdx.x * 256 + threadIdx.x, nthr = (long)gridDim.x * 256;
    const int c0 = (int)(gtid % CG) * 8;
    float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float mu[8], rs[8];  // (per-thread channel group: parameters in registers, all five loads of a column issued before the first use)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        mu[i] = saved[c0 + i];
        rs[i] = saved[C + c0 + i];
    }
    for (long nw = gtid / CG; nw < (long)N * W; nw += nthr / CG) {
        const int n = (int)(nw / W), w = (int)(nw - (long)n * W);
        float gv[8];
        Raw8<T> raw[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) raw[h] = load8_raw(z + (((long)n * H + h) * W + w) * C + c0);
        load8(gseq + ((long)w * N + n) * C + c0, gv);
        float zs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            float v[8];
            unpack8(raw[h], v);
#pragma unroll
            for (int i = 0; i < 8; ++i) zs[i] += (v[i] - mu[i]) * rs[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s1[i] += gv[i];
            s2[i] = fmaf(gv[i] * 0.25f, zs[i], s2[i]);
        }
    }
    group_sums_to_gsum(s1, s2, C, gsum);
}

template <class T>
__global__ __launch_bounds__(256) void k_avgpool_dz(const float* __restrict__ gseq, const T* __restrict__ z, const float* __restrict__ coef,
                                                    T* __restrict__ dz, int C, int N, int H, int W) {
    const int CG = C / 8;
    const long total = (long)N * H * W * CG;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
        const long p = it / CG;
        const int c0 = (int)(it - p * CG) * 8;
        const PixIdx q = decode_pixel(p, H, W);
        float zv[8], gv[8] = {0, 0, 0, 0, 0, 0, 0, 0}, o[8];
        load8(z + p * C + c0, zv);
        if (q.h < 4) load8(gseq + ((long)q.w * N + q.n) * C + c0, gv);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = fmaf(coef[c0 + i], gv[i] * 0.25f, fmaf(coef[C + c0 + i], zv[i], coef[2 * C + c0 + i]));
        store8(dz + p * C + c0, o);
    }
}

// per-column sum of a [rows][ld] fp32/bf16 matrix (bias gradients)
// ws (nullable, round 5): per-block partials [gridDim.x][C] instead of the cross-block float atomics -- summed in a fixed order by the deferred
// reduce launch (ocrs_bwd_defer_begin / _flush, det_bwd.hip).  The in-block sums are fixed-order slots (no LDS float atomics) either way.
template <class T>
__global__ __launch_bounds__(256) void k_col_sum(const T* __restrict__ a, int ld, int C, float* __restrict__ out, long rows, float* __restrict__ ws) {
    extern __shared__ float s_acc[];  // [2 row phases][C]
    const int c = threadIdx.x % 128;
    const int sub = threadIdx.x / 128;
    for (int cb = 0; cb < C; cb += 128) {
        if (cb + c < C) {
            float s = 0.f;
            for (long r = (long)blockIdx.x * 2 + sub; r < rows; r += (long)gridDim.x * 2) s += Elem<T>::ld(a + r * ld + cb + c);
            s_acc[sub * C + cb + c] = s;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256) {
        const float v = s_acc[i] + s_acc[C + i];
        if (ws) ws[(long)blockIdx.x * C + i] = v;
        else atomicAdd(&out[i], v);
    }
}

// The same for C % 4 == 0 and 4-element-aligned rows: a thread owns 4 consecutive columns (one 16 / 8-byte load per row), 64 column quads
// x 4 row phases per block, four rows in flight per thread; per-block partials pre-reduced in LDS, then C atomics per block (grid <= 2/CU).
template <class T>
__global__ __launch_bounds__(256) void k_col_sum4(const T* __restrict__ a, int ld, int C, float* __restrict__ out, long rows, float* __restrict__ ws) {
    extern __shared__ float s_acc[];  // [4 row phases][C]
    const int cq = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const long rstep = (long)gridDim.x * 4;
    for (int cb = 0; cb < C; cb += 256) {
        const int c = cb + cq * 4;
        if (c < C) {
            float s[4] = {0.f, 0.f, 0.f, 0.f};
            long r = (long)blockIdx.x * 4 + ph;
            for (; r + 7 * rstep < rows; r += 8 * rstep) {  // eight rows in flight per thread (2 blocks per CU: 64 KB in flight per CU)
                float v[8][4];
#pragma unroll
                for (int u = 0; u < 8; ++u) load4(a + (r + u * rstep) * ld + c, v[u]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    s[i] += ((v[0][i] + v[1][i]) + (v[2][i] + v[3][i])) + ((v[4][i] + v[5][i]) + (v[6][i] + v[7][i]));
            }
            for (; r + 3 * rstep < rows; r += 4 * rstep) {
                float v[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u) load4(a + (r + u * rstep) * ld + c, v[u]);
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i] += (v[0][i] + v[1][i]) + (v[2][i] + v[3][i]);
            }
            for (; r < rows; r += rstep) {
                float v[4];
                load4(a + r * ld + c, v);
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i] += v[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) s_acc[ph * C + c + i] = s[i];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256) {
        const float v = (s_acc[i] + s_acc[C + i]) + (s_acc[2 * C + i] + s_acc[3 * C + i]);
        if (ws) ws[(long)blockIdx.x * C + i] = v;
        else atomicAdd(&out[i], v);
    }
}

// Deferred mode (ocrs_bwd_defer_begin .. _flush): a launch that would end in cross-block float atomics onto out [n] writes per-block partials
// [nb][n] instead and queues their fixed-order column sum (k_reduce_multi, det_bwd.hip) -- the sums become bit-reproducible.  Returns the partial
// buffer, or null (not deferring / pool or queue full): the launch then falls back to the atomics.
float* rec_defer_partials(int nb, int n, float* out) {
    float* ws = bwd_defer_ws((long)nb * n);
    if (!ws) return nullptr;
    if (!bwd_defer_reduce(ws, nb, n, out, n, n, n, nullptr, 0)) return nullptr;
    return ws;
}
// ---------------------------------------------------------------------------------------------------------------------
static inline int ew_grid(long items) {
    long g = (items + 255) / 256;
    const long cap = (long)kNumCU * 8;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

bool conv3x3_rows_supported(int ldx, int ldo, int Cin, int M, int Hi, int Wi, int Ho, int Wo, int KH, int KW, int padh, int padw, int dtype);  // rec_conv3.hip
int conv3x3_rows_launch(const void* x, int ldx, const void* wpk, void* out, int ldo, const float* bias, int relu, double* gstat, int Cin, int M, int N, int Hi, int Wi,
                        int Ho, int Wo, int KW, int pad, hipStream_t st);
bool conv3x3_tile_supported(int ldx, int ldo, int Cin, int M, int Hi, int Wi, int Ho, int Wo, int KH, int KW, int padh, int padw, int dtype);  // rec_conv4.hip
int conv3x3_tile_launch(const void* x, int ldx, const void* wpk, void* out, int ldo, const float* bias, int relu, double* gstat, int Cin, int M, int N, int H, int W,
                        hipStream_t st);
bool conv3x3_c128_supported(int ldx, int ldo, int Cin, int M, int Hi, int Wi, int Ho, int Wo, int KH, int KW, int padh, int padw, int dtype);  // rec_conv2.hip
int conv3x3_c128_launch(const void* x, int ldx, const void* wpk, void* out, int ldo, const float* bias, int relu, double* gstat, int Cin, int N, int H, int W,
                        hipStream_t st);

template <class T, int TH, int TW>
static int launch_igemm(const void* x, int ldx, const void* wpk, void* out, int ldo, const float* bias, int relu, double* gstat, int Cin, int M,
                        int N, int Hi, int Wi, int Ho, int Wo, int KH, int KW, int padh, int padw, hipStream_t st) {
    const int MT_total = (M + 15) / 16;
    const int tiles = N * ((Wo + TW - 1) / TW) * ((Ho + TH - 1) / TH);
    const int HP = (TH + KH - 1) * (TW + KW - 1);
#define IG(MT_)                                                                                                                                  \
    {                                                                                                                                            \
        const size_t smem = ((HP * Mma<T>::LDS_PITCH * sizeof(T) + 15) & ~15) + 4 * 2 * MT_ * 16 * sizeof(float);                                   \
        const int gy = (MT_total + MT_ - 1) / MT_;                                                                                               \
        constexpr int WM_ = (Elem<T>::is_bf16 && TH > 1) ? (MT_ >= 8 ? 4 : (MT_ >= 4 ? 2 : 1)) : 1; /* measured: 1382 -> 974 us, 270 -> 230 us */ \
        hipLaunchKernelGGL((k_conv_igemm<T, MT_, TH, TW, WM_>), dim3(persistent_grid(tiles, gy >= 4 ? 2 : 4), gy), dim3(256), smem, st, (const T*)x, ldx, \
                           wpk, (T*)out, ldo, bias, relu, gstat, Cin, M, MT_total, N, Hi, Wi, Ho, Wo, KH, KW, padh, padw);                         \
    }
    if (MT_total % 8 == 0 || MT_total > 8)
        IG(8)
    else if (MT_total % 4 == 0 || MT_total > 4)
        IG(4)
    else if (MT_total >= 2)
        IG(2)
    else
        IG(1)
#undef IG
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// Split-bf16 GEMM for fp32 operands (the GRU input projections gi = x W_ih^T + b and their input gradients dx = dgi W_ih, throughput mode):
//   out[p][m] = sum_k X[p][k] * W(m, k) (+ bias[m]),   W(m, k) = KM ? Wm[k * ldw + m] : Wm[m * ldw + k]   (master layout, no packing).
// Same a*b ~ ah*bh + ah*bl + al*bh arithmetic as k_wgrad_gemm_x3.  Block = 128 pixels x 128 outputs, K in chunks of 32 staged as hi / lo
// bf16 planes; the pixel operand and the [m][k] weight operand are direct ds_read_b128 fragments, the [k][m] weight operand comes
// through the LDS transpose read.  Next chunk register-prefetched.
template <bool KM>
__global__ __launch_bounds__(256) void k_gemm_x3(const float* __restrict__ X, int ldx, const float* __restrict__ Wm, int ldw, const float* __restrict__ bias,
                                                 float* __restrict__ out, int ldo, int K, int M, long P, int Kw) {
    constexpr int BP = 128, BM = 128, KC = 32, PK = KC + 8, PM = BM + 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* Xh = reinterpret_cast<bf16*>(smem);  // [BP][PK]
    bf16* Xl = Xh + BP * PK;
    bf16* Wh = Xl + BP * PK;                   // KM ? [KC][PM] : [BM][PK]
    bf16* Wl = Wh + (KM ? KC * PM : BM * PK);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long p0 = (long)blockIdx.x * BP;
    const int m0 = blockIdx.y * BM;
    float4 rx[4], rw[4];
    auto issue = [&](int kc) {
        const int k0 = kc * KC;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + j * 256;
            {
                const int px = f >> 3, k4 = (f & 7) * 4;
                rx[j] = p0 + px < P ? *reinterpret_cast<const float4*>(X + (p0 + px) * ldx + k0 + k4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (KM) {
                const int k = f >> 5, m4 = (f & 31) * 4;
                rw[j] = (m0 + m4 < M && k0 + k < Kw) ? *reinterpret_cast<const float4*>(Wm + (long)(k0 + k) * ldw + m0 + m4) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                const int m = f >> 3, k4 = (f & 7) * 4;
                rw[j] = (m0 + m < M && k0 + k4 < Kw) ? *reinterpret_cast<const float4*>(Wm + (long)(m0 + m) * ldw + k0 + k4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto split_store = [&](const float4& v, bf16* hi, bf16* lo, int off) {
#ifdef X3_FLOOR  // (measurement build: what the kernel would cost if its operands arrived pre-split -- one v_perm per pair instead of the split)
        const unsigned b0 = __float_as_uint(v.x), b1 = __float_as_uint(v.y), b2 = __float_as_uint(v.z), b3 = __float_as_uint(v.w);
        *reinterpret_cast<uint2*>(hi + off) = make_uint2(__builtin_amdgcn_perm(b1, b0, 0x07060302u), __builtin_amdgcn_perm(b3, b2, 0x07060302u));
        *reinterpret_cast<uint2*>(lo + off) = make_uint2(__builtin_amdgcn_perm(b1, b0, 0x05040100u), __builtin_amdgcn_perm(b3, b2, 0x05040100u));
#else
        const float x[4] = {v.x, v.y, v.z, v.w};
        float h[4], l[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h[i] = Elem<bf16>::round(x[i]);
            l[i] = x[i] - h[i];
        }
        *reinterpret_cast<uint2*>(hi + off) = make_uint2(pack2bf(h[0], h[1]), pack2bf(h[2], h[3]));
        *reinterpret_cast<uint2*>(lo + off) = make_uint2(pack2bf(l[0], l[1]), pack2bf(l[2], l[3]));
#endif
    };
    const int wm = wave & 1, wn = wave >> 1;  // 2 x 2 waves: 4 m-tiles x 4 pixel-tiles each
    const int i16 = lane & 15, kg = lane >> 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nkc = K / KC;
    issue(0);
    for (int kc = 0; kc < nkc; ++kc) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + j * 256;
            split_store(rx[j], Xh, Xl, (f >> 3) * PK + (f & 7) * 4);
            if (KM)
                split_store(rw[j], Wh, Wl, (f >> 5) * PM + (f & 31) * 4);
            else
                split_store(rw[j], Wh, Wl, (f >> 3) * PK + (f & 7) * 4);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kc + 1 < nkc) issue(kc + 1);
        lds_barrier();
        bf16x8 wh[4], wl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (KM) {
                // K order must match the pixel operand's direct reads (lane group kg holds k = 8kg .. 8kg+7): rows 8kg + 0..3, then + 4..7
                const int o = (8 * kg + (i16 >> 2)) * PM + (wm * 4 + i) * 16 + (i16 & 3) * 4;
                wh[i] = lds_tr8(Wh + o, Wh + o + 4 * PM);
                wl[i] = lds_tr8(Wl + o, Wl + o + 4 * PM);
            } else {
                const int o = ((wm * 4 + i) * 16 + i16) * PK + kg * 8;
                wh[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Wh + o));
                wl[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Wl + o));
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = ((wn * 4 + j) * 16 + i16) * PK + kg * 8;
            const bf16x8 xh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Xh + o));
            const bf16x8 xl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Xl + o));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], xh, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], xl, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[i], xh, acc[i][j], 0, 0, 0);
            }
        }
        lds_barrier();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + (wm * 4 + i) * 16 + kg * 4;
        if (m >= M) continue;
        float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) {
            if (m + 3 < M)
                bs = *reinterpret_cast<const float4*>(bias + m);
            else  // last, partial quad of a ragged M (km = 0 only): columns [M, round_up(M, 4)) are written as 0
                bs = make_float4(bias[m], m + 1 < M ? bias[m + 1] : 0.f, m + 2 < M ? bias[m + 2] : 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long p = p0 + (wn * 4 + j) * 16 + i16;
            if (p < P) {
                const f32x4 v = acc[i][j];
                *reinterpret_cast<float4*>(out + p * ldo + m) = make_float4(v[0] + bs.x, v[1] + bs.y, v[2] + bs.z, v[3] + bs.w);
            }
        }
    }
}


extern "C" {

// Implicit-GEMM convolution / GEMM:  out[n][ho][wo][m] = sum_{ky,kx,c} W[m][(ky,kx,c)] * x[n][ho+ky-padh][wo+kx-padw][c]  (+bias, ReLU)
//   nn.Conv2d forward (models.py:189-240), its dgrad (flipped packed weights), GRU input projections and nn.Linear (KH=KW=1, Hi=Ho=1).
//   x [N][Hi][Wi][ldx] (Cin % 32 == 0, ldx >= Cin); wpk = ocrs_pack_frags(K = KH*KW*Cin ordered (tap, c), M); out [N][Ho][Wo][ldo];
//   gstat (nullable): [2][M] double batch sums of the (rounded) outputs, ACCUMULATED: the caller zeroes it (one fill for all layers of a step, like ocrs_dwpw_fwd).  Outputs rows m in [M, ldo) are written as 0(+0 bias).
int ocrs_conv_igemm(const void* x, int ldx, const void* wpk, void* out, int ldo, const float* bias, int relu, double* gstat, int Cin, int M, int N,
                    int Hi, int Wi, int Ho, int Wo, int KH, int KW, int padh, int padw, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(x && wpk && out && Cin % 32 == 0 && ldx >= Cin && M > 0 && ldo >= M && ldo % 4 == 0 && KH >= 1 && KW >= 1 && KH * KW <= 9);
    if (conv3x3_tile_supported(ldx, ldo, Cin, M, Hi, Wi, Ho, Wo, KH, KW, padh, padw, dtype))  // narrow layers, weights resident in LDS (rec_conv4.hip)
        return conv3x3_tile_launch(x, ldx, wpk, out, ldo, bias, relu, gstat, Cin, M, N, Hi, Wi, st);
    if (conv3x3_rows_supported(ldx, ldo, Cin, M, Hi, Wi, Ho, Wo, KH, KW, padh, padw, dtype))  // whole-row passes, one workgroup per CU (rec_conv3.hip)
        return conv3x3_rows_launch(x, ldx, wpk, out, ldo, bias, relu, gstat, Cin, M, N, Hi, Wi, Ho, Wo, KW, padh, st);
    if (conv3x3_c128_supported(ldx, ldo, Cin, M, Hi, Wi, Ho, Wo, KH, KW, padh, padw, dtype))  // the 128-output-channel 3x3 layers: 128 x 256 block tiles (rec_conv2.hip)
        return conv3x3_c128_launch(x, ldx, wpk, out, ldo, bias, relu, gstat, Cin, N, Hi, Wi, st);
    const bool gemm = Ho == 1 && KH == 1;
    if (dtype == 1)
        return gemm ? launch_igemm<bf16, 1, 128>(x, ldx, wpk, out, ldo, bias, relu, gstat, Cin, M, N, Hi, Wi, Ho, Wo, KH, KW, padh, padw, st)
                    : launch_igemm<bf16, 8, 16>(x, ldx, wpk, out, ldo, bias, relu, gstat, Cin, M, N, Hi, Wi, Ho, Wo, KH, KW, padh, padw, st);
    return gemm ? launch_igemm<float, 1, 128>(x, ldx, wpk, out, ldo, bias, relu, gstat, Cin, M, N, Hi, Wi, Ho, Wo, KH, KW, padh, padw, st)
                : launch_igemm<float, 8, 16>(x, ldx, wpk, out, ldo, bias, relu, gstat, Cin, M, N, Hi, Wi, Ho, Wo, KH, KW, padh, padw, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// The gathered weight gradient in the natural-layout + LDS-transpose-read form of k_wgrad_gemm_x3 (bf16, workspace flush):
//   D[ra][jc] = sum_pos A~[pos][ra] * B[pos * stride + tap(jc) - pad][cb(jc)],   jc = tap * CB + cb.
// k_wgrad_gather<bf16> builds TRANSPOSED [channel][position] tiles with two-byte scatter stores and holds 16 accumulator tiles + two tiles of
// prefetch per thread (256 VGPRs + 127 AGPRs: one block owns a CU, ~50 TFLOP/s -- and on the detection step's side stream such a block keeps a CU
// from the main stream's kernels for 100 us).  Here a chunk of 32 positions is staged as it lies in memory ([position][channel], 16-byte
// vectors; gathered rows outside B are zero) and both MFMA operands come from ds_read_b64_tr_b16; block = 128 x 128 outputs over a contiguous
// range of chunks, next chunk register-prefetched, partial -> the workspace layout of k_wgrad_gather_reduce.
__global__ __launch_bounds__(256, 2) void k_wgrad_gather_tr(const bf16* __restrict__ A, int ldA, int CA, const float* __restrict__ trA, const bf16* __restrict__ B,
                                                            int ldB, int CB, int N, int hA, int wA, int HB, int WB, int stride, int padh, int padw, int KW,
                                                            int ntaps, float* __restrict__ ws, int tiles_a, int chunks_per_block) {
    constexpr int BM = 128, KC = 32, PA = BM + 8;
    __shared__ __attribute__((aligned(16))) bf16 At[KC * PA];
    __shared__ __attribute__((aligned(16))) bf16 Bt[KC * PA];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ta = blockIdx.y % tiles_a, tj = blockIdx.y / tiles_a;
    const int a0 = ta * BM, j0 = tj * BM;
    const int J = ntaps * CB, CA8 = (CA + 7) & ~7;
    const long P = (long)N * hA * wA;
    const long nchunks = (P + KC - 1) / KC;
    const long c_first = (long)blockIdx.x * chunks_per_block, c_end = c_first + chunks_per_block < nchunks ? c_first + chunks_per_block : nchunks;
    // staging map: item f = tid + 256 j (j < 2): row f >> 4 of the chunk, columns (f & 15) * 8 .. + 7 of the block's 128
    const int col8 = (tid & 15) * 8;
    const bool cola = a0 + col8 < CA8, colb = j0 + col8 < J;
    const int tapb = colb ? (j0 + col8) / CB : 0, cb0 = j0 + col8 - tapb * CB;
    const int dyb = tapb / KW - padh, dxb = tapb % KW - padw;
    float sc[8], sh[8], lo[8];
    if (trA && cola) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = a0 + col8 + i < CA ? a0 + col8 + i : 0;
            sc[i] = trA[c];
            sh[i] = trA[CA + c];
            lo[i] = trA[2 * CA + c];
        }
    }
    uint4 ra[2], rb[2];
    unsigned ok = 0;  // bits 0-1: A rows inside P, bits 2-3: B rows inside the gathered tensor
    auto issue = [&](long c) {
        ok = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const long p = c * KC + (tid >> 4) + 16 * j;
            const bool inp = p < P;
            const PixIdx q = decode_pixel(inp ? p : 0, hA, wA);
            const bool oka = inp && cola;
            ra[j] = *reinterpret_cast<const uint4*>(oka ? A + p * ldA + a0 + col8 : A);
            const int Y = q.h * stride + dyb, X = q.w * stride + dxb;
            const bool okb = inp && colb && (unsigned)Y < (unsigned)HB && (unsigned)X < (unsigned)WB;
            rb[j] = *reinterpret_cast<const uint4*>(okb ? B + (((long)q.n * HB + Y) * WB + X) * ldB + cb0 : B);
            ok |= (oka ? 1u : 0u) << j | (okb ? 4u : 0u) << j;
        }
    };
    const int wm = wave & 1, wn = wave >> 1;  // 2 x 2 waves, each 4 x 4 MFMA tiles of 16 x 16
    const int prow = 4 * (lane >> 4) + ((lane & 15) >> 2), pcol = (lane & 3) * 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (c_first < c_end) issue(c_first);
    for (long c = c_first; c < c_end; ++c) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int off = ((tid >> 4) + 16 * j) * PA + col8;
            uint4 va = make_uint4(0, 0, 0, 0);
            if (ok & (1u << j)) {
                va = ra[j];
                if (trA) {
                    float v[8];
                    unpack8(Raw8<bf16>{va}, v);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fmaxf(fmaf(v[i], sc[i], sh[i]), lo[i]);
                    va = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
                }
            }
            *reinterpret_cast<uint4*>(At + off) = va;
            *reinterpret_cast<uint4*>(Bt + off) = (ok & (4u << j)) ? rb[j] : make_uint4(0, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < c_end) issue(c + 1);
        lds_barrier();
        bf16x8 af[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = prow * PA + (wm * 4 + i) * 16 + pcol;
            af[i] = lds_tr8(At + o, At + o + 16 * PA);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = prow * PA + (wn * 4 + j) * 16 + pcol;
            const bf16x8 bfr = lds_tr8(Bt + o, Bt + o + 16 * PA);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr, acc[i][j], 0, 0, 0);
        }
        lds_barrier();
    }
    // partial -> ws[blockIdx.x][jc][ra] (CA8-padded rows), 4 consecutive ra per lane
    float* wb = ws + (long)blockIdx.x * J * CA8;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r0 = a0 + (wm * 4 + i) * 16 + (lane >> 4) * 4, jc = j0 + (wn * 4 + j) * 16 + (lane & 15);
            if (r0 < CA8 && jc < J) {
                const f32x4 v = acc[i][j];
                *reinterpret_cast<float4*>(wb + (long)jc * CA8 + r0) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
}

// bf16 with a workspace: k_wgrad_gather_tr (OCRS_WGRAD_GATHER_TR=0: the transposed-tile kernel everywhere)
static bool wgrad_gather_tr_on(int dtype) {
    static const int on = env_int("OCRS_WGRAD_GATHER_TR", 1);
    return on && dtype == 1;
}
static void wgrad_gather_tr_grid(int CA, int CB, int ntaps, long P, int& gx, int& gy, int& tiles_a, int& cpb) {
    tiles_a = (((CA + 7) & ~7) + 127) / 128;
    gy = tiles_a * ((ntaps * CB + 127) / 128);
    const long nchunks = (P + 31) / 32;
    static const int target = env_int("OCRS_WGRAD_TR_BLOCKS", 1024);  // K splits: >= 8 chunks per block, ~4 blocks per CU in total
    long g = target / gy;
    if (g > nchunks / 8) g = nchunks / 8;
    // every K split writes (and the reducer re-reads) a full ntaps * CB * CA8 partial: keep the workspace around 16 MB
    const long part_bytes = (long)ntaps * CB * ((CA + 7) & ~7) * 4, by_ws = (16L << 20) / part_bytes;
    if (g > by_ws) g = by_ws < 8 ? 8 : by_ws;
    if (g < 1) g = 1;
    cpb = (int)((nchunks + g - 1) / g);
    gx = (int)((nchunks + cpb - 1) / cpb);
}
static void wgrad_gather_grid(int CA, int CB, int ntaps, long P, int dtype, long& gx, int& gy) {
    if (wgrad_gather_tr_on(dtype)) {
        int g, ta, cpb;
        wgrad_gather_tr_grid(CA, CB, ntaps, P, g, gy, ta, cpb);
        gx = g;
        return;
    }
    const int TP = dtype == 1 ? 128 : 64;
    const long ntiles = (P + TP - 1) / TP;
    const int CA8 = (CA + 7) & ~7;
    gy = ((CA8 + 127) / 128) * ((ntaps * CB + 127) / 128);
    static const int target = env_int("OCRS_WGRAD_BLOCKS", 512);  // two 4-wave blocks per CU (round 4: one wave per SIMD cannot keep the MFMA pipe busy -- tools/probes/mfma_issue_probe.hip; 256 -> 512: 0.563 -> 0.496 ms per step, 768: 0.59)  // few flushing blocks, each loops over many tiles
    gx = ntiles / 4;
    if (gx < 1) gx = 1;
    long cap = target / gy;
    if (cap < 8) cap = 8;
    if (gx > cap) gx = cap;
    if (gx >= 8) gx &= ~7L;
}

// workspace (floats) for the atomic-free flush of ocrs_wgrad_gather
long ocrs_wgrad_gather_ws_floats(int CA, int CB, int ntaps, long P, int dtype) {
    long gx;
    int gy;
    wgrad_gather_grid(CA, CB, ntaps, P, dtype, gx, gy);
    return gx * ntaps * CB * ((CA + 7) & ~7);
}

// Weight gradient by gathering: dW[(ra*CB + cb)*KH*KW + tap] += sum_pos A~[pos][ra] * B[pos*stride + tap - pad][cb].
//   Conv2d: A = dz [N][Ho][Wo][Cout], B = x; Linear / GRU: KH=KW=1, hA=1.  trA (nullable) = load transform of A.
//   ws: workspace of ocrs_wgrad_gather_ws_floats() floats (deterministic two-stage reduction) or null (float atomics into dW).
int ocrs_wgrad_gather(const void* A, int ldA, int CA, const float* trA, const void* B, int ldB, int CB, float* dW, float* ws, int N, int hA,
                      int wA, int HB, int WB, int stride, int padh, int padw, int KH, int KW, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(A && B && dW && CB % 8 == 0 && ldA % 8 == 0 && ldB % 8 == 0 && ldA >= ((CA + 7) & ~7) && ldB >= CB);
    const long P = (long)N * hA * wA;
    OCRS_CHECK_ARG(P < (1L << 31));
    long gx;
    int gy;
    wgrad_gather_grid(CA, CB, KH * KW, P, dtype, gx, gy);
    static DevOnce attr_set;
    if (attr_set.need()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_gather<float, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) !=
                hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_gather<bf16, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) !=
                hipSuccess)
            return OCRS_ERR_HIP;
        attr_set.done();
    }
    if (dtype == 1 && ws && wgrad_gather_tr_on(dtype)) {
        int g, gy2, ta, cpb;
        wgrad_gather_tr_grid(CA, CB, KH * KW, P, g, gy2, ta, cpb);
        hipLaunchKernelGGL(k_wgrad_gather_tr, dim3(g, gy2), dim3(256), 0, st, (const bf16*)A, ldA, CA, trA, (const bf16*)B, ldB, CB, N, hA, wA, HB, WB, stride,
                           padh, padw, KW, KH * KW, ws, ta, cpb);
    } else if (dtype == 1)
        hipLaunchKernelGGL((k_wgrad_gather<bf16, 128>), dim3((int)gx, gy), dim3(256), 2 * 128 * 136 * 2, st, (const bf16*)A, ldA, CA, trA, (const bf16*)B,
                           ldB, CB, dW, N, hA, wA, HB, WB, stride, padh, padw, KH, KW, ws);
    else
        hipLaunchKernelGGL((k_wgrad_gather<float, 64>), dim3((int)gx, gy), dim3(256), 2 * 128 * 68 * 4, st, (const float*)A, ldA, CA, trA,
                           (const float*)B, ldB, CB, dW, N, hA, wA, HB, WB, stride, padh, padw, KH, KW, ws);
    if (ws) {
        const int CA8 = (CA + 7) & ~7;
        const long n = (long)KH * KW * CB * CA8;
        hipLaunchKernelGGL(k_wgrad_gather_reduce, dim3((int)((n + 31) / 32)), dim3(256), 0, st, ws, (int)gx, CA, CA8, CB,
                           KH * KW, dW);
    }
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Split-bf16 ("bf16x3") weight-gradient GEMM for fp32 operands:  dW[ra][cb] += sum_p A[p][ra] * B[p][cb]   (K = P rows).
// The GRU / Linear weight gradients of the CRNN are fp32 (the reference keeps the GRU in fp32 under autocast, models.py:264-266) and
// were the largest item of the CRNN step on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32 is 1/16 of the bf16 rate).  In THROUGHPUT mode
// they run here as  a*b ~ ah*bh + ah*bl + al*bh  with  ah = bf16(a), al = bf16(a - ah)  (dropped term and residuals <= ~1.1e-5 relative
// per product, fp32 accumulation: the same order as the summation-order noise of an fp32 dot product of K = 25856 terms); parity
// (fp32) mode keeps the exact-fp32 kernel.  Operands are staged in natural [row][channel] order as hi / lo bf16 planes and read with
// the LDS transpose read (K = rows).  Block = 128 x 128 outputs, a contiguous range of 32-row chunks; partials -> workspace in the
// layout of k_wgrad_gather_reduce (ntaps = 1).
__global__ __launch_bounds__(256) void k_wgrad_gemm_x3(const float* __restrict__ A, int ldA, int CA, const float* __restrict__ B, int ldB, int CB, long P,
                                                       float* __restrict__ ws, int tiles_a, int chunks_per_block) {
    constexpr int BM = 128, BN = 128, KC = 32, PA = BM + 8;  // pitch (elements) of the bf16 planes
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* Ah = reinterpret_cast<bf16*>(smem);  // [KC][PA]
    bf16* Al = Ah + KC * PA;
    bf16* Bh = Al + KC * PA;
    bf16* Bl = Bh + KC * PA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ta = blockIdx.y % tiles_a, tb = blockIdx.y / tiles_a;
    const int a0 = ta * BM, b0 = tb * BN;
    const long nchunks = (P + KC - 1) / KC;
    const long c_first = (long)blockIdx.x * chunks_per_block, c_end = c_first + chunks_per_block < nchunks ? c_first + chunks_per_block : nchunks;
    // staging map: 4 float4 of A and 4 of B per thread and chunk: f = tid + 256*j -> row f/32, columns (f%32)*4..+3
    float4 ra[4], rb[4];
    auto issue = [&](long c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + j * 256, row = f >> 5, col = (f & 31) * 4;
            const long p = c * KC + row;
            const bool oka = p < P && a0 + col < CA, okb = p < P && b0 + col < CB;  // CB a multiple of 4, A rows padded to one (checked by the host)
            ra[j] = oka ? *reinterpret_cast<const float4*>(A + p * ldA + a0 + col) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[j] = okb ? *reinterpret_cast<const float4*>(B + p * ldB + b0 + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto split_store = [&](const float4& v, bf16* hi, bf16* lo, int off) {
#ifdef X3_FLOOR  // (measurement build: what the kernel would cost if its operands arrived pre-split -- one v_perm per pair instead of the split)
        const unsigned b0 = __float_as_uint(v.x), b1 = __float_as_uint(v.y), b2 = __float_as_uint(v.z), b3 = __float_as_uint(v.w);
        *reinterpret_cast<uint2*>(hi + off) = make_uint2(__builtin_amdgcn_perm(b1, b0, 0x07060302u), __builtin_amdgcn_perm(b3, b2, 0x07060302u));
        *reinterpret_cast<uint2*>(lo + off) = make_uint2(__builtin_amdgcn_perm(b1, b0, 0x05040100u), __builtin_amdgcn_perm(b3, b2, 0x05040100u));
#else
        const float x[4] = {v.x, v.y, v.z, v.w};
        float h[4], l[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h[i] = Elem<bf16>::round(x[i]);
            l[i] = x[i] - h[i];
        }
        *reinterpret_cast<uint2*>(hi + off) = make_uint2(pack2bf(h[0], h[1]), pack2bf(h[2], h[3]));
        *reinterpret_cast<uint2*>(lo + off) = make_uint2(pack2bf(l[0], l[1]), pack2bf(l[2], l[3]));
#endif
    };
    const int wm = wave & 1, wn = wave >> 1;  // 2 x 2 waves, each 4 x 4 MFMA tiles of 16 x 16
    const int prow = 4 * (lane >> 4) + ((lane & 15) >> 2), pcol = (lane & 3) * 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (c_first < c_end) issue(c_first);
    for (long c = c_first; c < c_end; ++c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + j * 256, off = (f >> 5) * PA + (f & 31) * 4;
            split_store(ra[j], Ah, Al, off);
            split_store(rb[j], Bh, Bl, off);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < c_end) issue(c + 1);
        lds_barrier();
        bf16x8 ah[4], al[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = prow * PA + (wm * 4 + i) * 16 + pcol;
            ah[i] = lds_tr8(Ah + o, Ah + o + 16 * PA);
            al[i] = lds_tr8(Al + o, Al + o + 16 * PA);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = prow * PA + (wn * 4 + j) * 16 + pcol;
            const bf16x8 bh = lds_tr8(Bh + o, Bh + o + 16 * PA), bl = lds_tr8(Bl + o, Bl + o + 16 * PA);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh, acc[i][j], 0, 0, 0);
            }
        }
        lds_barrier();
    }
    // partial -> ws[blockIdx.x][cb][ra] (CA8-padded rows), 4 consecutive ra per lane
    const int CA8 = (CA + 7) & ~7;
    float* wb = ws + (long)blockIdx.x * CB * CA8;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r0 = a0 + (wm * 4 + i) * 16 + (lane >> 4) * 4, cb = b0 + (wn * 4 + j) * 16 + (lane & 15);
            if (r0 < CA8 && cb < CB) {
                const f32x4 v = acc[i][j];
                *reinterpret_cast<float4*>(wb + (long)cb * CA8 + r0) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
}
static void wgrad_x3_grid(int CA, int CB, long P, int& gx, int& gy, int& tiles_a, int& cpb) {
    tiles_a = (CA + 127) / 128;
    gy = tiles_a * ((CB + 127) / 128);
    const long nchunks = (P + 31) / 32;
    static const int target = env_int("OCRS_WGRAD_X3_BLOCKS", 512);  // (two resident 4-wave workgroups per CU; 384 / 768 / 1024 measured: no better)
    long g = target / gy;
    if (g < 1) g = 1;
    if (g > nchunks) g = nchunks;
    cpb = (int)((nchunks + g - 1) / g);
    gx = (int)((nchunks + cpb - 1) / cpb);
}

static int wgrad3x3_gx(int Cin, int N, int H, int W) {
    const int ntiles = N * ((W + 15) / 16) * ((H + 7) / 8);
    const int gy = Cin / 32;
    static const int target = env_int("OCRS_WGRAD_BLOCKS", 512);  // two 4-wave blocks per CU (round 4: one wave per SIMD cannot keep the MFMA pipe busy -- tools/probes/mfma_issue_probe.hip; 256 -> 512: 0.563 -> 0.496 ms per step, 768: 0.59)
    long gx = ntiles / 4;
    if (gx < 1) gx = 1;
    long cap = target / gy;
    if (cap < 8) cap = 8;
    if (gx > cap) gx = cap;
    if (gx >= 8) gx &= ~7L;
    return (int)gx;
}

__global__ __launch_bounds__(256) void k_wgrad3x3_reduce(const float* __restrict__ ws, int gx, int Cout, int Cin, float* __restrict__ dW) {
    const long n = 9L * Cin * Cout;  // grid ceil(n / 32): see k_wgrad_gather_reduce
    const long i = (long)blockIdx.x * 32 + (threadIdx.x & 31);
    float s;
    if (!det_column_sum(ws, gx, n, i, s)) return;
    const int co = (int)(i % Cout), ci = (int)((i / Cout) % Cin), tap = (int)(i / ((long)Cout * Cin));
    dW[((long)co * Cin + ci) * 9 + tap] += s;
}

// Split-bf16 weight-gradient GEMM for fp32 operands (throughput mode of the GRU / Linear weight gradients):
//   dW [CA][CB] += A^T B,  A [P][ldA] (first CA columns), B [P][ldB] (first CB columns), CA, CB, ldA, ldB multiples of 4.
//   ws: ocrs_wgrad_gemm_x3_ws_floats() floats.  Products carry <= ~1.1e-5 relative error (see k_wgrad_gemm_x3), fp32 accumulation.
long ocrs_wgrad_gemm_x3_ws_floats(int CA, int CB, long P) {
    int gx, gy, ta, cpb;
    wgrad_x3_grid(CA, CB, P, gx, gy, ta, cpb);
    return (long)gx * CB * ((CA + 7) & ~7);
}
int ocrs_wgrad_gemm_x3(const float* A, int ldA, int CA, const float* B, int ldB, int CB, float* dW, float* ws, long P, hipStream_t st) {
    // (a ragged CA -- the class count of the output Linear -- is fine as long as the rows of A extend to the next multiple of 4: the extra
    // columns are loaded, their partials land in the CA8-padded workspace rows and the reduce skips them)
    OCRS_CHECK_ARG(A && B && dW && ws && P > 0 && CB % 4 == 0 && ldA % 4 == 0 && ldB % 4 == 0 && ldA >= ((CA + 3) & ~3) && ldB >= CB);
    OCRS_CHECK_ARG((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0);
    int gx, gy, ta, cpb;
    wgrad_x3_grid(CA, CB, P, gx, gy, ta, cpb);
    hipLaunchKernelGGL(k_wgrad_gemm_x3, dim3(gx, gy), dim3(256), 4 * 32 * 136 * 2, st, A, ldA, CA, B, ldB, CB, P, ws, ta, cpb);
    const int CA8 = (CA + 7) & ~7;
    const long n = (long)CB * CA8;
    hipLaunchKernelGGL(k_wgrad_gather_reduce, dim3((int)((n + 31) / 32)), dim3(256), 0, st, ws, gx, CA, CA8, CB, 1, dW);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// Split-bf16 GEMM for fp32 operands: out [P][ldo] (first M columns) = X [P][ldx] (first K columns) * W (+ bias [M]);
//   km = 0: W[m][k] at Wm[m * ldw + k];  km = 1: W[k][m] at Wm[k * ldw + m].   K % 32 == 0, M % 4 == 0, 16-byte aligned rows.
int ocrs_gemm_x3(const float* X, int ldx, int K, const float* Wm, int ldw, int km, const float* bias, float* out, int ldo, int M, long P, int Kw,
                 hipStream_t st) {
    // Kw (0: K): the k extent Wm really has -- X columns [Kw, K) meet zero weights (a K padded up to the 32-column chunk, e.g. the
    // zero-padded class columns of the output layer's gradient); km = 0 also takes a ragged M (rows of Wm), ldo >= round_up(M, 4)
    if (Kw <= 0) Kw = K;
    OCRS_CHECK_ARG(X && Wm && out && P > 0 && K > 0 && K % 32 == 0 && Kw <= K && ldx % 4 == 0 && ldw % 4 == 0 && ldo % 4 == 0 && ldx >= K &&
                   ldo >= ((M + 3) & ~3) && (km ? (M % 4 == 0) : (Kw % 4 == 0)));
    OCRS_CHECK_ARG(((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Wm) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0);
    const dim3 grid((unsigned)((P + 127) / 128), (unsigned)((M + 127) / 128));
    if (km)
        hipLaunchKernelGGL(k_gemm_x3<true>, grid, dim3(256), (2 * 128 * 40 + 2 * 32 * 136) * 2, st, X, ldx, Wm, ldw, bias, out, ldo, K, M, P, Kw);
    else
        hipLaunchKernelGGL(k_gemm_x3<false>, grid, dim3(256), (4 * 128 * 40) * 2, st, X, ldx, Wm, ldw, bias, out, ldo, K, M, P, Kw);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// workspace (floats) for the atomic-free flush of ocrs_conv3x3_wgrad
long ocrs_conv3x3_wgrad_ws_floats(int Cout, int Cin, int N, int H, int W) { return (long)wgrad3x3_gx(Cin, N, H, W) * 9 * Cin * Cout; }

// Conv2d 3x3 / stride 1 / pad 1 weight gradient (Cout <= 128, Cin % 32 == 0): dW [Cout][Cin][3][3] += ...
// ws: workspace of ocrs_conv3x3_wgrad_ws_floats() floats (deterministic two-stage reduction), or null (float atomics).
int ocrs_conv3x3_wgrad(const void* dz, int Cout, const void* x, int Cin, float* dW, float* ws, int N, int H, int W, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(dz && x && dW && Cout % 8 == 0 && Cout <= 128 && Cin % 32 == 0);
    const int gy = Cin / 32;
    const long gx = wgrad3x3_gx(Cin, N, H, W);  // few flushing blocks: each loops over many tiles
    static DevOnce attr_set;
    if (attr_set.need()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_wgrad<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024) !=
                hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_wgrad<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024) !=
                hipSuccess)
            return OCRS_ERR_HIP;
        attr_set.done();
    }
    static const int use_tr = env_int("OCRS_WGRAD3X3_TR", 1);
    if (dtype == 1 && use_tr) {
        static bool attr2 = false;
        if (!attr2) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_wgrad_tr<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) !=
                    hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_wgrad_tr<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) !=
                    hipSuccess)
                return OCRS_ERR_HIP;
            attr2 = true;
        }
        if (Cout > 64)
            hipLaunchKernelGGL(k_conv3x3_wgrad_tr<128>, dim3((int)gx, gy), dim3(256), (128 * 136 + 180 * 40) * 2, st, (const bf16*)dz, Cout, (const bf16*)x,
                               Cin, dW, N, H, W, ws);
        else
            hipLaunchKernelGGL(k_conv3x3_wgrad_tr<64>, dim3((int)gx, gy), dim3(256), (128 * 72 + 180 * 40) * 2, st, (const bf16*)dz, Cout, (const bf16*)x,
                               Cin, dW, N, H, W, ws);
    } else if (dtype == 1)
        hipLaunchKernelGGL(k_conv3x3_wgrad<bf16>, dim3((int)gx, gy), dim3(256), (128 * 136 + 96 * 168) * 2, st, (const bf16*)dz, Cout, (const bf16*)x, Cin,
                           dW, N, H, W, ws);
    else
        hipLaunchKernelGGL(k_conv3x3_wgrad<float>, dim3((int)gx, gy), dim3(256), (128 * 132 + 96 * 164) * 4, st, (const float*)dz, Cout,
                           (const float*)x, Cin, dW, N, H, W, ws);
    if (ws) hipLaunchKernelGGL(k_wgrad3x3_reduce, dim3((9 * Cin * Cout + 31) / 32), dim3(256), 0, st, ws, (int)gx, Cout, Cin, dW);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// Conv2d(1,32,3,p1)+ReLU+MaxPool2d(2) fused (models.py:180-187): img fp32 (N,1,H,W) -> out [N][H/2][W/2][32].
int ocrs_conv0_fwd(const float* img, const float* w, const float* bias, void* out, int N, int H, int W, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(img && w && bias && out && H >= 2 && W >= 2);  // (odd sizes: floor-mode pooling, the last row / column is in no window)
    const int grid = ew_grid((long)N * (H / 2) * (W / 2) * 4);
    if (dtype == 1)
        hipLaunchKernelGGL(k_conv0_fwd<bf16>, dim3(grid), dim3(256), 0, st, img, w, bias, (bf16*)out, N, H, W);
    else
        hipLaunchKernelGGL(k_conv0_fwd<float>, dim3(grid), dim3(256), 0, st, img, w, bias, (float*)out, N, H, W);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}
int conv0_bwd_mm_launch(const float* img, const float* w, const float* bias, const void* g, float* dW, float* db, int N, int H, int W, int dtype,
                        hipStream_t st);  // rec_conv0.hip: the matrix-core form (bf16 gradient, even H and W)
int ocrs_conv0_bwd(const float* img, const float* w, const float* bias, const void* g, float* dW, float* db, int N, int H, int W, int dtype,
                   hipStream_t st) {
    OCRS_CHECK_ARG(img && w && bias && g && dW && db);
    if (conv0_bwd_mm_launch(img, w, bias, g, dW, db, N, H, W, dtype, st)) {
        OCRS_LAUNCH_CHECK();
        return OCRS_OK;
    }
    const int grid = ew_grid((long)N * (H / 2) * (W / 2) * 4);
    if (dtype == 1)
        hipLaunchKernelGGL(k_conv0_bwd<bf16>, dim3(grid), dim3(256), 0, st, img, w, bias, (const bf16*)g, dW, db, N, H, W);
    else
        hipLaunchKernelGGL(k_conv0_bwd<float>, dim3(grid), dim3(256), 0, st, img, w, bias, (const float*)g, dW, db, N, H, W);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

#define DT_DISPATCH(KERNEL, GRID, SMEM, ...)                                                     \
    if (dtype == 1)                                                                              \
        hipLaunchKernelGGL(KERNEL<bf16>, dim3(GRID), dim3(256), SMEM, st, __VA_ARGS__);          \
    else                                                                                         \
        hipLaunchKernelGGL(KERNEL<float>, dim3(GRID), dim3(256), SMEM, st, __VA_ARGS__);

// BN+ReLU(+MaxPool PHxPW) forward on a pre-BN tensor z (models.py:197-199, 214-216, 231-233); tr = [3][C] load transform.
static bool window_fast() {  // OCRS_REC_WINDOW_FAST=0: the generic (run-time window shape) kernels everywhere
    static const int on = env_int("OCRS_REC_WINDOW_FAST", 1);
    return on != 0;
}
int ocrs_act_pool_fwd(const void* z, const float* tr, void* out, int C, int N, int H, int W, int PH, int PW, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(z && tr && out && C % 8 == 0 && PH * PW <= 4 && PH >= 1 && PW >= 1);
    const int grid = ew_grid((long)N * (H / PH) * (W / PW) * (C / 8));
#define APF(T_, PH_, PW_)                                                                                                          \
    if (PH == PH_ && PW == PW_) {                                                                                                  \
        hipLaunchKernelGGL((k_act_pool_fwd_t<T_, PH_, PW_>), dim3(grid), dim3(256), 0, st, (const T_*)z, tr, (T_*)out, C, N, H, W); \
        OCRS_LAUNCH_CHECK();                                                                                                       \
        return OCRS_OK;                                                                                                            \
    }
    if (window_fast() && 256 % (C / 8) == 0) {
        if (dtype == 1) { APF(bf16, 2, 2) APF(bf16, 2, 1) } else { APF(float, 2, 2) APF(float, 2, 1) }
    }
#undef APF
    if (dtype == 1)
        hipLaunchKernelGGL(k_act_pool_fwd<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)z, tr, (bf16*)out, C, N, H, W, PH, PW);
    else
        hipLaunchKernelGGL(k_act_pool_fwd<float>, dim3(grid), dim3(256), 0, st, (const float*)z, tr, (float*)out, C, N, H, W, PH, PW);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}
// backward reductions (gsum [2][C] double, ACCUMULATED: the caller zeroes it) and dz materialisation
int ocrs_rec_bn_reduce(const void* g, const void* z, const float* bn, const float* saved, double* gsum, int C, int N, int H, int W, int PH, int PW,
                       int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(g && z && bn && saved && gsum && C % 8 == 0 && 256 % (C / 8) == 0 && PH * PW <= 4);
    int grid = ew_grid((long)N * (H / PH) * (W / PW) * (C / 8));
    static const int bpc = env_int("OCRS_REC_REDUCE_BPC", 4);  // every block ends in 2 C same-address fp64 atomics (~11 ns each, serial per address)
    if (grid > bpc * kNumCU) grid = bpc * kNumCU;
#define RBR(T_, PH_, PW_)                                                                                                                      \
    if (PH == PH_ && PW == PW_) {                                                                                                              \
        hipLaunchKernelGGL((k_rec_bn_reduce_t<T_, PH_, PW_>), dim3(grid), dim3(256), 0, st, (const T_*)g, (const T_*)z, bn, saved, gsum, C, N, H, W); \
        OCRS_LAUNCH_CHECK();                                                                                                                   \
        return OCRS_OK;                                                                                                                        \
    }
    if (window_fast() && H % PH == 0 && W % PW == 0) {
        if (dtype == 1) { RBR(bf16, 2, 2) RBR(bf16, 2, 1) RBR(bf16, 1, 1) } else { RBR(float, 2, 2) RBR(float, 2, 1) RBR(float, 1, 1) }
    }
#undef RBR
    if (dtype == 1)
        hipLaunchKernelGGL(k_rec_bn_reduce<bf16>, dim3(grid), dim3(256), 2 * C * sizeof(float), st, (const bf16*)g, (const bf16*)z, bn, saved, gsum, C,
                           N, H, W, PH, PW);
    else
        hipLaunchKernelGGL(k_rec_bn_reduce<float>, dim3(grid), dim3(256), 2 * C * sizeof(float), st, (const float*)g, (const float*)z, bn, saved, gsum,
                           C, N, H, W, PH, PW);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}
int ocrs_col_sum(const void* a, int ld, int C, float* out, long rows, int dtype, hipStream_t st);
int ocrs_dz_apply(const void* g, const void* z, const float* bn, const float* coef, void* dz, int C, int N, int H, int W, int PH, int PW, int dtype,
                  float* dsum, hipStream_t st) {
    OCRS_CHECK_ARG(g && z && bn && coef && dz && C % 8 == 0 && PH * PW <= 4);
    int grid = ew_grid((long)N * ((H + PH - 1) / PH) * ((W + PW - 1) / PW) * (C / 8));
    const bool fast = window_fast() && H % PH == 0 && W % PW == 0 && 256 % (C / 8) == 0;
    static const int ds_bpc = env_int("OCRS_DZ_DSUM_BPC", 1);
    float* dsum_ws = nullptr;
    if (dsum && fast) {  // 1024-thread blocks, every one ending in C same-address atomics
        grid = (grid + 3) / 4;
        if (grid > ds_bpc * kNumCU) grid = ds_bpc * kNumCU;
        if ((PH == 2 && (PW == 2 || PW == 1)) || (PH == 1 && PW == 1)) dsum_ws = rec_defer_partials(grid, C, dsum);  // (the shapes DZA covers)
    }
#define DZA(T_, PH_, PW_)                                                                                                                    \
    if (PH == PH_ && PW == PW_) {                                                                                                            \
        if (dsum)                                                                                                                            \
            hipLaunchKernelGGL((k_dz_apply_t<T_, PH_, PW_, 1024>), dim3(grid), dim3(1024), 0, st, (const T_*)g, (const T_*)z, bn, coef, (T_*)dz, C, N, H, W, \
                               dsum, dsum_ws);                                                                                               \
        else                                                                                                                                 \
            hipLaunchKernelGGL((k_dz_apply_t<T_, PH_, PW_, 256>), dim3(grid), dim3(256), 0, st, (const T_*)g, (const T_*)z, bn, coef, (T_*)dz, C, N, H, W, \
                               dsum, dsum_ws);                                                                                               \
        OCRS_LAUNCH_CHECK();                                                                                                                 \
        return OCRS_OK;                                                                                                                      \
    }
    if (fast) {
        if (dtype == 1) { DZA(bf16, 2, 2) DZA(bf16, 2, 1) DZA(bf16, 1, 1) } else { DZA(float, 2, 2) DZA(float, 2, 1) DZA(float, 1, 1) }
    }
#undef DZA
    if (dtype == 1)
        hipLaunchKernelGGL(k_dz_apply<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)g, (const bf16*)z, bn, coef, (bf16*)dz, C, N, H, W, PH, PW);
    else
        hipLaunchKernelGGL(k_dz_apply<float>, dim3(grid), dim3(256), 0, st, (const float*)g, (const float*)z, bn, coef, (float*)dz, C, N, H, W, PH, PW);
    OCRS_LAUNCH_CHECK();
    if (dsum) return ocrs_col_sum(dz, C, C, dsum, (long)N * H * W, dtype, st);  // (generic window shapes: a separate pass over dz)
    return OCRS_OK;
}

// BatchNorm2d + AvgPool2d((4,1)) on H=5 + permute to (T=W, N, C) fp32 (models.py:241-242, 259-262), and its backward pieces.
int ocrs_avgpool_fwd(const void* z, const float* tr, float* seq, int C, int N, int H, int W, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(z && tr && seq && C % 8 == 0 && H >= 4);
    const int grid = ew_grid((long)N * W * (C / 8));
    if (dtype == 1)
        hipLaunchKernelGGL(k_avgpool_fwd<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)z, tr, seq, C, N, H, W);
    else
        hipLaunchKernelGGL(k_avgpool_fwd<float>, dim3(grid), dim3(256), 0, st, (const float*)z, tr, seq, C, N, H, W);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}
int ocrs_avgpool_bn_reduce(const float* gseq, const void* z, const float* saved, double* gsum, int C, int N, int H, int W, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(gseq && z && saved && gsum && C % 8 == 0 && 256 % (C / 8) == 0);
    int grid = ew_grid((long)N * W * (C / 8));
    if (grid > 4 * kNumCU) grid = 4 * kNumCU;  // (ends in 2 C same-address fp64 atomics per block: see ocrs_rec_bn_reduce)
    if (dtype == 1)
        hipLaunchKernelGGL(k_avgpool_bn_reduce<bf16>, dim3(grid), dim3(256), 2 * C * sizeof(float), st, gseq, (const bf16*)z, saved, gsum, C, N, H, W);
    else
        hipLaunchKernelGGL(k_avgpool_bn_reduce<float>, dim3(grid), dim3(256), 2 * C * sizeof(float), st, gseq, (const float*)z, saved, gsum, C, N, H, W);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}
int ocrs_avgpool_dz(const float* gseq, const void* z, const float* coef, void* dz, int C, int N, int H, int W, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(gseq && z && coef && dz && C % 8 == 0);
    const int grid = ew_grid((long)N * H * W * (C / 8));
    if (dtype == 1)
        hipLaunchKernelGGL(k_avgpool_dz<bf16>, dim3(grid), dim3(256), 0, st, gseq, (const bf16*)z, coef, (bf16*)dz, C, N, H, W);
    else
        hipLaunchKernelGGL(k_avgpool_dz<float>, dim3(grid), dim3(256), 0, st, gseq, (const float*)z, coef, (float*)dz, C, N, H, W);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// column sums (bias gradients): out[c] += sum_rows a[row][c]
int ocrs_col_sum(const void* a, int ld, int C, float* out, long rows, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(a && out && C > 0 && ld >= C && rows > 0);
    const int esz = dtype == 1 ? 2 : 4;
    if (C % 4 == 0 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(a) & (4 * esz - 1)) == 0) {
        long g4 = (rows + 15) / 16;  // >= 4 rows per thread before another block is worth its C trailing atomics
        if (g4 > 2 * kNumCU) g4 = 2 * kNumCU;
        if (g4 < 1) g4 = 1;
        float* ws = rec_defer_partials((int)g4, C, out);  // (deferring: per-block partials + one queued fixed-order reduce instead of float atomics)
        if (dtype == 1)
            hipLaunchKernelGGL(k_col_sum4<bf16>, dim3((int)g4), dim3(256), 4 * C * sizeof(float), st, (const bf16*)a, ld, C, out, rows, ws);
        else
            hipLaunchKernelGGL(k_col_sum4<float>, dim3((int)g4), dim3(256), 4 * C * sizeof(float), st, (const float*)a, ld, C, out, rows, ws);
        OCRS_LAUNCH_CHECK();
        return OCRS_OK;
    }
    long g = (rows + 1) / 2;
    if (g > 1024) g = 1024;
    float* ws = rec_defer_partials((int)g, C, out);
    if (dtype == 1)
        hipLaunchKernelGGL(k_col_sum<bf16>, dim3((int)g), dim3(256), 2 * C * sizeof(float), st, (const bf16*)a, ld, C, out, rows, ws);
    else
        hipLaunchKernelGGL(k_col_sum<float>, dim3((int)g), dim3(256), 2 * C * sizeof(float), st, (const float*)a, ld, C, out, rows, ws);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

}  // extern "C"
