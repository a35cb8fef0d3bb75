// 3x3 convolution of the CRNN's 128-output-channel layers (ocrs_models/models.py:179-236: conv.6 / 8 / 12 / 15 forward and the dgrad of
// 8 / 12 / 15) as an implicit GEMM with a 128 x 256 block tile (gfx950, bf16).  Same contract as k_conv_igemm (rec_conv.hip).
//
// k_conv_igemm gives every wave 2 x 8 accumulator tiles: each 16x16x32 MFMA needs 0.5 KB of fresh LDS operand, and with 16 waves per CU
// the LDS read time of a tap (128 KB at 128 B/clk) equals its MFMA time -- the kernel sits at 27-32 % MFMA-busy, 21 % of the bf16 peak.
// Here a wave owns 64 output channels x 128 pixels (4 x 8 tiles, 128 accumulator registers): 12 operand fragments feed 32 MFMAs
// (0.375 KB each, 96 B/clk per CU at full MFMA rate), the weight fragments of a (tap, 32-channel chunk) step come from LDS -- fetched
// once per block instead of once per wave -- and everything global is software-pipelined one step ahead:
//   step = (32-channel chunk cc, tap): A = 8 KB of packed weight fragments in wbuf[step & 1], B = the chunk's input tile + halo in
//   xbuf[cc & 1], read at the tap's pixel offset (no im2col);
//   the weights of step g + 2 and 1/8 of the next chunk's input tile are loaded (registers) at the top of step g and committed to the other
//   LDS buffers at the end of step g + 1 -- two steps of MFMA work cover the L2 / HBM round trip; ONE barrier per step.
// Tile = 16 rows x 16 pixels: one image at H >= 16, or two images x 8 rows (H = 8) -- a 16-pixel row is one MFMA N tile.
// Waves: 2 (M halves) x 2 (row halves).  Two blocks per CU (73 KB of LDS, <= 256 registers).
#include "det_common.h"

#ifndef OCRS_C128_UNROLL
#define OCRS_C128_UNROLL 1   // tap loop unroll factor
#endif
#ifndef OCRS_C128_SPLIT
#define OCRS_C128_SPLIT 4    // N tiles whose MFMAs are issued before the LDS commit of the prefetched data (8 = all; 4: 149 -> 145 us)
#endif

namespace {
constexpr int C2_TW = 16, C2_ROWS = 16, C2_HW = C2_TW + 2;  // tile width, stacked rows, halo width
constexpr int C2_PITCH = Mma<bf16>::LDS_PITCH;              // 40 bf16 = 80 B per staged pixel
constexpr int C2_MAXHP = 2 * 10 * C2_HW;                    // staged pixels: NI * (TH + 2) * 18 -- 324 (1 x 16 rows) or 360 (2 x 8 rows)
constexpr int C2_XBYTES = C2_MAXHP * C2_PITCH * 2;          // 28800
constexpr int C2_WBYTES = 8 * 64 * 16;                      // 8 M tiles x 64 lanes x 16 B
constexpr int C2_SMEM = 2 * C2_XBYTES + 2 * C2_WBYTES + 2 * 2 * 128 * 4;
}  // namespace

template <int TH /* rows per image in the tile: 16 (NI = 1) or 8 (NI = 2) */>
__global__ __launch_bounds__(256, 2) void k_conv3x3_c128(const bf16* __restrict__ x, int ldx, const void* __restrict__ wpk, bf16* __restrict__ out, int ldo,
                                                         const float* __restrict__ bias, int relu, double* __restrict__ gstat, int Cin, int N, int H, int W) {
    constexpr int NI = C2_ROWS / TH, HH = TH + 2, HP = NI * HH * C2_HW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* xbuf = reinterpret_cast<bf16*>(smem);                                  // [2][HP][PITCH]
    uint4* wbuf = reinterpret_cast<uint4*>(smem + 2 * C2_XBYTES);                // [2][8][64]
    float* s_stat = reinterpret_cast<float*>(smem + 2 * C2_XBYTES + 2 * C2_WBYTES);  // [2 row halves][2][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;  // M half (tiles 4 wm ..), row half (rows 8 wn ..)
    const int l15 = lane & 15, kq = lane >> 4;
    const int ncc = Cin / 32, nsteps = ncc * 9;
    const int tiles_x = (W + C2_TW - 1) / C2_TW, tiles_y = (H + TH - 1) / TH, tpi = tiles_x * tiles_y;
    const int ngrp = (N + NI - 1) / NI;
    const long ntiles = (long)ngrp * tpi;
    if (gstat) {
        for (int i = tid; i < 512; i += 256) s_stat[i] = 0.f;
        __syncthreads();
    }

    // staging roles.  Input: item = (staged pixel, 8-channel group); the HP * 4 items of a chunk are spread over taps 0..7 of the chunk before it.
    constexpr int XI = (HP * 4 + 7) / 8;            // items per step (<= 256: one per thread)
    static_assert(XI <= 256, "one staged input item per thread and step");
    auto tile_of = [&](long t, int& n0, int& h0, int& w0) {
        const int g = (int)(t / tpi), r = (int)(t - (long)g * tpi);
        n0 = g * NI;
        h0 = (r / tiles_x) * TH;
        w0 = (r % tiles_x) * C2_TW;
    };
    // raw vector of input item `it` of chunk cc of the tile at (n0, h0, w0) (zero outside the image / batch)
    auto load_x = [&](int it, int cc, int n0, int h0, int w0) -> uint4 {
        const int hp = it >> 2, g8 = it & 3;
        const int img = hp / (HH * C2_HW), rr = hp - img * (HH * C2_HW), hy = rr / C2_HW, hx = rr - hy * C2_HW;
        const int n = n0 + img, h = h0 + hy - 1, w = w0 + hx - 1;
        if (it < HP * 4 && n < N && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W)
            return *reinterpret_cast<const uint4*>(x + (((long)n * H + h) * W + w) * ldx + cc * 32 + g8 * 8);
        return make_uint4(0, 0, 0, 0);
    };
    auto store_x = [&](int it, int buf, const uint4& v) {
        if (it < HP * 4) *reinterpret_cast<uint4*>(xbuf + (long)buf * (C2_XBYTES / 2) + (it >> 2) * C2_PITCH + (it & 3) * 8) = v;
    };
    // weights of step (cc, tap): fragments kc = tap * ncc + cc, M tiles 0..7 -> 512 uint4, two per thread
    auto load_w = [&](int cc, int tap, uint4& a, uint4& b) {
        const uint4* src = reinterpret_cast<const uint4*>(wpk) + ((long)(tap * ncc + cc) * 8) * 64;
        a = src[tid];
        b = src[tid + 256];
    };

    TileSched ts(ntiles);
    if (ts.first >= ts.end) return;
    int n0, h0, w0;
    tile_of(ts.first, n0, h0, w0);
    const long my_tiles = (ts.end - ts.first + ts.step - 1) / ts.step;
    const long S = my_tiles * nsteps;  // steps of this block
    // Two-steps-deep register pipeline: the loads issued at the top of step g (weights of step g + 2, input group tap(g) of the next chunk)
    // are committed to LDS at the END of step g + 1 -- a step is ~600 cycles of MFMA work, an L2 / HBM round trip 1-2.5 thousand, and a
    // load issued and consumed within the same step stalled every step for most of that latency (first version: 2600 cycles per step).
    struct Pre {
        uint4 wa, wb, xv;
        int xit, xbuf_i;  // input item and target buffer (-1: none)
        bool w;
    };
    Pre cur, nxt;
    // prologue: chunk 0 of the first tile and the weights of step 0 synchronously; weights of step 1 into `cur`
    for (int it = tid; it < HP * 4; it += 256) store_x(it, 0, load_x(it, 0, n0, h0, w0));
    {
        uint4 a, b;
        load_w(0, 0, a, b);
        wbuf[tid] = a;
        wbuf[tid + 256] = b;
    }
    cur.w = S > 1;
    cur.xit = 0;
    cur.xbuf_i = -1;
    cur.wa = cur.wb = cur.xv = make_uint4(0, 0, 0, 0);
    if (cur.w) load_w(0, 1, cur.wa, cur.wb);  // step 1 = (chunk 0, tap 1)
    __syncthreads();

    int xcur = 0;  // xbuf holding the current chunk
    int wcur = 0;
    long g = 0;
    for (long t = ts.first; t < ts.end; t += ts.step) {
        tile_of(t, n0, h0, w0);
        const bool has_next = t + ts.step < ts.end;
        int nn0 = 0, nh0 = 0, nw0 = 0;
        if (has_next) tile_of(t + ts.step, nn0, nh0, nw0);
        f32x4 acc[4][8];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int cc = 0; cc < ncc; ++cc) {
            const bool last_cc = cc + 1 == ncc;
            const bool pre_x = !last_cc || has_next;  // a chunk follows (next chunk of this tile, or chunk 0 of the next tile)
#pragma unroll OCRS_C128_UNROLL
            for (int tap = 0; tap < 9; ++tap, ++g) {
                // ---- issue (registers): weights of step g + 2, group `tap` of the next chunk's input tile
                nxt.w = g + 2 < S;
                nxt.wa = nxt.wb = nxt.xv = make_uint4(0, 0, 0, 0);
                if (nxt.w) {
                    int c2 = cc, t2 = tap + 2;
                    if (t2 >= 9) {
                        t2 -= 9;
                        c2 = last_cc ? 0 : cc + 1;
                    }
                    load_w(c2, t2, nxt.wa, nxt.wb);
                }
                nxt.xit = tap * XI + tid;
                nxt.xbuf_i = (pre_x && tap < 8 && tid < XI) ? (xcur ^ 1) : -1;
                if (nxt.xbuf_i >= 0) nxt.xv = last_cc ? load_x(nxt.xit, 0, nn0, nh0, nw0) : load_x(nxt.xit, cc + 1, n0, h0, w0);
                // ---- MFMAs of this step
                const int ky = tap / 3, kx = tap - ky * 3;
                const bf16* xt = xbuf + (long)xcur * (C2_XBYTES / 2);
                const uint4* wt = wbuf + wcur * 512;
                uint4 af[4], bfr[8];
#pragma unroll
                for (int a = 0; a < 4; ++a) af[a] = wt[(wm * 4 + a) * 64 + lane];
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const int row = wn * 8 + b, img = row / TH, ry = row - img * TH;
                    bfr[b] = *reinterpret_cast<const uint4*>(xt + ((img * HH + ry + ky) * C2_HW + kx + l15) * C2_PITCH + kq * 8);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < OCRS_C128_SPLIT; ++b)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[a]), __builtin_bit_cast(bf16x8, bfr[b]), acc[a][b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // ---- commit what was issued ONE STEP AGO: weights of step g + 1 -> the other weight buffer, its input group -> the next chunk's buffer
                // (in the middle of the step's MFMAs: the wait for the loads and the LDS stores run under the first MFMAs still in the pipe)
                if (cur.w) {
                    uint4* wd = wbuf + (wcur ^ 1) * 512;
                    wd[tid] = cur.wa;
                    wd[tid + 256] = cur.wb;
                }
                if (cur.xbuf_i >= 0) store_x(cur.xit, cur.xbuf_i, cur.xv);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = OCRS_C128_SPLIT; b < 8; ++b)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[a]), __builtin_bit_cast(bf16x8, bfr[b]), acc[a][b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
                wcur ^= 1;
                __syncthreads();
            }
            xcur ^= 1;
        }
        // ---- epilogue: bias, ReLU, store, per-channel sums of the stored values
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int m0 = (wm * 4 + a) * 16 + kq * 4;
            float bs[4] = {0.f, 0.f, 0.f, 0.f};
            if (bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) bs[r] = bias[m0 + r];
            }
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int row = wn * 8 + b, img = row / TH, ry = row - img * TH;
                const int n = n0 + img, h = h0 + ry, w = w0 + l15;
                if (n < N && h < H && w < W) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = acc[a][b][r] + bs[r];
                        if (relu) v[r] = fmaxf(v[r], 0.f);
                    }
                    store4(out + (((long)n * H + h) * W + w) * ldo + m0, v[0], v[1], v[2], v[3]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float q = Elem<bf16>::round(v[r]);
                        s1[r] += q;
                        s2[r] = fmaf(q, q, s2[r]);
                    }
                }
            }
            if (gstat) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a1 = quad16_sum(s1[r]), a2 = quad16_sum(s2[r]);
                    if (l15 == 0) {  // (wave, channel) has exactly one owner lane: plain adds in program order -> run-to-run bit-stable
                        s_stat[wn * 256 + m0 + r] += a1;
                        s_stat[wn * 256 + 128 + m0 + r] += a2;
                    }
                }
            }
        }
    }
    if (gstat) {
        __syncthreads();
        for (int i = tid; i < 128; i += 256) {
            atomicAdd(&gstat[i], (double)(s_stat[i] + s_stat[256 + i]));  // fp64 sums of fp32 partials: exact, order-independent
            atomicAdd(&gstat[128 + i], (double)(s_stat[128 + i] + s_stat[256 + 128 + i]));
        }
    }
}

// 1 if ocrs_conv_igemm's arguments describe a layer this kernel covers
bool conv3x3_c128_supported(int ldx, int ldo, int Cin, int M, int Hi, int Wi, int Ho, int Wo, int KH, int KW, int padh, int padw, int dtype) {
    static const int on = env_int("OCRS_CONV_C128", 1);
    return on && dtype == 1 && M == 128 && Cin % 32 == 0 && Cin >= 64 && KH == 3 && KW == 3 && padh == 1 && padw == 1 && Ho == Hi && Wo == Wi && Hi >= 8 &&
           ldx % 8 == 0 && ldo % 4 == 0;
}

int conv3x3_c128_launch(const void* x, int ldx, const void* wpk, void* out, int ldo, const float* bias, int relu, double* gstat, int Cin, int N, int H, int W,
                        hipStream_t st) {
    const bool two = H < 16;  // two images x 8 rows per tile
    const int TH = two ? 8 : 16, NI = two ? 2 : 1;
    const long ntiles = (long)((N + NI - 1) / NI) * ((W + C2_TW - 1) / C2_TW) * ((H + TH - 1) / TH);
    static DevOnce attr_set;
    if (attr_set.need()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_c128<16>), hipFuncAttributeMaxDynamicSharedMemorySize, C2_SMEM) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_c128<8>), hipFuncAttributeMaxDynamicSharedMemorySize, C2_SMEM) != hipSuccess)
            return OCRS_ERR_HIP;
        attr_set.done();
    }
    const int grid = persistent_grid(ntiles, 2);
    if (two)
        hipLaunchKernelGGL(k_conv3x3_c128<8>, dim3(grid), dim3(256), C2_SMEM, st, (const bf16*)x, ldx, wpk, (bf16*)out, ldo, bias, relu, gstat, Cin, N, H, W);
    else
        hipLaunchKernelGGL(k_conv3x3_c128<16>, dim3(grid), dim3(256), C2_SMEM, st, (const bf16*)x, ldx, wpk, (bf16*)out, ldo, bias, relu, gstat, Cin, N, H, W);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}
