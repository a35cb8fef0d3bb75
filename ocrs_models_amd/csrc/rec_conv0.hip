// Backward of the CRNN's fused first layer, Conv2d(1, 32, 3, padding=1) + ReLU + MaxPool2d(2) (ocrs_models/models.py:181-187), on the matrix cores
// (round 5; bf16 gradient, gfx950).  Same contract as k_conv0_bwd (rec_conv.hip): dW [32][9], db [32] accumulated from img [N][H][W] fp32 and the
// gradient g [N][H/2][W/2][32] w.r.t. the pooled output; nothing of the forward is stored, the layer is recomputed.
//
// k_conv0_bwd does that on the VALU: per (pooled pixel, channel) 36 FMAs of recompute, a 4-way arg-max, 27 selects to pick the winner's patch and
// 10 accumulations -- 215 us at B = 256 x 64 x 400 with the VALU 76 % busy and 80 weight-gradient accumulators per thread.  Here both halves are GEMMs:
//   (1) recompute  s[o][ch][px] = sum_k W[ch][k] patch[px][o][k]  (+ bias as K slot 9 against a constant 1):  v_mfma_f32_32x32x2_f32 -- EXACT fp32
//       products like the forward's FMAs, M = the 32 channels, N = 32 pooled pixels of one row, K = 10; one accumulator set per pooling position o,
//       so that the arg-max over the window is a per-lane comparison of four accumulators (a lane holds 16 channels of ONE pixel);
//   (2) dW[ch][k] += sum_{(px, o)} a[ch][(px, o)] patch[px][o][k],  a = g[px][ch] where o is the window's first maximum and it is positive, else 0:
//       v_mfma_f32_16x16x32_bf16 with K = (pixel, position) = 128 per step.  a is the bf16 gradient itself (exact); the patch value goes in as
//       hi + lo bf16 halves (two MFMAs: products good to 2^-17), column 9 of the patch matrix is the constant 1 (-> db).  Both operands are
//       written to wave-private LDS in their natural [K][column] order and read back through the LDS transpose read (lds_tr8).
// A wave owns steps of 32 consecutive pooled pixels of one output row; its LDS (image rows, a, patches) is private: LDS operations of one wave
// execute in order, so there is no workgroup barrier in the loop.  The next step's image rows and gradient vectors are register-prefetched.
#include "det_common.h"

#ifndef C0_ABL
#define C0_ABL 0  // measurement builds: 1 no GEMM 2, 2 no a / patch writes and no GEMM 2, 3 no GEMM 1 MFMAs, 4 no prefetch loads in the loop
#endif
namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int C0_PITCH = 72;                         // floats per staged image row (68 used: columns 2 wp0 - 2 .. 2 wp0 + 65)
constexpr int C0_IMG_B = 4 * C0_PITCH * 4;           // 4 rows
// a / patch rows keep their natural 64 / 32-byte pitch but their 8-byte granules are permuted per row: the 32 lanes of a write instruction own 32
// consecutive K rows, which at these pitches share 4 / 8 bank groups (the writes were 69 of the first version's 198 us).  a: granule ^ ((row >> 2) & 7),
// patch: granule + ((row >> 3) & 3) mod 4 -- the four rows of a transpose read share row >> 2 and row >> 3, so a read stays 4 rows x 32 contiguous
// bytes (conflict-free) and only its address changes
constexpr int C0_S2_B = 64 * 32 * 2;                 // a  [K = (o & 1) * 32 + n][32 channels] bf16: two pooling positions at a time (C0_PH phases per step)
constexpr int C0_B2_B = 64 * 16 * 2;                 // patch [K][16 columns: 9 taps | 1 | 0 ...] bf16, hi and lo
constexpr int C0_WAVE_B = 2 * C0_IMG_B + C0_S2_B + 2 * C0_B2_B;  // image rows as fp32 and as packed (hi | lo << 16) bf16 pairs
constexpr int C0_SMEM = 4 * C0_WAVE_B;

__device__ __forceinline__ void wave_lds_fence() {  // order this wave's LDS writes before its later reads (the hardware runs them in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
}  // namespace

__global__ __launch_bounds__(256, 3) void k_conv0_bwd_mm(const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ bias,
                                                        const bf16* __restrict__ g, float* __restrict__ dW, float* __restrict__ db, int N, int H, int W,
                                                        float* __restrict__ ws /*nullable: per-block partials [gridDim.x][320] = dW [32][9] | db [32] instead of the atomics*/) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, half = lane >> 5;  // GEMM 1: this lane's pixel column / K slot (and 4-row block of the accumulator layout)
    char* base = smem + wave * C0_WAVE_B;
    float* imgT = reinterpret_cast<float*>(base);
    unsigned* imgHL = reinterpret_cast<unsigned*>(base + C0_IMG_B);  // the same rows split once per pixel: hi | lo << 16 (the patch matrix's operand form)
    bf16* S2 = reinterpret_cast<bf16*>(base + 2 * C0_IMG_B);
    bf16* Bh = reinterpret_cast<bf16*>(base + 2 * C0_IMG_B + C0_S2_B);
    bf16* Bl = reinterpret_cast<bf16*>(base + 2 * C0_IMG_B + C0_S2_B + C0_B2_B);
    for (int i = lane; i < 2 * C0_B2_B / 16; i += 64) reinterpret_cast<uint4*>(Bh)[i] = make_uint4(0, 0, 0, 0);  // (columns 10..15 stay zero)
    for (int i = lane; i < 2 * C0_IMG_B / 4; i += 64) imgT[i] = 0.f;

    const int Hp = H >> 1, Wp = W >> 1, SPR = (Wp + 31) >> 5;
    const int nsteps = N * Hp * SPR;  // (< 2^31: checked by the launcher)
    // GEMM 1 A operand: A[m = lane % 32][k = 2 i + lane / 32] = W[m][k] (k < 9) | bias[m] (k = 9)
    float aw[5];
    int koff[5];  // imgT offset of tap k for pooling position (0, 0): patch[ky][kx] = imgT[ky][2 n + kx + 1]
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int k = 2 * i + half;
        aw[i] = k < 9 ? w[n * 9 + k] : bias[n];
        const int kk = k < 9 ? k : 0;
        koff[i] = (kk / 3) * C0_PITCH + 2 * n + (kk % 3) + 1;
    }
    const bool kone = half == 1;  // (i = 4: K slot 9 = the constant 1)
    const unsigned psel = half ? 0x07060302u : 0x05040100u;  // v_perm_b32 selector: (odd column's word, even column's word) -> the pair's lo | hi dword
    // GEMM 2 transpose-read geometry (see lds_tr8): K rows 8 kg + (i >> 2) (+ 4), 4 columns at (i & 3) * 4
    const int l15 = lane & 15, kg = lane >> 4;
    const int trow = 8 * kg + (l15 >> 2), tcol = (l15 & 3) * 4;
    f32x4 acc2[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};

    const int gw = (int)blockIdx.x * 4 + wave, nw = (int)gridDim.x * 4;
    // ---- software pipeline: the raw image pairs (lanes 0..33: one float2 per staged row) and gradient quads of the NEXT step.  Loads are
    // unconditional (clamped address + select: a load under a branch makes hipcc wait for it before the next one) and the step decode is 32-bit
    // (a 64-bit division expands into branches)
    // (plain loads: hipcc sinks part of them towards their first use at the top of the next step.  Forcing them early was measured -- volatile loads:
    //  waited for one by one; hand-waited asm loads into named accumulation registers a[100:115]: issued a whole step ahead, 135 vs 131 us -- the step
    //  is not waiting for global memory but for its own chain of LDS round trips at two waves per SIMD)
    float2 pim[4];
    uint2 pg[4];
    unsigned pok = 0;  // bits 0..3: image row dy of the prefetched step is valid for this lane, bit 4: its pixel lies inside the output row
    const unsigned HpS = (unsigned)(Hp * SPR);
    auto issue = [&](int s) {
        const bool act = s < nsteps;
        const unsigned ss = act ? (unsigned)s : 0u;
        const unsigned ni = ss / HpS, rem = ss - ni * HpS, hp = rem / (unsigned)SPR, wp0 = (rem - hp * (unsigned)SPR) * 32u;
        const int col0 = 2 * (int)wp0 - 2 + 2 * lane;
        const bool cok = act && lane < 34 && col0 >= 0 && col0 < W;
        const float* ib = img + ((long)ni * H) * W + (cok ? col0 : 0);
        unsigned okb = 0;
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            const int r = 2 * (int)hp - 1 + dy;
            const bool ok = cok && r >= 0 && r < H;
            pim[dy] = *reinterpret_cast<const float2*>(ib + (long)(ok ? r : 0) * W);
            okb |= ok ? 1u << dy : 0u;
        }
        const bool pv = act && (int)wp0 + n < Wp;
        const bf16* gp = g + (pv ? (((long)ni * Hp + hp) * Wp + wp0 + n) * 32 + half * 4 : 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) pg[q] = *reinterpret_cast<const uint2*>(gp + q * 8);
        pok = okb | (pv ? 16u : 0u);
    };
    issue(gw);
    for (int s = gw; s < nsteps; s += nw) {
        // ---- stage the four image rows (zeros outside the image = the convolution's padding)
        if (lane < 34) {
#pragma unroll
            for (int dy = 0; dy < 4; ++dy) {
                const bool ok = (pok >> dy) & 1u;
                const float px = ok ? pim[dy].x : 0.f, py = ok ? pim[dy].y : 0.f;
                *reinterpret_cast<float2*>(imgT + dy * C0_PITCH + 2 * lane) = make_float2(px, py);
                const unsigned hx = f2bf(px), hy = f2bf(py);
                const unsigned lx = f2bf(px - bf2f((unsigned short)hx)), ly = f2bf(py - bf2f((unsigned short)hy));
                *reinterpret_cast<uint2*>(imgHL + dy * C0_PITCH + 2 * lane) = make_uint2(hx | (lx << 16), hy | (ly << 16));
            }
        }
        uint2 gq[4];
        {
            const bool pv = (pok >> 4) & 1u;
#pragma unroll
            for (int q = 0; q < 4; ++q) gq[q] = make_uint2(pv ? pg[q].x : 0u, pv ? pg[q].y : 0u);
        }
        wave_lds_fence();
        if (C0_ABL != 4) issue(s + nw);
        // ---- GEMM 1: the four pre-activation maps of the window, 16 channels of pixel n per lane
        f32x16 sacc[4];
        unsigned hv[4][5];  // the same operands as (hi | lo << 16) words
#pragma unroll
        for (int o = 0; o < 4; ++o) {
#pragma unroll
            for (int v = 0; v < 16; ++v) sacc[o][v] = 0.f;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                float b = imgT[koff[i] + (o >> 1) * C0_PITCH + (o & 1)];
                unsigned hw = imgHL[koff[i] + (o >> 1) * C0_PITCH + (o & 1)];
                if (i == 4) {
                    b = kone ? 1.f : b;
                    hw = kone ? 0x00003f80u : hw;
                }
                hv[o][i] = hw;
                if (C0_ABL != 3) sacc[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[i], b, sacc[o], 0, 0, 0);
                else sacc[o][i] += aw[i] * b;
            }
        }
        // ---- two phases of two pooling positions each: a / patch rows of 64 (position, pixel) pairs -> GEMM 2 over K = 64.  (Half the LDS per wave: three
        // workgroups per CU instead of two -- the step is a chain of LDS round trips, and what hides them is other waves.)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
        // ---- window arg-max (first maximum, ReLU floor 0: k_conv0_bwd's rule) -> a[o][channel] = g or 0, as bf16 bits
        // accumulator v of a lane = channel 8 (v / 4) + 4 half + v % 4 -> the lane's gradient quad q = v / 4, element v % 4
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // the winner of a window = the first position that attains the maximum, if that is positive: lane masks (compares + scalar mask logic), then the
            // gradient's half words are kept or zeroed in place (no 16-bit arithmetic)
            bool wsel[4][4];  // [element][position]
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int v = q * 4 + e;
                const float s0 = sacc[0][v], s1 = sacc[1][v], s2 = sacc[2][v], s3 = sacc[3][v];
                const float m = fmaxf(fmaxf(fmaxf(s0, s1), s2), s3);
                const bool pos = m > 0.f, e0 = s0 == m, e1 = s1 == m, e2 = s2 == m;
                wsel[e][0] = pos && e0;
                wsel[e][1] = pos && !e0 && e1;
                wsel[e][2] = pos && !e0 && !e1 && e2;
                wsel[e][3] = pos && !e0 && !e1 && !e2;
            }
            const unsigned g0l = gq[q].x & 0xffffu, g0h = gq[q].x & 0xffff0000u, g1l = gq[q].y & 0xffffu, g1h = gq[q].y & 0xffff0000u;
#pragma unroll
            for (int ol = 0; ol < 2; ++ol) {
                const int o = 2 * ph + ol;
                const unsigned x = (wsel[0][o] ? g0l : 0u) | (wsel[1][o] ? g0h : 0u), y = (wsel[2][o] ? g1l : 0u) | (wsel[3][o] ? g1h : 0u);
                if (C0_ABL != 2 || x == 0x1234)
                *reinterpret_cast<uint2*>(S2 + (ol * 32 + n) * 32 + (((q * 2 + half) ^ ((n >> 2) & 7)) << 2)) = make_uint2(x, y);
            }
        }
        // ---- the patch matrix: this lane's K slots of GEMM 1 are its columns (k = 2 i + half) of row (o, n), as hi / lo bf16.  The two lanes of a
        // pixel (half 0 / 1: the even / odd column of a pair) exchange their packed words (v_permlane32_swap with both operands the same register
        // leaves the even column's word in one result and the odd column's in the other, in both halves); the half-0 lane writes the pair's hi
        // dword, the half-1 lane its lo dword (one v_perm_b32 with a lane-constant selector): 20 dword writes per lane, no 16-bit arithmetic
#pragma unroll
        for (int ol = 0; ol < 2; ++ol)
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int o = 2 * ph + ol;
                const auto sw = __builtin_amdgcn_permlane32_swap(hv[o][i], hv[o][i], false, false);  // sw[0]: column 2 i, sw[1]: column 2 i + 1
                const unsigned word = __builtin_amdgcn_perm(sw[1], sw[0], psel);
                if (C0_ABL != 2 || word == 0x1234) reinterpret_cast<unsigned*>(half ? Bl : Bh)[(ol * 32 + n) * 8 + ((i + 2 * (n >> 3)) & 7)] = word;
            }
        wave_lds_fence();
        // ---- GEMM 2: dW[ch][k] += a^T patch, K = 64 (pixel, position) rows per phase
#pragma unroll
        for (int ks = 0; ks < ((C0_ABL == 1 || C0_ABL == 2) ? 0 : 2); ++ks) {
            const int r0 = ks * 32 + trow;
            // (rows r0 .. r0 + 3 and r0 + 4 .. r0 + 7: n = row & 31; both groups of four share (n >> 3), each shares its (n >> 2))
            const int nb0 = r0 & 31, nb1 = (r0 + 4) & 31;
            const int bc = (((l15 & 3) + (nb0 >> 3)) & 3) * 4;
            const bf16x8 bh = lds_tr8(Bh + r0 * 16 + bc, Bh + (r0 + 4) * 16 + bc);
            const bf16x8 bl = lds_tr8(Bl + r0 * 16 + bc, Bl + (r0 + 4) * 16 + bc);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int g0 = ((a * 4 + (l15 & 3)) ^ ((nb0 >> 2) & 7)) * 4, g1 = ((a * 4 + (l15 & 3)) ^ ((nb1 >> 2) & 7)) * 4;
                const bf16x8 af = lds_tr8(S2 + r0 * 32 + g0, S2 + (r0 + 4) * 32 + g1);
                acc2[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bh, acc2[a], 0, 0, 0);
                acc2[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bl, acc2[a], 0, 0, 0);
            }
        }
        wave_lds_fence();
        }
    }
    // ---- flush: lane (column l15, rows 16 a + 4 kg + r) -> workgroup sum through LDS (fixed order), one float atomic per element and workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [wave][32 ch][16 cols]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 32 + a * 16 + 4 * kg + r) * 16 + l15] = acc2[a][r];
    __syncthreads();
    for (int e = tid; e < 32 * 10; e += 256) {
        const int c = e / 10, k = e - c * 10;
        const float v = (red[(0 * 32 + c) * 16 + k] + red[(1 * 32 + c) * 16 + k]) + (red[(2 * 32 + c) * 16 + k] + red[(3 * 32 + c) * 16 + k]);
        if (ws)
            ws[(long)blockIdx.x * 320 + (k < 9 ? c * 9 + k : 288 + c)] = v;
        else if (k < 9)
            atomicAdd(&dW[c * 9 + k], v);
        else
            atomicAdd(&db[c], v);
    }
}

// launcher for ocrs_conv0_bwd (rec_conv.hip): 1 if this kernel took the call (bf16 gradient, even H and W)
extern "C" int conv0_bwd_mm_launch(const float* img, const float* w, const float* bias, const void* g, float* dW, float* db, int N, int H, int W, int dtype, hipStream_t st) {
    static const int on = env_int("OCRS_CONV0_MM", 1);
    if (!on || dtype != 1 || (H & 1) || (W & 1) || H < 2 || W < 2 || (long)N * (H / 2) * (((W / 2) + 31) / 32) >= (1L << 30)) return 0;
    static DevOnce attr;
    if (attr.need()) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv0_bwd_mm), hipFuncAttributeMaxDynamicSharedMemorySize, C0_SMEM);
        attr.done();
    }
    const long nsteps = (long)N * (H / 2) * (((W / 2) + 31) / 32);
    static const int bpc = env_int("OCRS_CONV0_MM_BPC", 3);
    long grid = (nsteps + 3) / 4;
    if (grid > (long)kNumCU * bpc) grid = (long)kNumCU * bpc;
    // deferring (ocrs_bwd_defer_begin): per-block partials + one queued fixed-order column sum instead of 320 float atomics per workgroup
    float* ws = bwd_defer_ws(grid * 320);
    if (ws && !bwd_defer_reduce(ws, (int)grid, 320, dW, 288, 288, 288, db, 32)) ws = nullptr;
    hipLaunchKernelGGL(k_conv0_bwd_mm, dim3((int)grid), dim3(256), C0_SMEM, st, img, w, bias, (const bf16*)g, dW, db, N, H, W, ws);
    return 1;
}
