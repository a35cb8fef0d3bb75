// ConvTranspose2d input gradient at the DEEP U-Net levels (gfx950, bf16, Cup in {64, 128, 256}) -- the same GEMM as k_convt_dgrad
// (det_bwd.hip:  dx[n,i,j,c] = sum_{tap,o} W[c][o][tap] g[n, 2i+ky, 2j+kx, o];  K = (tap, o) = 9 Cout,  M = Cup), restructured like
// det_ctf.hip: k_convt_dgrad re-loads a 32-channel slice of the gradient operand from global memory for each of its 9 Cout / 32 K chunks (36
// exposed round trips and 72 barriers per tile at Cout = 128).  Here the (2*8+1) x (2*8+1) output-gradient pixels under an 8 x 8 tile of input
// pixels are staged ONCE, all channels, in bf16 LDS; the K loop streams packed weight fragments from L2 one chunk ahead against gradient
// fragments read at the tap's (stride-2) offset.
#include "det_common.h"

namespace {
template <int CUP, int COUT>
struct CtdCfg {
    static constexpr int NT = 512, NW = 8, TW = 8, TH = 8, TP = 64, NNT = 4, SW = 2 * TW + 1, SP = SW * (2 * TH + 1);
    static constexpr int CG = COUT / 8, PXC = COUT + 8;
    static constexpr int NXI = (SP * CG + NT - 1) / NT;        // (staged pixel, channel group) items per thread
    static constexpr int MTD = CUP / 16;                       // M tiles
    static constexpr int MPW = MTD >= NW ? MTD / NW : 1;       // per wave
    static constexpr int NPW = MTD >= NW ? NNT : NNT * MTD / NW;
    static constexpr int NKC = 9 * COUT / 32;                  // K chunks: chunk kc = tap (32 kc) / COUT, channels (32 kc) % COUT ..
    static constexpr int SMEM = (SP * PXC * 2 + 15) & ~15;
    static_assert(NT % CG == 0 && COUT % 32 == 0 && NPW >= 1, "role mapping");
};
}  // namespace

// STATS (round 5): x is the raw output z of the block that this ConvTranspose alone consumes, tr / saved its load transform / [mean | rstd]: the launch also
// accumulates that block's BatchNorm-backward sums gsum [2][CUP] (fp64) += sum ghat | rstd sum ghat (z - mean), ghat = the STORED dx where bn(z) > 0 -- what
// k_bn_bwd_reduce computed in a pass of its own over (dx, z) (34 + 13 + 26 us per step at the three deep levels).  Per-lane register accumulators over the
// block's tiles, lanes -> DPP row sums -> waves through LDS in a fixed order -> one fp64 atomic per channel and block (exact sums of fp32 partials).
template <int CUP, int COUT, bool STATS>
__global__ __launch_bounds__(512) void k_ctd(const bf16* __restrict__ g, const void* __restrict__ wpk, bf16* __restrict__ dx, int h, int w, int H, int W,
                                             int N, const bf16* __restrict__ x, const float* __restrict__ tr, const float* __restrict__ saved,
                                             double* __restrict__ gsum) {
    using C = CtdCfg<CUP, COUT>;
    constexpr int NT = C::NT, TW = C::TW, TH = C::TH, SW = C::SW, SP = C::SP, CG = C::CG, PXC = C::PXC, NXI = C::NXI, MTD = C::MTD, MPW = C::MPW, NPW = C::NPW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* gs = reinterpret_cast<bf16*>(smem);  // [SP][PXC] output gradient under the tile (0 outside the output)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        for (int i = tid; i < C::SMEM / 16; i += NT) reinterpret_cast<uint4*>(smem)[i] = z4;
    }
    __syncthreads();
    const int tiles_x = (w + TW - 1) / TW, tiles_y = (h + TH - 1) / TH, tpi = tiles_x * tiles_y;
    const long ntiles = (long)N * tpi;
    const int cg = tid % CG;
    const int m0w = MTD >= C::NW ? wave * MPW : wave % MTD, n0w = MTD >= C::NW ? 0 : (wave / MTD) * NPW;
    const int l15 = lane & 15, kq = lane >> 4;

    float sc[STATS ? MPW : 1][4], sh[STATS ? MPW : 1][4], mu[STATS ? MPW : 1][4], st1[STATS ? MPW : 1][4], st2[STATS ? MPW : 1][4];
    if constexpr (STATS) {
#pragma unroll
        for (int a = 0; a < MPW; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = (m0w + a) * 16 + kq * 4 + r;
                sc[a][r] = tr[ch];
                sh[a][r] = tr[CUP + ch];
                mu[a][r] = saved[ch];
                st1[a][r] = st2[a][r] = 0.f;
            }
    }
    TileSched ts(ntiles);
    for (long t = ts.first; t < ts.end; t += ts.step) {
        const int n = (int)(t / tpi), r = (int)(t - (long)n * tpi);
        const int i0 = (r / tiles_x) * TH, j0 = (r % tiles_x) * TW;
        uint4 raw[NXI];
#pragma unroll
        for (int j = 0; j < NXI; ++j) {
            const int sp = (tid + j * NT) / CG, sy = sp / SW, sx = sp - sy * SW;
            const int Y = 2 * i0 + sy, X = 2 * j0 + sx;
            const bool v = (SP * CG % NT == 0 || tid + j * NT < SP * CG) && Y < H && X < W;
            raw[j] = v ? *reinterpret_cast<const uint4*>(g + (((long)n * H + Y) * W + X) * COUT + cg * 8) : make_uint4(0, 0, 0, 0);
        }
        Mma<bf16>::Frag wf[MPW], wn[MPW];
#pragma unroll
        for (int a = 0; a < MPW; ++a) wf[a] = Mma<bf16>::load_w(wpk, (long)0 * MTD + m0w + a, lane);
#pragma unroll
        for (int j = 0; j < NXI; ++j)
            if (SP * CG % NT == 0 || tid + j * NT < SP * CG) *reinterpret_cast<uint4*>(gs + ((tid + j * NT) / CG) * PXC + cg * 8) = raw[j];
        __syncthreads();
        f32x4 acc[MPW][NPW];
#pragma unroll
        for (int a = 0; a < MPW; ++a)
#pragma unroll
            for (int b = 0; b < NPW; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int kc = 0; kc < C::NKC; ++kc) {
            if (kc + 1 < C::NKC) {
#pragma unroll
                for (int a = 0; a < MPW; ++a) wn[a] = Mma<bf16>::load_w(wpk, (long)(kc + 1) * MTD + m0w + a, lane);
            }
            const int k0 = kc * 32, tap = k0 / COUT, o0 = k0 - tap * COUT;
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int b = 0; b < NPW; ++b) {
                const int p = (n0w + b) * 16 + l15, ty = p / TW, tx = p - ty * TW;
                const uint4 pf = *reinterpret_cast<const uint4*>(gs + ((2 * ty + ky) * SW + 2 * tx + kx) * PXC + o0 + kq * 8);
#pragma unroll
                for (int a = 0; a < MPW; ++a)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[a].q), __builtin_bit_cast(bf16x8, pf), acc[a][b], 0, 0, 0);
            }
#pragma unroll
            for (int a = 0; a < MPW; ++a) wf[a] = wn[a];
        }
#pragma unroll
        for (int b = 0; b < NPW; ++b) {
            const int p = (n0w + b) * 16 + l15, ty = p / TW, tx = p - ty * TW;
            const int qi = i0 + ty, qj = j0 + tx;
            if (qi < h && qj < w) {
                bf16* dst = dx + (((long)n * h + qi) * w + qj) * CUP + m0w * 16 + kq * 4;
#pragma unroll
                for (int a = 0; a < MPW; ++a) store4(dst + a * 16, acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
                if constexpr (STATS) {
                    const bf16* zs = x + (((long)n * h + qi) * w + qj) * CUP + m0w * 16 + kq * 4;
#pragma unroll
                    for (int a = 0; a < MPW; ++a) {
                        const uint2 zq = *reinterpret_cast<const uint2*>(zs + a * 16);
                        const float zv[4] = {__uint_as_float(zq.x << 16), __uint_as_float(zq.x & 0xffff0000u), __uint_as_float(zq.y << 16),
                                             __uint_as_float(zq.y & 0xffff0000u)};
#pragma unroll
                        for (int r2 = 0; r2 < 4; ++r2) {
                            const float gh = fmaf(zv[r2], sc[a][r2], sh[a][r2]) > 0.f ? bf2f(f2bf(acc[a][b][r2])) : 0.f;  // (the stored, rounded gradient)
                            st1[a][r2] += gh;
                            st2[a][r2] = fmaf(gh, zv[r2] - mu[a][r2], st2[a][r2]);
                        }
                    }
                }
            }
        }
        __syncthreads();  // the staged region is free
    }
    if constexpr (STATS) {
        float* slots = reinterpret_cast<float*>(smem);  // [wave][MPW * 16 channels][2]  (the staged region is free)
#pragma unroll
        for (int a = 0; a < MPW; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v1 = quad16_sum(st1[a][r]), v2 = quad16_sum(st2[a][r]);
                if (l15 == 0) {
                    slots[((wave * MPW + a) * 16 + kq * 4 + r) * 2 + 0] = v1;
                    slots[((wave * MPW + a) * 16 + kq * 4 + r) * 2 + 1] = v2;
                }
            }
        __syncthreads();
        for (int e = tid; e < 2 * CUP; e += NT) {
            const int c = e >> 1, which = e & 1, mt = c >> 4;
            float v = 0.f;
            for (int wv = 0; wv < C::NW; ++wv) {  // the waves that own M tile mt (fixed order)
                const int m0 = MTD >= C::NW ? wv * MPW : wv % MTD;
                if (mt >= m0 && mt < m0 + MPW) v += slots[((wv * MPW + (mt - m0)) * 16 + (c & 15)) * 2 + which];
            }
            atomicAdd(&gsum[which * CUP + c], (double)(which ? v * saved[CUP + c] : v));
        }
    }
}

extern "C" {

long det_ctd_supported(int Cup, int Cout, int dtype) {
    static const int on = env_int("OCRS_CTD", 1);
    return on && dtype == 1 && ((Cup == 256 && Cout == 128) || (Cup == 128 && Cout == 64) || (Cup == 64 && Cout == 32));
}

int det_ctd_launch(const void* g, const void* wpk, void* dx, int Cup, int Cout, int N, int h, int w, int H, int W, hipStream_t st, const void* x,
                   const float* tr, const float* saved, double* gsum) {
    OCRS_CHECK_ARG(det_ctd_supported(Cup, Cout, 1));
    const long ntiles = (long)N * ((h + 7) / 8) * ((w + 7) / 8);
    const long cap = (long)kNumCU * (Cout <= 64 ? 2 : 1);
    long gsz = ntiles < cap ? ntiles : cap;
    if (gsz >= 8) gsz &= ~7L;
    if (gsz < 1) gsz = 1;
#define CTD_CASE(CU_, CO_)                                                                                                                  \
    if (Cup == CU_ && Cout == CO_) {                                                                                                        \
        using CC = CtdCfg<CU_, CO_>;                                                                                                        \
        static DevOnce attr_set;                                                                                                       \
        if (attr_set.need()) {                                                                                                                    \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ctd<CU_, CO_, false>), hipFuncAttributeMaxDynamicSharedMemorySize, CC::SMEM) != \
                    hipSuccess ||                                                                                                           \
                hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ctd<CU_, CO_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, CC::SMEM) != \
                    hipSuccess)                                                                                                             \
                return OCRS_ERR_HIP;                                                                                                        \
            attr_set.done();                                                                                                                \
        }                                                                                                                                   \
        if (gsum)                                                                                                                           \
            hipLaunchKernelGGL((k_ctd<CU_, CO_, true>), dim3((int)gsz), dim3(512), CC::SMEM, st, (const bf16*)g, wpk, (bf16*)dx, h, w, H, W, N, \
                               (const bf16*)x, tr, saved, gsum);                                                                            \
        else                                                                                                                                \
            hipLaunchKernelGGL((k_ctd<CU_, CO_, false>), dim3((int)gsz), dim3(512), CC::SMEM, st, (const bf16*)g, wpk, (bf16*)dx, h, w, H, W, N, \
                               (const bf16*)nullptr, (const float*)nullptr, (const float*)nullptr, (double*)nullptr);                       \
    }
    CTD_CASE(256, 128) CTD_CASE(128, 64) CTD_CASE(64, 32)
#undef CTD_CASE
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

}  // extern "C"
