// ConvTranspose2d(Cup -> Cout, k3, s2, bias) + crop, forward, at the DEEP U-Net levels (gfx950, bf16, Cup in {64, 128, 256}) -- same GEMM as
// k_convt_fwd (det_fwd.hip: a GEMM "pixel" = an input-aligned position (i, j), i in [0, h], j in [0, w], producing the output quad
// (2i + py, 2j + px) from the four inputs x~[i - dy][j - dx]; K = (d, c) = 4 Cup, M = (q, o) = 4 Cout; reference ocrs_models/models.py:76-87),
// restructured like det_dwf.hip: k_convt_fwd re-loads a 32-channel slice of the pixel operand from global memory for each of its 4 Cup / 32
// K chunks (32 exposed round trips and 64 barriers per tile at Cup = 256: 110 us for a launch of 8.6 GFLOP).  Here the (8 + 1) x (8 + 1) input
// pixels of an 8 x 8 tile of positions are staged ONCE, all channels, as x~ in bf16 LDS; the K loop then only streams packed weight fragments
// from L2 (double-buffered in registers) against pixel fragments read at the neighbour's offset.  One barrier pair per tile.
#include "det_common.h"

namespace {
template <int CUP, int COUT>
struct CtfCfg {
    static constexpr int NT = 512, NW = 8, TW = 8, TH = 8, TP = 64, NNT = 4, SW = TW + 1, SP = SW * (TH + 1);
    static constexpr int CG = CUP / 8, PXC = CUP + 8;
    static constexpr int NXI = (SP * CG + NT - 1) / NT;   // (staged pixel, channel group) items per thread
    static constexpr int MTT = 4 * COUT / 16;             // M tiles (parity, output channel)
    static constexpr int MPW = MTT / NW;                  // per wave
    static constexpr int NKC = 4 * CUP / 32;              // K chunks: chunk kc = neighbour (32 kc) / CUP, channels (32 kc) % CUP ..
    static constexpr int SMEM = ((SP * PXC * 2 + 15) & ~15) + 3 * CUP * 4;
    static_assert(MTT % NW == 0 && NT % CG == 0 && CUP % 32 == 0, "role mapping");
};
__device__ __forceinline__ void unpack8c(const uint4& r, float (&v)[8]) {
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
    v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}
}  // namespace

template <int CUP, int COUT>
__global__ __launch_bounds__(512) void k_ctf(const bf16* __restrict__ x, const float* __restrict__ tr, const void* __restrict__ wpk,
                                             const float* __restrict__ bias, bf16* __restrict__ out, int h, int w, int H, int W, int N) {
    using C = CtfCfg<CUP, COUT>;
    constexpr int NT = C::NT, TW = C::TW, TH = C::TH, SW = C::SW, SP = C::SP, CG = C::CG, PXC = C::PXC, NXI = C::NXI, MTT = C::MTT, MPW = C::MPW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* xs = reinterpret_cast<bf16*>(smem);                                          // [SP][PXC] x~ (0 outside the input)
    float* s_tr = reinterpret_cast<float*>(smem + ((SP * PXC * 2 + 15) & ~15));        // [3][CUP]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 3 * CUP; i += NT) s_tr[i] = tr[i];
    {
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        for (int i = tid; i < ((SP * PXC * 2 + 15) & ~15) / 16; i += NT) reinterpret_cast<uint4*>(smem)[i] = z4;
    }
    __syncthreads();
    const int hp = h + 1, wp = w + 1;
    const int tiles_x = (wp + TW - 1) / TW, tiles_y = (hp + TH - 1) / TH, tpi = tiles_x * tiles_y;
    const long ntiles = (long)N * tpi;
    const int cg = tid % CG;
    float sc[8], sh[8], lo[8];
    load8(s_tr + cg * 8, sc);
    load8(s_tr + CUP + cg * 8, sh);
    load8(s_tr + 2 * CUP + cg * 8, lo);
    const int m0w = wave * MPW;
    const int l15 = lane & 15, kq = lane >> 4;

    TileSched ts(ntiles);
    for (long t = ts.first; t < ts.end; t += ts.step) {
        const int n = (int)(t / tpi), r = (int)(t - (long)n * tpi);
        const int i0 = (r / tiles_x) * TH, j0 = (r % tiles_x) * TW;
        // ---- stage the (TH + 1) x (TW + 1) input pixels (i0 - 1 .., j0 - 1 ..), all channels
        uint4 raw[NXI];
        unsigned ok = 0;
#pragma unroll
        for (int j = 0; j < NXI; ++j) {
            const int sp = (tid + j * NT) / CG, sy = sp / SW, sx = sp - sy * SW;
            const int ii = i0 - 1 + sy, jj = j0 - 1 + sx;
            const bool v = (SP * CG % NT == 0 || tid + j * NT < SP * CG) && (unsigned)ii < (unsigned)h && (unsigned)jj < (unsigned)w;
            raw[j] = *reinterpret_cast<const uint4*>(v ? x + (((long)n * h + ii) * w + jj) * CUP + cg * 8 : x);
            ok |= v ? 1u << j : 0u;
        }
        // first weight fragments: in flight with the pixel loads
        Mma<bf16>::Frag wf[MPW], wn[MPW];
#pragma unroll
        for (int a = 0; a < MPW; ++a) wf[a] = Mma<bf16>::load_w(wpk, (long)0 * MTT + m0w + a, lane);
#pragma unroll
        for (int j = 0; j < NXI; ++j) {
            if (SP * CG % NT != 0 && tid + j * NT >= SP * CG) break;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (ok & (1u << j)) {
                unpack8c(raw[j], v);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = max_lo(fmaf(v[i], sc[i], sh[i]), lo[i]);
            }
            store8_opaque(xs + ((tid + j * NT) / CG) * PXC + cg * 8, v);
        }
        __syncthreads();
        // ---- K loop: weight fragments streamed from L2 one chunk ahead, pixel fragments from the staged tile at the neighbour's offset
        f32x4 acc[MPW][4];
#pragma unroll
        for (int a = 0; a < MPW; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int kc = 0; kc < C::NKC; ++kc) {
            if (kc + 1 < C::NKC) {
#pragma unroll
                for (int a = 0; a < MPW; ++a) wn[a] = Mma<bf16>::load_w(wpk, (long)(kc + 1) * MTT + m0w + a, lane);
            }
            const int k0 = kc * 32, d = k0 / CUP, c0 = k0 - d * CUP;
            const int dy = d >> 1, dx = d & 1;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int p = b * 16 + l15, ty = p / TW, tx = p - ty * TW;
                const uint4 pf = *reinterpret_cast<const uint4*>(xs + ((ty - dy + 1) * SW + tx - dx + 1) * PXC + c0 + kq * 8);
#pragma unroll
                for (int a = 0; a < MPW; ++a)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[a].q), __builtin_bit_cast(bf16x8, pf), acc[a][b], 0, 0, 0);
            }
#pragma unroll
            for (int a = 0; a < MPW; ++a) wf[a] = wn[a];
        }
        // ---- epilogue: output quad of every position, + bias, crop
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int p = b * 16 + l15, ty = p / TW, tx = p - ty * TW;
            const int qi = i0 + ty, qj = j0 + tx;
            if (qi < hp && qj < wp) {
#pragma unroll
                for (int a = 0; a < MPW; ++a) {
                    const int m0 = (m0w + a) * 16 + kq * 4;
                    const int par = m0 / COUT, o0 = m0 - par * COUT;
                    const int Y = 2 * qi + (par >> 1), X = 2 * qj + (par & 1);
                    if (Y < H && X < W) {
                        const f32x4 v = acc[a][b];
                        store4(out + (((long)n * H + Y) * W + X) * COUT + o0, v[0] + bias[o0], v[1] + bias[o0 + 1], v[2] + bias[o0 + 2], v[3] + bias[o0 + 3]);
                    }
                }
            }
        }
        __syncthreads();  // the staged tile is free
    }
}

extern "C" {

long det_ctf_supported(int Cup, int Cout, int dtype) {
    static const int on = env_int("OCRS_CTF", 1);
    return on && dtype == 1 && ((Cup == 256 && Cout == 128) || (Cup == 128 && Cout == 64) || (Cup == 64 && Cout == 32));
}

int det_ctf_launch(const void* x, const float* tr, const void* wpk, const float* bias, void* out, int Cup, int Cout, int N, int h, int w, int H, int W,
                   hipStream_t st) {
    OCRS_CHECK_ARG(det_ctf_supported(Cup, Cout, 1));
    const long ntiles = (long)N * ((h + 1 + 7) / 8) * ((w + 1 + 7) / 8);
    long g = ntiles < 2L * kNumCU ? ntiles : 2L * kNumCU;
    if (g >= 8) g &= ~7L;
    if (g < 1) g = 1;
#define CTF2_CASE(CU_, CO_)                                                                                                                 \
    if (Cup == CU_ && Cout == CO_) {                                                                                                        \
        using CC = CtfCfg<CU_, CO_>;                                                                                                        \
        static DevOnce attr_set;                                                                                                       \
        if (attr_set.need()) {                                                                                                                    \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ctf<CU_, CO_>), hipFuncAttributeMaxDynamicSharedMemorySize, CC::SMEM) != \
                hipSuccess)                                                                                                                 \
                return OCRS_ERR_HIP;                                                                                                        \
            attr_set.done();                                                                                                                \
        }                                                                                                                                   \
        hipLaunchKernelGGL((k_ctf<CU_, CO_>), dim3((int)g), dim3(512), CC::SMEM, st, (const bf16*)x, tr, wpk, bias, (bf16*)out, h, w, H, W, N); \
    }
    CTF2_CASE(256, 128) CTF2_CASE(128, 64) CTF2_CASE(64, 32)
#undef CTF2_CASE
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

}  // extern "C"
