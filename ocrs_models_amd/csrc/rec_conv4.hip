// 3x3 convolution of the CRNN's NARROW layers (ocrs_models/models.py:189-196: conv.3 = Conv2d(32, 64) at 32 x W/2, and its dgrad 64 -> 32) as an
// implicit GEMM whose WEIGHTS STAY IN LDS (gfx950, bf16).  Same contract as k_conv_igemm (rec_conv.hip).
//
// These two layers are HBM-bound by arithmetic (105 + 210 MB per call at B = 256 x 64 x 400 against 60 GFLOP: 192 FLOP/B; the HBM floor is
// ~57 us, the MFMA floor 24 us) and k_conv_igemm runs them at 157 / 177 us (2.0 / 1.8 TB/s).  The whole-row kernel of rec_conv3.hip does not
// fit them: with K = 288 / 576 a pass is a handful of K = 32 steps of 8-16 MFMAs per wave, and its per-step costs (A fragments from L2 one
// step ahead, wait, scalar state) are then the step -- measured 324 us.  Here:
//   * the layer's packed weight fragments (9 ncc MTB KB <= 36 KB) are copied to LDS once per workgroup; a step's A fragments are four
//     ds_read_b128 at its top -- no global load, no wait, no register double-buffering, any number of steps per pass;
//   * a pass is a 2-D tile of R rows x TWc columns (4 x 100 at the CRNN's sizes: 25 N tiles of 16 pixels, flattened like rec_conv3), so the
//     staged halo is 6 x 102 pixels per 32-channel chunk whatever the image width;
//   * the input halo is staged by LDS-DMA into two planar chunk buffers as in rec_conv3.hip (units issued at taps 0..4 of the previous chunk,
//     ONE barrier per chunk at tap 8); B fragments are refilled in place one step ahead; the epilogue regroups a tile through LDS and
//     stores 16 bytes per lane.
// 8 waves = 8 pixel groups (WM = 1): a wave holds MH x NTW accumulator tiles (4 x 4 or 2 x 4).
#include "det_common.h"

#ifndef R4_ABL
#define R4_ABL 0
#endif
namespace {
__device__ uint4 g4_zero64[4];  // 64 zero bytes: source of the padding lanes' DMA

__device__ __forceinline__ void r4_dma16(const void* gsrc, unsigned lds_dst) {  // lane i's 16 bytes -> LDS byte lds_dst + 16 i (see rec_conv3.hip)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void r4_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ int r4_opaque_v(int v) {
    asm volatile("" : "+v"(v));
    return v;
}
}  // namespace

template <int MH /* 16-channel output tiles: Cout = 16 MH */, int NTW /* N tiles of 16 pixels per wave */>
__global__ __launch_bounds__(512, 2) void k_conv3x3_tile(const bf16* __restrict__ x, int ldx, const uint4* __restrict__ wpk, bf16* __restrict__ out, int ldo,
                                                         const float* __restrict__ bias, int relu, double* __restrict__ gstat, int Cin, int N, int H, int W,
                                                         int R, int TWc, int hppad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int WN = 8, NU = 5, M = MH * 16;
    const int wn = wave;
    const int l15 = lane & 15, kq = lane >> 4;
    const int HWp = TWc + 2;
    const unsigned PLANE = (unsigned)hppad * 16u;
    const int ncc = Cin / 32, nsteps = ncc * 9;
    const unsigned WB = 8u * PLANE;                                  // weight fragments [9 ncc][MH][64] uint4
    const unsigned DB = WB + (unsigned)nsteps * MH * 1024u;          // 1 KB nobody reads: target of the padding DMA instructions
    const unsigned SB = DB + 1024u;                                  // statistics slots [WN][2][M] floats
    const unsigned TB = SB + WN * 2 * M * 4;                         // halo DMA unit table [NU][512] int2
    const unsigned EB = TB + NU * 512 * 8;                           // epilogue staging: 2304 B per wave
    const int rbs = (H + R - 1) / R, cbs = (W + TWc - 1) / TWc, ppi = rbs * cbs, total = N * ppi;
    const int ppb = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int p_first = blockIdx.x * ppb, p_end = p_first + ppb < total ? p_first + ppb : total;
    if (p_first >= p_end) return;
    const int npx = R * TWc;
    float* s_stat = reinterpret_cast<float*>(smem + SB);
    if (gstat)
        for (int i = tid; i < WN * 2 * M; i += 512) s_stat[i] = 0.f;

    // ---- per-lane constants: B fragment base of N tile j (tile index j * 8 + wave), and the pixel's offset inside the tile (row << 16 | col)
    unsigned baddr[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        int p = (j * WN + wn) * 16 + l15;
        p = p < npx ? p : npx - 1;
        const int row = p / TWc, col = p - row * TWc;
        baddr[j] = kq * PLANE + (unsigned)(row * HWp + col) * 16u;
    }
    // halo DMA units (rec_conv3.hip): unit u = (group u >> 2 of 64 staged pixels, plane u & 3); wave w issues units w + 8 i at taps 1 + i.
    // Table entry: byte offset of the source relative to the tile's first pixel, staged (row << 16 | column) or -1 (padding unit / lane)
    const char* zsrc = reinterpret_cast<const char*>(g4_zero64);
    const int nunits = hppad / 16;
    int2* s_xt = reinterpret_cast<int2*>(smem + TB);
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int u = i * 8 + wave, grp = u >> 2, kg = u & 3;
        const int sp = grp * 64 + lane, hy = sp / HWp, hx = sp - hy * HWp;
        const bool inside = u < nunits && hy < R + 2;
        s_xt[i * 512 + tid] = make_int2((((hy - 1) * W + (hx - 1)) * ldx + kg * 8) * 2, inside ? (hy << 16 | hx) : -1);
    }
    auto tile_org = [&](int ps, int& n, int& r0, int& c0) {
        n = ps / ppi;
        const int q = ps - n * ppi, rb = q / cbs;
        r0 = rb * R;
        c0 = (q - rb * cbs) * TWc;
    };
    auto issue_x = [&](int i, const char* xc, int r0, int c0, int buf) {  // xc: first pixel of the tile + the chunk's channel offset
        const int u = i * 8 + wave, grp = u >> 2, kg = u & 3;
        const int2 e = s_xt[i * 512 + tid];
        const int hy = e.y >> 16, hx = e.y & 0xffff;
        const bool ok = e.y >= 0 && (unsigned)(r0 - 1 + hy) < (unsigned)H && (unsigned)(c0 - 1 + hx) < (unsigned)W;
        const char* src = ok ? xc + e.x : zsrc;
        const unsigned dst = (unsigned)(buf * 4 + kg) * PLANE + (unsigned)grp * 1024u;
        r4_dma16(src, __builtin_amdgcn_readfirstlane(u < nunits ? dst : DB));
    };
    auto tile_ptr = [&](int ps, int& r0, int& c0) -> const char* {
        int n;
        tile_org(ps, n, r0, c0);
        return reinterpret_cast<const char*>(x + (((long)n * H + r0) * W + c0) * ldx);
    };
    auto lds16 = [&](unsigned off) -> uint4 { return *reinterpret_cast<const uint4*>(smem + off); };

    // ---- prologue: the weights (once) and chunk 0 of the first pass
    {
        const char* wsrc = reinterpret_cast<const char*>(wpk);
        const int npieces = nsteps * MH;  // 1 KB pieces
        for (int pc = wave; pc < npieces; pc += 8) r4_dma16(wsrc + (long)pc * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(WB + (unsigned)pc * 1024u));
    }
    int r0c, c0c;
    const char* xpc = tile_ptr(p_first, r0c, c0c);
    __syncthreads();  // (unit table)
#pragma unroll
    for (int i = 0; i < NU; ++i) issue_x(i, xpc, r0c, c0c, 0);
    r4_wait_vm<0>();
    __syncthreads();

    f32x4 acc[MH][NTW];
    uint4 bq[NTW];
#pragma unroll
    for (int b = 0; b < NTW; ++b) bq[b] = lds16(baddr[b]);

    int tap = 0, kx = 0, cc = 0;
    unsigned tapoff = 0, xbuf = 0;
    const char* xnc = xpc;  // source of the halo units issued during the current chunk (the NEXT chunk's tile + channels)
    int rnc = r0c, cnc = c0c;
    const char* xnx = xpc;  // ... for the chunk after the pass's last one: chunk 0 of the next pass
    int rnx = 0, cnx = 0;
    bool has_nx = false;
    const unsigned rowjump = (unsigned)(HWp - 2) * 16u;
    unsigned wstep = WB + lane * 16u;  // this step's weight fragments

    // One K = 32 step (see rec_conv3.hip): A fragments from the LDS-resident weights at the top, MFMAs of the NTW tiles, bq[b] refilled with the
    // next step's tile b behind its MFMAs.  DMA: at taps 1..5 one halo unit of the next chunk; the wait for them (all issued >= 2 steps before)
    // and the ONE barrier per chunk sit at tap 8.
    auto step = [&](auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        uint4 af[MH];
#pragma unroll
        for (int a = 0; a < MH; ++a) af[a] = lds16(wstep + (unsigned)a * 1024u);
        // halo units of the next chunk at taps 0..4 (the buffer they overwrite was last read before the previous tap-8 / end-of-pass barrier)
        const bool with_x = !(R4_ABL & 1) && tap < 5 && (cc + 1 < ncc || has_nx);
        if (with_x) issue_x(tap, xnc, rnc, cnc, (int)(xbuf ^ 1u));
        constexpr bool last_step = false;  // (moving the pass's last barrier + refill behind the epilogue was measured: 135 -> 162 us)
        if (tap == 8) {  // every wave's halo units of the next chunk have landed; nobody reads the buffer the chunk after next is staged into
            r4_wait_vm<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        unsigned ntapoff = tapoff + (kx == 2 ? rowjump : 16u);
        if (tap == 8) ntapoff = (xbuf ^ 1u) * 4u * PLANE;
#pragma unroll
        for (int b = 0; b < NTW; ++b) {
#pragma unroll
            for (int a = 0; a < MH; ++a) {
                const f32x4 c = FIRST ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[a][b];
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[a]), __builtin_bit_cast(bf16x8, bq[b]), c, 0, 0, 0);
            }
            if (!last_step) bq[b] = lds16(baddr[b] + ntapoff);
            __builtin_amdgcn_sched_barrier(0);
        }
        tapoff = ntapoff;
        kx = kx == 2 ? 0 : kx + 1;
        wstep += (unsigned)ncc * MH * 1024u;  // next tap of this chunk
        if (++tap == 9) {                      // next chunk (after the pass's last chunk: chunk 0 of the next pass, same weights)
            tap = 0;
            xbuf ^= 1u;
            ++cc;
            const int cn = cc == ncc ? 0 : cc;
            wstep = WB + (unsigned)cn * MH * 1024u + lane * 16u;
            const bool lastc = cc + 1 >= ncc;
            xnc = lastc ? xnx : xpc + (cc + 1) * 64;
            rnc = lastc ? rnx : r0c;
            cnc = lastc ? cnx : c0c;
        }
    };

    for (int ps = p_first; ps < p_end; ++ps) {
        has_nx = ps + 1 < p_end;
        if (has_nx) xnx = tile_ptr(ps + 1, rnx, cnx);
        cc = 0;
        {
            const bool lastc = ncc == 1;
            xnc = lastc ? xnx : xpc + 64;
            rnc = lastc ? rnx : r0c;
            cnc = lastc ? cnx : c0c;
        }
        step(std::true_type{});
#pragma clang loop unroll(disable)
        for (int sidx = 1; sidx < nsteps; ++sidx) step(std::false_type{});
        // ---- epilogue: bias, ReLU, per-channel sums of the stored values; a tile at a time regrouped through 2 KB of LDS of the wave's own
        // ([16 pixels][M channels], 16-byte chunks XOR-ed with the pixel index) and stored 16 bytes per lane
        int n, r0, c0;
        tile_org(ps, n, r0, c0);
        const int lane_e = r4_opaque_v(lane), l15e = lane_e & 15, kqe = lane_e >> 4;
        constexpr int CPP = MH * 2;  // 16-byte chunks per pixel
        char* stg = smem + EB + wave * 2304;  // 2 KB of tile + the tile's 16 pixel offsets
        int* spix = reinterpret_cast<int*>(stg + 2048);  // the tile's 16 pixel offsets (elements of `out`), -1 = outside
        char* obase = reinterpret_cast<char*>(out + (((long)n * H + r0) * W + c0) * ldo);
        float bs[MH][4], s1[MH][4], s2[MH][4];
#pragma unroll
        for (int a = 0; a < MH; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bs[a][r] = bias ? bias[a * 16 + kqe * 4 + r] : 0.f;
                s1[a][r] = s2[a][r] = 0.f;
            }
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int pj = (j * WN + wn) * 16;
            if (pj >= npx) continue;  // (wave-uniform)
            const int p = pj + l15e, row = p / TWc, col = p - row * TWc;
            const bool mine = p < npx && r0 + row < H && c0 + col < W;
            if (kqe == 0) spix[l15e] = mine ? row * W + col : -1;
#pragma unroll
            for (int a = 0; a < MH; ++a) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[a][j][r] + bs[a][r];
                    if (relu) v[r] = fmaxf(v[r], 0.f);
                }
                const uint2 pk = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                if (gstat && mine) {
                    const float q0 = __uint_as_float(pk.x << 16), q1 = __uint_as_float(pk.x & 0xffff0000u), q2 = __uint_as_float(pk.y << 16),
                                q3 = __uint_as_float(pk.y & 0xffff0000u);
                    s1[a][0] += q0; s1[a][1] += q1; s1[a][2] += q2; s1[a][3] += q3;
                    s2[a][0] = fmaf(q0, q0, s2[a][0]); s2[a][1] = fmaf(q1, q1, s2[a][1]); s2[a][2] = fmaf(q2, q2, s2[a][2]); s2[a][3] = fmaf(q3, q3, s2[a][3]);
                }
                *reinterpret_cast<uint2*>(stg + l15e * (CPP * 16) + (((a * 2 + (kqe >> 1)) ^ (l15e & (CPP - 1))) << 4) + (kqe & 1) * 8) = pk;
            }
            asm volatile("" ::: "memory");  // (LDS operations of one wave execute in issue order; keep the compiler from reordering across lanes' dependences)
#pragma unroll
            for (int h = 0; h < (MH + 1) / 2; ++h) {
                const int ci = lane_e + 64 * h, px = ci / CPP, c16 = ci % CPP;
                if (px < 16) {
                    const uint4 q = *reinterpret_cast<const uint4*>(stg + px * (CPP * 16) + ((c16 ^ (px & (CPP - 1))) << 4));
                    const int po = spix[px];
                    if (po >= 0) *reinterpret_cast<uint4*>(obase + (long)po * ldo * 2 + c16 * 16) = q;
                }
            }
            asm volatile("" ::: "memory");
        }
        if (gstat) {
#pragma unroll
            for (int a = 0; a < MH; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a1 = quad16_sum(s1[a][r]), a2 = quad16_sum(s2[a][r]);
                    if (l15e == 0) {  // (pixel group, channel) has exactly one owner lane: plain adds in program order -> run-to-run bit-stable
                        s_stat[wn * 2 * M + a * 16 + kqe * 4 + r] += a1;
                        s_stat[wn * 2 * M + M + a * 16 + kqe * 4 + r] += a2;
                    }
                }
        }
        xpc = xnx;
        r0c = rnx;
        c0c = cnx;
    }
    r4_wait_vm<0>();
    if (gstat) {
        __syncthreads();
        for (int i = tid; i < 2 * M; i += 512) {
            float t = s_stat[i];  // pixel groups in a fixed order
#pragma unroll
            for (int q = 1; q < WN; ++q) t += s_stat[q * 2 * M + i];
            atomicAdd(&gstat[i], (double)t);  // fp64 sums of fp32 partials: exact, order-independent
        }
    }
}

// ---- launch side --------------------------------------------------------------------------------------------------------------------------
namespace {
struct R4Plan {
    int R, TWc, hppad, smem;
    double cost;
};
// tile = R rows x TWc columns with <= 32 N tiles (8 waves x 4) and <= 640 staged pixels (five halo units per wave and chunk): fewest passes
R4Plan r4_plan(int M, int Cin, int H, int W) {
    R4Plan best{0, 0, 0, 0, 1e30};
    const int wbytes = 9 * (Cin / 32) * (M / 16) * 1024;
    for (int TWc = 16; TWc <= W; ++TWc) {
        if (TWc != W && TWc % 4 != 0) continue;
        for (int R = 1; R <= H; ++R) {
            const int nt = (R * TWc + 15) / 16, hp = ((R + 2) * (TWc + 2) + 63) / 64 * 64;
            const int smem = 8 * hp * 16 + wbytes + 1024 + 8 * 2 * M * 4 + 5 * 512 * 8 + 8 * 2304;
            if (nt > 32 || hp > 640 || smem > 160 * 1024) break;
            const double cost = (double)((H + R - 1) / R) * ((W + TWc - 1) / TWc) * (1.0 + 0.02 * (32 - nt));
            if (cost < best.cost) best = R4Plan{R, TWc, hp, smem, cost};
        }
    }
    return best;
}
template <int MH>
int r4_launch(const R4Plan& pl, const void* x, int ldx, const void* wpk, void* out, int ldo, const float* bias, int relu, double* gstat, int Cin, int N, int H, int W,
              hipStream_t st) {
    static DevOnce attr_set;
    auto kern = &k_conv3x3_tile<MH, 4>;
    if (attr_set.need()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return OCRS_ERR_HIP;
        attr_set.done();
    }
    const int total = N * ((H + pl.R - 1) / pl.R) * ((W + pl.TWc - 1) / pl.TWc);
    int grid = total < kNumCU ? total : kNumCU;
    const int ppb = (total + grid - 1) / grid;
    grid = (total + ppb - 1) / ppb;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), pl.smem, st, (const bf16*)x, ldx, (const uint4*)wpk, (bf16*)out, ldo, bias, relu, gstat, Cin, N, H, W, pl.R,
                       pl.TWc, pl.hppad);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}
}  // namespace

bool conv3x3_tile_supported(int ldx, int ldo, int Cin, int M, int Hi, int Wi, int Ho, int Wo, int KH, int KW, int padh, int padw, int dtype) {
    const int on = env_int("OCRS_CONV_TILE", 1);
    return on && dtype == 1 && (M == 64 || M == 32) && (Cin == 32 || Cin == 64) && Cin * M <= 64 * 32 && KH == 3 && KW == 3 && padh == 1 && padw == 1 && Ho == Hi &&
           Wo == Wi && ldx % 8 == 0 && ldo % 8 == 0 && (long)Hi * Wi * ldx < (1L << 30) && r4_plan(M, Cin, Hi, Wi).R > 0;
}

int conv3x3_tile_launch(const void* x, int ldx, const void* wpk, void* out, int ldo, const float* bias, int relu, double* gstat, int Cin, int M, int N, int H, int W,
                        hipStream_t st) {
    const R4Plan pl = r4_plan(M, Cin, H, W);
    if (M == 64) return r4_launch<4>(pl, x, ldx, wpk, out, ldo, bias, relu, gstat, Cin, N, H, W, st);
    return r4_launch<2>(pl, x, ldx, wpk, out, ldo, bias, relu, gstat, Cin, N, H, W, st);
}
