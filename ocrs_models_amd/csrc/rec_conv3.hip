// 3x3 convolution of the CRNN's wide layers (ocrs_models/models.py:189-236: conv.6 / 8 / 12 / 15 forward and the dgrads of 8 / 12 / 15) as an
// implicit GEMM with ONE workgroup per CU that owns whole image rows (gfx950, bf16).  Same contract as k_conv_igemm (rec_conv.hip).
//
// Why another form (round 4).  k_conv3x3_c128 (rec_conv2.hip: 128 x 256 block tiles, two 4-wave blocks per CU) sits at 30-34 % MFMA-busy:
// every K = 32 step re-stages 8 KB of weights through registers and ds_write, waits for 12 fragments at the top of the step, and ends in a
// barrier; its 16-pixel-wide tiles waste 12 % of a 100-pixel row and its tile count (1792 / 896 over 512 slots) leaves a partial last round.
// At the benchmarked sizes a layer is exactly ONE image per CU (B = 256 crops, 256 CUs), so here:
//   * a PASS = R whole rows of one image (R x W <= 448 pixels; 4 x 100 for the 128-channel layers), flattened to N tiles of 16 pixels that may
//     straddle rows (a lane's halo position is a per-lane LDS address, the tap is a wave-uniform offset): no column waste, no partial round;
//   * 8 waves (512 threads), TWO per SIMD, as WM (channel groups) x WN (pixel groups): a wave holds MH x NTW accumulator tiles (4 x 7 for the
//     128-channel layers: 64 channels x 112 pixels).  Measured on the way (tools/probes/mfma_issue_probe.hip): one wave per SIMD issues
//     v_mfma_f32_16x16x32_bf16 at 43-64 % of the peak however clean its loop, two reach 73-78 %; and a wave with more than 256 accumulator
//     registers (8 x 13 tiles was the first form) makes hipcc shuffle accumulators between the AGPR and VGPR halves around every MFMA
//     (328 v_accvgpr moves per 80 MFMAs);
//   * the input halo reaches LDS by LDS-DMA (global_load_lds_dwordx4 in inline asm, so that hipcc's waitcnt bookkeeping does not drain it),
//     staged PLANAR -- [8-channel group][staged pixel][16 B], plane stride = 0 mod 256 B -- so a B fragment read (16 consecutive pixels x 4
//     channel groups) is conflict-free at every tap shift; one DMA unit fills 64 consecutive staged pixels of a plane (padding lanes read a
//     zero line): no VGPR, no ds_write, no VALU per byte.  Two chunk buffers: the next 32-channel chunk's halo (also the next pass's first) is
//     issued one unit per wave and step at taps 1..5 of the current chunk;
//   * the A (weight) fragments come straight from the packed array in L2 / L1 into registers, one step ahead, by asm loads with a hand-written
//     wait (two register sets, steps in pairs): no weight ring in LDS, hence nothing to synchronise per step -- ONE barrier per chunk (tap 8,
//     whose fragment refills are the first reads of the next chunk's buffer).  With a shared weight ring and a barrier per step the older
//     wave of every SIMD waited 22 % of the kernel for the younger one;
//   * B fragments are refilled in place a full step ahead of their MFMAs; the step's scalar state is incremental (a few dozen SALU; the first
//     version's (tap, chunk, pass) arithmetic was ~300 instructions per step next to 28 MFMAs);
//   * the epilogue regroups an N tile through 2 KB of LDS per wave and stores 16 bytes per lane (the CUs reach their epilogues together: the
//     layer's output is written in bursts, and 8-byte pieces of 32-byte segments made those bursts 40 % longer).
// 2 x 2 kernels and padding 0 / 1 (the CRNN's last conv and its dgrad) run through the same code with 4 steps per chunk.
#include "det_common.h"

#ifndef R3_ABL
#define R3_ABL 0  // ablation mask (measurement builds only): 1 no halo DMA, 2 no weight DMA, 4 no barrier / vmcnt wait, 8 no MFMA, 16 no B fragment reads
#endif
#ifndef R3_STAGGER
#define R3_STAGGER 0  // 1: stagger (measured: 89 -> 146 us, the older wave of a SIMD then waits for the younger at every barrier); 0: both waves of a SIMD take the step barrier at tile 1
#endif
#ifndef R3_SETPRIO
#define R3_SETPRIO 0  // s_setprio 1 around a step's MFMA / fragment-refill block
#endif
#ifndef R3_NT_OUT
#define R3_NT_OUT 1  // non-temporal output stores: the kernel 3-4 % faster (77.9 -> 75.1, 145 -> 139 us), CRNN step 4.92 -> 4.88 ms in A/B runs (the
                     // same in k_conv3x3_tile: no further change; in k_gemm_x3p, whose output the persistent GRU reads next: slower)
#endif
#ifndef R3_DBG
#define R3_DBG 0  // 1: per-phase cycle counters of every wave of block 0 (measurement builds; read with ocrs_conv_rows_dbg)
#endif
#if R3_DBG
__device__ long long g_r3dbg[8][8];
#define R3_T() __builtin_readcyclecounter()
#endif
namespace {
constexpr int R3_HPMAX = 1024;   // staged halo pixels per plane (<= 64 KB per 32-channel chunk, two chunks)
__device__ uint4 g_zero64[4];    // 64 zero bytes: source of the padding lanes' DMA (four channel groups)

// LDS-DMA of 16 bytes per lane: lane i's bytes land at LDS byte `lds_dst` + 16 i (lds_dst wave-uniform).  Invisible to hipcc's s_waitcnt
// bookkeeping (the point: a counted vmcnt below instead of vmcnt(0) in front of the next ds_read).
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
// the same with a wave-uniform base (SGPR pair) + a 32-bit per-lane byte offset: nothing per-lane for the compiler to pre-compute per step
__device__ __forceinline__ void dma16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
// hide a wave-uniform value from loop-invariant code motion (hipcc otherwise pre-computes every step's address set of the 18-step body
// outside the chunk loop -- ~150 registers -- and spills it; a spill reload inside the loop is a VMEM load hipcc waits vmcnt(0) for)
__device__ __forceinline__ unsigned opaque_s(unsigned v) {
    asm volatile("" : "+s"(v));
    return v;
}
__device__ __forceinline__ int opaque_v(int v) {  // the per-lane counterpart (a value derived from it cannot be kept in a register across the loop)
    asm volatile("" : "+v"(v));
    return v;
}
// A fragment straight from global memory / L2 into registers, hidden from hipcc's waitcnt bookkeeping like the DMAs (all VMEM operations of the
// K loop are counted by hand: mixing kinds makes hipcc over-wait by the number of hidden operations).  The destination is only valid behind
// wait_a() (the statement ties the registers, so no consumer is scheduled above it); tools/check_opaque_loads.py-style audit: no spills and no
// compiler copies of the fragment registers between load and wait (kernel-resource-usage: 0 spills; the loop's only v_mov are scalar moves).
typedef unsigned r3_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load_a(r3_u32x4& d, unsigned voff, const void* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(d) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void wait_a(r3_u32x4 (&f)[4]) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])::"memory");
}
__device__ __forceinline__ void wait_a(r3_u32x4 (&f)[2]) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(f[0]), "+v"(f[1])::"memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
}  // namespace

template <int MH /* 16-channel output tiles per wave */, int WM /* waves along the channels: Cout = 16 MH WM */, int NTW /* N tiles of 16 pixels per wave */,
          int UPS /* halo DMA units per wave and step (taps 1..5): 1 for <= 640 staged pixels, 2 up to 1280 */>
__global__ __launch_bounds__(512, 2) void k_conv3x3_rows(const bf16* __restrict__ x, int ldx, const uint4* __restrict__ wpk, bf16* __restrict__ out, int ldo,
                                                         const float* __restrict__ bias, int relu, double* __restrict__ gstat, int Cin, int N, int H, int W,
                                                         int R, int hppad, int Hi, int Wi, int KW, int pad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int MTB = MH * WM, WN = 8 / WM;
    // (H, W = OUTPUT size; Hi, Wi = input size; KW x KW taps, padding `pad`: 3 x 3 / pad 1 is Ho = Hi; the CRNN's last conv is 2 x 2 / pad 1, Ho = Hi + 1)
    const int NT = KW * KW;                       // taps = steps per chunk
    const int XT0 = NT == 9 ? 1 : 0, XTN = NT == 9 ? 5 : 2;  // halo units are issued at taps XT0 .. XT0 + XTN - 1 (landed two tops later: <= NT - 1)
    const int NUW = XTN * UPS;                    // units per wave and chunk
    const int wm = wave % WM, wn = wave / WM;  // waves w and w + 4 share a SIMD: pixel groups (0, 2) and (1, 3) -> 13 / 13 / 12 / 12 tiles of a 25-tile pass
    const int l15 = lane & 15, kq = lane >> 4;
    const int HWp = W + KW - 1;
    const unsigned PLANE = (unsigned)hppad * 16u;
    const unsigned DB = 8u * PLANE;                    // 1 KB nobody reads: target of the padding DMA instructions
    const unsigned SB = DB + 1024u;                    // statistics slots [WN pixel groups][2][MTB * 16] floats
    const unsigned TB = SB + WN * 2 * MTB * 16 * 4;    // halo DMA unit table [NUW][512] int2
    const unsigned EB = TB + (unsigned)NUW * 512 * 8;             // epilogue staging: 2 KB per wave
    const int ncc = Cin / 32, nsteps = ncc * KW * KW;        // ncc even (launch condition): chunk q of a pass lives in input buffer q & 1
    const int ppi = (H + R - 1) / R, total = N * ppi;
    const int ppb = (total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int p_first = blockIdx.x * ppb, p_end = p_first + ppb < total ? p_first + ppb : total;
    if (p_first >= p_end) return;
    const int npx = R * W;
    float* s_stat = reinterpret_cast<float*>(smem + SB);
    if (gstat) {
        for (int i = tid; i < WN * 2 * MTB * 16; i += 512) s_stat[i] = 0.f;
    }
#if R3_DBG
    long long t_start = R3_T(), t_bar = 0, t_epi = 0, t_pro = 0, t_x = 0;
#endif

    // ---- per-lane constants.  B fragment of N tile j (tile index j * WN + wn, pixel p = tile * 16 + l15 of the pass): LDS byte offset of
    // the pixel's halo position (tap (0, 0)) in plane kq
    unsigned baddr[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        int p = (j * WN + wn) * 16 + l15;
        p = p < npx ? p : npx - 1;
        const int row = p / W, col = p - row * W;
        baddr[j] = kq * PLANE + (unsigned)(row * HWp + col) * 16u;
    }
    // input DMA units: unit u = (group u >> 2 of 64 staged pixels, plane u & 3), hppad / 16 of them per chunk; wave w issues units w + 8 i, UPS
    // per step at taps 1 .. 5 (a unit is a 64-lane gather -- every lane another pixel's line, ~64 cycles of the CU's address path -- and a
    // chunk's halo must have landed two barriers before tap 8 starts reading it).  Units past the staged pixels
    // and lanes outside the image read the zero line (padding units go to the dump slot): every wave issues the same number of DMAs per step,
    // so the counted waits are compile-time constants.
    const char* zsrc = reinterpret_cast<const char*>(g_zero64);
    const int nunits = hppad / 16;
    // Per (unit, lane) table in LDS, filled once: byte offset of the source relative to the pass's first pixel, and the staged row (the only
    // part of the bounds test that depends on the pass; 1 << 24 = never valid) -- the issue path is a ds_read_b64, a compare and a select.
    int2* s_xt = reinterpret_cast<int2*>(smem + TB);
#pragma unroll
    for (int i = 0; i < 6 * UPS; ++i) {
        if (i >= NUW) break;
        const int u = i * 8 + wave, grp = u >> 2, kg = u & 3;
        const int sp = grp * 64 + lane, hy = sp / HWp, hx = sp - hy * HWp, w = hx - pad;
        const bool colok = u < nunits && hy < R + KW - 1 && (unsigned)w < (unsigned)Wi;
        s_xt[i * 512 + tid] = make_int2((((hy - pad) * Wi + w) * ldx + kg * 8) * 2, colok ? hy : (1 << 24));
    }
    // unit i (run-time, wave-uniform) of the chunk at `xc` (first pixel of its pass + the chunk's channel offset; r0 = first row of the pass)
    auto issue_x = [&](int i, const char* xc, int r0, int buf) {
        const int u = i * 8 + wave, grp = u >> 2, kg = u & 3;
        const int2 e = s_xt[i * 512 + tid];
        const bool ok = (unsigned)(r0 - pad + e.y) < (unsigned)Hi;
        const char* src = ok ? xc + e.x : zsrc;
        const unsigned dst = (unsigned)(buf * 4 + kg) * PLANE + (unsigned)grp * 1024u;
        dma16(src, __builtin_amdgcn_readfirstlane(u < nunits ? dst : DB));
    };
    // weights: the wave's MH fragments of a step straight from the packed array (wave-uniform step pointer + this per-lane offset): the 295 KB
    // of a layer's weights stay in L2 / L1 (four waves read the same fragments), and without a shared LDS ring there is nothing to
    // synchronise per step
    const unsigned wlane = (unsigned)((wm * MH) * 64 + lane) * 16u;
    auto load_af = [&](r3_u32x4 (&dst)[MH], const char* wstep) {
#pragma unroll
        for (int a = 0; a < MH; ++a) load_a(dst[a], wlane, wstep + a * 1024);
    };
    auto lds16 = [&](unsigned off) -> uint4 { return *reinterpret_cast<const uint4*>(smem + off); };
    auto pass_ptr = [&](int ps, int& r0) -> const char* {
        const int n = ps / ppi;
        r0 = (ps - n * ppi) * R;
        return reinterpret_cast<const char*>(x + ((long)n * Hi + r0) * Wi * ldx);  // (row r0 of the INPUT: may lie past its last row -- only an address base)
    };
    const char* wbase = reinterpret_cast<const char*>(wpk);
    const long tapstride = (long)ncc * (MTB * 1024);  // bytes between the fragments of consecutive taps of one chunk

    // ---- prologue: chunk 0 of the first pass, weights of steps 0, 1 and 2
    int r0c;
    const char* xpc = pass_ptr(p_first, r0c);
    __syncthreads();  // (unit table)
#pragma unroll
    for (int i = 0; i < 6 * UPS; ++i)
        if (i < NUW) issue_x(i, xpc, r0c, 0);
    f32x4 acc[MH][NTW];
    r3_u32x4 afA[MH], afB[MH];
    uint4 bq[NTW];
    load_af(afA, wbase);
    wait_a(afA);
    __syncthreads();
    load_af(afB, wbase + tapstride);  // step 1
#pragma unroll
    for (int b = 0; b < NTW; ++b) bq[b] = lds16(baddr[b]);
#if R3_DBG
    t_pro = R3_T() - t_start;
#endif

    // run-time step state, kept to a few scalar instructions per step (the first version's (tap, chunk, pass) state machine with its
    // divisions and 64-bit address arithmetic cost ~200 scalar + ~100 vector instructions per step -- as much issue time as the 28 MFMAs)
    int tap = 0, kx = 0;                 // current step's tap, its column
    int cc = 0;                          // current chunk
    unsigned tapoff = 0;                 // LDS byte offset of the current step's tap (incl. the input buffer)
    unsigned xbuf = 0;                   // input buffer of the current chunk (0 / 1)
    int tap2 = 2, cc2 = 0;               // (tap, chunk) of step + 2
    const char* wp2 = wbase + 2 * tapstride;
    const char* xnc = xpc;               // source of the halo units issued during the current chunk: first pixel of the NEXT chunk's pass + its channels
    int rnc = r0c;
    const char* xnx = xpc;               // the same for the chunk after the pass's last one: chunk 0 of the next pass
    int rnx = 0;
    const unsigned rowjump = (unsigned)(HWp - (KW - 1)) * 16u;

    // One K = 32 step: MFMAs of the NTW tiles from af / bq (B fragments loaded one step ago); behind tile b's MFMAs bq[b] is refilled with the
    // next step's tile b (a full step of distance).  afp -- the A register set the PREVIOUS step used -- receives the fragments of step + 1 at
    // the top, one step ahead of their use.  DMA: at taps 1..5 UPS halo units of the next chunk.  The top-of-step wait (this step's A
    // fragments; VMEM returns in order, so also every DMA this wave issued before them) is the only per-step synchronisation; the ONE barrier
    // per chunk sits at tap 8, whose fragment refills are the first reads of the next chunk's buffer.
    auto step = [&](auto first_tag, r3_u32x4 (&af)[MH], r3_u32x4 (&afp)[MH], bool peel) {
        constexpr bool FIRST = decltype(first_tag)::value;
        if (!peel) {
#if R3_DBG
            const long long tb0 = R3_T();
#endif
            wait_a(af);
#if R3_DBG
            t_x += R3_T() - tb0;
#endif
            if (!(R3_ABL & 2)) load_af(afp, wp2);
        }
        const bool with_x = !(R3_ABL & 1) && (unsigned)(tap - XT0) < (unsigned)XTN;
        if (with_x) {
#pragma unroll
            for (int q = 0; q < UPS; ++q) issue_x((tap - XT0) * UPS + q, xnc, rnc, (int)(xbuf ^ 1u));
        }
        if (tap == NT - 1 && !(R3_ABL & 4)) {  // every wave's halo units of the next chunk have landed (issued at taps <= 5, waited for at the top of tap 7)
#if R3_DBG
            const long long tb1 = R3_T();
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (and this wave's reads of the buffer the chunk after next will be staged into)
            __builtin_amdgcn_s_barrier();
#if R3_DBG
            t_bar += R3_T() - tb1;
#endif
        }
        // next step's tap offset: one pixel right, or to the start of the next row, or (after tap 8) tap 0 of the other input buffer
        unsigned ntapoff = tapoff + (kx == KW - 1 ? rowjump : 16u);
        if (tap == NT - 1) ntapoff = (xbuf ^ 1u) * 4u * PLANE;
#if R3_SETPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int b = 0; b < NTW; ++b) {
#pragma unroll
            for (int a = 0; a < MH; ++a) {
                const f32x4 c = FIRST ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[a][b];
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[a]), __builtin_bit_cast(bf16x8, bq[b]), c, 0, 0, 0);
            }
            if (!(R3_ABL & 16)) bq[b] = lds16(baddr[b] + ntapoff);
            __builtin_amdgcn_sched_barrier(0);
        }
#if R3_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        // ---- advance (scalar)
        tapoff = ntapoff;
        kx = kx == KW - 1 ? 0 : kx + 1;
        if (!peel) {
            wp2 += tapstride;
            if (++tap2 == NT) {  // step + 2 enters the next chunk (after a pass's last chunk: chunk 0 again, same weights)
                tap2 = 0;
                cc2 = cc2 + 1 == ncc ? 0 : cc2 + 1;
                wp2 = wbase + (long)cc2 * (MTB * 1024);
            }
        }
        if (++tap == NT) {  // next chunk; the halo units issued during it belong to the chunk after it
            tap = 0;
            xbuf ^= 1u;
            ++cc;
            const bool lastc = cc + 1 >= ncc;
            xnc = lastc ? xnx : xpc + (cc + 1) * 64;
            rnc = lastc ? rnx : r0c;
        }
    };

    for (int ps = p_first; ps < p_end; ++ps) {
        // successor of the pass's last chunk: chunk 0 of the next pass (none: every lane reads the zero line, nobody reads the buffer)
        rnx = -(1 << 20);
        xnx = xpc;
        if (ps + 1 < p_end) xnx = pass_ptr(ps + 1, rnx);
        cc = 0;
        xnc = ncc == 1 ? xnx : xpc + 64;
        rnc = ncc == 1 ? rnx : r0c;
        step(std::true_type{}, afA, afB, ps == p_first);
        step(std::false_type{}, afB, afA, false);
#pragma clang loop unroll(disable)
        for (int sidx = 2; sidx < nsteps; sidx += 2) {  // (nsteps = 9 ncc is even: ncc % 2 == 0 is a launch condition)
            step(std::false_type{}, afA, afB, false);
            step(std::false_type{}, afB, afA, false);
        }
#if R3_DBG
        const long long te0 = R3_T();
#endif
        // ---- epilogue of the pass: bias, ReLU, per-channel sums of the stored values, store.  A lane holds 4 consecutive channels of one pixel
        // per accumulator tile; stored directly that is 8-byte pieces of four different instructions per 32 bytes -- and every CU of the chip
        // reaches its epilogue at the same time, so the layer's output is written in bursts at the HBM write rate that partial lines allow.
        // Each wave therefore regroups one N tile at a time through 2 KB of LDS of its own ([16 pixels][64 channels], 16-byte chunks XOR-ed
        // with the pixel index): the stores are 16 bytes per lane, eight lanes per 128-byte line of a pixel's channel half.
        const int n = ps / ppi, r0 = r0c;
        const int lim = (H - r0) * W < npx ? (H - r0) * W : npx;
        const int lane_e = opaque_v(lane), l15 = lane_e & 15, kq = lane_e >> 4;  // (shadowing: nothing of the epilogue's addressing lives in registers across the K loop)
        constexpr int CPP = MH * 2;  // 16-byte chunks per pixel of the wave's channel group
        static_assert(MH == 4 || MH == 2, "epilogue staging layout");
        char* stg = smem + EB + wave * 2048;
        char* obase = reinterpret_cast<char*>(out + ((long)n * H + r0) * W * ldo + wm * MH * 16);
        float bs[MH][4], s1[MH][4], s2[MH][4];
#pragma unroll
        for (int a = 0; a < MH; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bs[a][r] = bias ? bias[(wm * MH + a) * 16 + kq * 4 + r] : 0.f;
                s1[a][r] = s2[a][r] = 0.f;
            }
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int pj = (j * WN + wn) * 16;
            if (pj >= lim || ((R3_ABL & 32) && acc[0][j][0] != 123.f)) continue;  // (wave-uniform)
            const bool mine = pj + l15 < lim;
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
                *reinterpret_cast<uint2*>(stg + l15 * (CPP * 16) + (((a * 2 + (kq >> 1)) ^ (l15 & (CPP - 1))) << 4) + (kq & 1) * 8) = pk;
            }
#pragma unroll
            for (int h = 0; h < MH / 2; ++h) {
                const int ci = lane_e + 64 * h, px = ci / CPP, c16 = ci % CPP;
                const uint4 q = *reinterpret_cast<const uint4*>(stg + px * (CPP * 16) + ((c16 ^ (px & (CPP - 1))) << 4));
#if R3_NT_OUT
                typedef unsigned u4v __attribute__((ext_vector_type(4)));
                if (pj + px < lim) __builtin_nontemporal_store((u4v){q.x, q.y, q.z, q.w}, reinterpret_cast<u4v*>(obase + (long)(pj + px) * ldo * 2 + c16 * 16));
#else
                if (pj + px < lim) *reinterpret_cast<uint4*>(obase + (long)(pj + px) * ldo * 2 + c16 * 16) = q;
#endif
            }
        }
        if (gstat) {
#pragma unroll
            for (int a = 0; a < MH; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a1 = quad16_sum(s1[a][r]), a2 = quad16_sum(s2[a][r]);
                    if (l15 == 0) {  // (pixel group, channel) has exactly one owner lane: plain adds in program order -> run-to-run bit-stable
                        s_stat[wn * 2 * MTB * 16 + (wm * MH + a) * 16 + kq * 4 + r] += a1;
                        s_stat[wn * 2 * MTB * 16 + MTB * 16 + (wm * MH + a) * 16 + kq * 4 + r] += a2;
                    }
                }
        }
#if R3_DBG
        t_epi += R3_T() - te0;
#endif
        xpc = xnx;
        r0c = rnx;
    }
#if R3_DBG
    if (blockIdx.x == 0 && lane == 0) {
        g_r3dbg[wave][0] = R3_T() - t_start;
        g_r3dbg[wave][1] = t_pro;
        g_r3dbg[wave][2] = t_x;
        g_r3dbg[wave][3] = t_bar;
        g_r3dbg[wave][4] = t_epi;
    }
#endif
    wait_vm<0>();  // (the wrapped-around weight loads of the last steps: no LDS-DMA may be in flight when the workgroup's LDS is released)
    if (gstat) {
        __syncthreads();
        constexpr int M = MTB * 16;
        for (int i = tid; i < 2 * M; i += 512) {
            float t = s_stat[i];  // pixel groups in a fixed order
#pragma unroll
            for (int q = 1; q < WN; ++q) t += s_stat[q * 2 * M + i];
            atomicAdd(&gstat[i], (double)t);  // fp64 sums of fp32 partials: exact, order-independent
        }
    }
}

// ---- launch side: instantiations and pass geometry -------------------------------------------------------------------------------------
namespace {
struct R3Variant {
    int M, ntw, ups, cap;  // output channels, N tiles per wave, halo units per step, N tiles per pass (WN * NTW)
};
// (KW = 2: a chunk has 4 steps and its halo units go out in 2 of them -- 3 per wave and step for <= 768 staged pixels)
// 128 output channels: 2 channel groups x 4 pixel groups; 64: 1 x 8.  NTW = 7 fits a 4 x 100 pass (25 tiles as 7 / 6 / 6 / 6), NTW = 8 power-of-two widths.
constexpr R3Variant R3_VARIANTS[] = {{128, 7, 1, 28}, {128, 8, 1, 32}, {128, 8, 2, 32}, {64, 4, 1, 32}, {64, 4, 2, 32}, {128, 7, 3, 28}, {128, 8, 3, 32}};  // (32 output channels x 4 tiles = 8 MFMAs per wave and step was measured: the per-step costs dominate, 324 vs 193 us for k_conv_igemm at 64 -> 32, 32 x 200)
constexpr int R3_NVAR = sizeof(R3_VARIANTS) / sizeof(R3_VARIANTS[0]);
constexpr int R3_LDS_FIXED = 1024 + 4 * 2 * 128 * 4 + 8 * 2048;  // dump slot, statistics, epilogue staging

struct R3Plan {
    int var, R, hppad, smem;
    double cost;
};
// rows per pass for one variant: fewest passes per image (a pass costs NTW tile times per step whatever it fills), then the tallest pass
R3Plan r3_plan_variant(int vi, int H, int W, int KW) {  // H, W: OUTPUT size
    const R3Variant& v = R3_VARIANTS[vi];
    R3Plan best{vi, 0, 0, 0, 1e30};
    const int xtn = KW == 3 ? 5 : 2;  // steps of a chunk that carry halo units
    if ((KW == 3) != (v.ups < 3)) return best;
    for (int R = 1; R <= H; ++R) {
        const int nt = (R * W + 15) / 16, hp = ((R + KW - 1) * (W + KW - 1) + 63) / 64 * 64;
        const int smem = 8 * hp * 16 + R3_LDS_FIXED + xtn * v.ups * 512 * 8;
        if (nt > v.cap || hp / 16 > 8 * xtn * v.ups || smem > 160 * 1024) break;
        const double cost = (double)((H + R - 1) / R) * v.ntw;
        if (cost <= best.cost) best = R3Plan{vi, R, hp, smem, cost};
    }
    return best;
}
R3Plan r3_plan(int M, int H, int W, int KW) {
    R3Plan best{-1, 0, 0, 0, 1e30};
    for (int vi = 0; vi < R3_NVAR; ++vi) {
        if (R3_VARIANTS[vi].M != M) continue;
        const R3Plan p = r3_plan_variant(vi, H, W, KW);
        if (p.R > 0 && p.cost < best.cost) best = p;
    }
    return best;
}
template <int MH, int WM, int NTW, int UPS>
int r3_launch(const R3Plan& pl, const void* x, int ldx, const void* wpk, void* out, int ldo, const float* bias, int relu, double* gstat, int Cin, int N, int H, int W,
              int Hi, int Wi, int KW, int pad, hipStream_t st) {
    static DevOnce attr_set;
    auto kern = &k_conv3x3_rows<MH, WM, NTW, UPS>;
    if (attr_set.need()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return OCRS_ERR_HIP;
        attr_set.done();
    }
    // whole passes per block, as evenly as the pass count allows (two / four passes = one image per CU at B = 256)
    const int total = N * ((H + pl.R - 1) / pl.R);
    int grid = total < kNumCU ? total : kNumCU;
    const int ppb = (total + grid - 1) / grid;
    grid = (total + ppb - 1) / ppb;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), pl.smem, st, (const bf16*)x, ldx, (const uint4*)wpk, (bf16*)out, ldo, bias, relu, gstat, Cin, N, H, W, pl.R,
                       pl.hppad, Hi, Wi, KW, pad);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}
}  // namespace

bool conv3x3_rows_supported(int ldx, int ldo, int Cin, int M, int Hi, int Wi, int Ho, int Wo, int KH, int KW, int padh, int padw, int dtype) {
    const int on = env_int("OCRS_CONV_ROWS", 1);  // 2: every shape the kernel can run (tests / measurements)
    // measured (tools/experiments/r4_conv_time.py, B = 256): ahead of k_conv3x3_c128 / k_conv_igemm by 10-30 % where a row is not a whole number
    // of 16-pixel tiles (100, 37, ...) or short (<= 64); behind at 128 / 192 pixels per row (568 / 425 vs 729 / 696 TF/s) -- those stay there
    if (on == 1 && KW == 3 && !(Wi % 16 != 0 || Wi <= 64)) return false;
    const bool k3 = KH == 3 && KW == 3 && padh == 1 && padw == 1 && Ho == Hi && Wo == Wi;
    const bool k2 = KH == 2 && KW == 2 && padh == padw && (padh == 0 || padh == 1) && Ho == Hi + 2 * padh - 1 && Wo == Wi + 2 * padw - 1 && M == 128;
    return on && dtype == 1 && (M == 128 || M == 64) && Cin % 64 == 0 && (k3 || k2) && ldx % 8 == 0 && ldo % 8 == 0 && (long)Hi * Wi * ldx < (1L << 30) &&
           r3_plan(M, Ho, Wo, KW).var >= 0;
}

int conv3x3_rows_launch(const void* x, int ldx, const void* wpk, void* out, int ldo, const float* bias, int relu, double* gstat, int Cin, int M, int N, int Hi, int Wi,
                        int Ho, int Wo, int KW, int pad, hipStream_t st) {
    const R3Plan pl = r3_plan(M, Ho, Wo, KW);
#define R3_GO(MH_, WM_, NTW_, UPS_) return r3_launch<MH_, WM_, NTW_, UPS_>(pl, x, ldx, wpk, out, ldo, bias, relu, gstat, Cin, N, Ho, Wo, Hi, Wi, KW, pad, st)
    switch (pl.var) {
        case 0: R3_GO(4, 2, 7, 1);
        case 1: R3_GO(4, 2, 8, 1);
        case 2: R3_GO(4, 2, 8, 2);
        case 3: R3_GO(4, 1, 4, 1);
        case 4: R3_GO(4, 1, 4, 2);
        case 5: R3_GO(4, 2, 7, 3);
        case 6: R3_GO(4, 2, 8, 3);
    }
#undef R3_GO
    return OCRS_ERR_ARG;
}

#if R3_DBG
extern "C" int ocrs_conv_rows_dbg(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_r3dbg), sizeof(long long) * 64) == hipSuccess ? 0 : 2; }
#endif
