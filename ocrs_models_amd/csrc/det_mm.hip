// "Everything on the matrix cores" forms of the DepthwiseConv block (reference ocrs_models/models.py:7-28) for the top U-Net levels
// (gfx950, bf16 storage, Cin, Cout in {8, 16, 32}): the depthwise 3x3 and the pointwise 1x1 are composed into ONE 3x3 implicit GEMM
// with the effective weight  Weff[o][(tap, c)] = Wpw[o][c] * Wdw[c][tap]  (rank-one per channel, built per launch from the fp32
// masters), so the 9-tap VALU loops, the depthwise-output tile and -- in the backward -- the du tensor of the separate kernels
// (k_dwpw_fwd; k_pw_bwd2 + k_dw_bwd) disappear.  The detection net is HBM-bound with the MFMA pipe ~4 % busy: spending
// 9x the pointwise FLOPs there is free, the VALU / LDS-issue time it removes was what kept those kernels at 3-3.5 TB/s.
//
// Backward of one block, per 16x32 (or 8x32) pixel tile, everything from ONE staged copy of (g, z) on the tile + 1-pixel ring and of x
// on the tile:
//     dz   = A * ghat + B * z + C                                  BatchNorm/ReLU backward (ghat optionally routed through MaxPool2d(2))
//     dx~[c][q]   = sum_{tap,o} Weff[o][(tap,c)] dz[q - off(tap)][o]          MFMA, K = 9 * Cout      (dgrad of both convs at once)
//     G_tap[c][o] = sum_q x~[q][c] dz[q - off(tap)][o]                        MFMA, K = pixels (LDS transpose reads)
//     dWpw[o][c]  = sum_tap Wdw[c][tap] G_tap[c][o],   dWdw[c][tap] = sum_o Wpw[o][c] G_tap[c][o]      (once per block, from G)
//     [+ the BatchNorm-backward sums of the block(s) that produced x:  sum ghat', sum ghat' x~  with ghat' = dx~ [x~ > 0]]
// HBM bytes per pixel: x (Cin) + z, g (Cout, 1.1-1.3x with the ring, mostly L2 hits) in, dx~ (Cin) out -- the 2 (Cin + Cout) of
// SURVEY.md 8(d); du is never formed.  All flushes are per-block partials reduced by ONE deterministic kernel (no atomics).
#include "det_common.h"
#include "det_rs.h"
#include <type_traits>

// ---- tile-shape / residency knobs (measurement builds: tools/build_variant.sh; the defaults are the measured optimum, profiles/README.md)
#ifndef OCRS_MM_TH
#define OCRS_MM_TH 8       // backward tile rows, Cin and Cout <= 16 (16 halves the ring re-reads but spills at 128 registers); per-shape overrides:
#endif
#ifndef OCRS_MM_TH_8_8
#define OCRS_MM_TH_8_8 12  // 12 rows: 476 of 512 threads hold a (z, g) item instead of 340 -- more bytes in flight from the same registers (650 -> 621 us)
#endif
#ifndef OCRS_MM_TH_16_8
#define OCRS_MM_TH_16_8 12  // (round 3, after the index-arithmetic diet freed registers: 890 -> 840 us, no spills in the FULL instantiations)
#endif
#ifndef OCRS_MM_TH_8_16
#define OCRS_MM_TH_8_16 12  // (946 -> 881 us; needs the single-buffered dgrad to fit 128 registers)
#endif
#ifndef OCRS_MM_TH_16_16
#define OCRS_MM_TH_16_16 OCRS_MM_TH
#endif
#ifndef OCRS_MM_TH_16_16_P
#define OCRS_MM_TH_16_16_P 12  // pooled gradient (the level-0 16 -> 16 block): round 2: 8 rows (1112 us) beat 12 (1151 us, 20 B of spills); round 3 (no spills after the index diet): 12 rows 1209 vs 1250 us
#endif
#define OCRS_MM_TH_OF(ci, co) ((ci) == 8 ? ((co) == 8 ? OCRS_MM_TH_8_8 : OCRS_MM_TH_8_16) : ((co) == 8 ? OCRS_MM_TH_16_8 : OCRS_MM_TH_16_16))
#ifndef OCRS_MM_SW
#define OCRS_MM_SW 1       // wave index in a scalar register
#endif
#ifndef OCRS_MM_EPI
#define OCRS_MM_EPI 1      // dgrad epilogue store addressing (0: per-lane 64-bit pixel index, 1: scalar row + lane offset, 2: 1 with both row pointers pinned scalar)
#endif
#ifndef OCRS_MM_BDB_EXCL
#define OCRS_MM_BDB_EXCL 1  // single-buffer the dgrad B fragments of the 12-row 8 -> 16 and pooled 16 -> 16 tiles (register diet)
#endif
#ifndef OCRS_MM_GPIPE
#define OCRS_MM_GPIPE 0     // weight-gradient phase: explicit one-step-ahead fragment prefetch (1) or straight-line code scheduled by hipcc (0)
#endif
#ifndef OCRS_MM_B3_16_8
#define OCRS_MM_B3_16_8 0   // backward, Cin = 16 -> Cout = 8: three blocks per CU (<= 85 registers: needs per-tile LDS statistics + single-buffered dgrad + 8 B of spills; measured 706 vs 673 us: off)
#endif
#ifndef OCRS_MM_C32_BPC
#define OCRS_MM_C32_BPC 1  // backward blocks per CU, Cin = 32 AND Cout = 32 (1: full register file, 16-row tiles: 144 / 506 us; 2: 8-row tiles, ~150 B of
                           // spills: 181 / 559 us).  Cin = 32, Cout = 16 always runs two blocks per CU (20-60 B of spills, 479 vs 529 us)
#endif
#ifndef OCRS_MM_S2G
#define OCRS_MM_S2G 1  // the producers' second BatchNorm-backward sum S2 = sum ghat' x~ from the weight-gradient accumulators at the flush (as k_rs_bwd does):
                       // sum_q dx~[q][c] x~[q][c] = sum_{tap,o} Weff[o][(tap,c)] G_tap[c][o] exactly (x~ [x~ > 0] = x~); the per-lane epilogue keeps S1 only.
                       // S2 then sums the UNROUNDED dx~ (an unbiased 2^-9 / sqrt(pixels) relative difference, the same as in det_rs.hip)
#endif
#ifndef OCRS_MM_C32_N256
#define OCRS_MM_C32_N256 1  // backward, Cin = 32 AND Cout = 32: TWO independent 256-thread workgroups per CU (256 registers each, 8-row tiles) instead of one
                            // 512-thread workgroup -- the same two waves per SIMD, but the barrier waits / commit / epilogue of one workgroup (35-40 % of a
                            // tile at one workgroup per CU: tools/runs/mm_prof.sh) run under the MFMA phases of the other
#endif
#ifndef OCRS_MM_C3216_N256
#define OCRS_MM_C3216_N256 1  // the same form for Cin = 32 -> Cout = 16 (which spills 12-44 B per lane at two 512-thread workgroups per CU): 382 -> 366 us at level 1
#endif
#ifndef OCRS_MM_C32_MERGE
#define OCRS_MM_C32_MERGE 1  // ... and both M tiles of the dgrad in ONE pass over the B fragments (the LDS reads of that phase halve: 16 more accumulator registers)
#endif
#ifndef OCRS_MM_C32_MERGE_BDB
#define OCRS_MM_C32_MERGE_BDB 0  // ... with the B fragments single-buffered (double-buffered, the merged pass spills 12-36 B per lane)
#endif
#ifndef OCRS_MM_BDB
#define OCRS_MM_BDB 1      // backward dgrad: double-buffer the B fragments across K chunks
#endif
#ifndef OCRS_MF_TH8
#define OCRS_MF_TH8 16     // forward tile rows, Cin = 8 (measured: 16 rows x 2 blocks/CU 300 us, 8 rows x 3 blocks/CU 329 us, 8 x 2: 356 us at level 0)
#endif
#ifndef OCRS_MF_TH16
#define OCRS_MF_TH16 8     // forward tile rows, Cin = 16 (16 spills)
#endif
#ifndef OCRS_MF_TH16_32
#define OCRS_MF_TH16_32 8  // forward tile rows, Cin = 16 -> Cout = 32
#endif
#ifndef OCRS_MF_TH32
#define OCRS_MF_TH32 8     // forward tile rows, Cin = 32
#endif
// forward tile rows must be a multiple of 8: the 8 waves share TH (row pair, column half) units (a 12-row build silently computes 8 rows --
// MfCfg asserts it)
#define OCRS_MF_TH_OF(ci, co, nst) ((nst) == 2 ? 8 : ((ci) == 32 ? OCRS_MF_TH32 : ((co) == 32 ? OCRS_MF_TH16_32 : ((ci) == 8 ? OCRS_MF_TH8 : OCRS_MF_TH16))))
#ifndef OCRS_MF_BPC16_8
#define OCRS_MF_BPC16_8 3  // forward blocks per CU, Cin = 16 -> Cout = 8 (80 registers: 490 -> 438 us at level 0; every other Cin = 16 shape spills at 85)
#endif
#ifndef OCRS_MF_BPC8
#define OCRS_MF_BPC8 2     // forward blocks per CU, Cin = 8
#endif

// floor-measurement builds (tools/r3): drop the global loads / the stores / the MFMA phases of k_mm_bwd
#ifndef OCRS_MM_NOLOAD
#define OCRS_MM_NOLOAD 0
#endif
#ifndef OCRS_MM_NOSTORE
#define OCRS_MM_NOSTORE 0
#endif
#ifndef OCRS_MM_NOCOMPUTE
#define OCRS_MM_NOCOMPUTE 0
#endif

namespace {

template <int C>
struct MmPitch {  // bf16 elements per pixel of an LDS tile: 16 / 32-byte rows are conflict-free as they are, 64-byte rows need the 16-byte pad
    static constexpr int V = (C == 32) ? 40 : C;
};

template <int CIN, int COUT, bool PPOOL = false>
struct MmCfg {
    static constexpr bool N256 = (OCRS_MM_C32_N256 && CIN == 32 && COUT == 32) || (OCRS_MM_C3216_N256 && CIN == 32 && COUT == 16);  // two 256-thread workgroups per CU with the full register file each
    static constexpr int NT = N256 ? 256 : 512, NW = NT / 64;          // 8 (4) waves
    // Cin = 32 (two M tiles, 16-24 more live registers than fit 128 without spilling -- and a scratch reload inside the pipelined loop waits for
    // every prefetched load): ONE block per CU with the full register file and 16-row tiles; everything else: two blocks per CU, 8-row tiles
    static constexpr bool T3 = OCRS_MM_B3_16_8 && CIN == 16 && COUT == 8 && !PPOOL;
    static constexpr int BPC = N256 ? 2 : ((CIN == 32 && COUT == 32) ? OCRS_MM_C32_BPC : (T3 ? 3 : 2));  // resident blocks per CU (launch bound: BPC * NW / 4 waves per SIMD)
    static constexpr int TW = 32;
    static constexpr int TH = N256 ? 8 : ((CIN == 32 || COUT == 32) ? (BPC == 1 ? 16 : 8) : ((PPOOL && CIN == 16 && COUT == 16) ? OCRS_MM_TH_16_16_P : OCRS_MM_TH_OF(CIN, COUT)));
    static constexpr bool MERGE = N256 && OCRS_MM_C32_MERGE;            // dgrad: both M tiles per pass
    static constexpr int TP = TW * TH;
    static constexpr bool T12P = CIN == 16 && COUT == 16 && TH == 12;  // 12-row 16 -> 16 tile (pooled or direct gradient): needs the two register diets below to fit 128
    static constexpr bool BDB = OCRS_MM_BDB && (!OCRS_MM_BDB_EXCL || (!(CIN == 8 && COUT == 16 && TH == 12) && !T12P)) && !T3 && !(N256 && OCRS_MM_C32_MERGE && !OCRS_MM_C32_MERGE_BDB);  // dgrad B fragments double-buffered across K chunks
    static constexpr int DW_ = TW + 2, DH_ = TH + 2, DP = DW_ * DH_;  // domain = tile + 1-pixel ring
    static constexpr int CGI = CIN / 8, CGO = COUT / 8;
    static constexpr int PD = MmPitch<COUT>::V, PX = MmPitch<CIN>::V;
    static constexpr int MT = (CIN + 15) / 16, NTO = (COUT + 15) / 16;
    static constexpr int KC = (9 * COUT + 31) / 32;                    // K chunks of the dgrad GEMM
    static constexpr int NPW = TP / 16 / NW;                           // dgrad N tiles (16 pixels) per wave
    static constexpr int KS = TP / 32;                                 // 32-pixel k-steps of the weight-gradient GEMM (one tile row each)
    static constexpr int TAPU = (COUT == 8) ? 5 : 9;                   // B-operand units: a tap, or a PAIR of taps when Cout = 8 (16 columns)
    static constexpr int NOWN = (TAPU == 9) ? 8 / NW : 1;              // weight-gradient units a wave owns (units 0..7; the ninth is shared)
    static constexpr int NGI = (DP * CGO + NT - 1) / NT;               // (z, g) items per thread
    static constexpr int NXI = (TP * CGI + NT - 1) / NT;               // x items per thread
    static constexpr int NWIN = (DH_ / 2) * (DW_ / 2);                 // 2x2 windows of the (window-aligned) domain
    static constexpr int NWI = (NWIN * 2 * CGO + NT - 1) / NT;       // pooled launches: (window, 4-channel quad) items per thread
    static_assert(DH_ % 2 == 0 && NT % (2 * CGO) == 0 && 4 * NWI <= 16, "pooled item layout");
    // LDS (bytes)
    static constexpr int OFF_D = 0;
    static constexpr int OFF_X = (OFF_D + DP * PD * 2 + 63) & ~63;
    static constexpr int OFF_WF = (OFF_X + TP * PX * 2 + 64 + 63) & ~63;  // +64: the Cin = 8 transpose reads run 16 bytes past the last pixel
    static constexpr int OFF_PAR = OFF_WF + MT * KC * 64 * 16;
    static constexpr int PAR_FLOATS = 3 * CIN + 6 * COUT + 9 * CIN + COUT * CIN + NW * 2 * MT * 16;  // trx | bn | coef | wdw [c][9] | wpw [o][c] | stats slots
    static constexpr int TILE_BYTES = OFF_PAR + PAR_FLOATS * 4;
    static constexpr int SLOT_FLOATS = (NW * NOWN * MT * NTO + NW) * 256 + NW * 2 * MT * 16;  // flush: G slots (own units | shared sub-tile) + stats slots
    static constexpr int SMEM = TILE_BYTES > SLOT_FLOATS * 4 ? TILE_BYTES : SLOT_FLOATS * 4;
    static constexpr int PART = COUT * CIN + 9 * CIN + 2 * CIN;        // floats per block partial: dWpw [COUT][CIN] | dWdw [CIN][9] | sums [2][CIN]
};

__device__ __forceinline__ f32x4 mfma16(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// half hf (channels 4*hf .. 4*hf+3) of a raw 8-channel bf16 vector
__device__ __forceinline__ void half4(const Raw8<bf16>& r, int hf, float (&v)[4]) {
    const unsigned a = hf ? r.a.z : r.a.x, b = hf ? r.a.w : r.a.y;
    v[0] = __uint_as_float(a << 16); v[1] = __uint_as_float(a & 0xffff0000u);
    v[2] = __uint_as_float(b << 16); v[3] = __uint_as_float(b & 0xffff0000u);
}
// 4 bf16 to LDS with the packing hidden from the optimiser (see store8_opaque)
__device__ __forceinline__ void st4bf(bf16* p, const float (&v)[4]) {
    unsigned lo, hi;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(v[0]), "v"(v[1]));
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(v[2]), "v"(v[3]));
    *reinterpret_cast<uint2*>(p) = make_uint2(lo, hi);
}

// ---- prefetch loads the compiler does not see (FULL kernels).  hipcc's own `s_waitcnt vmcnt` in front of the first use of a prefetched
// vector must hold on every control-flow path, and with stores issued between the load and its use (the previous tile's epilogue) any
// divergent branch or loop entry degrades it to "wait until (almost) nothing is outstanding": once per tile every wave then waits for the
// acknowledgement of its own stores.  Here the load is an opaque asm statement and the wait is written by hand with the exact number of
// younger operations (VMEM operations retire in order); the "+v" operand ties every use of the vector behind the wait.
// HAZARD the build checks for (tools/check_opaque_loads.py, run by ocrs_models_amd/build.py): the register allocator must not spill or copy a
// destination vector between the load and its wait -- it believes the asm statement has completed -- which would store stale bytes.  The
// checker disassembles every FULL kernel and fails the build if a register written by one of these loads is ever the source of a
// scratch store or of a plain v_mov / v_accvgpr_write (with one prefetch set hipcc keeps them in place; two sets at the 128-register cap
// do get spilled -- and an AGPR destination, which would rule spills out, makes hipcc split the 128 registers 64 | 64 and spill far more).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 gload16_opaque(const void* p) {
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(r) : "v"(p) : "memory");
    return r;
}
template <int N>
__device__ __forceinline__ void wait_vm_tied(u32x4& r) {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(N) : "memory");
}
// wait for loads 0 .. NL-1 (issue order) of a prefetch set; NS operations (stores) were issued after the last of them
template <int NL, int NS, int K = 0>
__device__ __forceinline__ void static_for_wait(u32x4* v) {
    if constexpr (K < NL) {
        wait_vm_tied<NS + NL - 1 - K>(v[K]);
        static_for_wait<NL, NS, K + 1>(v);
    }
}
__device__ __forceinline__ Raw8<bf16> raw8_of(const u32x4& v) {
    Raw8<bf16> r;
    r.a = make_uint4(v.x, v.y, v.z, v.w);
    return r;
}

// forward: Cin = 8 needs < 80 registers and ~10 KB of LDS: three blocks per CU (more bytes in flight: these launches are latency-bound)
template <int CINB, int COUT>
constexpr int mm_fwd_lb() { return CINB == 8 ? 2 * OCRS_MF_BPC8 : ((CINB == 16 && COUT == 8) ? 2 * OCRS_MF_BPC16_8 : 4); }
// the last-workgroup finalisation of the forward statistics is compiled into every shape but Cin = 16 -> Cout = 8: at that shape's 80-register cap (three
// blocks per CU) the extra code made hipcc spill a 64-bit index pair that the prefetch issue reloads from scratch -- behind an s_waitcnt vmcnt(0), i.e.
// behind the acknowledgement of the previous tile's stores (397 -> 454-472 us at level 0); the launcher runs k_bn_finalize_parts for it instead
template <int CINB, int COUT>
constexpr bool mm_fwd_fink() { return !(CINB == 16 && COUT == 8); }
template <int CIN, int COUT, bool PPOOL>
constexpr int mm_bwd_lb() { return MmCfg<CIN, COUT, PPOOL>::BPC * MmCfg<CIN, COUT, PPOOL>::NW / 4; }  // minimum waves per SIMD (2 = one 512-thread block per CU, 4 = two)
}  // namespace

// ----------------------------------------------------------------------------------------------------------------------------------
// backward
// ----------------------------------------------------------------------------------------------------------------------------------
// FULL: see k_mm_fwd (every tile inside the image, direct gradient: unconditional stores, opaque prefetch loads with hand-written waits,
// first tile peeled).
template <int CIN, int COUT, bool PPOOL, bool G2, bool STATS, bool FULL = false>
__global__ __launch_bounds__((MmCfg<CIN, COUT, PPOOL>::NT), (mm_bwd_lb<CIN, COUT, PPOOL>())) void k_mm_bwd(Src2<bf16> x, const float* __restrict__ tra, const float* __restrict__ trb,
                                                   const float* __restrict__ wdw /*[.][9], already offset to this launch's first channel*/,
                                                   const float* __restrict__ wpw /*[COUT][ldw], already offset*/, int ldw,
                                                   const bf16* __restrict__ g1, const bf16* __restrict__ g2, const bf16* __restrict__ z,
                                                   const float* __restrict__ bn, const float* __restrict__ coef, bf16* __restrict__ gxa,
                                                   bf16* __restrict__ gxb, float* __restrict__ ws, Tiling2 tg, BnFin fin, BwdLast bl) {
    using C = MmCfg<CIN, COUT, PPOOL>;
    constexpr int NT = C::NT, TW = C::TW, TH = C::TH, TP = C::TP, DW_ = C::DW_, DP = C::DP, CGI = C::CGI, CGO = C::CGO, PD = C::PD, PX = C::PX;
    constexpr int MT = C::MT, NTO = C::NTO, KC = C::KC, NPW = C::NPW, KS = C::KS;
    static_assert(!(FULL && PPOOL), "pooled launches have border tiles by construction (origin shift)");
    extern __shared__ __attribute__((aligned(64))) char smem[];
    bf16* tileD = reinterpret_cast<bf16*>(smem + C::OFF_D);   // [DP][PD]  dz on the domain (0 outside the image)
    bf16* tileX = reinterpret_cast<bf16*>(smem + C::OFF_X);   // [TP][PX]  x~ on the tile (0 outside the image)
    uint4* s_wf = reinterpret_cast<uint4*>(smem + C::OFF_WF); // [MT][KC][64] effective-weight A fragments
    float* s_trx = reinterpret_cast<float*>(smem + C::OFF_PAR);  // [CIN/8][3][8]
    float* s_bn = s_trx + 3 * CIN;                              // [3][COUT]
    float* s_cf = s_bn + 3 * COUT;                              // [3][COUT]
    float* s_w9 = s_cf + 3 * COUT;                              // [CIN][9]
    float* s_wp = s_w9 + 9 * CIN;                               // [COUT][CIN]
    const int H = tg.H, W = tg.W;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15;
#if OCRS_MM_SW
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar register: wave-dependent branches and offsets become scalar code)
#else
    const int wave = tid >> 6;
#endif

    // ---- prologue: parameters, effective-weight fragments
    fill_tr8(s_trx, x, tra, trb, CIN, tid);
    for (int i = tid; i < 3 * COUT; i += NT) s_bn[i] = bn[i];
    if (fin.gsum) {
        bn_fin_coef(fin, COUT, s_cf, tid, NT, blockIdx.x == 0);
    } else {
        for (int i = tid; i < 3 * COUT; i += NT) s_cf[i] = coef[i];
    }
    for (int i = tid; i < 9 * CIN; i += NT) s_w9[i] = wdw[i];
    for (int i = tid; i < COUT * CIN; i += NT) s_wp[i] = wpw[(i / CIN) * ldw + (i % CIN)];
    {   // zero both tiles once (pad columns / the slack behind tileX stay zero)
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        for (int i = tid; i < C::OFF_WF / 16; i += NT) reinterpret_cast<uint4*>(smem)[i] = z4;
    }
    __syncthreads();
    for (int f = tid; f < MT * KC * 64; f += NT) {
        const int l = f & 63, kc = (f >> 6) % KC, mt = (f >> 6) / KC;
        const int m = (FULL && CIN == 8) ? (l & 7) : mt * 16 + (l & 15);  // FULL, CIN = 8: M rows 8..15 duplicate rows 0..7 (unconditional stores)
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kc * 32 + (l >> 4) * 8 + j, tap = k / COUT, o = k % COUT;
            v[j] = (m < CIN && tap < 9) ? s_w9[m * 9 + tap] * s_wp[o * CIN + m] : 0.f;
        }
        s_wf[f] = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
    }
    __syncthreads();

    // ---- tile-invariant item descriptors
    // (z, g) items.  Direct gradient: one (domain pixel, 8-channel group) each.  Pooled gradient: one (2x2 window, 4-channel quad) each --
    // the domain of a pooled launch is window-aligned (tile origins sit at odd coordinates: org = tile * T - 1), so routing needs no
    // neighbours; quads instead of groups because a window item is four pixels of work: with groups only 170 of the 512 threads of a
    // 16-channel launch held one and the other waves idled at the barrier (the routing is the heaviest VALU phase of these launches).
    const int cgo = tid % CGO;
    constexpr int NQ = 2 * CGO;        // channel quads per pixel
    const int cq4 = (tid % NQ) * 4;    // this thread's quad (the same in every round: NT % NQ == 0)
    int gi_dyx[PPOOL ? C::NWI : C::NGI];  // dy | dx << 16 (domain coordinates; window top-left when pooled)
    if constexpr (!PPOOL) {
#pragma unroll
        for (int j = 0; j < C::NGI; ++j) {
            const int d = (tid + j * NT) / CGO, dy = d / DW_, dx = d - dy * DW_;
            gi_dyx[j] = dy | (dx << 16);
        }
    } else {
#pragma unroll
        for (int j = 0; j < C::NWI; ++j) {
            const int wd = (tid + j * NT) / NQ, wy = wd / (DW_ / 2), wx = wd - wy * (DW_ / 2);
            gi_dyx[j] = (2 * wy) | ((2 * wx) << 16);
        }
    }
    // x items: (tile pixel, 8-channel group)
    const int cgi = tid % CGI;
    const bool xi_a = cgi * 8 < x.Ca;
    const bf16* xi_base = xi_a ? x.a + cgi * 8 : x.b + (cgi * 8 - x.Ca);
    const int xi_pitch = xi_a ? x.Ca : x.Cb;
    constexpr int ORG = PPOOL ? -1 : 0;  // tile origin shift

    // ---- software pipeline state: raw vectors of the NEXT tile
    constexpr int NG = PPOOL ? 1 : C::NGI, NP = PPOOL ? C::NWI : 1;
    // one array, issue order (FULL waits walk it): [j][z | g1 | g2] for the NG (z, g) items, then the x items
    constexpr int GW = G2 ? 3 : 2, NLOADS = NG * GW + C::NXI;
    // prefetch depth: ONE vector set.  (Round 3 tried two sets -- the loads of tile t+2 issued behind commit(t): at the 128-register cap hipcc
    // spills the in-flight destination vectors, which an opaque load cannot survive; tools/check_opaque_loads.py catches exactly that.)
    constexpr int PFD = 1;
    u32x4 pf[PFD][NLOADS];
    constexpr int NSTORE = NPW * MT;  // FULL: dL/dx stores every wave issues per tile, all unconditional
    uint2 zp[4 * NP], gp1[NP], gp2[G2 ? NP : 1];  // pooled: raw quads of the window's four z and of the pooled gradient(s)
    unsigned okg_[PFD], okx_[PFD];  // validity bits of each set
#pragma unroll
    for (int q = 0; q < PFD; ++q) okg_[q] = okx_[q] = 0;
    auto ld16 = [&](const bf16* p) -> u32x4 {
        if constexpr (FULL) {
            return gload16_opaque(p);
        } else {
            const uint4 q = *reinterpret_cast<const uint4*>(p);
            return (u32x4){q.x, q.y, q.z, q.w};
        }
    };
    auto issue = [&](const TileOrg& o, auto SET) {
        constexpr int SI = decltype(SET)::value;
        unsigned okg = 0, okx = 0;
        const int h00 = o.h0 + ORG - 1, w00 = o.w0 + ORG - 1;  // image coordinates of the domain's corner pixel (may lie outside)
        // (pixel indices fit 32 bits: N (H + 2) (W + 2) < 2^31 is checked by the launcher; only the final element offset is 64-bit)
        const int corner = (o.n * H + h00) * W + w00;
        const bf16* zb = z + (long)corner * COUT;
        // FULL: a tile whose whole domain lies inside the image (all but the outermost ring of tiles: >= 85 % of them from 512^2 up) skips
        // the per-item bounds tests and address selects -- a scalar branch, both sides issue the same loads in the same order
        const bool inside = FULL && h00 >= 0 && w00 >= 0 && h00 + C::DH_ <= H && w00 + DW_ <= W;
        if constexpr (!PPOOL) {
            const bf16* g1b = g1 + (long)corner * COUT;
            const bf16* g2b = (G2 ? g2 : g1) + (long)corner * COUT;
            auto items = [&](auto INSIDE) {
#pragma unroll
                for (int j = 0; j < C::NGI; ++j) {
                    const int dy = gi_dyx[j] & 0xffff, dx = gi_dyx[j] >> 16, h = h00 + dy, w = w00 + dx;
                    const bool it_ok = (j + 1) * NT <= DP * CGO || tid + j * NT < DP * CGO;  // (a compile-time `true` for all but the last round)
                    const bool ok = it_ok && (decltype(INSIDE)::value || ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W));
                    const int goff = (dy * W + dx) * COUT + cgo * 8;
                    const bool ld = ok && !OCRS_MM_NOLOAD;
                    pf[SI][j * GW + 0] = ld16(ld ? zb + goff : z);
                    pf[SI][j * GW + 1] = ld16(ld ? g1b + goff : g1);
                    if constexpr (G2) pf[SI][j * GW + 2] = ld16(ld ? g2b + goff : g2);
                    okg |= ok ? 1u << j : 0u;
                }
            };
            if (inside) items(std::true_type{});
            else items(std::false_type{});
        } else {
            const int Hp = H >> 1, Wp = W >> 1;
#pragma unroll
            for (int j = 0; j < C::NWI; ++j) {
                const bool it_ok = C::NWIN * NQ % NT == 0 || tid + j * NT < C::NWIN * NQ;
                const int dy = gi_dyx[j] & 0xffff, dx = gi_dyx[j] >> 16, h = h00 + dy, w = w00 + dx;  // top-left pixel of the window (even coordinates)
                const int goff = (dy * W + dx) * COUT + cq4;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool ok = it_ok && (unsigned)(h + (k >> 1)) < (unsigned)H && (unsigned)(w + (k & 1)) < (unsigned)W;
                    zp[4 * j + k] = *reinterpret_cast<const uint2*>((ok && !OCRS_MM_NOLOAD) ? zb + goff + ((k >> 1) * W + (k & 1)) * COUT : z);
                    okg |= ok ? 1u << (4 * j + k) : 0u;
                }
                const int ph = h >> 1, pw = w >> 1;
                const bool gv = it_ok && h >= 0 && w >= 0 && ph < Hp && pw < Wp;  // floor mode: the last odd row / column is in no window
                const long pp = (o.n * Hp + ph) * Wp + pw;
                gp1[j] = *reinterpret_cast<const uint2*>((gv && !OCRS_MM_NOLOAD) ? g1 + pp * COUT + cq4 : g1);
                if constexpr (G2) gp2[j] = *reinterpret_cast<const uint2*>((gv && !OCRS_MM_NOLOAD) ? g2 + pp * COUT + cq4 : g2);
                okg |= gv ? 1u << (16 + j) : 0u;
            }
        }
        const int tb = (o.n * H + (o.h0 + ORG)) * W + (o.w0 + ORG);
        auto xitems = [&](auto INSIDE) {
#pragma unroll
            for (int j = 0; j < C::NXI; ++j) {
                const int p = (tid + j * NT) / CGI, ty = p / TW, tx = p % TW;
                const int h = o.h0 + ORG + ty, w = o.w0 + ORG + tx;
                const bool it_ok = (j + 1) * NT <= TP * CGI || tid + j * NT < TP * CGI;
                const bool ok = it_ok && (decltype(INSIDE)::value || ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W));
                pf[SI][NG * GW + j] = ld16((ok && !OCRS_MM_NOLOAD) ? xi_base + (long)(tb + ty * W + tx) * xi_pitch : xi_base);
                okx |= ok ? 1u << j : 0u;
            }
        };
        if (FULL && inside) xitems(std::true_type{});
        else xitems(std::false_type{});
        okg_[SI] = okg;
        okx_[SI] = okx;
    };

    // ---- dgrad (dx~) MFMA bookkeeping: B-fragment offset of chunk kc for this lane.  k = kc*32 + (lane>>4)*8 .. +7 lies inside ONE tap;
    // the tap is a compile-time function of kc and a 1- or 2-bit selector from the lane id (selects between literal offsets).
    const int kg = lane >> 4;
    auto tap_off = [](int tap) constexpr -> int { return tap < 9 ? ((2 - tap / 3) * DW_ + (2 - tap % 3)) * PD : 0; };
    auto boff_of = [&](int kc, bool& valid) -> int {
        if constexpr (COUT == 32) {
            valid = true;
            return tap_off(kc) + kg * 8;
        } else if constexpr (COUT == 16) {
            const int t0 = 2 * kc, t1 = 2 * kc + 1;
            valid = (kg < 2) ? t0 < 9 : t1 < 9;
            return ((kg < 2) ? tap_off(t0) : tap_off(t1)) + (kg & 1) * 8;
        } else {
            const int t0 = 4 * kc;
            valid = t0 + kg < 9;
            const int a = (kg & 1) ? tap_off(t0 + 1) : tap_off(t0), b = (kg & 1) ? tap_off(t0 + 3) : tap_off(t0 + 2);
            return (kg & 2) ? b : a;
        }
    };
    // ---- weight-gradient (G) bookkeeping.  Transpose-read lane geometry: pixel row prow (+16 for the second half), 4 channels at pcol.
    const int prow = 4 * (lane >> 4) + ((lane & 15) >> 2), pcol = (lane & 3) * 4;
    // unit -> per-lane B offset relative to pixel (ks, 0) of the tile, i.e. element offset of tileD[((ks + 2 - ky) * DW_ + prow + 2 - kx)][...]
    auto unit_off = [&](int u) -> int {
        if constexpr (COUT == 8) {
            const int sel = (lane & 3) >> 1;
            int tap = 2 * u + sel;
            tap = tap > 8 ? 8 : tap;  // unit 4 = tap 8 twice (columns 8..15 are ignored at the flush)
            const int ky = tap / 3, kx = tap - ky * 3;
            return ((2 - ky) * DW_ + prow + 2 - kx) * PD + (lane & 1) * 4;
        } else {
            const int ky = u / 3, kx = u - ky * 3;
            return ((2 - ky) * DW_ + prow + 2 - kx) * PD + pcol;
        }
    };
    // own unit / k-step range of this wave, and its share of the last unit: ONE (M tile, N tile) sub-tile of it over a range of k-steps
    // (a whole-unit share would need a second full accumulator set: 16 more registers at 32 x 32 channels)
    constexpr int NSUB = MT * NTO, NKR = C::NW / NSUB;
    static_assert(C::NW % NSUB == 0 && KS >= NKR, "shared-unit split");
    const int u_own = (C::TAPU == 9) ? wave : (wave & 3);
    const int ks_own0 = (C::TAPU == 9) ? 0 : (wave >> 2) * (KS / 2), ks_own1 = (C::TAPU == 9) ? KS : ks_own0 + KS / 2;
    const int sh_a = (wave % NSUB) % MT, sh_b = (wave % NSUB) / MT, sh_k0 = (wave / NSUB) * KS / NKR, sh_k1 = (wave / NSUB + 1) * KS / NKR;
    constexpr int NOWN = C::NOWN;  // own units of this wave: u_own, u_own + NW, ...
    int off_own[NOWN];
#pragma unroll
    for (int j = 0; j < NOWN; ++j) off_own[j] = unit_off(u_own + j * C::NW);
    const int off_sh = unit_off(C::TAPU - 1) + sh_b * 16;
    f32x4 accO[NOWN][MT][NTO], accS = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NOWN; ++j)
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NTO; ++b) accO[j][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // producers' BatchNorm-backward sums: per-lane register accumulators (one M tile), or -- with two M tiles, where 16 more live registers
    // spill -- per-tile sums added to this wave's own LDS slots (single writer, fixed order: deterministic)
    constexpr bool STL = STATS && (MT == 2 || C::T12P || C::T3);
    float* s_st = s_wp + COUT * CIN;  // [wave][2][MT*16] (STL only)
    float st1[(STATS && !STL) ? MT : 1][4], st2[(STATS && !STL) ? MT : 1][4];
    if constexpr (STATS && !STL) {
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) st1[a][i] = st2[a][i] = 0.f;
    }
    if constexpr (STL) {
        for (int i = tid; i < C::NW * 2 * MT * 16; i += NT) s_st[i] = 0.f;  // (ordered before its first use by the tile loop's barriers)
    }

#ifdef OCRS_MM_PROF  // (debug build: per-phase cycle totals of block 0 / thread 0 land in the first bytes of gxa)
    unsigned long long pt[5] = {0, 0, 0, 0, 0}, pc = __builtin_readcyclecounter();
#define MM_MARK(i) { const unsigned long long now_ = __builtin_readcyclecounter(); pt[i] += now_ - pc; pc = now_; }
#else
#define MM_MARK(i)
#endif
    TileSched ts(tg.ntiles);
    TileIter<TW, TH> tit(tg, ts.first < ts.end ? ts.first : 0, ts.step);  // (points at the newest tile whose loads were issued)
    TileOrg orgs[PFD];
    orgs[0] = tit.org();
    using I0 = std::integral_constant<int, 0>;
    if (ts.first < ts.end) issue(orgs[0], I0{});
    // KIND: 0 = first tile of the block, 2 = steady state -- the number of operations issued after this tile's loads differs
    auto tile_body = [&](long t, auto KIND, auto SET) __attribute__((always_inline)) {
        constexpr int SI = decltype(SET)::value, KI = decltype(KIND)::value;
        const TileOrg org = orgs[SI];
        const unsigned okg = okg_[SI], okx = okx_[SI];
        if constexpr (FULL) {
            // the hand-written waits.  Issue order: L0 | L1 S0 | L2 S1 ... (loads of the next tile behind the commit, then this tile's NSTORE
            // epilogue stores) -> operations younger than Lt at the top of tile t: 0 for the block's first tile, NSTORE afterwards
            constexpr int YOUNGER = KI == 0 ? 0 : NSTORE;
            static_for_wait<NLOADS, YOUNGER>(pf[SI]);
        }
        MM_MARK(4)
        // ================= phase 1: commit the prefetched tile: dz -> tileD, x~ -> tileX =================
        // Four channels at a time (the five per-channel coefficient vectors of a half are 20 registers instead of 40; 8-byte LDS stores).
        if constexpr (!PPOOL) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int c4 = cgo * 8 + hf * 4;
                const f32x4 bs = *reinterpret_cast<const f32x4*>(s_bn + c4), bt = *reinterpret_cast<const f32x4*>(s_bn + COUT + c4);
                const f32x4 ca = *reinterpret_cast<const f32x4*>(s_cf + c4), cb = *reinterpret_cast<const f32x4*>(s_cf + COUT + c4),
                            cc = *reinterpret_cast<const f32x4*>(s_cf + 2 * COUT + c4);
#pragma unroll
                for (int j = 0; j < C::NGI; ++j) {
                    const int it = tid + j * NT;
                    if (DP * CGO % NT == 0 || it < DP * CGO) {
                        float dz[4] = {0.f, 0.f, 0.f, 0.f};
                        if (okg & (1u << j)) {
                            float zv[4], ga[4];
                            half4(raw8_of(pf[SI][j * GW + 0]), hf, zv);
                            half4(raw8_of(pf[SI][j * GW + 1]), hf, ga);
                            if constexpr (G2) {
                                float gb[4];
                                half4(raw8_of(pf[SI][j * GW + 2]), hf, gb);
#pragma unroll
                                for (int i = 0; i < 4; ++i) ga[i] += gb[i];
                            }
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float gh = fmaf(zv[i], bs[i], bt[i]) > 0.f ? ga[i] : 0.f;
                                dz[i] = fmaf(ca[i], gh, fmaf(cb[i], zv[i], cc[i]));
                            }
                        }
                        st4bf(tileD + (it / CGO) * PD + c4, dz);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            const f32x4 bs = *reinterpret_cast<const f32x4*>(s_bn + cq4), bt = *reinterpret_cast<const f32x4*>(s_bn + COUT + cq4);
            const f32x4 ca = *reinterpret_cast<const f32x4*>(s_cf + cq4), cb = *reinterpret_cast<const f32x4*>(s_cf + COUT + cq4),
                        cc = *reinterpret_cast<const f32x4*>(s_cf + 2 * COUT + cq4);
#pragma unroll
            for (int j = 0; j < C::NWI; ++j) {
                const int it = tid + j * NT;
                if (C::NWIN * NQ % NT == 0 || it < C::NWIN * NQ) {
                    float gs[4], dz[4][4];
                    gs[0] = __uint_as_float(gp1[j].x << 16); gs[1] = __uint_as_float(gp1[j].x & 0xffff0000u);
                    gs[2] = __uint_as_float(gp1[j].y << 16); gs[3] = __uint_as_float(gp1[j].y & 0xffff0000u);
                    if constexpr (G2) {
                        gs[0] += __uint_as_float(gp2[j].x << 16); gs[1] += __uint_as_float(gp2[j].x & 0xffff0000u);
                        gs[2] += __uint_as_float(gp2[j].y << 16); gs[3] += __uint_as_float(gp2[j].y & 0xffff0000u);
                    }
                    const bool gv = (okg >> (16 + j)) & 1u;
                    // first maximum of the window in post-ReLU space (row-major scan, strict > keeps the first), must be > 0: the pooled
                    // value passes the ReLU.  When gv holds all four pixels lie inside the image.
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float zk[4], mk[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const unsigned w2 = (i < 2) ? zp[4 * j + k].x : zp[4 * j + k].y;
                            zk[k] = (i & 1) ? __uint_as_float(w2 & 0xffff0000u) : __uint_as_float(w2 << 16);
                            mk[k] = max_lo(fmaf(zk[k], bs[i], bt[i]), 0.f);
                        }
                        float best = mk[0];
                        int sel = 0;
#pragma unroll
                        for (int k = 1; k < 4; ++k) {
                            const bool gt = mk[k] > best;
                            best = gt ? mk[k] : best;
                            sel = gt ? k : sel;
                        }
                        const float gsel = (gv && best > 0.f) ? gs[i] : 0.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const bool ok = (okg >> (4 * j + k)) & 1u;
                            const float gh = sel == k ? gsel : 0.f;
                            dz[k][i] = ok ? fmaf(ca[i], gh, fmaf(cb[i], zk[k], cc[i])) : 0.f;
                        }
                    }
                    const int wd = it / NQ, wy = wd / (DW_ / 2), wx = wd - wy * (DW_ / 2);
#pragma unroll
                    for (int k = 0; k < 4; ++k) st4bf(tileD + ((2 * wy + (k >> 1)) * DW_ + 2 * wx + (k & 1)) * PD + cq4, dz[k]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const float* tp = s_trx + cgi * 24 + hf * 4;
            const f32x4 sc = *reinterpret_cast<const f32x4*>(tp), sh = *reinterpret_cast<const f32x4*>(tp + 8), lo = *reinterpret_cast<const f32x4*>(tp + 16);
#pragma unroll
            for (int j = 0; j < C::NXI; ++j) {
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (okx & (1u << j)) {
                    half4(raw8_of(pf[SI][NG * GW + j]), hf, v);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = max_lo(fmaf(v[i], sc[i], sh[i]), lo[i]);
                }
                if (TP * CGI % NT == 0 || tid + j * NT < TP * CGI) st4bf(tileX + ((tid + j * NT) / CGI) * PX + cgi * 8 + hf * 4, v);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        MM_MARK(0)
        if (t + ts.step < ts.end) {
            tit.next();
            orgs[SI] = tit.org();
            issue(orgs[SI], SET);
        }
        lds_barrier();
        MM_MARK(1)
        // ================= phase 2a: dx~ = Weff * dz (shifted), MFMA; epilogue: store + the producers' BatchNorm-backward sums =================
        // One M tile (16 input channels) at a time: the second pass re-reads the B fragments from LDS instead of holding 2x the accumulators.
        {
            // N tile a of this wave = 16 consecutive pixels of ONE tile row: row and first column are wave-uniform (scalar registers), the
            // lane only adds (lane & 15) -- the per-store 64-bit address arithmetic was a quarter of the kernel's vector instructions
            int pbase[NPW];
#pragma unroll
            for (int a = 0; a < NPW; ++a) {
                const int k = wave * NPW + a, ty = k / (TW / 16), tx0 = (k % (TW / 16)) * 16;
                pbase[a] = (ty * DW_ + tx0 + l15) * PD;
            }
            const int tb = (org.n * H + (org.h0 + ORG)) * W + (org.w0 + ORG);
            constexpr int NBM = C::MERGE ? MT : 1;  // M tiles per pass over the B fragments
#pragma unroll
            for (int b0 = 0; b0 < MT; b0 += NBM) {
                f32x4 accm[NBM][NPW];
#pragma unroll
                for (int bm = 0; bm < NBM; ++bm)
#pragma unroll
                    for (int a = 0; a < NPW; ++a) accm[bm][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
                uint4 bcur[NPW], bnxt[NPW];
                auto load_b = [&](uint4 (&dst)[NPW], int kc) {
                    bool bv;
                    const int bo = boff_of(kc, bv);
#pragma unroll
                    for (int a = 0; a < NPW; ++a) {
                        dst[a] = *reinterpret_cast<const uint4*>(tileD + pbase[a] + bo);
                        if (!bv) dst[a] = make_uint4(0, 0, 0, 0);
                    }
                };
                if constexpr (OCRS_MM_NOCOMPUTE) {
                } else if constexpr (C::BDB) {  // B fragments double-buffered across K chunks
                    load_b(bcur, 0);
#pragma unroll
                    for (int kc = 0; kc < KC; ++kc) {
                        uint4 wf[NBM];
#pragma unroll
                        for (int bm = 0; bm < NBM; ++bm) wf[bm] = s_wf[((b0 + bm) * KC + kc) * 64 + lane];
                        if (kc + 1 < KC) load_b(bnxt, kc + 1);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int bm = 0; bm < NBM; ++bm)
#pragma unroll
                            for (int a = 0; a < NPW; ++a) accm[bm][a] = mfma16(wf[bm], bcur[a], accm[bm][a]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int a = 0; a < NPW; ++a) bcur[a] = bnxt[a];
                    }
                } else {  // (8 fewer live registers: what lets the 12-row tile of the 8 -> 16 channel shape fit 128)
#pragma unroll
                    for (int kc = 0; kc < KC; ++kc) {
                        uint4 wf[NBM];
#pragma unroll
                        for (int bm = 0; bm < NBM; ++bm) wf[bm] = s_wf[((b0 + bm) * KC + kc) * 64 + lane];
                        load_b(bcur, kc);
#pragma unroll
                        for (int bm = 0; bm < NBM; ++bm)
#pragma unroll
                            for (int a = 0; a < NPW; ++a) accm[bm][a] = mfma16(wf[bm], bcur[a], accm[bm][a]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
              for (int bm = 0; bm < NBM; ++bm) {
                const int b = b0 + bm;
                f32x4 (&acc)[NPW] = accm[bm];
                const int m0 = b * 16 + (lane >> 4) * 4;
                const int ms = (FULL && CIN == 8) ? (m0 & 7) : m0;  // (duplicate rows store to the address of the row they duplicate)
                const bool in_a = ms < x.Ca;
                const int voff = l15 * (in_a ? x.Ca : x.Cb) + (in_a ? ms : ms - x.Ca);  // element offset of this lane's 4 channels from the N tile's first pixel
                float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int a = 0; a < NPW; ++a) {
                    const int k = wave * NPW + a, ty = k / (TW / 16), tx0 = (k % (TW / 16)) * 16;
                    const bool pv = FULL || (!OCRS_MM_NOSTORE && (unsigned)(org.h0 + ORG + ty) < (unsigned)H && (unsigned)(org.w0 + ORG + tx0 + l15) < (unsigned)W);
                    const long srow = tb + ty * W + tx0;  // (scalar)
                    if constexpr (FULL) {  // unconditional store, no divergent branch around it
                        const f32x4 v = acc[a];
                        bf16* dst = (in_a ? gxa + srow * x.Ca : gxb + srow * x.Cb) + voff;
                        // (H need not be a multiple of the tile height: an N tile is one tile row -- rows below the image go to the block's
                        //  scratch lines behind the partials in ws, a wave-uniform select)
                        const bool row_ok = org.h0 + ty < H;
                        dst = row_ok ? dst : reinterpret_cast<bf16*>(ws + (long)gridDim.x * C::PART + (long)blockIdx.x * 1024) + tid * 4;
                        store4(dst, v[0], v[1], v[2], v[3]);
                    }
                    if (m0 < CIN) {
                        const f32x4 v = acc[a];
                        if (pv && !FULL) {
#if OCRS_MM_EPI == 2
                            bf16* pa = gxa + srow * x.Ca;
                            bf16* pb = gxb + srow * x.Cb;
                            asm volatile("" : "+s"(pa), "+s"(pb));  // both row pointers stay scalar (hipcc re-associates the select into a per-lane 64-bit multiply)
                            store4((in_a ? pa : pb) + voff, v[0], v[1], v[2], v[3]);
#elif OCRS_MM_EPI == 1
                            bf16* dst = (in_a ? gxa + srow * x.Ca : gxb + srow * x.Cb) + voff;
                            store4(dst, v[0], v[1], v[2], v[3]);
#else
                            const long pix = srow + l15;
                            if (in_a)
                                store4(gxa + pix * x.Ca + m0, v[0], v[1], v[2], v[3]);
                            else
                                store4(gxb + pix * x.Cb + (m0 - x.Ca), v[0], v[1], v[2], v[3]);
#endif
                        }
                        if constexpr (STATS) {
                            float xq[4];
                            load4(tileX + (k * 16 + l15) * PX + m0, xq);  // 0 outside the image: contributes nothing
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                // the producer's backward reads the STORED (rounded) gradient; x~ > 0 <=> bn(z) > 0 for the ReLU producers that ask
                                const float gh = xq[i] > 0.f ? Elem<bf16>::round(v[i]) : 0.f;
                                t1[i] += gh;
                                if constexpr (!OCRS_MM_S2G) t2[i] = fmaf(gh, xq[i], t2[i]);
                            }
                        }
                    }
                }
                if constexpr (STATS && !STL) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        st1[b][i] += t1[i];
                        if constexpr (!OCRS_MM_S2G) st2[b][i] += t2[i];
                    }
                }
                if constexpr (STL) {
                    float r1[4], r2[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        r1[i] = quad16_sum(t1[i]);
                        if constexpr (!OCRS_MM_S2G) r2[i] = quad16_sum(t2[i]);
                    }
                    if ((lane & 15) == 0) {
                        float* q1 = s_st + (wave * 2 + 0) * MT * 16 + m0;
                        f32x4 o1 = *reinterpret_cast<f32x4*>(q1);
#pragma unroll
                        for (int i = 0; i < 4; ++i) o1[i] += r1[i];
                        *reinterpret_cast<f32x4*>(q1) = o1;
                        if constexpr (!OCRS_MM_S2G) {
                            float* q2 = s_st + (wave * 2 + 1) * MT * 16 + m0;
                            f32x4 o2 = *reinterpret_cast<f32x4*>(q2);
#pragma unroll
                            for (int i = 0; i < 4; ++i) o2[i] += r2[i];
                            *reinterpret_cast<f32x4*>(q2) = o2;
                        }
                    }
                }
              }
            }
        }
        MM_MARK(2)
        // ================= phase 2b: G_tap += x~^T dz(shifted), K = the tile's pixels, operands by LDS transpose reads =================
        // Straight-line code: the wave's own unit has a compile-time number of k-steps starting at a scalar first step, so every read address
        // is (per-tile base register + immediate) and hipcc can issue the transpose reads of the following steps under the current MFMAs.  (As a
        // loop over all k-steps with wave-uniform `continue`s every step was: 4 reads -> s_waitcnt lgkmcnt(0) -> 1 MFMA, a full LDS round trip
        // per MFMA.)
        {
            constexpr int LOWN = OCRS_MM_NOCOMPUTE ? 0 : ((C::TAPU == 9) ? KS : KS / 2);
            const bf16* xo = tileX + (ks_own0 * 32 + prow) * PX + pcol;
            const bf16* dq = tileD + ks_own0 * DW_ * PD + off_own[0];
#if OCRS_MM_GPIPE
            static_assert(NOWN == 1, "the explicit fragment pipeline is written for one own unit");
            bf16x8 afc[MT], bfc[NTO];
#pragma unroll
            for (int a = 0; a < MT; ++a) afc[a] = lds_tr8(xo + a * 16, xo + a * 16 + 16 * PX);
#pragma unroll
            for (int bb = 0; bb < NTO; ++bb) bfc[bb] = lds_tr8(dq + bb * 16, dq + bb * 16 + 16 * PD);
#pragma unroll
            for (int i = 0; i < LOWN; ++i) {
                bf16x8 afn[MT], bfn[NTO];
                if (i + 1 < LOWN) {
                    const bf16* xn = xo + (i + 1) * 32 * PX;
                    const bf16* dn = dq + (i + 1) * DW_ * PD;
#pragma unroll
                    for (int a = 0; a < MT; ++a) afn[a] = lds_tr8(xn + a * 16, xn + a * 16 + 16 * PX);
#pragma unroll
                    for (int bb = 0; bb < NTO; ++bb) bfn[bb] = lds_tr8(dn + bb * 16, dn + bb * 16 + 16 * PD);
                }
#pragma unroll
                for (int bb = 0; bb < NTO; ++bb)
#pragma unroll
                    for (int a = 0; a < MT; ++a) accO[0][a][bb] = mfma16(afc[a], bfc[bb], accO[0][a][bb]);
                if (i + 1 < LOWN) {
#pragma unroll
                    for (int a = 0; a < MT; ++a) afc[a] = afn[a];
#pragma unroll
                    for (int bb = 0; bb < NTO; ++bb) bfc[bb] = bfn[bb];
                }
            }
#else
#pragma unroll
            for (int i = 0; i < LOWN; ++i) {
                const bf16* xn = xo + i * 32 * PX;
                const bf16* dn = dq + i * DW_ * PD;
                bf16x8 af[MT];
#pragma unroll
                for (int a = 0; a < MT; ++a) af[a] = lds_tr8(xn + a * 16, xn + a * 16 + 16 * PX);
#pragma unroll
                for (int j = 0; j < NOWN; ++j) {  // (the x~ fragments of a k-step serve all own units)
                    const bf16* dj = dn + (off_own[j] - off_own[0]);
#pragma unroll
                    for (int bb = 0; bb < NTO; ++bb) {
                        const bf16x8 bfr = lds_tr8(dj + bb * 16, dj + bb * 16 + 16 * PD);
#pragma unroll
                        for (int a = 0; a < MT; ++a) accO[j][a][bb] = mfma16(af[a], bfr, accO[j][a][bb]);
                    }
                }
            }
#endif
            // this wave's sub-tile of the shared last unit over its k-step range (1 .. KS / NKR + 1 steps)
            const bf16* xs0 = tileX + prow * PX + sh_a * 16 + pcol;
            const bf16* ds0 = tileD + off_sh;
            for (int ks = sh_k0; ks < (OCRS_MM_NOCOMPUTE ? sh_k0 : sh_k1); ++ks) {
                const bf16* xa = xs0 + ks * 32 * PX;
                const bf16* da = ds0 + ks * DW_ * PD;
                accS = mfma16(lds_tr8(xa, xa + 16 * PX), lds_tr8(da, da + 16 * PD), accS);
            }
        }
        MM_MARK(3)
        lds_barrier();  // all readers of the tiles are done before the next commit
    };
    {
        using K0 = std::integral_constant<int, 0>;
        using K2 = std::integral_constant<int, 2>;
        long t = ts.first;
        if constexpr (FULL) {
            if (t < ts.end) {
                tile_body(t, K0{}, I0{});
                t += ts.step;
            }
            for (; t < ts.end; t += ts.step) tile_body(t, K2{}, I0{});
            // (nothing is in flight here -- the last tile issues no prefetch -- but only this wait lets tools/check_opaque_loads.py, which
            //  follows every static path, see it)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            for (; t < ts.end; t += ts.step) tile_body(t, K2{}, I0{});
        }
    }
#ifdef OCRS_MM_PROF
    if (blockIdx.x == 0 && tid == 0 && gxa) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(gxa);
        for (int i = 0; i < 5; ++i) o[i] = pt[i];
        o[5] = (ts.end - ts.first + ts.step - 1) / ts.step;
    }
#endif

    // ================= flush: G slots -> dWpw / dWdw partials of this block (workspace); stats partials =================
    __syncthreads();
    float* slots = reinterpret_cast<float*>(smem);        // [own unit j * NW + wave][MT][NTO][4][64]
    float* slotS = slots + C::NW * NOWN * MT * NTO * 256;  // [wave][4][64] this wave's sub-tile of the shared unit
    float* sstat = slotS + C::NW * 256;                    // [wave][2][MT*16] (register-accumulated stats)
#pragma unroll
    for (int j = 0; j < NOWN; ++j)
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NTO; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) slots[(((j * C::NW + wave) * MT + a) * NTO + b) * 256 + r * 64 + lane] = accO[j][a][b][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) slotS[wave * 256 + r * 64 + lane] = accS[r];
    if constexpr (STATS && !STL) {
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v1 = quad16_sum(st1[a][i]), v2 = OCRS_MM_S2G ? 0.f : quad16_sum(st2[a][i]);
                if ((lane & 15) == 0) {
                    sstat[(wave * 2 + 0) * MT * 16 + a * 16 + (lane >> 4) * 4 + i] = v1;
                    sstat[(wave * 2 + 1) * MT * 16 + a * 16 + (lane >> 4) * 4 + i] = v2;
                }
            }
    }
    if constexpr (STL) {
        static_assert(!STL || (C::NW * NOWN * MT * NTO + C::NW) * 256 * 4 <= C::OFF_PAR, "the G flush slots must not reach the LDS-resident stats");
        sstat = s_st;
    }
    __syncthreads();
    // G[tap][c][o] from the slots (fixed summation order -> deterministic)
    auto Gval = [&](int tap, int c, int o) -> float {
        const int a = c >> 4, r = c & 3, lrow = (c & 15) >> 2;
        const int b = (COUT == 8) ? 0 : (o >> 4);
        const int n = (COUT == 8) ? ((tap < 8 ? (tap & 1) * 8 : 0) + o) : (o & 15);
        const int ln = lrow * 16 + n;
        if (tap < 8) {
            if constexpr (COUT == 8) {
                const int u = tap >> 1;
                return slots[((u * MT + a) * NTO + b) * 256 + r * 64 + ln] + slots[(((u + 4) * MT + a) * NTO + b) * 256 + r * 64 + ln];
            } else
                return slots[((tap * MT + a) * NTO + b) * 256 + r * 64 + ln];
        }
        float sum = 0.f;
        const int sub = b * MT + a;
        for (int kr = 0; kr < NKR; ++kr) sum += slotS[(sub + NSUB * kr) * 256 + r * 64 + ln];
        return sum;
    };
    float* part = ws + (long)blockIdx.x * C::PART;
    // (s_w9 / s_wp were overwritten by the slots: re-read the masters -- once per block)
    for (int e = tid; e < COUT * CIN; e += NT) {
        const int o = e / CIN, c = e - o * CIN;
        float s = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) s = fmaf(wdw[c * 9 + tap], Gval(tap, c, o), s);
        part[e] = s;
    }
    for (int e = tid; e < 9 * CIN; e += NT) {
        const int c = e / 9, tap = e - c * 9;
        float s = 0.f;
        for (int o = 0; o < COUT; ++o) s = fmaf(wpw[o * ldw + c], Gval(tap, c, o), s);
        part[COUT * CIN + e] = s;
    }
    for (int e = tid; e < 2 * CIN; e += NT) {  // stats partials interleaved per channel: [CIN][S1 | S2]
        float s = 0.f;
        if constexpr (STATS) {
            const int c = e >> 1, which = e & 1;
            if (OCRS_MM_S2G && which) {  // S2 from G, with the bf16 effective weights the dgrad used
                for (int tap = 0; tap < 9; ++tap)
                    for (int o = 0; o < COUT; ++o) s = fmaf(bf2f(f2bf(wdw[c * 9 + tap] * wpw[o * ldw + c])), Gval(tap, c, o), s);
            } else {
                for (int w = 0; w < C::NW; ++w) s += sstat[(w * 2 + which) * MT * 16 + c];
            }
        }
        part[COUT * CIN + 9 * CIN + e] = s;
        if (STATS && bl.raw) bwd_last_add(bl, CIN, e >> 1, e & 1, s);
    }
    if constexpr (STATS) {
        if (bl.raw) {  // the producers' sums finalised here by the last workgroup (BwdLast): the reduce kernel leaves the dependency chain
            __syncthreads();  // (every read of the flush slots is done: smem[0] is free)
            bwd_last_finish(bl, CIN, tra, trb, tid, NT, reinterpret_cast<int*>(smem));
        }
    }
}

// Deterministic second stage of every flush of k_mm_bwd: one thread-column per output element, the nb block partials are summed in a fixed
// order (8 interleaved chains per element, combined through LDS in a fixed tree), a single writer per element, no atomics.
//   dwpw [COUT][ldw] (+c_off) += ,  dwdw [(c_off + c)][9] += ,  and for the producers of the input (if asked):
//   gsum [2][Cs] (fp64) += { S1, rstd * ((S2 - shift*S1)/scale - mean*S1) }   (S1 = sum ghat', S2 = sum ghat' x~,  x~ = z*scale + shift where ghat' != 0)
__global__ __launch_bounds__(256) void k_mm_bwd_reduce(const float* __restrict__ ws, int nb, int CIN, int COUT, int Ca, float* __restrict__ dwpw, int ldw,
                                                       float* __restrict__ dwdw, double* __restrict__ gsum_a, double* __restrict__ gsum_b,
                                                       const float* __restrict__ saved_a, const float* __restrict__ saved_b,
                                                       const float* __restrict__ tra, const float* __restrict__ trb) {
    __shared__ float red[8][32];
    const int nelem = COUT * CIN + 11 * CIN;
    const int col = threadIdx.x & 31, chain = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + col;
    // chain c sums partials c, c+8, c+16, ...: eight independent accumulators keep eight loads in flight (the loop is pure L2 latency); the
    // association order is fixed, so the result is bit-reproducible
    float s = 0.f;
    if (e < nelem) {
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int b = chain;
        for (; b + 56 < nb; b += 64) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += ws[(long)(b + 8 * u) * nelem + e];
        }
        for (int u = 0; b < nb; b += 8, ++u) a[u & 7] += ws[(long)b * nelem + e];
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    red[chain][col] = s;
    __syncthreads();
    __shared__ float fin[32];
    const float v = ((red[0][col] + red[1][col]) + (red[2][col] + red[3][col])) + ((red[4][col] + red[5][col]) + (red[6][col] + red[7][col]));
    if (chain == 0) fin[col] = v;
    __syncthreads();
    if (chain != 0 || e >= nelem) return;
    if (e >= COUT * CIN + 9 * CIN) {
        const int r2 = e - (COUT * CIN + 9 * CIN);
        if (r2 & 1) return;  // S2 is consumed by its channel's S1 thread
        const int c = r2 >> 1, Cb = CIN - Ca;
        const bool in_a = c < Ca;
        double* gs = in_a ? gsum_a : gsum_b;
        if (!gs) return;
        const int cc = in_a ? c : c - Ca, Cs = in_a ? Ca : Cb;
        const float* sv = in_a ? saved_a : saved_b;
        const float* tr = in_a ? tra : trb;
        const double S1 = v, S2 = fin[col + 1], sc = tr[cc], sh = tr[Cs + cc], mean = sv[cc], rstd = sv[Cs + cc];
        gs[cc] += S1;
        if (sc != 0.0) gs[Cs + cc] += rstd * ((S2 - sh * S1) / sc - mean * S1);
        return;
    }
    if (e < COUT * CIN) {
        const int o = e / CIN, c = e - o * CIN;
        dwpw[o * ldw + c] += v;
        return;
    }
    const int r = e - COUT * CIN;
    if (r < 9 * CIN) {
        dwdw[r] += v;
        return;
    }
    // stats: [CIN][S1 | S2] pairs; the pair of a channel sits in two adjacent columns of the same 32-wide window (the stats base is even)
    return;
}
static int mm_grid(int th, int N, int H, int W, int pooled, int bpc = 2) {
    const long ntiles = (long)N * ((W + pooled + 31) / 32) * ((H + pooled + th - 1) / th);
    return persistent_grid(ntiles, bpc);
}
static int mm_th(int Cin, int Cout, int nst) { return OCRS_MF_TH_OF(Cin, Cout, nst); }  // forward tiles
static int mm_bwd_bpc(int Cin, int Cout, int pooled = 0) {
    if (Cin == 32 && Cout == 32) return OCRS_MM_C32_N256 ? 2 : OCRS_MM_C32_BPC;
    if (Cin == 32 && Cout == 16 && OCRS_MM_C3216_N256) return 2;
    return (OCRS_MM_B3_16_8 && Cin == 16 && Cout == 8 && !pooled) ? 3 : 2;
}
static int mm_bwd_th(int Cin, int Cout, int pooled) {
    if (Cin == 32 || Cout == 32) return (mm_bwd_bpc(Cin, Cout) == 1 && !(OCRS_MM_C32_N256 && Cin == 32 && Cout == 32)) ? 16 : 8;
    return (pooled && Cin == 16 && Cout == 16) ? OCRS_MM_TH_16_16_P : OCRS_MM_TH_OF(Cin, Cout);
}

template <int CIN, int COUT, bool PPOOL, bool G2, bool STATS>
static void mm_bwd_launch1(const Src2<bf16>& x, const float* tra, const float* trb, const float* wdw, const float* wpw, int ldw, const bf16* g1, const bf16* g2,
                           const bf16* z, const float* bn, const float* coef, bf16* gxa, bf16* gxb, float* ws, int N, int H, int W, int nb, const BnFin& fin,
                           const BwdLast& bl, hipStream_t st) {
    using CC = MmCfg<CIN, COUT, PPOOL>;
    Tiling2 tg = make_tiling2(N, H + (PPOOL ? 1 : 0), W + (PPOOL ? 1 : 0), CC::TW, CC::TH);  // pooled: origins shifted by -1 -> one more row / column of tiles may be needed
    tg.H = H;
    tg.W = W;
    // (the attribute is set per call: it is per device, this runs on any thread, and it is a cheap host-side update -- ADVICE r02)
    static const int full_on = env_int("OCRS_MM_FULL", 1);
    if constexpr (!PPOOL) {
        if (full_on && W % CC::TW == 0) {  // every tile COLUMN inside the image: unconditional stores + hand-written prefetch waits
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mm_bwd<CIN, COUT, PPOOL, G2, STATS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, CC::SMEM);
            OCRS_LAUNCH_T((k_mm_bwd<CIN, COUT, PPOOL, G2, STATS, true>), dim3(nb), dim3(CC::NT), CC::SMEM, st, x, tra, trb, wdw, wpw, ldw, g1, g2, z, bn, coef,
                          gxa, gxb, ws, tg, fin, bl);
            return;
        }
    }
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mm_bwd<CIN, COUT, PPOOL, G2, STATS, false>), hipFuncAttributeMaxDynamicSharedMemorySize, CC::SMEM);
    OCRS_LAUNCH_T((k_mm_bwd<CIN, COUT, PPOOL, G2, STATS, false>), dim3(nb), dim3(CC::NT), CC::SMEM, st, x, tra, trb, wdw, wpw, ldw, g1, g2, z, bn, coef, gxa, gxb,
                  ws, tg, fin, bl);
}

template <int CIN, int COUT>
static void mm_bwd_dispatch(const Src2<bf16>& x, const float* tra, const float* trb, const float* wdw, const float* wpw, int ldw, const bf16* g1, const bf16* g2,
                            int pooled, const bf16* z, const float* bn, const float* coef, bf16* gxa, bf16* gxb, float* ws, bool stats, int N, int H, int W,
                            int nb, const BnFin& fin, const BwdLast& bl, hipStream_t st) {
#define MMB(PP, GG, SS) mm_bwd_launch1<CIN, COUT, PP, GG, SS>(x, tra, trb, wdw, wpw, ldw, g1, g2, z, bn, coef, gxa, gxb, ws, N, H, W, nb, fin, bl, st)
    if (pooled) {
        if (g2) { if (stats) MMB(true, true, true); else MMB(true, true, false); }
        else    { if (stats) MMB(true, false, true); else MMB(true, false, false); }
    } else {
        if (g2) { if (stats) MMB(false, true, true); else MMB(false, true, false); }
        else    { if (stats) MMB(false, false, true); else MMB(false, false, false); }
    }
#undef MMB
}

// in-kernel finalisation state for one block-backward launch (off: raw = null) and the launch's second stage
static BwdLast mm_bwd_last(bool want, int Cin, int Ca, double* gsA, double* gsB, const float* svA, const float* svB) {
    BwdLast bl{nullptr, nullptr, gsA, gsB, svA, svB, Ca, 0};
    static const int on = env_int("OCRS_BWD_LAST", 1);
    if (!want || !on) return bl;
    double* p = bwd_defer_scratch(BWD_LAST_SLOTS * 2 * Cin + 2);
    if (p) {
        bl.raw = p;
        bl.counter = reinterpret_cast<unsigned*>(p + BWD_LAST_SLOTS * 2 * Cin);
    }
    return bl;
}
static void mm_bwd_second_stage(const float* ws, int nb, int Cin, int Cout, int Ca, float* dwpw, int ldw, float* dwdw, double* gsA, double* gsB, const float* svA,
                                const float* svB, const float* tA, const float* tB, const BwdLast& bl, bool may_defer, hipStream_t st) {
    const int ne = Cout * Cin + 11 * Cin;
    if (bl.raw) {  // the sums are done in the block kernel: only the weight gradients are left, and nothing in the backward reads them
        gsA = gsB = nullptr;
        if (may_defer && bwd_defer_reduce(ws, nb, ne, dwpw, Cout * Cin, Cin, ldw, dwdw, 9 * Cin)) return;
    } else if (may_defer && !gsA && !gsB && bwd_defer_reduce(ws, nb, ne, dwpw, Cout * Cin, Cin, ldw, dwdw, 9 * Cin)) {
        return;
    }
    OCRS_LAUNCH_T(k_mm_bwd_reduce, dim3((ne + 31) / 32), dim3(256), 0, st, ws, nb, Cin, Cout, Ca, dwpw, ldw, dwdw, gsA, gsB, svA, svB, tA, tB);
}

extern "C" {

// 1 if the matrix-core block backward covers this shape (bf16; Cin, Cout in {8,16,32}; a 32|32 concat input runs as two launches)
long ocrs_mm_bwd_supported(int Ca, int Cb, int Cout, int dtype) {
    if (dtype != 1 || !(Cout == 8 || Cout == 16 || Cout == 32)) return 0;
    const int Cin = Ca + Cb;
    if (Ca == 32 && Cb == 32) return 1;
    if ((Cin == 8 && Cout == 32) || (Cin == 32 && Cout == 8)) return 0;  // (not in the net: no instantiation)
    return (Cin == 8 || Cin == 16 || Cin == 32) && Ca % 8 == 0 && Cb % 8 == 0;
}
long ocrs_mm_bwd_ws_floats(int Ca, int Cb, int Cout, int N, int H, int W) {
    const int Cin = (Ca == 32 && Cb == 32) ? 32 : Ca + Cb;
    const int nb0 = mm_grid(mm_bwd_th(Cin, Cout, 0), N, H, W, 0, mm_bwd_bpc(Cin, Cout, 0)), nb1 = mm_grid(mm_bwd_th(Cin, Cout, 1), N, H, W, 1, mm_bwd_bpc(Cin, Cout, 1));
    // per block: its partials + 4 KB of scratch lines (stores of rows below the image); a 32 | 32 concat input runs as two launches, each with its own half
    // (so that both second stages can be deferred: ocrs_bwd_defer_begin)
    const long tiled = (long)(nb0 > nb1 ? nb0 : nb1) * (Cout * Cin + 11 * Cin + 1024) * ((Ca == 32 && Cb == 32) ? 2 : 1);
    const long rs = rs_bwd_supported(Ca, Cb, Cout, 0, N, H, W) ? (long)rs_bwd_blocks(Cin, Cout, N, H, W, 0) * (Cout * Cin + 11 * Cin) : 0;  // row-streaming form (det_rs.hip)
    return tiled > rs ? tiled : rs;
}

// Backward of one DepthwiseConv block on the matrix cores (replaces ocrs_pw_bwd + ocrs_dw_bwd [+ ocrs_bn_bwd_reduce of the producers]):
//   xa | xb (Ca | Cb channels) with load transforms tra | trb: the block input;  wdw [Cin][9], wpw [Cout][Cin]: fp32 master weights;
//   g1 (+ g2): gradient w.r.t. the block output, at half resolution when pooled (routed through MaxPool2d(2));  z, bn, coef: as ocrs_pw_bwd;
//   gxa | gxb: dL/dx~;  dwpw / dwdw: ACCUMULATED (+=, single writer: deterministic);  ws: ocrs_mm_bwd_ws_floats() floats;
//   saved_a/gsum_a, saved_b/gsum_b (nullable): as ocrs_dw_bwd.
static int mm_bwd_impl(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const float* wpw, const void* g1,
                       const void* g2, int pooled, const void* z, const float* bn, const float* coef, const BnFin& fin, void* gxa, void* gxb, float* dwpw,
                       float* dwdw, float* ws, const float* saved_a, double* gsum_a, const float* saved_b, double* gsum_b, int Cout, int N, int H, int W,
                       int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(xa && tra && wdw && wpw && g1 && z && bn && (coef || fin.gsum) && gxa && dwpw && dwdw && ws && (Cb == 0 || (xb && trb && gxb)));
    OCRS_CHECK_ARG(ocrs_mm_bwd_supported(Ca, Cb, Cout, dtype) && (long)N * (H + 2) * (W + 2) < (1L << 31) && H >= 2 && W >= 2);
    OCRS_CHECK_ARG((!gsum_a || saved_a) && (!gsum_b || (saved_b && Cb > 0)));
    const int CinTot = Ca + Cb;
    const bool split = Ca == 32 && Cb == 32;
    const int nlaunch = split ? 2 : 1;
    for (int part = 0; part < nlaunch; ++part) {
        // a 32 | 32 concat input: z = Wpw[:, :32] u_a + Wpw[:, 32:] u_b, so dx~, dWpw and dWdw separate by source; dz is the same for both launches
        const int Cin = split ? 32 : CinTot, c_off = part * 32;
        Src2<bf16> x{(const bf16*)(part ? xb : xa), (const bf16*)(split ? nullptr : xb), split ? 32 : Ca, split ? 0 : Cb};
        const float* tA = part ? trb : tra;
        const float* tB = split ? nullptr : trb;
        bf16* ga = (bf16*)(part ? gxb : gxa);
        bf16* gb = split ? nullptr : (bf16*)gxb;
        const float* svA = part ? saved_b : saved_a;
        double* gsA = part ? gsum_b : gsum_a;
        const float* svB = split ? nullptr : saved_b;
        double* gsB = split ? nullptr : gsum_b;
        const bool stats = gsA || gsB;
        const bool rs = !split && rs_bwd_supported(Ca, Cb, Cout, pooled, N, H, W);  // the row-streaming kernel (det_rs.hip) covers this launch
        const int nb = rs ? rs_bwd_blocks(Cin, Cout, N, H, W, g2 != nullptr)
                          : mm_grid(mm_bwd_th(Cin, Cout, pooled ? 1 : 0), N, H, W, pooled ? 1 : 0, mm_bwd_bpc(Cin, Cout, pooled ? 1 : 0));
        const float* wd = wdw + c_off * 9;
        const float* wp = wpw + c_off;
        // deferred second stage (ocrs_bwd_defer_begin): the producers' sums by the last workgroup of the block kernel, the weight-gradient reduce queued
        // (the two launches of a split pair write their partials to separate halves of ws)
        float* const ws_full = ws;
        ws = ws_full + (size_t)part * nb * (Cout * Cin + 11 * Cin + 1024);
        const BwdLast bl = mm_bwd_last(stats, Cin, x.Ca, gsA, gsB, svA, svB);
        if (rs) rs_bwd_launch(x, tA, tB, wd, wp, CinTot, (const bf16*)g1, (const bf16*)g2, (const bf16*)z, bn, coef, ga, gb, ws, stats, Cout, N, H, W, fin, st, nullptr, nullptr,
                              nullptr, nullptr, bl);
#define MM_CASE(CI_, CO_)                                                                                                             \
    if (!rs && Cin == CI_ && Cout == CO_)                                                                                             \
        mm_bwd_dispatch<CI_, CO_>(x, tA, tB, wd, wp, CinTot, (const bf16*)g1, (const bf16*)g2, pooled, (const bf16*)z, bn, coef, ga, gb, ws, stats, N, H, W, nb, fin, bl, st);
        MM_CASE(8, 8) MM_CASE(8, 16) MM_CASE(16, 8) MM_CASE(16, 16) MM_CASE(16, 32) MM_CASE(32, 16) MM_CASE(32, 32)
#undef MM_CASE
        mm_bwd_second_stage(ws, nb, Cin, Cout, x.Ca, dwpw + c_off, CinTot, dwdw + c_off * 9, gsA, gsB, svA, svB, tA, tB, bl, true, st);
        ws = ws_full;
    }
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

long ocrs_mm_bwd_head_supported(int Ca, int Cb, int Cout, int N, int H, int W, int dtype) {
    return dtype == 1 && Ca == 8 && Cb == 0 && Cout == 8 && rs_bwd_supported(Ca, Cb, Cout, 0, N, H, W) ? 1 : 0;
}
// ocrs_mm_bwd_fin with the gradient w.r.t. the block output formed on the fly from out_conv's backward (gl [N H W] fp32, whead [8]): see k_rs_bwd<..., HEAD>
int ocrs_mm_bwd_fin_head(const void* xa, int Ca, const float* tra, const float* wdw, const float* wpw, const float* gl, const float* whead, const void* z,
                         const float* bn, const double* gsum, const float* gamma, const float* saved, float* dgamma, float* dbeta, void* gxa, float* dwpw,
                         float* dwdw, float* ws, const float* saved_a, double* gsum_a, int Cout, int N, int H, int W, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(xa && tra && wdw && wpw && gl && whead && z && bn && gsum && gamma && saved && dgamma && dbeta && gxa && dwpw && dwdw && ws);
    OCRS_CHECK_ARG(ocrs_mm_bwd_head_supported(Ca, 0, Cout, N, H, W, dtype) && (!gsum_a || saved_a));
    const BnFin fin{gsum, gamma, saved, dgamma, dbeta, (long)N * H * W};
    Src2<bf16> x{(const bf16*)xa, nullptr, Ca, 0};
    const int nb = rs_bwd_blocks(Ca, Cout, N, H, W, 0);
    const BwdLast bl = mm_bwd_last(gsum_a != nullptr, Ca, Ca, gsum_a, nullptr, saved_a, nullptr);
    rs_bwd_launch(x, tra, nullptr, wdw, wpw, Ca, nullptr, nullptr, (const bf16*)z, bn, nullptr, (bf16*)gxa, nullptr, ws, gsum_a != nullptr, Cout, N, H, W, fin, st, gl,
                  whead, nullptr, nullptr, bl);
    mm_bwd_second_stage(ws, nb, Ca, Cout, Ca, dwpw, Ca, dwdw, gsum_a, nullptr, saved_a, nullptr, tra, nullptr, bl, true, st);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// ocrs_mm_bwd_fin for the block behind the first block (in_conv.seq.1): its input is given as the first block's u plane (ocrs_dwpw_c1_fwd_u) + that block's
// pointwise weight wexp [8], x[p][c] = round(wexp[c] * u[p]) (the stored values); g1 (+ g2): direct gradients.  Needs ocrs_mm_bwd_head_supported(8, 0, Cout, ...).
int ocrs_mm_bwd_fin_xu(const void* xu, const float* wexp, const float* tra, const float* wdw, const float* wpw, const void* g1, const void* g2, const void* z,
                       const float* bn, const double* gsum, const float* gamma, const float* saved, float* dgamma, float* dbeta, void* gxa, float* dwpw,
                       float* dwdw, float* ws, const float* saved_a, double* gsum_a, int Cout, int N, int H, int W, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(xu && wexp && tra && wdw && wpw && g1 && z && bn && gsum && gamma && saved && dgamma && dbeta && gxa && dwpw && dwdw && ws);
    OCRS_CHECK_ARG(ocrs_mm_bwd_head_supported(8, 0, Cout, N, H, W, dtype) && (!gsum_a || saved_a));
    const BnFin fin{gsum, gamma, saved, dgamma, dbeta, (long)N * H * W};
    Src2<bf16> x{nullptr, nullptr, 8, 0};
    const int nb = rs_bwd_blocks(8, Cout, N, H, W, g2 != nullptr);
    const BwdLast bl = mm_bwd_last(gsum_a != nullptr, 8, 8, gsum_a, nullptr, saved_a, nullptr);
    rs_bwd_launch(x, tra, nullptr, wdw, wpw, 8, (const bf16*)g1, (const bf16*)g2, (const bf16*)z, bn, nullptr, (bf16*)gxa, nullptr, ws, gsum_a != nullptr, Cout, N, H, W,
                  fin, st, nullptr, nullptr, (const bf16*)xu, wexp, bl);
    mm_bwd_second_stage(ws, nb, 8, Cout, 8, dwpw, 8, dwdw, gsum_a, nullptr, saved_a, nullptr, tra, nullptr, bl, true, st);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// ocrs_mm_bwd_fin_xu that ALSO accumulates the first block's weight-gradient sums (k_rs_bwd<..., C1>: see det_rs.hip) from the network input img [N H W] fp32 into
// c1acc [8][32] fp64 (zeroed by the caller) and stores NO input gradient: the first block (models.py:115) is this block's only producer and needs dL/dx~ only
// for its weight gradient, which ocrs_c1_bwd_fin then forms from the sums.
int ocrs_mm_bwd_fin_xu_c1(const void* xu, const float* wexp, const float* tra, const float* wdw, const float* wpw, const void* g1, const void* g2, const void* z,
                          const float* bn, const double* gsum, const float* gamma, const float* saved, float* dgamma, float* dbeta, float* dwpw, float* dwdw,
                          float* ws, const float* saved_a, double* gsum_a, const float* img, double* c1acc, int Cout, int N, int H, int W, int dtype,
                          hipStream_t st) {
    OCRS_CHECK_ARG(xu && wexp && tra && wdw && wpw && g1 && z && bn && gsum && gamma && saved && dgamma && dbeta && dwpw && dwdw && ws && img && c1acc);
    OCRS_CHECK_ARG(Cout == 8 && ocrs_mm_bwd_head_supported(8, 0, Cout, N, H, W, dtype) && (!gsum_a || saved_a));
    const BnFin fin{gsum, gamma, saved, dgamma, dbeta, (long)N * H * W};
    Src2<bf16> x{nullptr, nullptr, 8, 0};
    const int nb = rs_bwd_blocks(8, Cout, N, H, W, g2 != nullptr);
    const BwdLast bl = mm_bwd_last(gsum_a != nullptr, 8, 8, gsum_a, nullptr, saved_a, nullptr);
    rs_bwd_launch(x, tra, nullptr, wdw, wpw, 8, (const bf16*)g1, (const bf16*)g2, (const bf16*)z, bn, nullptr, nullptr, nullptr, ws, gsum_a != nullptr, Cout, N, H, W,
                  fin, st, nullptr, nullptr, (const bf16*)xu, wexp, bl, img, c1acc);
    mm_bwd_second_stage(ws, nb, 8, Cout, 8, dwpw, 8, dwdw, gsum_a, nullptr, saved_a, nullptr, tra, nullptr, bl, true, st);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

int ocrs_mm_bwd(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const float* wpw, const void* g1,
                const void* g2, int pooled, const void* z, const float* bn, const float* coef, void* gxa, void* gxb, float* dwpw, float* dwdw, float* ws,
                const float* saved_a, double* gsum_a, const float* saved_b, double* gsum_b, int Cout, int N, int H, int W, int dtype, hipStream_t st) {
    const BnFin fin{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    return mm_bwd_impl(xa, xb, Ca, Cb, tra, trb, wdw, wpw, g1, g2, pooled, z, bn, coef, fin, gxa, gxb, dwpw, dwdw, ws, saved_a, gsum_a, saved_b, gsum_b, Cout, N, H,
                       W, dtype, st);
}

// ocrs_mm_bwd with ocrs_bn_bwd_finalize folded in: instead of `coef`, the block's complete BatchNorm-backward sums gsum [2][Cout] (fp64), its gamma and
// saved [mean | rstd]; the kernel derives the dz coefficients in its prologue and writes dgamma / dbeta [Cout] (one launch less per block).
int ocrs_mm_bwd_fin(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const float* wpw, const void* g1,
                    const void* g2, int pooled, const void* z, const float* bn, const double* gsum, const float* gamma, const float* saved, float* dgamma,
                    float* dbeta, void* gxa, void* gxb, float* dwpw, float* dwdw, float* ws, const float* saved_a, double* gsum_a, const float* saved_b,
                    double* gsum_b, int Cout, int N, int H, int W, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(gsum && gamma && saved && dgamma && dbeta);
    const BnFin fin{gsum, gamma, saved, dgamma, dbeta, (long)N * H * W};
    return mm_bwd_impl(xa, xb, Ca, Cb, tra, trb, wdw, wpw, g1, g2, pooled, z, bn, nullptr, fin, gxa, gxb, dwpw, dwdw, ws, saved_a, gsum_a, saved_b, gsum_b, Cout, N,
                       H, W, dtype, st);
}

}  // extern "C"

// ----------------------------------------------------------------------------------------------------------------------------------
// forward:  z[o][p] = sum_{tap,c} Weff[o][(tap,c)] x~[p + off(tap)][c]   (MFMA, K = 9 * Cin; x~ = the producers' BatchNorm+ReLU applied on load)
//           + per-channel batch sums of the stored z (BatchNorm statistics, deterministic per-block partials)
//           + optionally MaxPool2d(2) of the block output in its pre-BatchNorm form (see k_dwpw_fwd).
// CINB channels per stage (8 / 16 / 32), NST stages (2 for the 32 | 32 concat: one stage per source, same accumulators).
// ----------------------------------------------------------------------------------------------------------------------------------
namespace {
template <int CINB, int NST, int COUT>
struct MfCfg {
    static constexpr int NT = 512, NW = 8;
    static constexpr int TW = 32, TH = OCRS_MF_TH_OF(CINB, COUT, NST), TP = TW * TH;
    static constexpr int DW_ = TW + 2, DH_ = TH + 2, DP = DW_ * DH_;
    static constexpr int CGB = CINB / 8;
    static constexpr int PX = MmPitch<CINB>::V;
    static constexpr int MT = (COUT + 15) / 16;
    static constexpr int KC = (9 * CINB + 31) / 32;
    static_assert(TH % 8 == 0, "forward tiles: TH (row pair, column half) units over 8 waves");
    static constexpr int UPW = TH / 8, NPW = 2 * UPW;                  // (row pair, column half) units per wave; 16-pixel N tiles per wave
    static constexpr int NXI = (DP * CGB + NT - 1) / NT;               // x items per thread and stage
    static constexpr int OFF_X = 0;
    static constexpr int OFF_WF = (DP * PX * 2 + 63) & ~63;
    static constexpr int OFF_PAR = OFF_WF + NST * MT * KC * 64 * 16;
    static constexpr int PAR_FLOATS = 3 * CINB * NST + 9 * CINB * NST + COUT * CINB * NST + NW * MT * 16 * 2;
    static constexpr int SMEM = OFF_PAR + PAR_FLOATS * 4;
};
}  // namespace

// BatchNorm statistics finalisation folded into the forward launch ("last workgroup done", round 5): every workgroup publishes its [COUT][sum | sum^2]
// partial with agent-scope (sc1, write-through) stores, drains them, takes a ticket from an agent-scope counter; the workgroup that draws the last
// ticket reads all partials with agent-scope loads (they may sit behind another XCD's L2) and does what k_bn_finalize_parts does, in the same
// association order (bit-identical results) -- one ~6 us launch less per block on the forward's critical path.  counter: one zeroed word per launch.
// the arithmetic of k_bn_finalize_parts for element column `col` (0..31) of 32-element window `win`, chain `chain` (0..7); red: [windows][8][32] doubles
template <bool AGENT>
__device__ __forceinline__ double bn_parts_chain_sum(const float* __restrict__ parts, int nparts, int C, int e, int chain) {
    double s = 0.0;
    if (e < 2 * C) {
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // eight loads in flight per chain, fixed association order
        auto ld = [&](long i) -> double {
            if constexpr (AGENT) return (double)__hip_atomic_load(parts + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else return (double)parts[i];
        };
        int b = chain;
        for (; b + 56 < nparts; b += 64) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += ld((long)(b + 8 * u) * 2 * C + e);
        }
        for (int u = 0; b < nparts; b += 8, ++u) a[u & 7] += ld((long)b * 2 * C + e);
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    return s;
}

// FULL (the launcher sets it when every tile lies inside the image: H % TH == 0, W % TW == 0, even sizes when pooling): every global store of
// the tile loop is then UNCONDITIONAL for every lane (no divergent branch around a store) and the first tile is peeled out of the loop.
// Why it matters: hipcc derives the `s_waitcnt vmcnt(N)` in front of the first use of a prefetched vector from the operations that are
// pending on EVERY path into that point.  The prefetch loads of tile t+1 are issued before tile t's epilogue stores, and VMEM operations
// retire in order, so the right wait is vmcnt(#stores + younger loads).  With a store inside a divergent `if` (a skippable block) or with the
// loop entered straight from the prologue (no stores pending on that path) the count that holds on every path is vmcnt(younger loads) -- and
// then every wave waits, once per tile, until the previous tile's stores are ACKNOWLEDGED by memory before it may touch the next tile's
// input (measured: that wait, not HBM latency or bandwidth, was the largest stall of these kernels).  Lanes that hold no output channel
// (M rows 8..15 of the 16-row MFMA tile when COUT = 8) carry a duplicate of rows 0..7 (duplicated weight rows) and store the same bytes to
// the same address; the two lanes of a max-pool pair likewise.
// XU (round 5; Cin = 8, one stage): the input is the first block's output given as its rank-one generator -- the u plane (bf16 [N][H][W], k_c1_fwd2) and that
// block's pointwise weight wexp [8]: x[p][c] = round(wexp[c] * u[p]), the values the first block would have stored, from 2 instead of 16 bytes per pixel.
struct XuSrc {
    const bf16* u;
    const float* wexp;
};
template <int CINB, int NST, int COUT, bool POOL, bool FULL, bool XU = false>
__global__ __launch_bounds__(512, (mm_fwd_lb<CINB, COUT>())) void k_mm_fwd(Src2<bf16> x, const float* __restrict__ tra, const float* __restrict__ trb,
                                                   const float* __restrict__ wdw /*[Cin][9]*/, const float* __restrict__ wpw /*[COUT][Cin]*/,
                                                   bf16* __restrict__ z, float* __restrict__ ws /*[grid][COUT][2]*/, Tiling2 tg,
                                                   const float* __restrict__ gamma, bf16* __restrict__ pooled, FwdFin fin, XuSrc xu) {
    static_assert(!XU || (CINB == 8 && NST == 1), "u plane: the 8-channel output of the first block");
    using C = MfCfg<CINB, NST, COUT>;
    constexpr int NT = C::NT, TW = C::TW, TH = C::TH, DW_ = C::DW_, DP = C::DP, CGB = C::CGB, PX = C::PX, MT = C::MT, KC = C::KC, NPW = C::NPW;
    constexpr int CIN = CINB * NST;
    extern __shared__ __attribute__((aligned(64))) char smem[];
    bf16* tileX = reinterpret_cast<bf16*>(smem + C::OFF_X);     // [DP][PX] x~ on the domain of the current stage (0 outside the image)
    uint4* s_wf = reinterpret_cast<uint4*>(smem + C::OFF_WF);   // [NST][MT][KC][64] effective-weight A fragments
    float* s_trx = reinterpret_cast<float*>(smem + C::OFF_PAR); // [CIN/8][3][8]
    float* s_w9 = s_trx + 3 * CIN;                               // [CIN][9]
    float* s_wp = s_w9 + 9 * CIN;                                // [COUT][CIN]
    float* s_stat = s_wp + COUT * CIN;                           // [wave][MT*16][2]
    const int H = tg.H, W = tg.W;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15;
#if OCRS_MM_SW
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: tile rows / first columns of the wave's N tiles become scalar code)
#else
    const int wave = tid >> 6;
#endif

    fill_tr8(s_trx, x, tra, trb, CIN, tid);
    for (int i = tid; i < 9 * CIN; i += NT) s_w9[i] = wdw[i];
    for (int i = tid; i < COUT * CIN; i += NT) s_wp[i] = wpw[i];
    {
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        for (int i = tid; i < C::OFF_WF / 16; i += NT) reinterpret_cast<uint4*>(smem)[i] = z4;
    }
    __syncthreads();
    for (int f = tid; f < NST * MT * KC * 64; f += NT) {
        const int l = f & 63, kc = (f >> 6) % KC, mt = ((f >> 6) / KC) % MT, st = (f >> 6) / (KC * MT);
        const int m = (FULL && COUT == 8) ? (l & 7) : mt * 16 + (l & 15);  // FULL, COUT = 8: rows 8..15 duplicate rows 0..7
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kc * 32 + (l >> 4) * 8 + j, tap = k / CINB, c = st * CINB + k % CINB;
            v[j] = (m < COUT && tap < 9) ? s_w9[c * 9 + tap] * s_wp[m * CIN + c] : 0.f;
        }
        s_wf[f] = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
    }
    __syncthreads();

    // x items: (domain pixel, 8-channel group) of stage st; with one stage the source (a | b) depends on the channel group, with two stages on the stage
    const int cgb = tid % CGB;
    int xi_dyx[C::NXI];
#pragma unroll
    for (int j = 0; j < C::NXI; ++j) {
        const int d = (tid + j * NT) / CGB, dy = d / DW_, dx = d - dy * DW_;
        xi_dyx[j] = dy | (dx << 16);
    }
    u32x4 xr[NST][C::NXI];  // (XU: the aligned 16-byte group of eight u values that holds the item's pixel)
    float we[XU ? 8 : 1];
    if constexpr (XU) {
#pragma unroll
        for (int i = 0; i < 8; ++i) we[i] = xu.wexp[i];
    }
    unsigned okx = 0;
    constexpr int NSTORE = NPW * MT + (POOL ? (NPW / 2) * MT : 0);  // FULL: stores every wave issues per tile, all unconditional
    constexpr int NLOAD = NST * C::NXI;
    auto issue = [&](const TileOrg& o) {
        okx = 0;
        const int corner = (o.n * H + (o.h0 - 1)) * W + (o.w0 - 1);  // (pixel indices fit 32 bits, see the launcher's check)
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const int c0 = st * CINB + cgb * 8;
            const bool in_a = c0 < x.Ca;
            const bf16* base = in_a ? x.a + c0 : x.b + (c0 - x.Ca);
            const int pitch = in_a ? x.Ca : x.Cb;
            const bf16* cb = base + (long)corner * pitch;
            auto items = [&](auto INSIDE) {
#pragma unroll
                for (int j = 0; j < C::NXI; ++j) {
                    const int dy = xi_dyx[j] & 0xffff, dx = xi_dyx[j] >> 16, h = o.h0 - 1 + dy, w = o.w0 - 1 + dx;
                    const bool it_ok = (j + 1) * NT <= DP * CGB || tid + j * NT < DP * CGB;  // (a compile-time `true` for all but the last round)
                    const bool ok = it_ok && (decltype(INSIDE)::value || ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W));
                    // XU: a 2-byte item would be a lone 32-bit asm destination (hipcc then pairs it as the don't-care high half of 64-bit address
                    // arithmetic while the load is in flight); every lane loads the aligned group of 8 pixels around its own instead -- the same
                    // instruction as the other forms, an eighth of their lines -- and picks its pixel at the commit (row starts and tile origins
                    // are multiples of 8 pixels: the position inside the group is (dx - 1) & 7, tile-invariant)
                    const bf16* src = XU ? (ok ? xu.u + ((corner + dy * W + dx) & ~7) : xu.u) : (ok ? cb + (dy * W + dx) * pitch : base);
                    if constexpr (FULL) {
                        xr[st][j] = gload16_opaque(src);
                    } else {
                        const uint4 q = *reinterpret_cast<const uint4*>(src);
                        xr[st][j] = (u32x4){q.x, q.y, q.z, q.w};
                    }
                    okx |= ok ? 1u << (st * C::NXI + j) : 0u;
                }
            };
            // FULL: a tile whose whole domain lies inside the image skips the per-item bounds tests and address selects (scalar branch; both
            // sides issue the same loads in the same order)
            if (FULL && o.h0 >= 1 && o.w0 >= 1 && o.h0 + TH + 1 <= H && o.w0 + TW + 1 <= W) items(std::true_type{});
            else items(std::false_type{});
        }
    };
    // FULL: the hand-written waits.  `pend_stores`: the previous tile's epilogue stores were issued after these loads (false for the first tile)
    auto wait_loads = [&](bool pend_stores) {
        if constexpr (FULL) {
            if (pend_stores) {
                static_for_wait<NLOAD, NSTORE>(&xr[0][0]);
            } else {
                static_for_wait<NLOAD, 0>(&xr[0][0]);
            }
        }
    };
    const int kg = lane >> 4;
    auto tap_off = [](int tap) constexpr -> int { return tap < 9 ? ((tap / 3) * DW_ + tap % 3) * PX : 0; };
    auto boff_of = [&](int kc, bool& valid) -> int {
        if constexpr (CINB == 32) {
            valid = true;
            return tap_off(kc) + kg * 8;
        } else if constexpr (CINB == 16) {
            const int t0 = 2 * kc, t1 = 2 * kc + 1;
            valid = (kg < 2) ? t0 < 9 : t1 < 9;
            return ((kg < 2) ? tap_off(t0) : tap_off(t1)) + (kg & 1) * 8;
        } else {
            const int t0 = 4 * kc;
            valid = t0 + kg < 9;
            const int a = (kg & 1) ? tap_off(t0 + 1) : tap_off(t0), b = (kg & 1) ? tap_off(t0 + 3) : tap_off(t0 + 2);
            return (kg & 2) ? b : a;
        }
    };
    // N tiles of this wave: unit u = wave * UPW + i -> row pair u >> 1, column half u & 1; tile a = 2 * i + (row within the pair)
    int pty[NPW], ptx0[NPW];  // (wave-uniform: row and first column of N tile a; the lane adds lane & 15)
#pragma unroll
    for (int a = 0; a < NPW; ++a) {
        const int u = wave * C::UPW + (a >> 1);
        pty[a] = 2 * (u >> 1) + (a & 1);
        ptx0[a] = (u & 1) * 16;
    }
    float s1[MT][4], s2[MT][4];
#pragma unroll
    for (int b = 0; b < MT; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) s1[b][i] = s2[b][i] = 0.f;
    float sg[POOL ? MT : 1][4];
    if constexpr (POOL) {
#pragma unroll
        for (int b = 0; b < MT; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int m = b * 16 + (lane >> 4) * 4 + i;
                if (FULL && COUT == 8) m &= 7;  // (the duplicate rows must select like the rows they duplicate)
                sg[b][i] = (m < COUT && gamma[m] < 0.f) ? -1.f : 1.f;
            }
    }

    TileSched ts(tg.ntiles);
    TileIter<TW, TH> tit(tg, ts.first < ts.end ? ts.first : 0, ts.step);
    TileOrg org_next = tit.org();
    if (ts.first < ts.end) issue(org_next);
    auto tile_body = [&](long t, bool first) __attribute__((always_inline)) {
        const TileOrg org = org_next;
        wait_loads(!first);
        f32x4 acc[NPW][MT];
#pragma unroll
        for (int a = 0; a < NPW; ++a)
#pragma unroll
            for (int b = 0; b < MT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            if (st) lds_barrier();  // the previous stage's readers are done
            // ---- commit stage st: x~ -> tileX (four channels at a time)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const float* tp = s_trx + (st * CGB + cgb) * 24 + hf * 4;
                const f32x4 sc = *reinterpret_cast<const f32x4*>(tp), sh = *reinterpret_cast<const f32x4*>(tp + 8), lo = *reinterpret_cast<const f32x4*>(tp + 16);
#pragma unroll
                for (int j = 0; j < C::NXI; ++j) {
                    const int it = tid + j * NT;
                    if (DP * CGB % NT == 0 || it < DP * CGB) {
                        float v[4] = {0.f, 0.f, 0.f, 0.f};
                        if (okx & (1u << (st * C::NXI + j))) {
                            if constexpr (XU) {
                                const int sel = ((xi_dyx[j] >> 16) + 7) & 7;
                                const u32x4 q = xr[0][j];
                                const unsigned w2 = (sel & 4) ? ((sel & 2) ? q.w : q.z) : ((sel & 2) ? q.y : q.x);
                                const float uv = __uint_as_float((sel & 1) ? (w2 & 0xffff0000u) : (w2 << 16));
#pragma unroll
                                for (int i = 0; i < 4; ++i) v[i] = Elem<bf16>::round(we[hf * 4 + i] * uv);  // (cgb = 0: one 8-channel group)
                            } else {
                                half4(raw8_of(xr[st][j]), hf, v);
                            }
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = max_lo(fmaf(v[i], sc[i], sh[i]), lo[i]);
                        }
                        st4bf(tileX + (it / CGB) * PX + cgb * 8 + hf * 4, v);
                    }
                }
            }
            if (st == NST - 1) {
                __builtin_amdgcn_sched_barrier(0);
                if (t + ts.step < ts.end) {
                    tit.next();
                    org_next = tit.org();
                    issue(org_next);
                }
            }
            lds_barrier();
            // ---- MFMA: all K chunks of this stage
            int pbase[NPW];
#pragma unroll
            for (int a = 0; a < NPW; ++a) pbase[a] = (pty[a] * DW_ + ptx0[a] + l15) * PX;
            uint4 bcur[NPW], bnxt[NPW];
            auto load_b = [&](uint4 (&dst)[NPW], int kc) {
                bool bv;
                const int bo = boff_of(kc, bv);
#pragma unroll
                for (int a = 0; a < NPW; ++a) {
                    dst[a] = *reinterpret_cast<const uint4*>(tileX + pbase[a] + bo);
                    if (!bv) dst[a] = make_uint4(0, 0, 0, 0);
                }
            };
            load_b(bcur, 0);
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                uint4 wf[MT];
#pragma unroll
                for (int b = 0; b < MT; ++b) wf[b] = s_wf[((st * MT + b) * KC + kc) * 64 + lane];
                if (kc + 1 < KC) load_b(bnxt, kc + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < NPW; ++a)
#pragma unroll
                    for (int b = 0; b < MT; ++b) acc[a][b] = mfma16(wf[b], bcur[a], acc[a][b]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < NPW; ++a) bcur[a] = bnxt[a];
            }
        }
        // ---- epilogue: store z (4 consecutive channels per lane), statistics of the STORED values, optional 2x2 max-pool
        const int tb = (org.n * H + org.h0) * W + org.w0;
#pragma unroll
        for (int a = 0; a < NPW; ++a) {
            const bool pv = FULL || (org.h0 + pty[a] < H && org.w0 + ptx0[a] + l15 < W);
            bf16* zrow = z + (long)(tb + pty[a] * W + ptx0[a]) * COUT;  // (scalar)
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                const int m0 = b * 16 + (lane >> 4) * 4;
                const f32x4 v = acc[a][b];
                if constexpr (FULL) {  // unconditional store (duplicate rows -> same bytes, same address); statistics from the real rows only
                    store4(zrow + l15 * COUT + (COUT == 8 ? (m0 & 7) : m0), v[0], v[1], v[2], v[3]);
                } else {
                    if (pv && m0 < COUT) store4(zrow + l15 * COUT + m0, v[0], v[1], v[2], v[3]);
                }
                if (pv && m0 < COUT) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float q = Elem<bf16>::round(v[i]);
                        s1[b][i] += q;
                        s2[b][i] = fmaf(q, q, s2[b][i]);
                    }
                }
            }
        }
        if constexpr (POOL) {
            // relu(bn(z)) is monotone in z with the sign of gamma: the window's selected element is max z (gamma >= 0) or min z (gamma < 0); a window =
            // this lane's pixel and lane ^ 1 (same row) of N tiles a (even row) and a + 1 (the row below); see k_dwpw_fwd
            const int Hp = H >> 1, Wp = W >> 1;
#pragma unroll
            for (int a = 0; a < NPW; a += 2) {
                const int ph = (org.h0 + pty[a]) >> 1, pw = (org.w0 + ptx0[a] + l15) >> 1;
#pragma unroll
                for (int b = 0; b < MT; ++b) {
                    const int m0 = b * 16 + (lane >> 4) * 4;
                    float m4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float v0 = sg[b][i] * Elem<bf16>::round(acc[a][b][i]), v1 = sg[b][i] * Elem<bf16>::round(acc[a + 1][b][i]);
                        const float vv = fmaxf(v0, v1);
                        m4[i] = sg[b][i] * fmaxf(vv, dpp_f<0xB1>(vv));  // quad_perm [1,0,3,2]: the horizontally adjacent pixel
                    }
                    if constexpr (FULL) {  // (both lanes of a pair hold the same maximum and store it to the same address)
                        store4(pooled + (long)((org.n * Hp + ph) * Wp + pw) * COUT + (COUT == 8 ? (m0 & 7) : m0), m4[0], m4[1], m4[2], m4[3]);
                    } else {
                        if ((lane & 1) == 0 && ph < Hp && pw < Wp && m0 < COUT)
                            store4(pooled + (long)((org.n * Hp + ph) * Wp + pw) * COUT + m0, m4[0], m4[1], m4[2], m4[3]);
                    }
                }
            }
        }
        lds_barrier();  // all readers of tileX are done before the next commit
    };
    {
        long t = ts.first;
        if constexpr (FULL) {  // peeled first tile: the loop header then only sees states with the previous tile's stores pending
            if (t < ts.end) {
                tile_body(t, true);
                t += ts.step;
            }
            for (; t < ts.end; t += ts.step) tile_body(t, false);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (for the static checker: see k_mm_bwd)
        } else {
            for (; t < ts.end; t += ts.step) tile_body(t, false);
        }
    }
    // ---- statistics: lanes -> wave slots -> block partial [COUT][sum | sum of squares] (fixed order: deterministic)
#pragma unroll
    for (int b = 0; b < MT; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a1 = quad16_sum(s1[b][i]), a2 = quad16_sum(s2[b][i]);
            if ((lane & 15) == 0) {
                const int m = b * 16 + (lane >> 4) * 4 + i;
                s_stat[(wave * MT * 16 + m) * 2 + 0] = a1;
                s_stat[(wave * MT * 16 + m) * 2 + 1] = a2;
            }
        }
    __syncthreads();
    for (int e = tid; e < 2 * COUT; e += NT) {
        float s = 0.f;
        for (int w = 0; w < C::NW; ++w) s += s_stat[w * MT * 32 + e];
        if (mm_fwd_fink<CINB, COUT>() && fin.counter) __hip_atomic_store(ws + (long)blockIdx.x * (2 * COUT) + e, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (sc1: visible across XCDs once drained)
        else ws[(long)blockIdx.x * (2 * COUT) + e] = s;
    }
    if constexpr (!mm_fwd_fink<CINB, COUT>()) return;
    if (!fin.counter) return;
    // ---- last workgroup done: finalise the statistics here (see FwdFin)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* s_flag = reinterpret_cast<int*>(smem);              // (the tiles are dead)
    double* red = reinterpret_cast<double*>(smem + 64);      // [2 windows][8 chains][32 columns]
    if (tid == 0) *s_flag = __hip_atomic_fetch_add(fin.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (*s_flag == 0) return;
    static_assert(2 * COUT <= 64 && NT == 512, "two 32-element windows x 8 chains x 32 columns = the 512 threads");
    const int win = tid >> 8, t8 = tid & 255, col = t8 & 31, chain = t8 >> 5, e = win * 32 + col;
    red[(win * 8 + chain) * 32 + col] = bn_parts_chain_sum<true>(ws, gridDim.x, COUT, e, chain);
    __syncthreads();
    if (chain == 0) {
        const double* r = red + win * 256;
        const double tot = ((r[col] + r[32 + col]) + (r[64 + col] + r[96 + col])) + ((r[128 + col] + r[160 + col]) + (r[192 + col] + r[224 + col]));
        red[512 + win * 32 + col] = tot;
    }
    __syncthreads();
    if (tid == 0) {
        if (fin.nbt) *fin.nbt += 1;
        __hip_atomic_store(fin.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (ready for a replay of the same launch, e.g. a captured graph)
    }
    if (chain == 0 && e < 2 * COUT && !(e & 1))
        bn_finalize_channel(red[512 + e], red[512 + e + 1], fin.count, e >> 1, COUT, fin.gamma, fin.beta, fin.eps, fin.momentum, fin.tr, fin.saved, fin.run_mean,
                            fin.run_var, fin.lo);
}

// BatchNorm2d training statistics from per-block partials [nparts][C][sum | sum of squares] (fp32 partials, fp64 total, fixed summation order:
// bit-reproducible) -> load transform, saved mean | rstd, running statistics.  Same arithmetic as k_bn_finalize (det_fwd.hip).
__global__ __launch_bounds__(256) void k_bn_finalize_parts(const float* __restrict__ parts, int nparts, long count, int C, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, float momentum, float* __restrict__ tr,
                                                           float* __restrict__ saved, float* __restrict__ run_mean, float* __restrict__ run_var,
                                                           long long* __restrict__ nbt, float lo) {
    __shared__ double red[8][32];
    const int col = threadIdx.x & 31, chain = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + col;  // element of [C][2]
    red[chain][col] = bn_parts_chain_sum<false>(parts, nparts, C, e, chain);
    __syncthreads();
    if (chain != 0) return;
    const double tot = ((red[0][col] + red[1][col]) + (red[2][col] + red[3][col])) + ((red[4][col] + red[5][col]) + (red[6][col] + red[7][col]));
    red[0][col] = tot;
    __syncthreads();  // (only chain 0 = one wave half reaches here: 32 lanes of the first wave)
    if (e == 0 && nbt) *nbt += 1;
    if (e >= 2 * C || (e & 1)) return;
    bn_finalize_channel(tot, red[0][col + 1], count, e >> 1, C, gamma, beta, eps, momentum, tr, saved, run_mean, run_var, lo);
}

template <int CINB, int NST, int COUT>
static void mm_fwd_launch(const Src2<bf16>& x, const float* tra, const float* trb, const float* wdw, const float* wpw, bf16* z, float* ws, const float* gamma,
                          bf16* pooled, int N, int H, int W, int nb, const FwdFin& fin, hipStream_t st, const XuSrc& xu = XuSrc{nullptr, nullptr}) {
    using CC = MfCfg<CINB, NST, COUT>;
    const Tiling2 tg = make_tiling2(N, H, W, CC::TW, CC::TH);
    // (set per call: the attribute is per device and this is called from any thread; it is a cheap host-side table update)
    static const int full_on = env_int("OCRS_MM_FULL", 1);
    const bool full = full_on && H % CC::TH == 0 && W % CC::TW == 0 && (!pooled || ((H | W) & 1) == 0);
#define MF_LAUNCH(PO_, FU_)                                                                                                                          \
    {                                                                                                                                                \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mm_fwd<CINB, NST, COUT, PO_, FU_>), hipFuncAttributeMaxDynamicSharedMemorySize, CC::SMEM); \
        hipLaunchKernelGGL((k_mm_fwd<CINB, NST, COUT, PO_, FU_>), dim3(nb), dim3(CC::NT), CC::SMEM, st, x, tra, trb, wdw, wpw, z, ws, tg, gamma, pooled, fin, xu); \
    }
    if constexpr (CINB == 8 && NST == 1) {
        if (xu.u) {  // the block behind the first block (never pooled)
#define MF_LAUNCH_XU(FU_)                                                                                                                                 \
    {                                                                                                                                                     \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mm_fwd<CINB, NST, COUT, false, FU_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, CC::SMEM); \
        hipLaunchKernelGGL((k_mm_fwd<CINB, NST, COUT, false, FU_, true>), dim3(nb), dim3(CC::NT), CC::SMEM, st, x, tra, trb, wdw, wpw, z, ws, tg, gamma, pooled, fin, xu); \
    }
            if (full) MF_LAUNCH_XU(true) else MF_LAUNCH_XU(false)
#undef MF_LAUNCH_XU
            return;
        }
    }
    if (pooled) {
        if (full) MF_LAUNCH(true, true) else MF_LAUNCH(true, false)
    } else {
        if (full) MF_LAUNCH(false, true) else MF_LAUNCH(false, false)
    }
#undef MF_LAUNCH
}

extern "C" {

long ocrs_mm_fwd_supported(int Ca, int Cb, int Cout, int dtype) {
    if (dtype != 1 || !(Cout == 8 || Cout == 16 || Cout == 32)) return 0;
    const int Cin = Ca + Cb;
    if (Ca == 32 && Cb == 32) return Cout == 32;
    if ((Cin == 8 && Cout == 32) || (Cin == 32 && Cout == 8)) return 0;
    return (Cin == 8 || Cin == 16 || Cin == 32) && Ca % 8 == 0 && Cb % 8 == 0;
}
// number of per-block statistics partials ocrs_mm_fwd writes (ws = that many x 2 * Cout floats)
long ocrs_mm_fwd_nparts(int Ca, int Cb, int Cout, int N, int H, int W) {
    if (rs_fwd_supported(Ca, Cb, Cout, N, H, W)) return rs_fwd_blocks(N, H, W);  // (also the grid of the tiled kernel when such a launch pools: persistent, any size)
    const int cinb = (Ca == 32 && Cb == 32) ? 32 : Ca + Cb;
    const int th = mm_th(cinb, Cout, (Ca == 32 && Cb == 32) ? 2 : 1);
    return mm_grid(th, N, H, W, 0, cinb == 8 ? OCRS_MF_BPC8 : ((cinb == 16 && Cout == 8) ? OCRS_MF_BPC16_8 : 2));
}

// DepthwiseConv block forward on the matrix cores up to the pre-BatchNorm output (replaces ocrs_dwpw_fwd for bf16, Cin / Cout in {8, 16, 32}
// and the 32 | 32 concat): wdw [Cin][9] / wpw [Cout][Cin] fp32 masters; z [P][Cout]; ws: ocrs_mm_fwd_nparts() x [Cout][sum | sum^2] fp32
// per-block partials of the batch statistics (-> ocrs_bn_finalize_parts); gamma / pooled (nullable): as ocrs_dwpw_fwd.
static int mm_fwd_impl(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const float* wpw, void* z, float* ws,
                       const float* gamma, void* pooled, int Cout, int N, int H, int W, int dtype, const FwdFin& fin, hipStream_t st,
                       const XuSrc& xu = XuSrc{nullptr, nullptr}) {
    OCRS_CHECK_ARG((xa || xu.u) && tra && wdw && wpw && z && ws && (Cb == 0 || (xb && trb)) && (!pooled || gamma));
    OCRS_CHECK_ARG(ocrs_mm_fwd_supported(Ca, Cb, Cout, dtype) && (long)N * (H + 2) * (W + 2) < (1L << 31));
    const int Cin = Ca + Cb;
    Src2<bf16> x{(const bf16*)xa, (const bf16*)xb, Ca, Cb};
    const int nb = (int)ocrs_mm_fwd_nparts(Ca, Cb, Cout, N, H, W);
    if (!pooled && rs_fwd_supported(Ca, Cb, Cout, N, H, W)) {  // the row-streaming kernel (det_rs.hip)
        rs_fwd_launch(x, xu.u, xu.wexp, tra, trb, wdw, wpw, (bf16*)z, ws, Cout, N, H, W, nb, fin, st);
        OCRS_LAUNCH_CHECK();
        return OCRS_OK;
    }
#define MF_CASE(CB_, NS_, CO_) \
    if (Cin == CB_ * NS_ && Cout == CO_ && (NS_ == 1 || Ca == 32)) mm_fwd_launch<CB_, NS_, CO_>(x, tra, trb, wdw, wpw, (bf16*)z, ws, gamma, (bf16*)pooled, N, H, W, nb, fin, st, xu);
    MF_CASE(8, 1, 8) MF_CASE(8, 1, 16) MF_CASE(16, 1, 8) MF_CASE(16, 1, 16) MF_CASE(16, 1, 32) MF_CASE(32, 1, 16) MF_CASE(32, 1, 32) MF_CASE(32, 2, 32)
#undef MF_CASE
    if (fin.counter && Cin == 16 && Cout == 8)  // (see mm_fwd_fink: this shape's statistics are finalised by their own launch)
        hipLaunchKernelGGL(k_bn_finalize_parts, dim3((2 * Cout + 31) / 32), dim3(256), 0, st, ws, nb, fin.count, Cout, fin.gamma, fin.beta, fin.eps, fin.momentum,
                           fin.tr, fin.saved, fin.run_mean, fin.run_var, fin.nbt, fin.lo);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

int ocrs_mm_fwd(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const float* wpw, void* z, float* ws,
                const float* gamma, void* pooled, int Cout, int N, int H, int W, int dtype, hipStream_t st) {
    const FwdFin fin{nullptr, 0, nullptr, nullptr, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f};
    return mm_fwd_impl(xa, xb, Ca, Cb, tra, trb, wdw, wpw, z, ws, gamma, pooled, Cout, N, H, W, dtype, fin, st);
}

// ocrs_mm_fwd with ocrs_bn_finalize_parts folded into the launch (the last workgroup to finish finalises the statistics, bit-identical to the
// separate call): bn_w / bn_b = the block's BatchNorm weight / bias, counter = ONE zeroed 32-bit word (left zero), the rest as ocrs_bn_finalize_parts.
int ocrs_mm_fwd_fin(const void* xa, const void* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const float* wpw, void* z, float* ws,
                    const float* gamma, void* pooled, unsigned* counter, long count, const float* bn_w, const float* bn_b, float eps, float momentum, float* tr,
                    float* saved, float* run_mean, float* run_var, long long* nbt, float lo, int Cout, int N, int H, int W, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(counter && count > 0 && bn_w && bn_b && tr && saved);
    const FwdFin fin{counter, count, bn_w, bn_b, eps, momentum, tr, saved, run_mean, run_var, nbt, lo};
    return mm_fwd_impl(xa, xb, Ca, Cb, tra, trb, wdw, wpw, z, ws, gamma, pooled, Cout, N, H, W, dtype, fin, st);
}

// ocrs_mm_fwd_fin for the block behind the first block: the input is the first block's u plane + pointwise weight wexp [8] (see k_mm_fwd<..., XU>), tra its
// load transform [3][8]; no pooling.
int ocrs_mm_fwd_fin_xu(const void* xu, const float* wexp, const float* tra, const float* wdw, const float* wpw, void* z, float* ws, unsigned* counter, long count,
                       const float* bn_w, const float* bn_b, float eps, float momentum, float* tr, float* saved, float* run_mean, float* run_var, long long* nbt,
                       float lo, int Cout, int N, int H, int W, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(xu && wexp && counter && count > 0 && bn_w && bn_b && tr && saved && (Cout == 8 || Cout == 16) && W % 8 == 0);  // (aligned 8-pixel groups)
    const FwdFin fin{counter, count, bn_w, bn_b, eps, momentum, tr, saved, run_mean, run_var, nbt, lo};
    return mm_fwd_impl(nullptr, nullptr, 8, 0, tra, nullptr, wdw, wpw, z, ws, nullptr, nullptr, Cout, N, H, W, dtype, fin, st, XuSrc{(const bf16*)xu, wexp});
}

// nn.BatchNorm2d training statistics from ocrs_mm_fwd's per-block partials (models.py:23): see ocrs_bn_finalize.
int ocrs_bn_finalize_parts(const float* parts, int nparts, long count, int C, const float* gamma, const float* beta, float eps, float momentum, float* tr,
                           float* saved, float* run_mean, float* run_var, long long* nbt, float lo, hipStream_t st) {
    OCRS_CHECK_ARG(parts && nparts > 0 && gamma && beta && tr && saved && C > 0 && count > 0);
    hipLaunchKernelGGL(k_bn_finalize_parts, dim3((2 * C + 31) / 32), dim3(256), 0, st, parts, nparts, count, C, gamma, beta, eps, momentum, tr, saved, run_mean,
                       run_var, nbt, lo);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

}  // extern "C"
