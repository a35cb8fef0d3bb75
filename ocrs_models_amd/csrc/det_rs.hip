// Row-streaming form of the "everything on the matrix cores" DepthwiseConv block backward (reference ocrs_models/models.py:7-28; same
// mathematics as k_mm_bwd in det_mm.hip: Weff[o][(tap, c)] = Wpw[o][c] * Wdw[c][tap], dz = A * ghat + B * z + C,
// dx~ = Weff^T (*) dz, G_tap[c][o] = sum_q x~[q][c] dz[q - off(tap)][o], dWpw / dWdw from G) with a different execution structure.
//
// k_mm_bwd stages 12 x 32-pixel tiles per 512-thread block: two barriers per tile, a 2-D tile decode per tile, 24 % ring re-reads, the
// producers' BatchNorm-backward sums as a per-lane VALU epilogue.  Round 3 / 4 measured that its time is neither bytes nor any one
// unit but the per-tile chain of dependent phases (4.8 VALU wave-instructions and 0.63 KB of LDS traffic per pixel for 8 -> 8 channels).  Here:
//
//   * ONE WAVE owns a strip of 30 output columns (32 input columns: one halo column each side) and marches down the image two rows per tick.
//     Its dz rows live in a wave-private 6-row LDS ring, its x~ rows in a 2-slot tile: LDS operations of one wave execute in order, so there
//     is NO workgroup barrier in the main loop and the 16 waves of a CU drift apart freely (their VALU / LDS / MFMA / memory phases interleave).
//   * A lane is a (row of the pair, pixel) and holds ALL channels of its pixel: every per-channel BatchNorm / transform coefficient is
//     wave-uniform, i.e. a scalar-register operand of v_pk_fma_f32 (one of the two non-data operands; the other one sits in a VGPR pair) --
//     no LDS parameter reads (k_mm_bwd: 16 ds_read_b128 per tile and wave).
//   * Rows are contiguous in HBM: a tick's loads are 512-byte runs at (scalar tick offset + constant lane offset) through buffer descriptors
//     (out-of-range lanes read 0 / their stores are dropped by the hardware bounds check: no address selects, no divergent branches, so hipcc's
//     vmcnt bookkeeping is exact), issued TWO ticks ahead.
//   * Cin = 8: the 16-row MFMA M tile carries (row parity, channel): one MFMA set yields both rows of the pair from the 4 ring rows around them
//     (K = 4 x 3 taps x Cout = 96 for Cout = 8: no K padding, half the MFMAs / B-fragment reads / stores of the one-row form).
//   * The producers' BatchNorm-backward sums S1 = sum ghat', S2 = sum ghat' x~ (ghat' = dx~ [x~ > 0]) come out of the weight-gradient GEMM:
//     S2[c] = sum_{tap,o} Weff[o][(tap,c)] G_tap[c][o]  (dx~ is linear in dz), and S1 likewise from a second set of M rows that holds the
//     0/1 mask [x~ > 0] instead of x~ (for Cin = 8 those are rows 8..15 of the same M tile: free).  The per-lane epilogue (29 % of k_mm_bwd's
//     VALU work) is gone.  The sums are those of the UNROUNDED dx~ (k_mm_bwd sums the stored bf16 values): an unbiased 2^-9 / sqrt(pixels) difference.
//   * Work split: the (image, strip, row pair) steps are cut into equal contiguous ranges, one per wave (two warm-up ticks per range start /
//     strip change): no tail, no tile scheduler.
// HBM bytes per pixel: x (Cin) + z, g (Cout) in, dx~ (Cin) out = the 2 (Cin + Cout) of SURVEY.md 8(d), x 32/30 for the halo columns of the reads.
#include "det_rs.h"
#include <type_traits>

#ifndef OCRS_RS_WPS
#define OCRS_RS_WPS 3  // waves per SIMD the kernel is compiled for (register cap 512 / WPS) = resident 4-wave workgroups per CU.  Measured in the step (8 -> 8
                       // channels, level 0): 3 per CU 480-507 us, 4 per CU 546 us, 2 per CU 480 us -- the memory side prefers fewer, more coherent streams
#endif
#ifndef OCRS_RS_LD_NT
#define OCRS_RS_LD_NT 0          // 1: non-temporal prefetch loads (measurement knob)
#endif
#if OCRS_RS_LD_NT
#define OCRS_RS_LD_HINT " nt"
#else
#define OCRS_RS_LD_HINT ""
#endif
#ifndef OCRS_RS_ST_AUX
#define OCRS_RS_ST_AUX 0         // aux bits of the dx~ stores (measurement knob: 2 = nt)
#endif
#ifndef OCRS_RS_TICK_BARRIER
#define OCRS_RS_TICK_BARRIER 0   // 1: the four waves of a workgroup meet at every tick (measurement knob: keeps their four adjacent strips in step)
#endif
#ifndef OCRS_RS_WPS_G2
#define OCRS_RS_WPS_G2 3  // two gradient tensors: 8 more prefetch registers per set (139 VGPRs; at the 128 cap hipcc spills loop invariants, and a scratch reload is a vmcnt(0))
#endif
#ifndef OCRS_RS_8_16
#define OCRS_RS_8_16 1  // 8 -> 16 channels on this kernel (two waves per SIMD, per-channel coefficients of the 16-channel side in vector registers): 772 -> 725 us at
                        // level 0.  16 -> 8 and 16 -> 16 were measured too (-DOCRS_RS_16): 754 vs 645 us and 660 vs 227 us -- they stay on k_mm_bwd
#endif
#ifndef OCRS_RS_816_SC
#define OCRS_RS_816_SC 1
#endif
#ifndef OCRS_RS_WPS_16
#define OCRS_RS_WPS_16 2  // a 16-channel side: twice the prefetch registers / coefficient pairs / G accumulators (spills at the 168-register cap of 3)
#endif
#ifndef OCRS_RS_WPS_8_16
#define OCRS_RS_WPS_8_16 2  // 8 -> 16 channels, one gradient tensor: fits 168 registers without spills, but three workgroups per CU measured 831 us against 725 us
                            // with two (level 0; k_mm_bwd: 772 us) -- as for 8 -> 8 channels the memory side prefers fewer streams
#endif
template <int CIN, int COUT, bool G2>
constexpr int rs_wps() {
    return (CIN == 8 && COUT == 8) ? (G2 ? OCRS_RS_WPS_G2 : OCRS_RS_WPS) : ((CIN == 8 && COUT == 16 && !G2) ? OCRS_RS_WPS_8_16 : OCRS_RS_WPS_16);
}
static int rs_wps_rt(int Cin, int Cout, int g2) {
    return (Cin == 8 && Cout == 8) ? (g2 ? OCRS_RS_WPS_G2 : OCRS_RS_WPS) : ((Cin == 8 && Cout == 16 && !g2) ? OCRS_RS_WPS_8_16 : OCRS_RS_WPS_16);
}

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct RsArgs {
    const bf16 *xa, *xb;
    int Ca, Cb;
    const float *tra, *trb, *wdw, *wpw;
    int ldw;
    const bf16 *g1, *g2, *z;
    const float *gl, *whead;  // HEAD: dL/dlogit [N H W] fp32 and out_conv's weight [8] instead of g1 / g2
    const bf16* xu;           // XU: the block input as the first block's rank-one generator u [N H W] (bf16) ...
    const float* wexp;        //     ... and its pointwise weight [8]: x[p][c] = round(wexp[c] * u[p])
    const float* img;         // C1: the network input [N H W] fp32 (the first block's input) ...
    double* c1acc;            //     ... and its sums [8 replicas][32] fp64 (zeroed): [0, 8) R[c] = sum ghat1[c] u, [8, 17) T[tap] = sum du img(tap)
    const float *bn, *coef;
    bf16 *gxa, *gxb;
    float* ws;
    int N, H, W, NS, NP;
    int NB, PB, njobs;  // row blocks per image, row pairs per block, jobs = N * NB * NS
    BnFin fin;
    BwdLast bl;
};

template <int CIN, int COUT>
struct RsCfg {
    static constexpr int NW = 4, NT = 256;
    static constexpr int SW = 30;                    // output columns per strip (the 32 staged columns minus one halo column each side)
    static constexpr int PDB = COUT * 2;             // bytes per pixel of a ring row
    static constexpr int ROWB = 34 * PDB;            // ring row: the 32 staged pixels at positions 1..32, positions 0 / 33 stay zero (the reads of the
                                                     // discarded edge outputs land there)
    static constexpr int RROWS = 6;                  // rows 2cp - 1 .. 2cp + 2 of the pair being computed + the pair being committed
    static constexpr int RINGB = RROWS * ROWB + 64;  // (+ slack: a zero-weight K slot reads one position past the last row)
    static constexpr int XPB = CIN * 2;              // x tile: bytes per pixel of one plane (plane 0: x~, plane 1: [x~ > 0] as 0 / 1)
    static constexpr int XROWB = 32 * XPB, XPLANEB = 2 * XROWB, XSLOTB = 2 * XPLANEB, XB = 2 * XSLOTB;  // [slot 2][plane 2][row 2][32 px]
    static constexpr bool DUAL = CIN == 8;           // M tile = (row parity, channel)
    static constexpr int NPASS = DUAL ? 1 : 2;       // dgrad passes per tick (one per row of the pair when M = 16 channels)
    static constexpr int NDY = DUAL ? 4 : 3;         // ring rows a dgrad pass reads
    static constexpr int G8 = COUT / 8;              // 8-channel groups per dz pixel
    static constexpr int TPC = 4 / G8;               // column offsets (dxi) per 32-deep K chunk: a chunk = ONE ring row (wave-uniform row base)
    static constexpr int CPD = (3 + TPC - 1) / TPC;  // chunks per ring row (the K slots past dxi = 2 carry zero weights)
    static constexpr int KC = NDY * CPD;
    static constexpr int MTG = (2 * CIN) / 16;       // weight-gradient M tiles: [x~ | mask]
    // weight-gradient N units of 16 (tap, o) columns, each inside ONE kernel row ky: Cout = 8: {kx 0 | kx 1}, {kx 2 | (dup)}; Cout = 16: one tap
    static constexpr int UPK = COUT == 8 ? 2 : 3;    // units per kernel row
    static constexpr int NU = 3 * UPK;
    static constexpr int NZ = COUT / 8, NX = CIN / 8;  // 16-byte items per lane and tick
    static constexpr int OFF_RING = 64;              // (+ 64: position -1 of wave 0's first row)
    static constexpr int OFF_XT = OFF_RING + NW * RINGB;
    static constexpr int STB = 2 * 32 * CIN * 2;     // dx~ staging of a row pair (the MFMA output layout regrouped to one pixel per lane: 16-byte stores)
    static constexpr int OFF_ST = OFF_XT + NW * XB;
    static constexpr int OFF_WF = OFF_ST + NW * STB;
    static constexpr int OFF_PAR = OFF_WF + KC * 1024;
    static constexpr int PAR_FLOATS = 6 * COUT + 3 * CIN + 9 * CIN + COUT * CIN;  // bn [3][COUT] | coef [3][COUT] | trx [3][CIN] | w9 [CIN][9] | wp [COUT][CIN]
    static constexpr int SMEM_MAIN = OFF_PAR + PAR_FLOATS * 4;
    static constexpr int SLOT_BYTES = NW * MTG * NU * 1024;  // flush: every wave's G accumulators
    static constexpr int SMEM = SMEM_MAIN > SLOT_BYTES ? SMEM_MAIN : SLOT_BYTES;
    static constexpr int PART = COUT * CIN + 9 * CIN + 2 * CIN;  // = MmCfg::PART (k_mm_bwd_reduce's element layout)
};

__device__ __forceinline__ f32x4 mfma_bf(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_bf(const bf16x8& a, const bf16x8& b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ f32x2 unpk(unsigned w) { return (f32x2){__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}; }
__device__ __forceinline__ float usc(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }  // -> scalar register
__device__ __forceinline__ f32x2 vreg(f32x2 v) {  // pin a (uniform) pair into vector registers: the second non-data operand of a v_pk_fma_f32
    asm volatile("" : "+v"(v));
    return v;
}
// ---- buffer descriptor as four scalar words (raw buffer, 32-bit offsets, hardware range check against `bytes`)
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}
// Prefetch loads the compiler does not see, with hand-written waits (see det_mm.hip: with stores pending in the same counter hipcc waits
// vmcnt(0) in front of the first use of a loaded register, i.e. for the NEXT tick's loads and for the acknowledgement of the last stores).
// tools/check_rs_loads.py verifies that hipcc leaves the destination registers alone between the load and its wait.
__device__ __forceinline__ u32x4 bload16_opaque(const i32x4& rsrc, int voff) {
    u32x4 r;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" OCRS_RS_LD_HINT : "=&v"(r) : "v"(voff), "s"(rsrc) : "memory");
    return r;
}
__device__ __forceinline__ unsigned bload2_opaque(const i32x4& rsrc, int voff) {
    unsigned r;
    asm volatile("buffer_load_ushort %0, %1, %2, 0 offen" OCRS_RS_LD_HINT : "=&v"(r) : "v"(voff), "s"(rsrc) : "memory");
    return r;
}
__device__ __forceinline__ unsigned bload4_opaque(const i32x4& rsrc, int voff) {
    unsigned r;
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen" OCRS_RS_LD_HINT : "=&v"(r) : "v"(voff), "s"(rsrc) : "memory");
    return r;
}
// in-place forms ("+v": the load's destination IS the variable's current register -- with a fresh "=&v" output hipcc may give the value a new register
// and join the two at a loop header with a copy issued while the load is still in flight, which tools/check_rs_loads.py caught in k_rs_fwd)
__device__ __forceinline__ void bload16_inplace(u32x4& r, const i32x4& rsrc, int voff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" OCRS_RS_LD_HINT : "+v"(r) : "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void bload2_inplace(unsigned& r, const i32x4& rsrc, int voff) {
    asm volatile("buffer_load_ushort %0, %1, %2, 0 offen" OCRS_RS_LD_HINT : "+v"(r) : "v"(voff), "s"(rsrc) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm(unsigned& r) {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%1) ; releases %0" : "+v"(r) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm(u32x4& r) {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%1) ; releases %0" : "+v"(r) : "n"(N) : "memory");  // (the comment names the registers for tools/check_rs_loads.py)
}
__device__ __forceinline__ const bf16* lds_at(const char* smem, unsigned off) { return reinterpret_cast<const bf16*>(smem + off); }

// one tick of a wave's program: commit row pair q of strip s of image n into the ring / x tile, then compute pair q - 1 (comp = 0: a warm-up
// tick -- the same vector-memory instructions with the stores switched off, so that the hand-counted vmcnt waits hold on every path)
struct RsTick {
    int n, s, q, qm3, comp, valid;  // qm3 = q mod 3 (ring rows 2 qm3, 2 qm3 + 1)
};
// Work order.  A job = one strip x one block of PB row pairs; jobs are numbered (image, row block, strip) with the strip fastest and dealt
// round-robin to the waves (wave v runs jobs v, v + nwaves, ...; v numbered XCD-major), so that at any time neighbouring waves -- of one
// workgroup, then of one XCD -- walk NEIGHBOURING strips down the same rows: the 128-byte lines two strips share (a strip's 480 / 512-byte
// row segments are not line-aligned) are fetched / completed once in the L2 they share.  tools/probes/bw_probe.hip measured the alternative
// (a wave walking one strip top to bottom, its neighbours' rows far away in time): 3.65 TB/s instead of 5.2-5.5 for 3 reads + 1 write -- every
// shared line goes to HBM twice and every partial line write back as a read-modify-write (the first form of this kernel: 680 us for 8 -> 8
// channels at level 0 with its memory operations compiled out at 271).  The last, partial round of jobs costs little: the kernel is
// bandwidth-bound, the waves that are left run faster.
struct RsGen {  // (all wave-uniform: scalar registers)
    int j, njobs, nw, NS, NB, PB, NP;
    int n, s, p, pend, pm3, phase, have;
    __device__ __forceinline__ void load_job() {
        have = j < njobs;
        if (!have) return;  // (n, s, p stay on the last valid position: the dead ticks re-load it)
        const int per = NB * NS;
        n = j / per;
        const int r = j - n * per, rb = r / NS;
        s = r - rb * NS;
        p = rb * PB;
        pend = p + PB < NP ? p + PB : NP;
        pm3 = p % 3;
        phase = 0;
    }
    __device__ __forceinline__ RsGen(int first, int njobs_, int nw_, int NS_, int NB_, int PB_, int NP_)
        : j(first), njobs(njobs_), nw(nw_), NS(NS_), NB(NB_), PB(PB_), NP(NP_), n(0), s(0), p(0), pend(0), pm3(0), phase(0) {
        load_job();
    }
    static __device__ __forceinline__ int wrap3(int v) { return v < 0 ? v + 3 : (v >= 3 ? v - 3 : v); }
    __device__ __forceinline__ RsTick next() {
        RsTick t;
        t.n = n;
        t.s = s;
        t.valid = have;
        t.comp = 0;
        if (!have) {  // past the end: a harmless re-load of the last position (keeps the loop body free of conditional loads)
            t.q = p < NP ? p : NP - 1;
            t.qm3 = pm3;
            return t;
        }
        if (phase == 0) {
            t.q = p - 1;
            t.qm3 = wrap3(pm3 - 1);
            phase = 1;
        } else if (phase == 1) {
            t.q = p;
            t.qm3 = pm3;
            phase = 2;
        } else {
            t.q = p + 1;
            t.qm3 = wrap3(pm3 + 1);
            t.comp = 1;
            ++p;
            pm3 = wrap3(pm3 + 1);
            if (p == pend) {
                j += nw;
                load_job();
            }
        }
        return t;
    }
};

// element e of [C][2] summed over the per-block partials by chain `chain` (of 8): k_bn_finalize_parts's association order (det_mm.hip: bn_parts_chain_sum<true>)
__device__ __forceinline__ double rs_parts_chain_sum(const float* __restrict__ parts, int nparts, int C, int e, int chain) {
    double s = 0.0;
    if (e < 2 * C) {
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        auto ld = [&](long i) -> double { return (double)__hip_atomic_load(parts + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
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

}  // namespace

// HEAD: the block in front of out_conv -- dL/dy[p][c] = round(gl[p] * whead[c]) formed here from ocrs_head_bwd_gl's 4-byte-per-pixel gl (see there)
// XU: the block behind the first block (in_conv.seq.1) -- its input x[p][c] = round(wexp[c] * u[p]) is rebuilt from the 2-byte-per-pixel u plane (k_c1_fwd2)
// C1 (round 5, with XU): the weight gradient of the FIRST block (models.py:115, Conv2d(1, 8) depthwise + pointwise) is linear in what this launch has in
// its hands -- dz1[c] = A[c] ghat1[c] + B[c] z1[c] + C[c] with A = gamma rstd known from the forward and ghat1 = dx~ [x~ > 0] formed here -- so instead
// of writing dx~ (16 B / pixel) for a separate pass (k_c1_bwd2: reads it back + the image, 170 us) a lane accumulates, for its pixel,
//   R[c] += ghat1[c] u   and   T[tap] += (sum_c wexp[c] A[c] ghat1[c]) img[p + tap]
// (17 per-lane accumulators; the image comes in as one more prefetched dword per lane and tick and lives in a wave-private LDS ring next to u);
// k_c1_bwd_fin combines them with the forward-only sums (k_c1_fwd2: sum u, u^2, u img(tap), img(tap)) once the batch sums S1 / S2 are complete.
// This launch then stores no dx~ at all.
template <int CIN, int COUT, bool G2, bool SPLIT, bool HEAD = false, bool XU = false, bool C1 = false>
__global__ __launch_bounds__(256, (rs_wps<CIN, COUT, G2>())) void k_rs_bwd(RsArgs A) {
    static_assert(!HEAD || (!G2 && COUT == 8), "head gradient: one 8-channel source");
    static_assert(!XU || (CIN == 8 && !SPLIT), "u plane: one 8-channel source");
    static_assert(!C1 || XU, "first-block sums: the block behind the first block");
    using C = RsCfg<CIN, COUT>;
    constexpr int PDB = C::PDB, ROWB = C::ROWB, RINGB = C::RINGB, XPB = C::XPB, KC = C::KC, NU = C::NU, MTG = C::MTG, NZ = C::NZ, NX = C::NX, SW = C::SW;
    constexpr int G8 = C::G8, TPC = C::TPC, CPD = C::CPD, UPK = C::UPK;
    constexpr bool DUAL = C::DUAL;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    uint4* s_wf = reinterpret_cast<uint4*>(smem + C::OFF_WF);
    float* s_bn = reinterpret_cast<float*>(smem + C::OFF_PAR);  // [3][COUT]
    float* s_cf = s_bn + 3 * COUT;                               // [3][COUT]
    float* s_trx = s_cf + 3 * COUT;                              // [3][CIN] scale | shift | lo
    float* s_w9 = s_trx + 3 * CIN;                               // [CIN][9]
    float* s_wp = s_w9 + 9 * CIN;                                // [COUT][CIN]
    const int H = A.H, W = A.W;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- prologue (block-wide): parameters, effective-weight A fragments, zeroed rings / x tiles (the pad positions stay zero)
    for (int i = tid; i < 3 * COUT; i += C::NT) s_bn[i] = A.bn[i];
    if (A.fin.gsum) {
        bn_fin_coef(A.fin, COUT, s_cf, tid, C::NT, blockIdx.x == 0);
    } else {
        for (int i = tid; i < 3 * COUT; i += C::NT) s_cf[i] = A.coef[i];
    }
    for (int i = tid; i < 3 * CIN; i += C::NT) {
        const int r = i / CIN, c = i - r * CIN;
        s_trx[i] = c < A.Ca ? A.tra[r * A.Ca + c] : A.trb[r * A.Cb + (c - A.Ca)];
    }
    for (int i = tid; i < 9 * CIN; i += C::NT) s_w9[i] = A.wdw[i];
    for (int i = tid; i < COUT * CIN; i += C::NT) s_wp[i] = A.wpw[(i / CIN) * A.ldw + (i % CIN)];
    for (int i = tid; i < C::OFF_WF / 16; i += C::NT) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
    constexpr int C1ROW = 34 * 4, C1PLANE = 6 * C1ROW, C1B = 2 * C1PLANE + 32;  // per wave: image ring | u ring, fp32 [6 rows][34] (positions 0 / 33 stay zero)
    if constexpr (C1) {
        for (int i = tid; i < C::NW * C1B / 4; i += C::NT) reinterpret_cast<float*>(smem + C::SMEM)[i] = 0.f;
    }
    __syncthreads();
    // A[m][k] of the dgrad GEMM.  Chunk kc = (ring row dy, part h); K slot (lane group kgl, j): column offset dxi = h * TPC + kgl / G8 (dz pixel =
    // output pixel + dxi - 1), channel o = (kgl % G8) * 8 + j.  The dz pixel (dy, dxi) is conv tap (ky, kx) = (rp + 2 - dy, 2 - dxi) of the
    // output row rp.  DUAL: m = (rp, c), dy in 0..3; otherwise m = c, and the fragment is that of rp = 0 (pass ps reads ring rows ps + dy).
    for (int f = tid; f < KC * 64; f += C::NT) {
        const int l = f & 63, kc = f >> 6, m = l & 15, kgl = l >> 4;
        const int dy = kc / CPD, h = kc - dy * CPD;
        const int rp = DUAL ? (m >> 3) : 0, c = DUAL ? (m & 7) : m;
        const int dxi = h * TPC + kgl / G8, ky = rp + 2 - dy, kx = 2 - dxi;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int o = (kgl % G8) * 8 + j;
            v[j] = (dxi <= 2 && ky >= 0 && ky <= 2) ? s_w9[c * 9 + ky * 3 + kx] * s_wp[o * CIN + c] : 0.f;
        }
        s_wf[f] = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
    }
    __syncthreads();

    // ---- per-channel coefficients: one operand of every packed fma from scalar registers, the other one pinned into a VGPR pair
    // (a 16-channel side: its coefficients do not fit the scalar register file next to the descriptors -- hipcc then spills SGPRs inside the tick
    //  loop -- so that side keeps all of them in vector registers: these instantiations run two waves per SIMD with 256 registers each)
    constexpr bool SO = COUT == 8, SI = CIN == 8;  // scalar-register coefficients
    auto sv2 = [](auto SC, float a, float b) -> f32x2 {
        if constexpr (decltype(SC)::value) return (f32x2){usc(a), usc(b)};
        else return vreg((f32x2){a, b});
    };
    using BO = std::integral_constant<bool, SO>;
    using BI = std::integral_constant<bool, SI>;
    using BC = std::integral_constant<bool, SO || (SI && OCRS_RS_816_SC)>;  // 8 -> 16 channels: the dz coefficients A, B stay scalar (32 SGPRs), the BatchNorm scale goes to vector registers
    f32x2 bs2[COUT / 2], bt2[COUT / 2], ca2[COUT / 2], cb2[COUT / 2], cc2[COUT / 2], sc2[CIN / 2], sh2[CIN / 2];
    float lo1[CIN];
#pragma unroll
    for (int i = 0; i < COUT / 2; ++i) {
        bs2[i] = sv2(BO{}, s_bn[2 * i], s_bn[2 * i + 1]);
        bt2[i] = vreg((f32x2){s_bn[COUT + 2 * i], s_bn[COUT + 2 * i + 1]});
        ca2[i] = sv2(BC{}, s_cf[2 * i], s_cf[2 * i + 1]);
        cb2[i] = sv2(BC{}, s_cf[COUT + 2 * i], s_cf[COUT + 2 * i + 1]);
        cc2[i] = vreg((f32x2){s_cf[2 * COUT + 2 * i], s_cf[2 * COUT + 2 * i + 1]});
    }
#pragma unroll
    for (int i = 0; i < CIN / 2; ++i) {
        sc2[i] = sv2(BI{}, s_trx[2 * i], s_trx[2 * i + 1]);
        sh2[i] = vreg((f32x2){s_trx[CIN + 2 * i], s_trx[CIN + 2 * i + 1]});
    }
#pragma unroll
    for (int i = 0; i < CIN; ++i) {
        if constexpr (SI) lo1[i] = usc(s_trx[2 * CIN + i]);
        else {
            lo1[i] = s_trx[2 * CIN + i];
            asm volatile("" : "+v"(lo1[i]));
        }
    }

    float wh[HEAD ? 8 : 1], we[XU ? 8 : 1];
    if constexpr (HEAD) {
#pragma unroll
        for (int i = 0; i < 8; ++i) wh[i] = usc(A.whead[i]);
    }
    if constexpr (XU) {
#pragma unroll
        for (int i = 0; i < 8; ++i) we[i] = usc(A.wexp[i]);
    }
    // ---- buffer descriptors (hardware bounds check: out-of-range loads return 0, out-of-range stores are dropped)
    const unsigned npix = (unsigned)A.N * (unsigned)H * (unsigned)W;
    const int Ca = A.Ca, Cb = A.Cb;
    const i32x4 r_z = make_rsrc(A.z, npix * PDB), r_g1 = HEAD ? make_rsrc(A.gl, npix * 4) : make_rsrc(A.g1, npix * PDB),
                r_g2 = make_rsrc(G2 ? A.g2 : A.g1, npix * PDB);
    const i32x4 r_xa = XU ? make_rsrc(A.xu, npix * 2) : make_rsrc(A.xa, npix * Ca * 2), r_xb = make_rsrc(SPLIT ? A.xb : A.xa, npix * (SPLIT ? Cb : Ca) * 2);
    const i32x4 r_img = make_rsrc(C1 ? (const void*)A.img : (const void*)A.z, npix * (C1 ? 4u : (unsigned)PDB));
    const __amdgpu_buffer_rsrc_t w_a = __builtin_amdgcn_make_buffer_rsrc((void*)A.gxa, 0, npix * Ca * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_b = __builtin_amdgcn_make_buffer_rsrc((void*)(SPLIT ? A.gxb : A.gxa), 0, npix * (SPLIT ? Cb : Ca) * 2, 0x00020000);

    // ---- lane constants
    const int rr = lane >> 5, px = lane & 31;                       // commit: this lane's (row of the pair, staged pixel)
    const int lz = (rr * W + px) * PDB;                             // z / g byte offset from the tick's corner pixel
    int lxo[NX];                                                    // x items: byte offset, source by item
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const bool in_a = !SPLIT || 8 * j < Ca;
        lxo[j] = (rr * W + px) * (in_a ? Ca : Cb) * 2 + (in_a ? 8 * j : 8 * j - Ca) * 2;
    }
    const unsigned ring_w = C::OFF_RING + wave * RINGB, xt_w = C::OFF_XT + wave * C::XB;
    const unsigned wl = ring_w + rr * ROWB + (px + 1) * PDB;       // ring write (+ the pair's row base)
    const unsigned xl = xt_w + rr * C::XROWB + px * XPB;           // x tile write (+ slot)
    // dgrad B fragments: output pixel l15 (+ 16 per N tile) reads position l15 + dxi, K slot kg -> (dxi = kg / G8 (+ h * TPC), 8-channel group kg % G8)
    const unsigned lbl = (l15 + kg / G8) * PDB + (kg % G8) * 16;
    // dgrad stores: this lane's 4 channels of pixel l15 (+ 16 per N tile)
    const int st_rp = DUAL ? (kg >> 1) : 0;                          // (DUAL: M rows 8..15 are the second row of the pair)
    const int st_ch = DUAL ? (kg & 1) * 4 : kg * 4;
    const unsigned stw = C::OFF_ST + wave * C::STB + (st_rp * 32 + l15) * (CIN * 2) + st_ch * 2;  // staging write: this lane's 4 channels of (row st_rp [+ pass], pixel l15 [+ 16])
    const unsigned str = C::OFF_ST + wave * C::STB + (rr * 32 + px) * (CIN * 2);                   // staging read: (row rr, pixel px), 16 bytes per item
    // weight gradient: transpose-read geometry (see lds_tr4): K = pixels prow (+16), 4 consecutive M / N columns at quad cq
    const int prow = 4 * kg + (l15 >> 2), cq = lane & 3;
    const unsigned lxa = xt_w + prow * XPB + (DUAL ? (cq >> 1) * C::XPLANEB + (cq & 1) * 8 : cq * 8);  // A: x~ | mask (DUAL: mask = M rows 8..15)
    // B, Cout = 8: unit type 0 = {kx 0 | kx 1} by column half, type 1 = {kx 2 | kx 2}; x pixel q pairs with dz pixel q + 1 - kx = position q + 2 - kx.
    // Cout = 16: unit = one tap, the kx shift is an immediate
    const unsigned lw0 = COUT == 8 ? (prow + 2 - (cq >> 1)) * PDB + (cq & 1) * 8 : (prow + 2) * PDB + cq * 8;
    const unsigned lw1 = prow * PDB + (cq & 1) * 8;

    f32x4 accG[MTG][NU];
#pragma unroll
    for (int a = 0; a < MTG; ++a)
#pragma unroll
        for (int u = 0; u < NU; ++u) accG[a][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr bool WREG = KC <= 4 && rs_wps<CIN, COUT, G2>() <= 3;  // the A fragments stay in registers when they are few
    u32x4 wfr[WREG ? KC : 1];
    if constexpr (WREG) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const uint4 q = s_wf[kc * 64 + lane];
            wfr[kc] = (u32x4){q.x, q.y, q.z, q.w};
        }
    }

    // ---- this wave's jobs (workgroup b runs on XCD b % 8: number the waves XCD-major)
    const int nblk = gridDim.x;
    const int vblk = (nblk & 7) == 0 ? (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    RsGen gen(vblk * C::NW + wave, A.njobs, nblk * C::NW, A.NS, A.NB, A.PB, A.NP);

    u32x4 pz[2][NZ], pg[2][HEAD ? 1 : NZ], pg2[2][G2 ? NZ : 1], pxr[2][NX];
    unsigned pgl[2] = {0u, 0u};  // HEAD: this lane's gl
    unsigned pxu[2] = {0u, 0u};  // XU: this lane's u (bf16 bits)
    unsigned pim[2] = {0u, 0u};  // C1: this lane's image pixel (fp32 bits)
    const unsigned c1_w = C::SMEM + wave * C1B;  // C1: this wave's image / u rings
    f32x2 c1R[C1 ? 4 : 1], c1k[C1 ? 4 : 1];  // (channel pairs: packed fp32 FMAs)
    float c1T[C1 ? 9 : 1];
    if constexpr (C1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            c1R[i] = (f32x2){0.f, 0.f};
            c1k[i] = (f32x2){usc(A.wexp[2 * i] * s_trx[2 * i]), usc(A.wexp[2 * i + 1] * s_trx[2 * i + 1])};  // wexp[c] * (gamma rstd)[c]: the load transform's scale IS A[c]
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) c1T[i] = 0.f;
    }
    auto corner = [&](const RsTick& t) -> int {  // pixel index of (row 2q clamped into the image, column 30 s - 1): may be -1 / beyond a row end
        const int qc = t.q < 0 ? 0 : (t.q >= A.NP ? A.NP - 1 : t.q);
#ifdef OCRS_RS_NOLOAD  // (floor-measurement build: every tick re-reads the same lines)
        return (wave * 2) * W;
#endif
        return (t.n * H + 2 * qc) * W + SW * t.s - 1;
    };
    auto issue = [&](auto ST, const RsTick& t) {
        constexpr int S = decltype(ST)::value;
        const int cp = corner(t);
        const int oz = cp * PDB;
#pragma unroll
        for (int j = 0; j < NZ; ++j) {
            pz[S][j] = bload16_opaque(r_z, oz + lz + 16 * j);
            if constexpr (!HEAD) pg[S][j] = bload16_opaque(r_g1, oz + lz + 16 * j);
            if constexpr (G2) pg2[S][j] = bload16_opaque(r_g2, oz + lz + 16 * j);
        }
        if constexpr (HEAD) pgl[S] = bload4_opaque(r_g1, (cp + rr * W + px) * 4);
        if constexpr (C1) pim[S] = bload4_opaque(r_img, (cp + rr * W + px) * 4);
        if constexpr (XU) {
            pxu[S] = bload2_opaque(r_xa, (cp + rr * W + px) * 2);
        } else {
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                const bool in_a = !SPLIT || 8 * j < Ca;
                pxr[S][j] = bload16_opaque(in_a ? r_xa : r_xb, cp * ((in_a ? Ca : Cb) * 2) + lxo[j]);
            }
        }
    };
    // operations a tick issues: its loads (behind its commit), then its compute's stores
#ifdef OCRS_RS_NOCOMPUTE
    constexpr int NLOADS = NZ * (G2 ? 3 : (HEAD ? 1 : 2)) + (HEAD ? 1 : 0) + NX + (C1 ? 1 : 0), NSTORE = 0;
#else
    constexpr int NLOADS = NZ * (G2 ? 3 : (HEAD ? 1 : 2)) + (HEAD ? 1 : 0) + NX + (C1 ? 1 : 0), NSTORE = C1 ? 0 : NX;  // (C1: dx~ is consumed here, not stored)
#endif
    auto commit = [&](auto ST, auto YOUNGER, const RsTick& t) {
        constexpr int S = decltype(ST)::value;
        {   // the hand-written wait: YOUNGER = operations issued after this set's last load (vector memory operations retire in order)
#ifdef OCRS_RS_DRAIN  // (debug build: wait for everything)
            constexpr int Y = 0;
#else
            constexpr int Y = decltype(YOUNGER)::value;
#endif
            if constexpr (XU) wait_vm<Y>(pxu[S]);
            else wait_vm<Y>(pxr[S][NX - 1]);
            if constexpr (C1) wait_vm<Y>(pim[S]);
#pragma unroll
            for (int j = 0; j < NZ; ++j) {
                wait_vm<Y>(pz[S][j]);
                if constexpr (!HEAD) wait_vm<Y>(pg[S][j]);
                if constexpr (G2) wait_vm<Y>(pg2[S][j]);
            }
            if constexpr (HEAD) wait_vm<Y>(pgl[S]);
            if constexpr (!XU) {
#pragma unroll
                for (int j = 0; j + 1 < NX; ++j) wait_vm<Y>(pxr[S][j]);
            }
        }
        const int col = SW * t.s - 1 + px;
        const int row = 2 * t.q + rr;
        const bool ok = (unsigned)col < (unsigned)W && (unsigned)row < (unsigned)H;
        const bool okx = ok && px >= 1 && px <= SW;
        const unsigned wrow = (unsigned)(2 * t.qm3) * ROWB;
        if constexpr (C1) {  // image and u of this lane's pixel into the wave's rings (row 2 qm3 + rr, position px + 1; zero outside the image)
            const unsigned ro = c1_w + (unsigned)((2 * t.qm3 + rr) * C1ROW + (px + 1) * 4);
            *reinterpret_cast<float*>(smem + ro) = ok ? __uint_as_float(pim[S]) : 0.f;
            *reinterpret_cast<float*>(smem + ro + C1PLANE) = ok ? __uint_as_float(pxu[S] << 16) : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NZ; ++j) {
            unsigned w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ci = 4 * j + k;  // channel pair
                const f32x2 zv = unpk(pz[S][j][k]);
                f32x2 gv;
                if constexpr (HEAD) {  // the stored gradient of the separate path: gl * w[c] rounded to bf16
                    const float glv = __uint_as_float(pgl[S]);
                    gv = unpk(cvt_pk(glv * wh[2 * ci], glv * wh[2 * ci + 1]));
                } else {
                    gv = unpk(pg[S][j][k]);
                }
                if constexpr (G2) gv += unpk(pg2[S][j][k]);
                const f32x2 y = __builtin_elementwise_fma(zv, bs2[ci], bt2[ci]);
                f32x2 gh;
                gh.x = y.x > 0.f ? gv.x : 0.f;
                gh.y = y.y > 0.f ? gv.y : 0.f;
                const f32x2 d = __builtin_elementwise_fma(ca2[ci], gh, __builtin_elementwise_fma(cb2[ci], zv, cc2[ci]));
                const unsigned pk = cvt_pk(d.x, d.y);
                w[k] = ok ? pk : 0u;
            }
            *reinterpret_cast<uint4*>(smem + wl + wrow + 16 * j) = make_uint4(w[0], w[1], w[2], w[3]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned xs = (unsigned)(t.q & 1) * C::XSLOTB;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            unsigned w[4], mk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ci = 4 * j + k;
                f32x2 xin;
                if constexpr (XU) {  // the first block's stored output: wexp[c] * u rounded to bf16
                    const float uv = __uint_as_float(pxu[S] << 16);
                    xin = unpk(cvt_pk(uv * we[2 * ci], uv * we[2 * ci + 1]));
                } else {
                    xin = unpk(pxr[S][j][k]);
                }
                f32x2 v = __builtin_elementwise_fma(xin, sc2[ci], sh2[ci]);
                if constexpr (SI) {
                    asm("v_max_f32 %0, %1, %2" : "=v"(v.x) : "v"(v.x), "s"(lo1[2 * ci]));
                    asm("v_max_f32 %0, %1, %2" : "=v"(v.y) : "v"(v.y), "s"(lo1[2 * ci + 1]));
                } else {
                    asm("v_max_f32 %0, %1, %2" : "=v"(v.x) : "v"(v.x), "v"(lo1[2 * ci]));
                    asm("v_max_f32 %0, %1, %2" : "=v"(v.y) : "v"(v.y), "v"(lo1[2 * ci + 1]));
                }
                const unsigned pk = cvt_pk(v.x, v.y);
                w[k] = okx ? pk : 0u;
                // [x~ > 0] per half word: x~ is a non-negative bf16 (ReLU producers: lo = +0 and v_max(-0, +0) = +0), so bits != 0 <=> x~ > 0
                unsigned one;
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(one) : "v"(w[k]), "v"(0x00010001u));
                asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(mk[k]) : "v"(one), "v"(0x3f803f80u));
            }
            *reinterpret_cast<uint4*>(smem + xl + xs + 16 * j) = make_uint4(w[0], w[1], w[2], w[3]);
            *reinterpret_cast<uint4*>(smem + xl + xs + C::XPLANEB + 16 * j) = make_uint4(mk[0], mk[1], mk[2], mk[3]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto compute = [&](const RsTick& t) {
        const int cp = t.q - 1;  // the pair computed: rows 2cp, 2cp + 1 from image rows 2cp - 1 .. 2cp + 2 = ring rows (2 qm3 + 3 + dy) mod 6
        unsigned rbase[4];        // (scalar)
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            const int r = 2 * t.qm3 + 3 + dy;
            rbase[dy] = ring_w + (unsigned)(r >= 6 ? r - 6 : r) * ROWB;
        }
        const int colbase = SW * t.s - 1;
        const int pix0 = (t.n * H + 2 * cp) * W + colbase;
#ifdef OCRS_RS_NOSTORE  // (floor-measurement build: every store is dropped by the range check)
        const bool comp = false;
#else
        const bool comp = t.comp != 0;
#endif
        // ================= dx~ = Weff^T (*) dz =================
        unsigned ra[4];
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) ra[dy] = rbase[dy] + lbl;
#pragma unroll
        for (int ps = 0; ps < C::NPASS; ++ps) {
            f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                constexpr int dummy = 0;
                (void)dummy;
                const int dy = kc / CPD, h = kc - dy * CPD;
                u32x4 wf;
                if constexpr (WREG) {
                    wf = wfr[kc];
                } else {
                    const uint4 q = s_wf[kc * 64 + lane];
                    wf = (u32x4){q.x, q.y, q.z, q.w};
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const uint4 q = *reinterpret_cast<const uint4*>(smem + ra[ps + dy] + (h * TPC + nt * 16) * PDB);
                    acc[nt] = mfma_bf(wf, (u32x4){q.x, q.y, q.z, q.w}, acc[nt]);
                }
            }
            // regroup through LDS: the accumulator layout (4 channels of a pixel per lane, 8-byte pieces interleaved over the lane groups) -> one
            // pixel per lane.  (8-byte stores: half-filled 16-byte granules per lane group -- the memory side then ran far below what the same
            // bytes reach as 16-byte-per-lane, line-contiguous stores: tools/probes/bw_probe2.hip vs the first form of this kernel.)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                // (compiler-visible converts: an asm statement that reads MFMA results is invisible to hipcc's hazard recogniser -- with the asm
                //  v_cvt_pk of the commit phase here, one register allocation read accumulators before the matrix core had written them: sparse NaNs)
                *reinterpret_cast<uint2*>(smem + stw + ((DUAL ? 0 : ps * 32) + nt * 16) * (CIN * 2)) =
                    make_uint2(pack2bf(acc[nt][0], acc[nt][1]), pack2bf(acc[nt][2], acc[nt][3]));
            }
        }
        uint4 c1q = make_uint4(0, 0, 0, 0);
        bool c1ok = false;
        {
            const bool ok = comp && 2 * cp + rr < H && px >= 1 && px <= SW && colbase + px < W;
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                const uint4 q = *reinterpret_cast<const uint4*>(smem + str + 16 * j);
                if constexpr (C1) {  // (dx~ of this lane's pixel stays here: see below)
                    c1q = q;
                    c1ok = ok;
                } else {
                    const bool in_a = !SPLIT || 8 * j < Ca;
                    const int off = pix0 * ((in_a ? Ca : Cb) * 2) + lxo[j];
                    __builtin_amdgcn_raw_buffer_store_b128((u32x4){q.x, q.y, q.z, q.w}, in_a ? w_a : w_b, ok ? off : -1, 0, OCRS_RS_ST_AUX);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ================= G_tap[c][o] += x~^T dz(shifted), and the same with [x~ > 0] for the producers' BatchNorm-backward sums =================
        // (warm-up ticks skip this part: it holds no vector-memory operation, so the branch does not disturb the hand-counted waits)
        if (!comp) return;
        if constexpr (C1) {
            // ---- the first block's sums from this lane's pixel (row 2 cp + rr, column colbase + px): ghat1 = dx~ [x~ > 0] (the stored bf16 dx~ -- what
            // k_c1_bwd2 read -- times the 0 / 1 mask plane of the x tile), du = sum_c (wexp A)[c] ghat1[c], then R += ghat1 u and T += du img(3 x 3)
            const uint4 mk = *reinterpret_cast<const uint4*>(smem + xt_w + (unsigned)(cp & 1) * C::XSLOTB + C::XPLANEB + rr * C::XROWB + px * XPB);
            const unsigned qw[4] = {c1q.x, c1q.y, c1q.z, c1q.w};
            const unsigned mw[4] = {mk.x, mk.y, mk.z, mk.w};
            // ring rows of image rows 2 cp - 1 .. 2 cp + 2 are (2 qm3 + 3 + dy) mod 6 (as the dz ring); this lane's rows: dy = rr .. rr + 2, u: dy = rr + 1
            unsigned cr[4];
#pragma unroll
            for (int dy = 0; dy < 4; ++dy) {
                const int r = 2 * t.qm3 + 3 + dy;
                cr[dy] = c1_w + (unsigned)((r >= 6 ? r - 6 : r) * C1ROW);
            }
            const unsigned r0 = rr ? cr[1] : cr[0], r1 = rr ? cr[2] : cr[1], r2 = rr ? cr[3] : cr[2];
            // (a pixel outside the output -- halo lane, warm-up row -- contributes nothing: its u and du are zeroed, two selects instead of four)
            const float uu = c1ok ? *reinterpret_cast<const float*>(smem + r1 + C1PLANE + (px + 1) * 4) : 0.f;
            const f32x2 uu2 = (f32x2){uu, uu};
            f32x2 du2 = (f32x2){0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x2 gp = unpk(qw[k]) * unpk(mw[k]);
                du2 = __builtin_elementwise_fma(c1k[k], gp, du2);
                c1R[k] = __builtin_elementwise_fma(gp, uu2, c1R[k]);
            }
            const float du = c1ok ? du2.x + du2.y : 0.f;
            const unsigned rrow[3] = {r0, r1, r2};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) c1T[ky * 3 + kx] = fmaf(du, *reinterpret_cast<const float*>(smem + rrow[ky] + (px + kx) * 4), c1T[ky * 3 + kx]);
        }
        const unsigned xa0 = lxa + (unsigned)(cp & 1) * C::XSLOTB;
        unsigned rw0[4], rw1[COUT == 8 ? 4 : 1];
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            rw0[dy] = rbase[dy] + lw0;
            if constexpr (COUT == 8) rw1[dy] = rbase[dy] + lw1;
        }
        // a dz ring row dy serves kernel row ky = 2 - dy of the pair's first x row and ky = 3 - dy of its second one: each B fragment is read once
        bf16x8 af[2][MTG];
#pragma unroll
        for (int rp = 0; rp < 2; ++rp)
#pragma unroll
            for (int a = 0; a < MTG; ++a) {
                const unsigned p = xa0 + rp * C::XROWB + (DUAL ? 0 : a * C::XPLANEB);
                af[rp][a] = lds_tr8(lds_at(smem, p), lds_at(smem, p + 16 * XPB));
            }
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
#pragma unroll
            for (int ut = 0; ut < UPK; ++ut) {
                unsigned p;
                if constexpr (COUT == 8) {
                    p = ut == 0 ? rw0[dy] : rw1[dy];
                } else {
                    p = rw0[dy] - ut * PDB;  // kx = ut
                }
                const bf16x8 bfr = lds_tr8(lds_at(smem, p), lds_at(smem, p + 16 * PDB));
#pragma unroll
                for (int rp = 0; rp < 2; ++rp) {
                    const int ky = rp + 2 - dy;
                    if (ky >= 0 && ky <= 2) {
#pragma unroll
                        for (int a = 0; a < MTG; ++a) accG[a][ky * UPK + ut] = mfma_bf(af[rp][a], bfr, accG[a][ky * UPK + ut]);
                    }
                }
            }
            if (dy == 1) __builtin_amdgcn_sched_barrier(0);
        }
    };

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    RsTick t0 = gen.next(), t1 = gen.next();
    issue(S0{}, t0);
    issue(S1{}, t1);
    auto tick = [&](auto ST, auto YOUNGER, RsTick& t) __attribute__((always_inline)) {
#if OCRS_RS_TICK_BARRIER
        __builtin_amdgcn_s_barrier();  // (only valid while all four waves run the same number of ticks: a measurement build for full-size launches)
#endif
        commit(ST, YOUNGER, t);
        const RsTick tn = gen.next();
        issue(ST, tn);
#ifndef OCRS_RS_NOCOMPUTE  // (floor-measurement build: loads + commit only)
        compute(t);
#endif
        t = tn;
    };
    // issue order: L0 L1 | [commit 0] L0' St | [commit 1] L1' St | ...: younger than a set's loads are the other set's loads and the stores of the
    // one (second tick) or two (steady state) computes in between
    if (t0.valid) {
        tick(S0{}, std::integral_constant<int, NLOADS>{}, t0);
        if (t1.valid) {
            tick(S1{}, std::integral_constant<int, NLOADS + NSTORE>{}, t1);
            while (t0.valid) {
                tick(S0{}, std::integral_constant<int, NLOADS + 2 * NSTORE>{}, t0);
                if (!t1.valid) break;
                tick(S1{}, std::integral_constant<int, NLOADS + 2 * NSTORE>{}, t1);
            }
        }
    }
    // the dead ticks' loads are still in flight: keep every prefetch register allocated (a "+v" tie) up to the drain -- hipcc re-uses a register
    // whose loaded value is never read the moment the asm statement ends, and the late load then overwrites the new contents (caught by
    // tools/check_rs_loads.py)
#pragma unroll
    for (int S = 0; S < 2; ++S) {
#pragma unroll
        for (int j = 0; j < NZ; ++j) {
            wait_vm<0>(pz[S][j]);
            if constexpr (!HEAD) wait_vm<0>(pg[S][j]);
            if constexpr (G2) wait_vm<0>(pg2[S][j]);
        }
        if constexpr (HEAD) wait_vm<0>(pgl[S]);
        if constexpr (XU) wait_vm<0>(pxu[S]);
        if constexpr (C1) wait_vm<0>(pim[S]);
        if constexpr (!XU) {
#pragma unroll
            for (int j = 0; j < NX; ++j) wait_vm<0>(pxr[S][j]);
        }
    }

    // ================= flush: G of the four waves -> dWpw / dWdw / producers' sums partial of this block =================
    __syncthreads();
    float* slots = reinterpret_cast<float*>(smem);  // [wave][MTG][NU][4][64]
#pragma unroll
    for (int a = 0; a < MTG; ++a)
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) slots[((wave * MTG + a) * NU + u) * 256 + r * 64 + lane] = accG[a][u][r];
    __syncthreads();
    // G (mrow, tap, o): mrow = c for the x~ rows, CIN + c for the mask rows; fixed summation order over the waves -> deterministic
    auto Gv = [&](int mrow, int tap, int o) -> float {
        const int a = mrow >> 4, m = mrow & 15;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int u = COUT == 8 ? ky * UPK + (kx >> 1) : tap, n = COUT == 8 ? ((kx & 1) * 8 + o) : o;
        const int idx = (a * NU + u) * 256 + (m & 3) * 64 + (m >> 2) * 16 + n;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < C::NW; ++w) s += slots[w * MTG * NU * 256 + idx];
        return s;
    };
    float* part = A.ws + (long)blockIdx.x * C::PART;
    for (int e = tid; e < COUT * CIN; e += C::NT) {
        const int o = e / CIN, c = e - o * CIN;
        float s = 0.f;
#pragma unroll 1  // (once per block: not worth the registers -- unrolled, the 8 -> 16 channel instantiation spills here)
        for (int tap = 0; tap < 9; ++tap) s = fmaf(A.wdw[c * 9 + tap], Gv(c, tap, o), s);
        part[e] = s;
    }
    for (int e = tid; e < 9 * CIN; e += C::NT) {
        const int c = e / 9, tap = e - c * 9;
        float s = 0.f;
#pragma unroll 1
        for (int o = 0; o < COUT; ++o) s = fmaf(A.wpw[o * A.ldw + c], Gv(c, tap, o), s);
        part[COUT * CIN + e] = s;
    }
    for (int e = tid; e < 2 * CIN; e += C::NT) {  // [CIN][S1 | S2] with the bf16 effective weights the dgrad used
        const int c = e >> 1, which = e & 1;
        float s = 0.f;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll 1
            for (int o = 0; o < COUT; ++o) {
                const float we = bf2f(f2bf(A.wdw[c * 9 + tap] * A.wpw[o * A.ldw + c]));
                s = fmaf(we, Gv(which ? c : CIN + c, tap, o), s);
            }
        part[COUT * CIN + 9 * CIN + e] = s;
        if (A.bl.raw) bwd_last_add(A.bl, CIN, c, which, s);
    }
    if constexpr (C1) {  // the first block's sums: lanes -> wave -> workgroup (fixed order) -> fp64 device atomics, 8 replicas by workgroup index (exact sums)
        __syncthreads();
        float* cred = reinterpret_cast<float*>(smem) + 64;  // [wave][17]
#pragma unroll
        for (int i = 0; i < 17; ++i) {
            const float v = wave_sum(i < 8 ? ((i & 1) ? c1R[i >> 1].y : c1R[i >> 1].x) : c1T[i - 8]);
            if (lane == 0) cred[wave * 17 + i] = v;
        }
        __syncthreads();
        if (tid < 17) {
            const float v = (cred[tid] + cred[17 + tid]) + (cred[34 + tid] + cred[51 + tid]);
            atomicAdd(A.c1acc + (blockIdx.x & 7) * 32 + tid, (double)v);
        }
    }
    if (A.bl.raw) {  // the producers' sums finalised here by the last workgroup (BwdLast in det_common.h)
        __syncthreads();  // (every read of the flush slots is done: smem[0] is free)
        bwd_last_finish(A.bl, CIN, A.tra, A.trb, tid, C::NT, reinterpret_cast<int*>(smem));
    }
}

// =============================================================================================================================================
// Row-streaming FORWARD of a DepthwiseConv block with 8 input channels (reference ocrs_models/models.py:7-28; levels 0 of the U-Net: in_conv.seq.1,
// down.0's first block, the block in front of out_conv): z = Weff (*) x~ as in k_mm_fwd (det_mm.hip), with k_rs_bwd's execution structure -- a wave
// owns a 30-column strip and marches down the image two rows per tick, its x~ rows live in a wave-private 6-row LDS ring (no workgroup barrier in
// the main loop), rows are loaded two ticks ahead through buffer descriptors with hand-counted waits, per-channel load-transform coefficients are
// scalar-register operands.  Cout = 8: the 16-row MFMA M tile carries (row parity, output channel): one MFMA set per tick yields both rows of the
// pair (K = 4 ring rows x 3 columns x 8 channels = 96: three chunks of 32 plus one of zero padding -- 8 MFMAs per 60 output pixels); Cout = 16:
// one pass per row.  The output goes through a per-wave LDS staging area so that every lane stores one pixel's channels (16-byte stores), the
// per-channel batch sums of the STORED values are per-lane register accumulators (reduced once, at the end), written as per-block partials in
// k_mm_fwd's layout and finalised by the last workgroup (FwdFin).  XU: the input is the first block's u plane (see k_rs_bwd).
// The tiled kernel needs ~8-10 ps per pixel for these shapes whatever they read (in_conv.seq.1 from the 2-byte u plane: 340 us for 18 B / pixel);
// their per-tile instruction stream, not memory, is what bounds it.
// k_rs_fwd's prefetch vectors live in AGPRs named in the asm text (see the kernel): vector IDX = a[32 + 4 IDX : 35 + 4 IDX] (hipcc allocates its own
// accumulation registers, if any, from a0 upwards; tools/check_rs_loads.py fails the build if compiler-generated code ever mentions one of these)
template <int IDX>
__device__ __forceinline__ void rsf_load16(const i32x4& rsrc, int voff);
template <int IDX, int Y>
__device__ __forceinline__ u32x4 rsf_take16();
#define RSF_VEC(IDX_, R0, R1, R2, R3)                                                                                                                 \
    template <>                                                                                                                                       \
    __device__ __forceinline__ void rsf_load16<IDX_>(const i32x4& rsrc, int voff) {                                                                   \
        asm volatile("buffer_load_dwordx4 a[" #R0 ":" #R3 "], %0, %1, 0 offen" : : "v"(voff), "s"(rsrc) : "memory", "a" #R0, "a" #R1, "a" #R2, "a" #R3); \
    }                                                                                                                                                 \
    template <int Y>                                                                                                                                  \
    __device__ __forceinline__ u32x4 rsf_take16_##IDX_() {                                                                                            \
        static_assert(Y >= 0 && Y < 64, "vmcnt is a 6-bit field");                                                                                    \
        unsigned x, y, z, w;                                                                                                                          \
        asm volatile("s_waitcnt vmcnt(%4)\n\tv_accvgpr_read_b32 %0, a" #R0 "\n\tv_accvgpr_read_b32 %1, a" #R1 "\n\tv_accvgpr_read_b32 %2, a" #R2              \
                     "\n\tv_accvgpr_read_b32 %3, a" #R3                                                                                               \
                     : "=v"(x), "=v"(y), "=v"(z), "=v"(w) : "n"(Y) : "memory");                                                                       \
        return (u32x4){x, y, z, w};                                                                                                                   \
    }
RSF_VEC(0, 32, 33, 34, 35)
RSF_VEC(1, 36, 37, 38, 39)
RSF_VEC(2, 40, 41, 42, 43)
RSF_VEC(3, 44, 45, 46, 47)
#undef RSF_VEC
template <int IDX, int Y>
__device__ __forceinline__ u32x4 rsf_take16() {
    if constexpr (IDX == 0) return rsf_take16_0<Y>();
    else if constexpr (IDX == 1) return rsf_take16_1<Y>();
    else if constexpr (IDX == 2) return rsf_take16_2<Y>();
    else return rsf_take16_3<Y>();
}
template <int S>
__device__ __forceinline__ void rsf_load2(const i32x4& rsrc, int voff) {
    if constexpr (S == 0) asm volatile("buffer_load_ushort a32, %0, %1, 0 offen" : : "v"(voff), "s"(rsrc) : "memory", "a32");
    else asm volatile("buffer_load_ushort a36, %0, %1, 0 offen" : : "v"(voff), "s"(rsrc) : "memory", "a36");
}
template <int S, int Y>
__device__ __forceinline__ unsigned rsf_take2() {
    unsigned x;
    if constexpr (S == 0) asm volatile("s_waitcnt vmcnt(%1)\n\tv_accvgpr_read_b32 %0, a32" : "=v"(x) : "n"(Y) : "memory");
    else asm volatile("s_waitcnt vmcnt(%1)\n\tv_accvgpr_read_b32 %0, a36" : "=v"(x) : "n"(Y) : "memory");
    return x;
}
// a 16-byte store the hardware range check drops (offset 0xffffffff): keeps the prologue's operation count equal to a tick's
__device__ __forceinline__ void rsf_dropped_store(const i32x4& rsrc) {
    const int off = -1;
    const u32x4 zero = (u32x4){0u, 0u, 0u, 0u};
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" : : "v"(zero), "v"(off), "s"(rsrc) : "memory");
}
struct RsfArgs {
    const bf16 *xa, *xb, *xu;  // xa | xb: the (concatenated) input, Ca | Cb channels; xu: the u plane instead (Cin = 8)
    int Ca, Cb;
    const float *wexp, *tra, *trb, *wdw, *wpw;
    bf16* z;
    float* ws;
    int N, H, W, NS, NP, NB, PB, njobs;
    FwdFin fin;
};
#ifndef OCRS_RSF_WPS
#define OCRS_RSF_WPS 3  // resident 4-wave workgroups per CU (per launch at level 0, XU | plain: 3: 240 | 226 us, 4: 255 | 254 us; the tiled kernel: 340 | 279 us)
#endif
template <int CIN, int COUT>
struct RsfCfg {
    static_assert(CIN == 8 || (CIN == 16 && COUT == 8), "shapes of level 0: 8 -> 8 (| 16), 16 -> 8");
    static constexpr int NW = 4, NT = 256, SW = 30;
    static constexpr int PXB = CIN * 2, ROWB = 34 * PXB, RROWS = 6, RINGB = RROWS * ROWB + 64;
    static constexpr bool DUAL = COUT == 8;
    static constexpr int G8 = CIN / 8;                  // 8-channel groups per input pixel
    static constexpr int TPC = 4 / G8;                  // column slots per 32-deep K chunk (a chunk lies inside ONE ring row)
    static constexpr int CPD = (3 + TPC - 1) / TPC;     // chunks per ring row (slots past kx = 2 carry zero weights)
    static constexpr int NPASS = DUAL ? 1 : 2, NDY = DUAL ? 4 : 3, KC = NDY * CPD;
    static constexpr int NX = CIN / 8, NZ = COUT / 8;   // 16-byte input / output items per lane and tick
    static constexpr int STB = 2 * 32 * COUT * 2;       // staging of a row pair's output
    static constexpr int OFF_RING = 64, OFF_ST = OFF_RING + NW * RINGB, OFF_WF = OFF_ST + NW * STB, OFF_PAR = OFF_WF + KC * 1024;
    static constexpr int PAR_FLOATS = 3 * CIN + 9 * CIN + COUT * CIN + NW * 2 * COUT;  // trx | w9 | wp | per-wave stat slots
    static constexpr int SMEM = OFF_PAR + PAR_FLOATS * 4;
    static_assert(64 + (8 * 32 + 32) * 8 <= OFF_ST, "the finalisation's reduction area fits the dead rings");
};

template <int CIN, int COUT, bool XU>
__global__ __launch_bounds__(256, OCRS_RSF_WPS) void k_rs_fwd(RsfArgs A) {
    static_assert(!XU || CIN == 8, "u plane: the 8-channel output of the first block");
    using C = RsfCfg<CIN, COUT>;
    constexpr int PXB = C::PXB, ROWB = C::ROWB, RINGB = C::RINGB, KC = C::KC, NX = C::NX, NZ = C::NZ, SW = C::SW, G8 = C::G8, TPC = C::TPC, CPD = C::CPD;
    constexpr bool DUAL = C::DUAL;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    uint4* s_wf = reinterpret_cast<uint4*>(smem + C::OFF_WF);
    float* s_trx = reinterpret_cast<float*>(smem + C::OFF_PAR);  // [3][CIN] scale | shift | lo
    float* s_w9 = s_trx + 3 * CIN;                               // [CIN][9]
    float* s_wp = s_w9 + 9 * CIN;                                // [COUT][CIN]
    float* s_st = s_wp + COUT * CIN;                             // [wave][COUT][2]
    const int H = A.H, W = A.W, Ca = A.Ca, Cb = A.Cb;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int i = tid; i < 3 * CIN; i += C::NT) {
        const int r = i / CIN, c = i - r * CIN;
        s_trx[i] = c < Ca ? A.tra[r * Ca + c] : A.trb[r * Cb + (c - Ca)];
    }
    for (int i = tid; i < 9 * CIN; i += C::NT) s_w9[i] = A.wdw[i];
    for (int i = tid; i < COUT * CIN; i += C::NT) s_wp[i] = A.wpw[i];
    for (int i = tid; i < C::OFF_WF / 16; i += C::NT) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);  // (the rings' pad positions stay zero)
    __syncthreads();
    // A[m][k]: chunk kc = (ring row dy, part h); K slot (lane group kgl, j) = (column offset kx = h * TPC + kgl / G8, input channel (kgl % G8) * 8 + j); the
    // input pixel (dy, kx) is conv tap (ky, kx) = (dy - rp, kx) of output row rp.  DUAL: m = (rp, o); otherwise m = o and the fragment is that of
    // rp = 0 (pass ps reads rows ps + dy)
    for (int f = tid; f < KC * 64; f += C::NT) {
        const int l = f & 63, kc = f >> 6, dy = kc / CPD, h = kc - dy * CPD, m = l & 15, kgl = l >> 4;
        const int kx = h * TPC + kgl / G8, c0 = (kgl % G8) * 8;
        const int rp = DUAL ? (m >> 3) : 0, o = DUAL ? (m & 7) : m, ky = dy - rp;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (kx <= 2 && ky >= 0 && ky <= 2) ? s_w9[(c0 + j) * 9 + ky * 3 + kx] * s_wp[o * CIN + c0 + j] : 0.f;
        s_wf[f] = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
    }
    __syncthreads();
    f32x2 sc2[CIN / 2], sh2[CIN / 2];
    float lo1[CIN], we[XU ? 8 : 1];
#pragma unroll
    for (int i = 0; i < CIN / 2; ++i) {
        sc2[i] = (f32x2){usc(s_trx[2 * i]), usc(s_trx[2 * i + 1])};
        sh2[i] = vreg((f32x2){s_trx[CIN + 2 * i], s_trx[CIN + 2 * i + 1]});
    }
#pragma unroll
    for (int i = 0; i < CIN; ++i) lo1[i] = usc(s_trx[2 * CIN + i]);
    if constexpr (XU) {
#pragma unroll
        for (int i = 0; i < 8; ++i) we[i] = usc(A.wexp[i]);
    }
    u32x4 wfr[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        const uint4 q = s_wf[kc * 64 + lane];
        wfr[kc] = (u32x4){q.x, q.y, q.z, q.w};
    }
    const unsigned npix = (unsigned)A.N * (unsigned)H * (unsigned)W;
    const i32x4 r_xa = XU ? make_rsrc(A.xu, npix * 2) : make_rsrc(A.xa, npix * Ca * 2);
    const i32x4 r_xb = make_rsrc(Cb ? A.xb : A.xa, npix * (Cb ? Cb : Ca) * 2);
    const __amdgpu_buffer_rsrc_t w_z = __builtin_amdgcn_make_buffer_rsrc((void*)A.z, 0, npix * COUT * 2, 0x00020000);
    const i32x4 w_zi = make_rsrc(A.z, npix * COUT * 2);

    const int rr = lane >> 5, px = lane & 31;  // commit / store: this lane's (row of the pair, staged pixel)
    const unsigned ring_w = C::OFF_RING + wave * RINGB;
    const unsigned wl = ring_w + rr * ROWB + (px + 1) * PXB;
    const unsigned lbl = (l15 + kg / G8) * PXB + (kg % G8) * 16;  // B fragments: output pixel l15 (+ 16 per N tile) reads ring position l15 + kx
    const int st_rp = DUAL ? (kg >> 1) : 0, st_ch = DUAL ? (kg & 1) * 4 : kg * 4;
    const unsigned stw = C::OFF_ST + wave * C::STB + (st_rp * 32 + l15) * (COUT * 2) + st_ch * 2;
    const unsigned str = C::OFF_ST + wave * C::STB + (rr * 32 + px) * (COUT * 2);

    f32x2 s1[COUT / 2], s2[COUT / 2];  // batch sums of this lane's stored values (channel pairs)
#pragma unroll
    for (int i = 0; i < COUT / 2; ++i) s1[i] = s2[i] = (f32x2){0.f, 0.f};

    const int nblk = gridDim.x;
    const int vblk = (nblk & 7) == 0 ? (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    RsGen gen(vblk * C::NW + wave, A.njobs, nblk * C::NW, A.NS, A.NB, A.PB, A.NP);

    auto corner = [&](const RsTick& t) -> int {
        const int qc = t.q < 0 ? 0 : (t.q >= A.NP ? A.NP - 1 : t.q);
        return (t.n * H + 2 * qc) * W + SW * t.s - 1;
    };
    auto issue = [&](auto ST, const RsTick& t) {
        constexpr int S = decltype(ST)::value;
        const int cp = corner(t);
        if constexpr (XU) {
            rsf_load2<S>(r_xa, (cp + rr * W + px) * 2);
        } else {
            const int pi = cp + rr * W + px;
            // item j = channels 8j .. 8j + 7 of the concatenated input: from xa when 8j < Ca, else from xb
            if constexpr (NX == 1) {
                rsf_load16<S>(r_xa, pi * (Ca * 2));
            } else {
                rsf_load16<S * 2>(r_xa, pi * (Ca * 2));
                if (Ca > 8) rsf_load16<S * 2 + 1>(r_xa, pi * (Ca * 2) + 16);  // (wave-uniform branch: both sides issue ONE load)
                else rsf_load16<S * 2 + 1>(r_xb, pi * (Cb * 2));
            }
        }
    };
    constexpr int NLOADS = NX, NSTORE = NZ;
    auto commit = [&](auto ST, auto YOUNGER, const RsTick& t) {
        constexpr int S = decltype(ST)::value;
        constexpr int Y = decltype(YOUNGER)::value;
        const int col = SW * t.s - 1 + px, row = 2 * t.q + rr;
        const bool ok = (unsigned)col < (unsigned)W && (unsigned)row < (unsigned)H;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            u32x4 raw = (u32x4){0u, 0u, 0u, 0u};
            // (the set's loads were issued together and retire in order: the wait for its last one, written on the first take, covers them all)
            if constexpr (XU) raw.x = rsf_take2<S, Y>();
            else if (j == 0) raw = rsf_take16<S * NX, Y>();
            else raw = rsf_take16<S * NX + (NX - 1), Y>();
            unsigned w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ci = 4 * j + k;
                f32x2 xin;
                if constexpr (XU) {
                    const float uv = __uint_as_float(raw.x << 16);
                    xin = unpk(cvt_pk(uv * we[2 * k], uv * we[2 * k + 1]));
                } else {
                    xin = unpk(raw[k]);
                }
                f32x2 v = __builtin_elementwise_fma(xin, sc2[ci], sh2[ci]);
                asm("v_max_f32 %0, %1, %2" : "=v"(v.x) : "v"(v.x), "s"(lo1[2 * ci]));
                asm("v_max_f32 %0, %1, %2" : "=v"(v.y) : "v"(v.y), "s"(lo1[2 * ci + 1]));
                const unsigned pk = cvt_pk(v.x, v.y);
                w[k] = ok ? pk : 0u;  // (zeros outside the image: the convolution's padding)
            }
            *reinterpret_cast<uint4*>(smem + wl + (unsigned)(2 * t.qm3) * ROWB + 16 * j) = make_uint4(w[0], w[1], w[2], w[3]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto compute = [&](const RsTick& t) {
        const int cp = t.q - 1;  // the pair computed: output rows 2cp, 2cp + 1 from input rows 2cp - 1 .. 2cp + 2 = ring rows (2 qm3 + 3 + dy) mod 6
        unsigned ra[4];
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            const int r = 2 * t.qm3 + 3 + dy;
            ra[dy] = ring_w + (unsigned)(r >= 6 ? r - 6 : r) * ROWB + lbl;
        }
        const int colbase = SW * t.s - 1;
        const int pix0 = (t.n * H + 2 * cp) * W + colbase;
        const bool comp = t.comp != 0;
#pragma unroll
        for (int ps = 0; ps < C::NPASS; ++ps) {
            f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                const int dy = kc / CPD, h = kc - dy * CPD;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const uint4 q = *reinterpret_cast<const uint4*>(smem + ra[ps + dy] + (h * TPC + nt * 16) * PXB);
                    acc[nt] = mfma_bf(wfr[kc], (u32x4){q.x, q.y, q.z, q.w}, acc[nt]);
                }
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)  // (compiler-visible converts: see k_rs_bwd)
                *reinterpret_cast<uint2*>(smem + stw + ((DUAL ? 0 : ps * 32) + nt * 16) * (COUT * 2)) =
                    make_uint2(pack2bf(acc[nt][0], acc[nt][1]), pack2bf(acc[nt][2], acc[nt][3]));
        }
        const bool ok = comp && 2 * cp + rr < H && px >= 1 && px <= SW && colbase + px < W;
#pragma unroll
        for (int j = 0; j < NZ; ++j) {
            const uint4 q = *reinterpret_cast<const uint4*>(smem + str + 16 * j);
            const int off = (pix0 + rr * W + px) * (COUT * 2) + 16 * j;
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){q.x, q.y, q.z, q.w}, w_z, ok ? off : -1, 0, 0);
            const unsigned qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f32x2 v = unpk(qq[k]);
                if (!ok) v = (f32x2){0.f, 0.f};
                s1[4 * j + k] += v;
                s2[4 * j + k] = __builtin_elementwise_fma(v, v, s2[4 * j + k]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    RsTick t0 = gen.next(), t1 = gen.next();
    // Prefetch registers.  The row loads of a tick are issued two ticks ahead and must not be waited for early, so they are asm statements hipcc does
    // not see through, with hand-counted waits (vector memory operations retire in order) -- and their destinations are ACCUMULATION registers named
    // in the asm text (set 0: a[32:35], set 1: a[36:39]): hipcc allocates no value of its own there (the kernel needs ~100 of its 168 registers; the
    // build checks that no AGPR access exists outside these statements), so no live-range split or copy can ever touch a register with a load in
    // flight.  (With VGPR destinations tied to C++ variables -- k_rs_bwd's form -- hipcc kept a set in different registers in the prologue and in the
    // loop and joined them with copies issued while the loads were in flight; tools/check_rs_loads.py caught it.  Compiler-visible loads got
    // vmcnt(1) instead of vmcnt(3): every tick then waited for the load it had just issued.)
    // ONE loop without peeled first ticks: the prologue issues the steady state's operation sequence L0 St L1 St with stores the range check drops.
    auto dummy_stores = [&]() {
#pragma unroll
        for (int j = 0; j < NSTORE; ++j) rsf_dropped_store(w_zi);
    };
    issue(S0{}, t0);
    dummy_stores();
    issue(S1{}, t1);
    dummy_stores();
    auto tick = [&](auto ST, RsTick& t) __attribute__((always_inline)) {
        commit(ST, std::integral_constant<int, NLOADS + 2 * NSTORE>{}, t);
        const RsTick tn = gen.next();
        issue(ST, tn);
        compute(t);
        t = tn;
    };
    while (t0.valid) {
        tick(S0{}, t0);
        if (!t1.valid) break;
        tick(S1{}, t1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the dead ticks' loads)

    // ---- batch sums: lanes -> wave -> workgroup partial [COUT][sum | sum of squares] (k_mm_fwd's layout), then the last workgroup finalises
#pragma unroll
    for (int i = 0; i < COUT / 2; ++i) {
        const float a0 = wave_sum(s1[i].x), a1 = wave_sum(s1[i].y), b0 = wave_sum(s2[i].x), b1 = wave_sum(s2[i].y);
        if (lane == 0) {
            s_st[(wave * COUT + 2 * i) * 2 + 0] = a0;
            s_st[(wave * COUT + 2 * i) * 2 + 1] = b0;
            s_st[(wave * COUT + 2 * i + 1) * 2 + 0] = a1;
            s_st[(wave * COUT + 2 * i + 1) * 2 + 1] = b1;
        }
    }
    __syncthreads();
    const FwdFin& fin = A.fin;
    for (int e = tid; e < 2 * COUT; e += C::NT) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < C::NW; ++w) s += s_st[w * COUT * 2 + e];
        if (fin.counter) __hip_atomic_store(A.ws + (long)blockIdx.x * (2 * COUT) + e, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else A.ws[(long)blockIdx.x * (2 * COUT) + e] = s;
    }
    if (!fin.counter) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* s_flag = reinterpret_cast<int*>(smem);
    double* red = reinterpret_cast<double*>(smem + 64);  // [8 chains][32 columns] + [32] (the rings are dead)
    if (tid == 0) *s_flag = __hip_atomic_fetch_add(fin.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (*s_flag == 0) return;
    static_assert(2 * COUT <= 32 && C::NT == 256, "one 32-element window x 8 chains x 32 columns = the 256 threads");
    const int col = tid & 31, chain = tid >> 5;
    red[chain * 32 + col] = rs_parts_chain_sum(A.ws, gridDim.x, COUT, col, chain);
    __syncthreads();
    if (chain == 0) {
        const double tot = ((red[col] + red[32 + col]) + (red[64 + col] + red[96 + col])) + ((red[128 + col] + red[160 + col]) + (red[192 + col] + red[224 + col]));
        red[256 + col] = tot;
    }
    __syncthreads();
    if (tid == 0) {
        if (fin.nbt) *fin.nbt += 1;
        __hip_atomic_store(fin.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (chain == 0 && col < 2 * COUT && !(col & 1))
        bn_finalize_channel(red[256 + col], red[256 + col + 1], fin.count, col >> 1, COUT, fin.gamma, fin.beta, fin.eps, fin.momentum, fin.tr, fin.saved, fin.run_mean,
                            fin.run_var, fin.lo);
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
static int rs_env() {
    static const int v = env_int("OCRS_RS", 1);
    return v;
}
bool rs_bwd_supported(int Ca, int Cb, int Cout, int pooled, int N, int H, int W) {
    if (!rs_env() || pooled) return false;
    const int Cin = Ca + Cb;
    // (the kernel is written for Cin / Cout in {8, 16}, but only 8 -> 8 is instantiated: with 16 channels on either side the per-channel coefficients
    //  no longer fit the scalar register file next to the descriptors -- hipcc spills SGPRs to scratch inside the tick loop, 256 VGPRs + 40-88 B of
    //  scratch -- and a scratch reload in that loop is a vmcnt(0); those shapes need LDS-resident coefficients first)
#ifdef OCRS_RS_16
    if (!((Cin == 8 || Cin == 16) && (Cout == 8 || Cout == 16))) return false;
#else
    if (!(Cin == 8 && (Cout == 8 || (OCRS_RS_8_16 && Cout == 16)))) return false;
#endif
    if (Cb != 0 && !(Ca == 8 && Cb == 8)) return false;
    const long bytes = (long)N * H * W * (Cin > Cout ? Cin : Cout) * 2;
    return bytes < (1L << 31) && H >= 2 && W >= 2;
}
static void rs_geometry(int N, int H, int W, int wps, int& NS, int& NP, int& NB, int& PB, int& njobs, int& nblocks) {
    NS = (W + 29) / 30;
    NP = (H + 1) / 2;
    static const int rb_env = env_int("OCRS_RS_RB", 64);          // rows per job (two warm-up ticks per job: 64 rows = 6 %)
    static const int blocks_env = env_int("OCRS_RS_BLOCKS", 0);  // (tests: a small grid makes every wave run several jobs)
    PB = rb_env / 2 > 0 ? rb_env / 2 : 1;
    NB = (NP + PB - 1) / PB;
    njobs = N * NB * NS;
    const int cap = blocks_env > 0 ? blocks_env : kNumCU * wps;  // resident blocks (4 waves each) at wps waves per SIMD
    const int need = (njobs + 3) / 4;
    nblocks = need < cap ? need : cap;
    if (nblocks >= 8) nblocks &= ~7;
}
int rs_bwd_blocks(int Cin, int Cout, int N, int H, int W, int g2) {
    int NS, NP, NB, PB, njobs, nb;
    rs_geometry(N, H, W, rs_wps_rt(Cin, Cout, g2), NS, NP, NB, PB, njobs, nb);
    return nb;
}
void rs_bwd_launch(const Src2<bf16>& x, const float* tra, const float* trb, const float* wdw, const float* wpw, int ldw, const bf16* g1, const bf16* g2,
                   const bf16* z, const float* bn, const float* coef, bf16* gxa, bf16* gxb, float* ws, bool stats, int Cout, int N, int H, int W,
                   const BnFin& fin, hipStream_t st, const float* gl, const float* whead, const bf16* xu, const float* wexp, const BwdLast& bl, const float* img,
                   double* c1acc) {
    RsArgs a;
    a.bl = bl;
    a.img = img;
    a.c1acc = c1acc;
    a.gl = gl;
    a.whead = whead;
    a.xu = xu;
    a.wexp = wexp;
    a.xa = x.a; a.xb = x.b; a.Ca = x.Ca; a.Cb = x.Cb;
    a.tra = tra; a.trb = trb; a.wdw = wdw; a.wpw = wpw; a.ldw = ldw;
    a.g1 = g1; a.g2 = g2; a.z = z; a.bn = bn; a.coef = coef; a.gxa = gxa; a.gxb = gxb; a.ws = ws;
    a.N = N; a.H = H; a.W = W;
    int nb;
    rs_geometry(N, H, W, rs_wps_rt(x.Ca + x.Cb, Cout, g2 != nullptr), a.NS, a.NP, a.NB, a.PB, a.njobs, nb);
    a.fin = fin;
    (void)stats;
    const int Cin = x.Ca + x.Cb;
#define RS_LAUNCH(CI_, CO_, G2_, SP_)                                                                                                        \
    {                                                                                                                                        \
        using CC = RsCfg<CI_, CO_>;                                                                                                          \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rs_bwd<CI_, CO_, G2_, SP_>), hipFuncAttributeMaxDynamicSharedMemorySize, CC::SMEM); \
        OCRS_LAUNCH_T((k_rs_bwd<CI_, CO_, G2_, SP_>), dim3(nb), dim3(CC::NT), CC::SMEM, st, a);                                             \
    }
    if (xu) {  // the block behind the first block (8 -> 8 channels, input rebuilt from the u plane; one or two gradients)
        using CC = RsCfg<8, 8>;
        if (img && c1acc) {  // C1: also the first block's weight-gradient sums, no dx~ store (+ the image / u rings behind the regular LDS layout)
            constexpr int SM = CC::SMEM + CC::NW * (2 * 6 * 34 * 4 + 32);
            if (g2) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rs_bwd<8, 8, true, false, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SM);
                OCRS_LAUNCH_T((k_rs_bwd<8, 8, true, false, false, true, true>), dim3(nb), dim3(CC::NT), SM, st, a);
            } else {
                hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rs_bwd<8, 8, false, false, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SM);
                OCRS_LAUNCH_T((k_rs_bwd<8, 8, false, false, false, true, true>), dim3(nb), dim3(CC::NT), SM, st, a);
            }
            return;
        }
        if (g2) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rs_bwd<8, 8, true, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, CC::SMEM);
            OCRS_LAUNCH_T((k_rs_bwd<8, 8, true, false, false, true>), dim3(nb), dim3(CC::NT), CC::SMEM, st, a);
        } else {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rs_bwd<8, 8, false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, CC::SMEM);
            OCRS_LAUNCH_T((k_rs_bwd<8, 8, false, false, false, true>), dim3(nb), dim3(CC::NT), CC::SMEM, st, a);
        }
        return;
    }
    if (gl) {  // the block in front of out_conv (8 -> 8 channels, one source, one gradient)
        using CC = RsCfg<8, 8>;
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rs_bwd<8, 8, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, CC::SMEM);
        OCRS_LAUNCH_T((k_rs_bwd<8, 8, false, false, true>), dim3(nb), dim3(CC::NT), CC::SMEM, st, a);
        return;
    }
    const bool sp = x.Cb > 0;
#define RS_CASE(CI_, CO_)                                                                 \
    if (Cin == CI_ && Cout == CO_) {                                                      \
        if (g2) {                                                                         \
            if (sp) RS_LAUNCH(CI_, CO_, true, (CI_ > 8)) else RS_LAUNCH(CI_, CO_, true, false)   \
        } else {                                                                          \
            if (sp) RS_LAUNCH(CI_, CO_, false, (CI_ > 8)) else RS_LAUNCH(CI_, CO_, false, false) \
        }                                                                                 \
    }
    RS_CASE(8, 8)
#if OCRS_RS_8_16
    RS_CASE(8, 16)
#endif
#ifdef OCRS_RS_16
    RS_CASE(16, 8) RS_CASE(16, 16)
#endif
#undef RS_CASE
#undef RS_LAUNCH
}

// ---- row-streaming forward (k_rs_fwd): Cin = 8 (one source or the u plane) or 16 (one source or the 8 | 8 concat), Cout = 8, no fused pooling
bool rs_fwd_supported(int Ca, int Cb, int Cout, int N, int H, int W) {
    static const int on = env_int("OCRS_RSF", 1);
    static const int on16 = env_int("OCRS_RSF_C16", 1);  // the 16 -> 8 channel block of level 0
    // (Cout = 16 -- one pass per row, 12 MFMAs per tick; the kernel is written for it, -DOCRS_RSF_16 instantiates it at OCRS_RSF_WPS <= 3 -- measured
    //  432 us against k_mm_fwd's 346 us at level 0: not built by default)
    const int Cin = Ca + Cb;
    bool shape = (Ca == 8 && Cb == 0 && Cout == 8) || (on16 && Cin == 16 && (Cb == 0 || Ca == 8) && Cout == 8);
#ifdef OCRS_RSF_16
    shape = shape || (Ca == 8 && Cb == 0 && Cout == 16);
#endif
    if (!on || !shape) return false;
    return (long)N * H * W * (Cin > Cout ? Cin : Cout) * 2 < (1L << 31) && H >= 2 && W >= 2;
}
int rs_fwd_blocks(int N, int H, int W) {
    int NS, NP, NB, PB, njobs, nb;
    rs_geometry(N, H, W, OCRS_RSF_WPS, NS, NP, NB, PB, njobs, nb);
    return nb;
}
void rs_fwd_launch(const Src2<bf16>& x, const bf16* xu, const float* wexp, const float* tra, const float* trb, const float* wdw, const float* wpw, bf16* z, float* ws,
                   int Cout, int N, int H, int W, int nb, const FwdFin& fin, hipStream_t st) {
    RsfArgs a;
    a.xa = x.a; a.xb = x.b; a.Ca = x.Ca; a.Cb = x.Cb; a.xu = xu; a.wexp = wexp; a.tra = tra; a.trb = trb; a.wdw = wdw; a.wpw = wpw; a.z = z; a.ws = ws;
    a.N = N; a.H = H; a.W = W;
    int nb_geo;
    rs_geometry(N, H, W, OCRS_RSF_WPS, a.NS, a.NP, a.NB, a.PB, a.njobs, nb_geo);
    (void)nb_geo;  // (the caller's nb = rs_fwd_blocks(): the number of partials it allocated)
    a.fin = fin;
#define RSF_LAUNCH(CI_, CO_, XU_)                                                                                                              \
    {                                                                                                                                          \
        using CC = RsfCfg<CI_, CO_>;                                                                                                           \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rs_fwd<CI_, CO_, XU_>), hipFuncAttributeMaxDynamicSharedMemorySize, CC::SMEM);    \
        OCRS_LAUNCH_T((k_rs_fwd<CI_, CO_, XU_>), dim3(nb), dim3(CC::NT), CC::SMEM, st, a);                                                    \
    }
    const int Cin = x.Ca + x.Cb;
    if (Cin == 16) RSF_LAUNCH(16, 8, false)
    else if (Cout == 8) {
        if (xu) RSF_LAUNCH(8, 8, true) else RSF_LAUNCH(8, 8, false)
    }
#ifdef OCRS_RSF_16
    else {
        if (xu) RSF_LAUNCH(8, 16, true) else RSF_LAUNCH(8, 16, false)
    }
#endif
#undef RSF_LAUNCH
}
