// fp32 (parity-mode) DepthwiseConv block kernels of the wide levels as register-resident ROW-STREAMING waves (round 6).
//
// The reference trains detection in fp32 (ocrs_models/train_detection.py:92-97, no autocast); fp32 is the only mode that meets north_star's
// 1e-4.  Until round 5 that mode ran the round-1 tile kernels (k_dwpw_fwd / k_pw_bwd / k_dw_bwd, csrc/det_fwd.hip / det_bwd.hip): LDS-bandwidth
// bound forward (1.6-2.0 TB/s on the fp32 bytes), a `du` round trip through HBM between the two backward kernels.  These kernels are the fp32
// counterpart of csrc/det_rs.hip, but simpler: an fp32 pixel is 32-64 B, so ONE register layout serves every tensor and no LDS tile exists at all.
//
//   Layout "pixel x channel quad":  lane = (n, q), n = lane & 15 = pixel column of a 16-column strip, q = lane >> 4 = channel quad;
//   a register set (4 VGPRs r = 0..3) holds channels 16 s + 4 q + r of that pixel.
//     * it is what a 16-byte NHWC access gives (dwordx4 per lane, 1 KB contiguous per wave instruction for 16 channels),
//     * it IS the D layout of v_mfma_f32_16x16x4_f32 (D[m = 4 q + r][n]) -- the pointwise output needs no re-arrangement -- and
//     * it is a valid B operand of the same MFMA if K step r is made to contract over the channels {4 k + r}: the A fragments are packed that
//       way once per wave (A_r[lane (m, k)] = W[m][4 k + r]), K steps are summed anyway.  Exact fp32 products, fp32 accumulation.
//   The depthwise 3x3 is per channel: left / right neighbours are the adjacent lanes of the 16-lane DPP row (row_shr:1 / row_shl:1, folded into the
//   VALU op), upper / lower neighbours are the previous rows, kept as three partial sums per channel.  A strip is 16 lanes = 14 output columns
//   + one halo column each side (their outputs are discarded).  Rows are loaded through buffer descriptors (hardware range check: out-of-range
//   lanes read 0 and their stores are dropped -- no divergent branch around any memory instruction, so hipcc's vmcnt counts stay exact), P rows
//   ahead into a register ring; there is no LDS traffic and no barrier in the loop.  BatchNorm batch statistics: per-lane fp32 partials ->
//   DPP row sums -> one fp64 atomic per (workgroup, channel) (an fp64 sum of fp32 values is exact, hence order-independent: bit-reproducible),
//   finalised by the launch's last workgroup (bn_finalize_last_block).  Fused MaxPool2d(2) in its pre-BatchNorm form as in k_dwpw_fwd<.., POOL>.
#include "det_common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
using rsrc_t = __amdgpu_buffer_rsrc_t;

constexpr int DPP_SHR1 = 0x111;  // row_shr:1  result[i] = src[i - 1] within the 16-lane row (lane 0: 0)
constexpr int DPP_SHL1 = 0x101;  // row_shl:1  result[i] = src[i + 1]                         (lane 15: 0)

__device__ __forceinline__ rsrc_t mk_rsrc(const void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ f32x4 bld16(rsrc_t r, int off) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0)); }
__device__ __forceinline__ float bld4(rsrc_t r, int off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0)); }
__device__ __forceinline__ void bst16(rsrc_t r, int off, const f32x4& v) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, 0); }
__device__ __forceinline__ f32x4 or4(const f32x4& a, const f32x4& b) {  // lanes get their value from exactly one of two range-checked loads (the other returned 0)
    return __builtin_bit_cast(f32x4, __builtin_bit_cast(u32x4, a) | __builtin_bit_cast(u32x4, b));
}

constexpr int RS32_COLS = 14;  // output columns of a strip (16 lanes - 2 halo lanes)

struct Rs32Jobs {
    int ncg, nrb, rb, njobs;  // column groups per row, row blocks per image, rows per block, jobs = N * nrb * ncg (column group fastest)
};
// rb = the rows per job wanted (64 at full size: 3 % halo rows).  A wave walks its job row by row (~1 us per row), so a SMALL tensor cut into 64-row jobs
// leaves most wave slots empty and every kernel takes >= 66 row times (the config-1-sized fp32 step, 2 x 512^2, went 3.9 -> 5.3 ms on the first form):
// halve the rows per job (even values down to 8) until there are two jobs per resident wave slot
static inline Rs32Jobs rs32_jobs(int N, int H, int W, int cols, int rb) {
    Rs32Jobs j;
    j.ncg = (W + cols - 1) / cols;
    const long want = 2L * kNumCU * 3 * 4;
    while (rb > 8 && (long)N * ((H + rb - 1) / rb) * j.ncg < want) rb = (rb / 2 + 1) & ~1;
    j.rb = rb;
    j.nrb = (H + rb - 1) / rb;
    j.njobs = N * j.nrb * j.ncg;
    return j;
}

struct Rs32F {
    const float *xa, *xb, *tra, *trb, *wdw, *wpw, *gamma;
    float *z, *pooled;
    double* gstat;
    int Ca, Cb, Cout, N, H, W;
    Rs32Jobs jb;
    FwdFin fin;
};

// XCD-aware static job schedule of a persistent grid (workgroup b runs on XCD b % 8): every XCD owns a contiguous eighth of the jobs and the four
// waves of a workgroup take neighbouring jobs (neighbouring strips share their halo columns' cache lines in that XCD's L2)
struct Rs32Sched {
    int first, step, end;
    __device__ __forceinline__ Rs32Sched(int njobs, int wave) {
        const int nb = gridDim.x;
        if ((nb & 7) == 0) {
            const int per = (njobs + 7) >> 3, xcd = blockIdx.x & 7;
            first = xcd * per + (blockIdx.x >> 3) * 4 + wave;
            step = (nb >> 3) * 4;
            end = (xcd + 1) * per < njobs ? (xcd + 1) * per : njobs;
        } else {
            first = blockIdx.x * 4 + wave;
            step = nb * 4;
            end = njobs;
        }
    }
};

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------------------------
// forward: x~ = max(x * scale + shift, lo) on load (producer's BatchNorm + ReLU; channel concat [a | b]) -> depthwise 3x3 (zero padding)
// -> pointwise 1x1 on the fp32 matrix cores -> z store + sum z, sum z^2 (+ fused 2x2 max-pool of the pre-BatchNorm z).
// NSET: register sets of 16 input channels (Cin <= 16 NSET); MT: 16-row M tiles of output channels; SPLIT: two sources; P: rows in flight.
// ---------------------------------------------------------------------------------------------------------------------------------------------
template <int NSET, int MT, bool SPLIT, bool POOL, int P>
__global__ __launch_bounds__(256, (NSET * MT == 1) ? 3 : 2) void k_rs32_fwd(const Rs32F A) {
    __shared__ float s_stat[4][2][16 * MT];
    __shared__ int s_flag;
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Ca = A.Ca, Cb = A.Cb, Cin = Ca + Cb, Cout = A.Cout, H = A.H, W = A.W;

    // ---- per-lane constants: the lane's channel quad of every set
    float sc[NSET][4], sh[NSET][4], lo[NSET][4], wd[NSET][4][9], af[MT][NSET][4];
    int qa[NSET], qb[NSET];  // byte offset of the quad inside a pixel of source a / b, or -1: the quad does not come from that source
    bool chok[NSET];
#pragma unroll
    for (int s = 0; s < NSET; ++s) {
        const int cb = 16 * s + 4 * q;
        chok[s] = cb < Cin;
        const bool in_a = cb < Ca;
        qa[s] = (chok[s] && in_a) ? cb * 4 : -1;
        qb[s] = (chok[s] && !in_a) ? (cb - Ca) * 4 : -1;
        const float* tr = in_a ? A.tra : A.trb;
        const int Cs = in_a ? Ca : Cb, cc = in_a ? cb : cb - Ca;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[s][r] = chok[s] ? tr[cc + r] : 0.f;
            sh[s][r] = chok[s] ? tr[Cs + cc + r] : 0.f;
            lo[s][r] = chok[s] ? tr[2 * Cs + cc + r] : 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) wd[s][r][t] = chok[s] ? A.wdw[(cb + r) * 9 + t] : 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt][s][r] = (chok[s] && n + 16 * mt < Cout) ? A.wpw[(n + 16 * mt) * Cin + cb + r] : 0.f;  // A_r[(m, k)] = W[m][4 k + r]
        }
    }
    float sg[MT][4];  // sign of the BatchNorm weight (POOL: the window's selected element is max z or min z)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sg[mt][r] = (POOL && 16 * mt + 4 * q + r < Cout && A.gamma[16 * mt + 4 * q + r] < 0.f) ? -1.f : 1.f;

    const unsigned npix = (unsigned)A.N * H * W;
    const rsrc_t ra = mk_rsrc(A.xa, npix * Ca * 4), rbs = mk_rsrc(SPLIT ? A.xb : A.xa, npix * (SPLIT ? Cb : Ca) * 4);
    const rsrc_t rz = mk_rsrc(A.z, npix * Cout * 4);
    const int Hp = H >> 1, Wp = W >> 1;
    const rsrc_t rp = mk_rsrc(POOL ? A.pooled : A.z, POOL ? (unsigned)A.N * Hp * Wp * Cout * 4 : npix * Cout * 4);
    const int pa = Ca * 4, pb = Cb * 4, pz = Cout * 4;

    float s1[MT][4], s2[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) s1[mt][r] = s2[mt][r] = 0.f;

    const Rs32Sched sched(A.jb.njobs, wave);
    for (int job = sched.first; job < sched.end; job += sched.step) {
        const int cg = job % A.jb.ncg, t = job / A.jb.ncg, rbk = t % A.jb.nrb, img = t / A.jb.nrb;
        const int y0 = rbk * A.jb.rb, y1 = (y0 + A.jb.rb < H) ? y0 + A.jb.rb : H;
        const int col = cg * RS32_COLS + n - 1;
        const bool colok = (unsigned)col < (unsigned)W;
        const bool useful = n >= 1 && n <= RS32_COLS && col < W;
        const int rowpix0 = img * H * W + col;  // + yy * W = the lane's pixel of row yy

        f32x4 ring[P][NSET], ringb[SPLIT ? P : 1][NSET];
        auto issue = [&](int k, int yy) {  // (unconditional instructions: rows outside the job / image are range-check misses, not branches)
            const bool ok = colok && (unsigned)yy < (unsigned)H && yy <= y1;
            const int pix = rowpix0 + yy * W;
#pragma unroll
            for (int s = 0; s < NSET; ++s) {
                ring[k][s] = bld16(ra, (ok && qa[s] >= 0) ? (int)((unsigned)pix * (unsigned)pa + (unsigned)qa[s]) : -1);
                if constexpr (SPLIT) ringb[k][s] = bld16(rbs, (ok && qb[s] >= 0) ? (int)((unsigned)pix * (unsigned)pb + (unsigned)qb[s]) : -1);
            }
        };
#pragma unroll
        for (int k = 0; k < P; ++k) issue(k, y0 - 1 + k);

        float accA[NSET][4], accB[NSET][4];
        f32x4 dprev[MT];
#pragma unroll
        for (int s = 0; s < NSET; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) accA[s][r] = accB[s][r] = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) dprev[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int yb = y0 - 1; yb <= y1; yb += P) {
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const int yy = yb + k;  // (the last pass may run up to P - 1 rows past y1: loads miss the range check, stores are dropped)
                // ---- x~ of input row yy (zero outside the image: the convolution's padding), then the next row's loads into the freed slot
                const bool ok = colok && (unsigned)yy < (unsigned)H;
                float X[NSET][4];
#pragma unroll
                for (int s = 0; s < NSET; ++s) {
                    f32x4 v = ring[k][s];
                    if constexpr (SPLIT) v = or4(v, ringb[k][s]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) X[s][r] = ((ok && chok[s]) ? 1.f : 0.f) * fmaxf(fmaf(v[r], sc[s][r], sh[s][r]), lo[s][r]);  // (0 / 1 factor: branch-free)
                }
                issue(k, yy + P);
                // ---- depthwise: row yy completes output row yy - 1 (kernel row 2), continues row yy (row 1), opens row yy + 1 (row 0)
                float accC[NSET][4];
#pragma unroll
                for (int s = 0; s < NSET; ++s)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x = X[s][r], l = dpp_f<DPP_SHR1>(x), rr = dpp_f<DPP_SHL1>(x);
                        const float* w = wd[s][r];
                        accA[s][r] = fmaf(w[8], rr, fmaf(w[7], x, fmaf(w[6], l, accA[s][r])));
                        accB[s][r] = fmaf(w[5], rr, fmaf(w[4], x, fmaf(w[3], l, accB[s][r])));
                        accC[s][r] = fmaf(w[2], rr, fmaf(w[1], x, w[0] * l));
                    }
                const int yo = yy - 1;
                {
                    const bool rowo = yo >= y0 && yo < y1;
                    const int opix = img * H * W + yo * W + col;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int s = 0; s < NSET; ++s)
#pragma unroll
                            for (int r = 0; r < 4; ++r) d = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt][s][r], accA[s][r], d, 0, 0, 0);
                        const bool st = rowo && useful && 16 * mt + 4 * q < Cout;
                        bst16(rz, st ? (int)((unsigned)opix * (unsigned)pz + (unsigned)(16 * mt + 4 * q) * 4u) : -1, d);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = (st ? 1.f : 0.f) * d[r];
                            s1[mt][r] += v;
                            s2[mt][r] = fmaf(v, v, s2[mt][r]);
                        }
                        if constexpr (POOL) {
                            {  // yo odd = the window's second row: this lane's pixel, the next lane's, and the same two of the row above
                                f32x4 m4;
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const float vv = fmaxf(sg[mt][r] * dprev[mt][r], sg[mt][r] * d[r]);
                                    m4[r] = sg[mt][r] * fmaxf(vv, dpp_f<DPP_SHL1>(vv));
                                }
                                const bool ps = st && (yo & 1) && (n & 1) && (col >> 1) < Wp;  // (n odd <=> even column: col = 14 cg + n - 1)
                                bst16(rp, ps ? ((img * Hp + (yo >> 1)) * Wp + (col >> 1)) * pz + (16 * mt + 4 * q) * 4 : -1, m4);
                            }
                            dprev[mt] = d;
                        }
                    }
                }
#pragma unroll
                for (int s = 0; s < NSET; ++s)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        accA[s][r] = accB[s][r];
                        accB[s][r] = accC[s][r];
                    }
            }
        }
    }
    // ---- batch statistics: DPP sum over the strip's 16 lanes -> one slot per (wave, channel) -> fp64 atomics per workgroup
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a1 = quad16_sum(s1[mt][r]), a2 = quad16_sum(s2[mt][r]);
            if (n == 0) {
                s_stat[wave][0][16 * mt + 4 * q + r] = a1;
                s_stat[wave][1][16 * mt + 4 * q + r] = a2;
            }
        }
    __syncthreads();
    for (int c = tid; c < Cout; c += 256) {
        atomicAdd(&A.gstat[c], (double)((s_stat[0][0][c] + s_stat[1][0][c]) + (s_stat[2][0][c] + s_stat[3][0][c])));
        atomicAdd(&A.gstat[Cout + c], (double)((s_stat[0][1][c] + s_stat[1][1][c]) + (s_stat[2][1][c] + s_stat[3][1][c])));
    }
    bn_finalize_last_block(A.fin, A.gstat, Cout, tid, 256, &s_flag);
}


// ---------------------------------------------------------------------------------------------------------------------------------------------
// backward of one block in ONE pass (Cin, Cout <= 16; direct gradient source): per pixel it reads g (+ g2), z and x once and writes dL/dx~ once
// -- the 2 (Cin + Cout) elements of SURVEY 8(d); `du` is never stored (the round-1 pair k_pw_bwd + k_dw_bwd moved 2 Cout + 5 Cin).
//   dz = A ghat + B z + C (BatchNorm / ReLU backward; A, B, C derived from the block's complete sums in the prologue: bn_fin_coef)
//   du = Wpw^T dz                         fp32 MFMA, D layout over the input channels, valid on the halo lanes too (pointwise)
//   dx~ = depthwise3x3^T(du)              rows yy-2 .. yy of du in registers, left / right by DPP -> stored for the centre row c = yy - 1
//   dWdw[ch][tap] += du(c) x~(c + tap)    36 per-lane accumulators, x~ rows c-1 .. c+1 in registers
//   u(c) = depthwise3x3(x~)  recomputed;  dWpw[o][ch] += sum_px dz(c)[o][px] u(c)[ch][px]: K = pixels -> both operands through a wave-private LDS
//                                         transpose ([channel][pixel], written and read by the same wave: in order, no barrier) -> 4 fp32 MFMAs per row
//   STATS: the producers' BatchNorm-backward sums  S1 = sum ghat', S2 = sum ghat' x~, ghat' = dx~ [x~ > 0]  (BwdLast mode 0: converted by the last workgroup)
// Per-workgroup partials of the weight gradients go to ws (fixed-order second stage: bwd_reduce_or_defer), the producers' sums through exact fp64
// atomics to the launch's last workgroup (bwd_last_finish): bit-reproducible.
// ---------------------------------------------------------------------------------------------------------------------------------------------
struct Rs32B {
    const float *xa, *xb, *tra, *trb, *wdw, *wpw, *g1, *g2, *z, *bn;
    float *gxa, *gxb, *ws;
    int Ca, Cb, Cout, N, H, W;
    Rs32Jobs jb;
    BnFin fin;
    BwdLast bl;
    int ldw;  // row pitch of wpw (= Cin, or the concat's total when the launch handles ONE 16-channel source of a 16 | 16 block: see ocrs_rs32_bwd)
    const float *gl, *whead;  // HEAD: the block in front of out_conv forms its output gradient g[c] = gl * whead[c] from out_conv's dL/dlogit (4 B per pixel)
};
constexpr int RS32_TP = 20;  // pitch (floats) of a [channel][16 pixels] row of the transpose buffers: 16-byte aligned rows, conflict-light

// P = 1: the next row's loads are issued the moment this row's have been consumed and have the whole row's arithmetic (~1 us at two waves per SIMD)
// to arrive; unrolling two rows (P = 2) costs ~100 registers (256 + spills) for nothing.
// DUAL (Cin, Cout <= 8, single source): an 8-channel tensor fills only the lane groups q = 0, 1 -- the other half of every VALU instruction would idle
// (the kernel is VALU-issue bound: ~450 instructions per strip row).  The wave then runs TWO strips side by side, strip q >> 1 in lane groups (2 s,
// 2 s + 1): every per-channel instruction serves 28 columns, and ONE set of four MFMAs still yields du of both strips in place because the weight
// fragment is block-diagonal over (strip, channel): A_r[(m, k)] = [m >> 3 == k >> 1] Wpw[4 (k & 1) + r][m & 7]  ->  D rows 0-7 = strip 0, 8-15 = strip 1.
template <bool SPLIT, bool G2, bool STATS, int P, bool DUAL = false, bool HEAD = false>
__global__ __launch_bounds__(256, (DUAL && G2) ? 2 : 3) void k_rs32_bwd(const Rs32B A) {
    static_assert(!(HEAD && (SPLIT || G2)), "HEAD: the single-source block in front of out_conv");  // (branch-free form: 134 .. 174 registers -> three workgroups per CU)
    static_assert(!(DUAL && SPLIT), "two strips per wave: single-source 8-channel blocks only");
    constexpr int TP2 = DUAL ? 36 : RS32_TP;  // pitch of a transpose row: 16 (or 2 x 16) pixels + pad
    __shared__ float s_cf[3 * 16];
    __shared__ __attribute__((aligned(16))) float s_t[4][3][16 * TP2];  // per wave: dz^T of rows yy (slot yy & 1) and yy - 1 | u^T
    __shared__ float s_red[4][16 * 16 + 9 * 16 + 2 * 16];                   // per wave: dWpw | dWdw | S1 | S2
    // per-channel constants, [row][16 channels]: a lane reads its quad of a row as ONE ds_read_b128 per use instead of holding 36 + 36 registers for the
    // whole launch (at 256 registers the kernel spilled 23 .. 190 of them).  s_w: the 9 depthwise taps; s_k: 0 A, 1 B, 2 C (dz coefficients), 3 / 4 scale /
    // shift of this block's ReLU mask, 5 / 6 / 7 scale / shift / lo of the input's load transform, 8 the producers' saved mean (STATS)
    __shared__ __attribute__((aligned(16))) float s_w[9 * 16];
    __shared__ __attribute__((aligned(16))) float s_k[9 * 16];
    __shared__ int s_flag;
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Ca = A.Ca, Cb = A.Cb, Cin = Ca + Cb, Cout = A.Cout, H = A.H, W = A.W;
    bn_fin_coef(A.fin, Cout, s_cf, tid, 256, blockIdx.x == 0);
    // zeroed: the first tick of a wave reads the dz^T slot of a row it never wrote (times u = 0: but uninitialised LDS may hold NaN / Inf bit patterns of an
    // earlier kernel, and NaN x 0 = NaN in the MFMA -- caught by the full GPU suite in round 6); DUAL: rows 8 .. 15 are never written at all
    for (int i = tid; i < 4 * 3 * 16 * TP2; i += 256) (&s_t[0][0][0])[i] = 0.f;
    __syncthreads();
    if (tid < 16) {
        const int c = tid;
        const bool o = c < Cout, i = c < Cin, ia = c < Ca;
        const float* tr = ia ? A.tra : A.trb;
        const float* sv = ia ? A.bl.saved_a : A.bl.saved_b;
        const bool want = STATS && (ia ? A.bl.gsum_a : A.bl.gsum_b) != nullptr;
        const int Cs = ia ? Ca : Cb, cc = ia ? c : c - Ca;
        s_k[0 * 16 + c] = o ? s_cf[c] : 0.f;
        s_k[1 * 16 + c] = o ? s_cf[Cout + c] : 0.f;
        s_k[2 * 16 + c] = o ? s_cf[2 * Cout + c] : 0.f;
        s_k[3 * 16 + c] = o ? A.bn[c] : 0.f;
        s_k[4 * 16 + c] = o ? A.bn[Cout + c] : 0.f;
        s_k[5 * 16 + c] = i ? tr[cc] : 0.f;
        s_k[6 * 16 + c] = i ? tr[Cs + cc] : 0.f;
        s_k[7 * 16 + c] = i ? tr[2 * Cs + cc] : 0.f;
        s_k[8 * 16 + c] = HEAD ? (o ? A.whead[c] : 0.f) : ((i && want) ? sv[cc] : 0.f);  // (row 8: out_conv's weight in the HEAD form)
        for (int t = 0; t < 9; ++t) s_w[t * 16 + c] = i ? A.wdw[c * 9 + t] : 0.f;
    }
    __syncthreads();

    // ---- per-lane constants: output-channel quad 4 q .. 4 q + 3 (g, z, dz) and input-channel quad 4 q .. 4 q + 3 (x, du, dx)
    const int sB = DUAL ? (q >> 1) : 0;             // the lane's strip
    const int c4 = DUAL ? 4 * (q & 1) : 4 * q;      // the lane's channel quad
    const bool okO = c4 < Cout, okI = c4 < Cin, in_a = c4 < Ca;
    float afd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // A_r[(m = input channel, k)] = Wpw[4 k + r][m]  (DUAL: block-diagonal over the two strips)
        const int mc = DUAL ? (n & 7) : n;
        afd[r] = (c4 + r < Cout && mc < Cin && (!DUAL || (n >> 3) == sB)) ? A.wpw[(c4 + r) * A.ldw + mc] : 0.f;
    }
    const unsigned npix = (unsigned)A.N * H * W;
    const rsrc_t ra = mk_rsrc(A.xa, npix * Ca * 4), rbs = mk_rsrc(SPLIT ? A.xb : A.xa, npix * (SPLIT ? Cb : Ca) * 4);
    const rsrc_t rg1 = HEAD ? mk_rsrc(A.gl, npix * 4) : mk_rsrc(A.g1, npix * Cout * 4), rg2 = mk_rsrc(G2 ? A.g2 : (HEAD ? A.z : A.g1), npix * Cout * 4), rz = mk_rsrc(A.z, npix * Cout * 4);
    const rsrc_t wa = mk_rsrc(A.gxa, npix * Ca * 4), wb = mk_rsrc(SPLIT ? A.gxb : A.gxa, npix * (SPLIT ? Cb : Ca) * 4);
    const unsigned pa = Ca * 4, pb = Cb * 4, po = Cout * 4;
    const int qa = (okI && in_a) ? c4 * 4 : -1, qb = (okI && !in_a) ? (c4 - Ca) * 4 : -1;

    float aw[4][9];  // dWdw partials of the lane's four channels
    f32x4 dwpw = {0.f, 0.f, 0.f, 0.f};  // D[m = o][n = input channel]: lane (ch, k) reg r = dWpw[4 k + r][ch]
    float st1[4], st2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        st1[r] = st2[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) aw[r][t] = 0.f;
    }
    float* tu = s_t[wave][2];

    const Rs32Sched sched(A.jb.njobs, wave);
    for (int job = sched.first; job < sched.end; job += sched.step) {
        const int cg = job % A.jb.ncg, t = job / A.jb.ncg, rbk = t % A.jb.nrb, img = t / A.jb.nrb;
        const int y0 = rbk * A.jb.rb, y1 = (y0 + A.jb.rb < H) ? y0 + A.jb.rb : H;
        const int col = cg * (DUAL ? 2 * RS32_COLS : RS32_COLS) + sB * RS32_COLS + n - 1;
        const bool colok = (unsigned)col < (unsigned)W;
        const bool useful = n >= 1 && n <= RS32_COLS && col < W;
        const int rowpix0 = img * H * W + col;

        f32x4 rg[HEAD ? 1 : P], rg2v[G2 ? P : 1], rzv[P], rx[P], rxb[SPLIT ? P : 1];
        float rgl[HEAD ? P : 1];
        auto issue = [&](int k, int yy) {
            const bool ok = colok && (unsigned)yy < (unsigned)H && yy <= y1;
            const unsigned pix = (unsigned)(rowpix0 + yy * W);
            const int oo = (ok && okO) ? (int)(pix * po + (unsigned)c4 * 4u) : -1;
            if constexpr (HEAD)
                rgl[k] = bld4(rg1, (ok && okO) ? (int)(pix * 4u) : -1);
            else
                rg[k] = bld16(rg1, oo);
            if constexpr (G2) rg2v[k] = bld16(rg2, oo);
            rzv[k] = bld16(rz, oo);
            rx[k] = bld16(ra, (ok && qa >= 0) ? (int)(pix * pa + (unsigned)qa) : -1);
            if constexpr (SPLIT) rxb[k] = bld16(rbs, (ok && qb >= 0) ? (int)(pix * pb + (unsigned)qb) : -1);
        };
#pragma unroll
        for (int k = 0; k < P; ++k) issue(k, y0 - 1 + k);

        // rings: rows yy-2 (index 0), yy-1 (1), yy (2)
        float X[3][4], DU[3][4], dzc[4];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) X[i][r] = DU[i][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) dzc[r] = 0.f;

        for (int yb = y0 - 1; yb <= y1; yb += P) {
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const int yy = yb + k;
                const bool ok = colok && (unsigned)yy < (unsigned)H && yy <= y1;
                // ---- rotate the rings, then fill index 2 with row yy
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    X[0][r] = X[1][r]; X[1][r] = X[2][r];
                    DU[0][r] = DU[1][r]; DU[1][r] = DU[2][r];
                }
                int oz = 0;
                asm volatile("" : "+v"(oz));  // (an opaque zero in the table addresses: keeps hipcc from hoisting the 18 constant vectors out of the loop)
                const float* kk = s_k + c4 + oz;
                const float* kw = s_w + c4 + oz;
                {
                    f32x4 gv;
                    if constexpr (HEAD) {
                        const f32x4 wh = *reinterpret_cast<const f32x4*>(kk + 128);  // g[c] = gl * whead[c]: the product k_head_bwd would have stored
                        gv = wh * rgl[k];
                    } else {
                        gv = rg[k];
                    }
                    if constexpr (G2) gv = gv + rg2v[k];
                    const f32x4 zv = rzv[k];
                    f32x4 xv = rx[k];
                    if constexpr (SPLIT) xv = or4(xv, rxb[k]);
                    const f32x4 cA = *reinterpret_cast<const f32x4*>(kk), cB = *reinterpret_cast<const f32x4*>(kk + 16), cC = *reinterpret_cast<const f32x4*>(kk + 32);
                    const f32x4 msc = *reinterpret_cast<const f32x4*>(kk + 48), msh = *reinterpret_cast<const f32x4*>(kk + 64);
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(kk + 80), sh = *reinterpret_cast<const f32x4*>(kk + 96), lo = *reinterpret_cast<const f32x4*>(kk + 112);
                    // (masks as 0 / 1 factors: hipcc turned per-lane ternaries around these expressions into exec-masked regions -- 17 s_and_saveexec per
                    // row, each a scheduling barrier; out-of-range loads returned 0, so every masked value is finite)
                    const float mO = (ok && okO) ? 1.f : 0.f, mI = (ok && okI) ? 1.f : 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float gh = fmaf(zv[r], msc[r], msh[r]) > 0.f ? gv[r] : 0.f;
                        dzc[r] = mO * fmaf(cA[r], gh, fmaf(cB[r], zv[r], cC[r]));
                        X[2][r] = mI * fmaxf(fmaf(xv[r], sc[r], sh[r]), lo[r]);
                    }
                }
                issue(k, yy + P);
                {
                    float* tzw = s_t[wave][yy & 1];
                    f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        tzw[(c4 + r) * TP2 + sB * 16 + n] = dzc[r];
                        d = __builtin_amdgcn_mfma_f32_16x16x4f32(afd[r], dzc[r], d, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) DU[2][r] = d[r];
                }
                // ---- centre row c = yy - 1
                const int c = yy - 1;
                const bool rowo = c >= y0 && c < y1;
                const bool act = rowo && useful;
                const float actf = act ? 1.f : 0.f;
                f32x4 dx = {0.f, 0.f, 0.f, 0.f};
                float u[4] = {0.f, 0.f, 0.f, 0.f}, dm[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) dm[r] = actf * DU[1][r];
                // dx~(c) = sum_{ky,kx} w[ky][kx] du(c + 1 - ky, col + 1 - kx):  ky = 0 -> row c + 1 (ring 2), kx = 0 -> column + 1 (row_shl)
                // u(c)   = sum_{ky,kx} w[ky][kx] x~(c + ky - 1, col + kx - 1);  dWdw[ky][kx] += du(c) x~(c + ky - 1, col + kx - 1)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(kw + (ky * 3 + 0) * 16), w1 = *reinterpret_cast<const f32x4*>(kw + (ky * 3 + 1) * 16),
                                w2 = *reinterpret_cast<const f32x4*>(kw + (ky * 3 + 2) * 16);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dv = DU[2 - ky][r];
                        dx[r] = fmaf(w2[r], dpp_f<DPP_SHR1>(dv), fmaf(w1[r], dv, fmaf(w0[r], dpp_f<DPP_SHL1>(dv), dx[r])));
                        const float xv = X[ky][r], xl = dpp_f<DPP_SHR1>(xv), xr = dpp_f<DPP_SHL1>(xv);
                        u[r] = fmaf(w2[r], xr, fmaf(w1[r], xv, fmaf(w0[r], xl, u[r])));
                        aw[r][ky * 3 + 0] = fmaf(dm[r], xl, aw[r][ky * 3 + 0]);
                        aw[r][ky * 3 + 1] = fmaf(dm[r], xv, aw[r][ky * 3 + 1]);
                        aw[r][ky * 3 + 2] = fmaf(dm[r], xr, aw[r][ky * 3 + 2]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    u[r] *= actf;
                    if constexpr (STATS) {
                        const float gh = X[1][r] > 0.f ? actf * dx[r] : 0.f;
                        st1[r] += gh;
                        st2[r] = fmaf(gh, X[1][r], st2[r]);
                    }
                }
                {
                    const unsigned opix = (unsigned)(img * H * W + c * W + col);
                    bst16(wa, (act && qa >= 0) ? (int)(opix * pa + (unsigned)qa) : -1, dx);
                    if constexpr (SPLIT) bst16(wb, (act && qb >= 0) ? (int)(opix * pb + (unsigned)qb) : -1, dx);
                }
                // ---- dWpw += dz(c) u(c)^T over the strip's 16 pixels (u is zero on the halo lanes and outside the job's rows)
#pragma unroll
                for (int r = 0; r < 4; ++r) tu[(c4 + r) * TP2 + sB * 16 + n] = u[r];
#pragma unroll
                for (int sp = 0; sp < (DUAL ? 2 : 1); ++sp) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(s_t[wave][c & 1] + n * TP2 + sp * 16 + 4 * q);  // A_t[(m = o, k)] = dz(c)[o][pixel 4 k + t] (written one tick ago)
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(tu + n * TP2 + sp * 16 + 4 * q);  // B_t[(k, n = ch)] = u[ch][pixel 4 k + t]
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) dwpw = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[tt], b4[tt], dwpw, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);  // (keeps hipcc from interleaving the P unrolled rows: their temporaries would all be live at once -> spills)
            }
        }
    }
    // ---- per-workgroup partials: lanes -> wave (DPP row sums) -> LDS slot per wave -> fixed-order sum of the four waves
    float* red = s_red[wave];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[(4 * q + r) * 16 + n] = dwpw[r];  // dWpw[o = 4 k + r][ch = n]: complete over the wave's pixels already
        // (DUAL: rows 4 q + r >= 8 of the per-channel sums are the second strip's partials of channel 4 (q & 1) + r: added below)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float v = quad16_sum(aw[r][t]);
            if (n == 0) red[256 + (4 * q + r) * 9 + t] = v;
        }
        if constexpr (STATS) {
            const float v1 = quad16_sum(st1[r]), v2 = quad16_sum(st2[r]);
            if (n == 0) {
                red[256 + 144 + 4 * q + r] = v1;
                red[256 + 144 + 16 + 4 * q + r] = v2;
            }
        }
    }
    __syncthreads();
    const int ne = Cout * Cin + 9 * Cin;
    for (int e = tid; e < ne + (STATS ? 2 * Cin : 0); e += 256) {
        int idx;
        if (e < Cout * Cin)
            idx = (e / Cin) * 16 + e % Cin;
        else if (e < ne)
            idx = 256 + (e - Cout * Cin);
        else
            idx = 256 + 144 + ((e - ne) / Cin) * 16 + (e - ne) % Cin;
        float v = (s_red[0][idx] + s_red[1][idx]) + (s_red[2][idx] + s_red[3][idx]);
        if (DUAL && e >= Cout * Cin) {
            const int i2 = idx + (e < ne ? 72 : 8);
            v += (s_red[0][i2] + s_red[1][i2]) + (s_red[2][i2] + s_red[3][i2]);
        }
        if (e < ne)
            A.ws[(long)blockIdx.x * ne + e] = v;
        else if (A.bl.raw)
            bwd_last_add(A.bl, Cin, (e - ne) % Cin, (e - ne) / Cin, v);
    }
    if constexpr (STATS) {
        if (A.bl.raw) bwd_last_finish(A.bl, Cin, A.tra, A.trb, tid, 256, &s_flag);
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------------------
// The one-pass backward for the 32-channel shapes of level 1 (NSET register sets of 16 input channels, MT tiles of 16 output channels; (2, 2) does not fit
// the register file): 16 | 16 -> 16 (up.1.contract.seq.0) and 16 -> 32 (down.1.seq.0).  Same structure as k_rs32_bwd, every per-set / per-tile piece
// indexed; a concat source is one whole register set (no per-lane source select).
// ---------------------------------------------------------------------------------------------------------------------------------------------
template <int NSET, int MT, bool SPLIT, bool G2, bool STATS>
__global__ __launch_bounds__(256, 2) void k_rs32_bwdx(const Rs32B A) {
    constexpr int CI = 16 * NSET, CO = 16 * MT;
    static_assert(NSET * MT <= 2 && (!SPLIT || NSET == 2), "shapes of level 1");
    __shared__ float s_cf[3 * CO];
    __shared__ __attribute__((aligned(16))) float s_tz[4][2][CO * RS32_TP];  // per wave: dz^T of rows yy (slot yy & 1) and yy - 1
    __shared__ __attribute__((aligned(16))) float s_tu[4][CI * RS32_TP];     // per wave: u^T
    __shared__ float s_red[4][CO * CI + 9 * CI + 2 * CI];
    __shared__ __attribute__((aligned(16))) float s_w[9 * CI];               // [tap][input channel]
    __shared__ __attribute__((aligned(16))) float s_ko[5 * CO];              // [A | B | C | mask scale | mask shift][output channel]
    __shared__ __attribute__((aligned(16))) float s_ki[3 * CI];              // [scale | shift | lo][input channel]
    __shared__ int s_flag;
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Ca = A.Ca, Cb = A.Cb, Cin = Ca + Cb, Cout = A.Cout, H = A.H, W = A.W;
    bn_fin_coef(A.fin, Cout, s_cf, tid, 256, blockIdx.x == 0);
    for (int i = tid; i < 4 * 2 * CO * RS32_TP; i += 256) (&s_tz[0][0][0])[i] = 0.f;  // (see k_rs32_bwd: the first tick reads a slot it never wrote)
    for (int i = tid; i < 4 * CI * RS32_TP; i += 256) (&s_tu[0][0])[i] = 0.f;
    __syncthreads();
    if (tid < 32) {
        const int c = tid;
        if (c < CO) {
            const bool o = c < Cout;
            s_ko[0 * CO + c] = o ? s_cf[c] : 0.f;
            s_ko[1 * CO + c] = o ? s_cf[Cout + c] : 0.f;
            s_ko[2 * CO + c] = o ? s_cf[2 * Cout + c] : 0.f;
            s_ko[3 * CO + c] = o ? A.bn[c] : 0.f;
            s_ko[4 * CO + c] = o ? A.bn[Cout + c] : 0.f;
        }
        if (c < CI) {
            const bool i = c < Cin, ia = c < Ca;
            const float* tr = ia ? A.tra : A.trb;
            const int Cs = ia ? Ca : Cb, cc = ia ? c : c - Ca;
            s_ki[0 * CI + c] = i ? tr[cc] : 0.f;
            s_ki[1 * CI + c] = i ? tr[Cs + cc] : 0.f;
            s_ki[2 * CI + c] = i ? tr[2 * Cs + cc] : 0.f;
            for (int t = 0; t < 9; ++t) s_w[t * CI + c] = i ? A.wdw[c * 9 + t] : 0.f;
        }
    }
    __syncthreads();
    const int c4 = 4 * q;
    bool okO[MT], okI[NSET];
    float afd[MT][NSET][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) okO[mt] = 16 * mt + c4 < Cout;
#pragma unroll
    for (int st = 0; st < NSET; ++st) okI[st] = 16 * st + c4 < Cin;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int st = 0; st < NSET; ++st)
#pragma unroll
            for (int r = 0; r < 4; ++r)  // A_r[(m = input channel of set st, k)] = Wpw[16 mt + 4 k + r][16 st + m]
                afd[mt][st][r] = (16 * mt + c4 + r < Cout && 16 * st + n < Cin) ? A.wpw[(16 * mt + c4 + r) * A.ldw + 16 * st + n] : 0.f;
    const unsigned npix = (unsigned)A.N * H * W;
    const rsrc_t ra = mk_rsrc(A.xa, npix * Ca * 4), rbs = mk_rsrc(SPLIT ? A.xb : A.xa, npix * (SPLIT ? Cb : Ca) * 4);
    const rsrc_t rg1 = mk_rsrc(A.g1, npix * Cout * 4), rg2 = mk_rsrc(G2 ? A.g2 : A.g1, npix * Cout * 4), rz = mk_rsrc(A.z, npix * Cout * 4);
    const rsrc_t wa = mk_rsrc(A.gxa, npix * Ca * 4), wb = mk_rsrc(SPLIT ? A.gxb : A.gxa, npix * (SPLIT ? Cb : Ca) * 4);
    const unsigned pa = Ca * 4, pb = Cb * 4, po = Cout * 4;

    float aw[NSET][4][9], st1[NSET][4], st2[NSET][4];
    f32x4 dwpw[MT][NSET];
#pragma unroll
    for (int st = 0; st < NSET; ++st) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) dwpw[mt][st] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            st1[st][r] = st2[st][r] = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) aw[st][r][t] = 0.f;
        }
    }
    float* tu = s_tu[wave];

    const Rs32Sched sched(A.jb.njobs, wave);
    for (int job = sched.first; job < sched.end; job += sched.step) {
        const int cg = job % A.jb.ncg, t = job / A.jb.ncg, rbk = t % A.jb.nrb, img = t / A.jb.nrb;
        const int y0 = rbk * A.jb.rb, y1 = (y0 + A.jb.rb < H) ? y0 + A.jb.rb : H;
        const int col = cg * RS32_COLS + n - 1;
        const bool colok = (unsigned)col < (unsigned)W;
        const bool useful = n >= 1 && n <= RS32_COLS && col < W;
        const int rowpix0 = img * H * W + col;

        f32x4 rg[MT], rg2v[G2 ? MT : 1], rzv[MT], rx[NSET];
        auto issue = [&](int yy) {
            const bool ok = colok && (unsigned)yy < (unsigned)H && yy <= y1;
            const unsigned pix = (unsigned)(rowpix0 + yy * W);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int oo = (ok && okO[mt]) ? (int)(pix * po + (unsigned)(16 * mt + c4) * 4u) : -1;
                rg[mt] = bld16(rg1, oo);
                if constexpr (G2) rg2v[mt] = bld16(rg2, oo);
                rzv[mt] = bld16(rz, oo);
            }
#pragma unroll
            for (int st = 0; st < NSET; ++st) {
                const bool fromb = SPLIT && st == 1;  // (a concat source is one whole set: 16 | 16)
                const unsigned off = fromb ? pix * pb + (unsigned)c4 * 4u : pix * pa + (unsigned)(16 * st + c4) * 4u;
                rx[st] = bld16(fromb ? rbs : ra, (ok && okI[st]) ? (int)off : -1);
            }
        };
        issue(y0 - 1);

        float X[3][NSET][4], DU[3][NSET][4];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int st = 0; st < NSET; ++st)
#pragma unroll
                for (int r = 0; r < 4; ++r) X[i][st][r] = DU[i][st][r] = 0.f;

        for (int yy = y0 - 1; yy <= y1; ++yy) {
            const bool ok = colok && (unsigned)yy < (unsigned)H;
#pragma unroll
            for (int st = 0; st < NSET; ++st)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    X[0][st][r] = X[1][st][r]; X[1][st][r] = X[2][st][r];
                    DU[0][st][r] = DU[1][st][r]; DU[1][st][r] = DU[2][st][r];
                }
            int oz = 0;
            asm volatile("" : "+v"(oz));  // (opaque zero: the constant vectors stay LDS reads inside the loop)
            const float* ko = s_ko + c4 + oz;
            const float* ki = s_ki + c4 + oz;
            const float* kw = s_w + c4 + oz;
            float dzc[MT][4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 gv = rg[mt];
                if constexpr (G2) gv = gv + rg2v[mt];
                const f32x4 zv = rzv[mt];
                const f32x4 cA = *reinterpret_cast<const f32x4*>(ko + 16 * mt), cB = *reinterpret_cast<const f32x4*>(ko + CO + 16 * mt),
                            cC = *reinterpret_cast<const f32x4*>(ko + 2 * CO + 16 * mt), msc = *reinterpret_cast<const f32x4*>(ko + 3 * CO + 16 * mt),
                            msh = *reinterpret_cast<const f32x4*>(ko + 4 * CO + 16 * mt);
                const float mO = (ok && okO[mt]) ? 1.f : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gh = fmaf(zv[r], msc[r], msh[r]) > 0.f ? gv[r] : 0.f;
                    dzc[mt][r] = mO * fmaf(cA[r], gh, fmaf(cB[r], zv[r], cC[r]));
                }
            }
#pragma unroll
            for (int st = 0; st < NSET; ++st) {
                const f32x4 xv = rx[st];
                const f32x4 sc = *reinterpret_cast<const f32x4*>(ki + 16 * st), sh = *reinterpret_cast<const f32x4*>(ki + CI + 16 * st),
                            lo = *reinterpret_cast<const f32x4*>(ki + 2 * CI + 16 * st);
                const float mI = (ok && okI[st]) ? 1.f : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) X[2][st][r] = mI * fmaxf(fmaf(xv[r], sc[r], sh[r]), lo[r]);
            }
            issue(yy + 1);
            {
                float* tzw = s_tz[wave][yy & 1];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tzw[(16 * mt + c4 + r) * RS32_TP + n] = dzc[mt][r];
#pragma unroll
                for (int st = 0; st < NSET; ++st) {
                    f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) d = __builtin_amdgcn_mfma_f32_16x16x4f32(afd[mt][st][r], dzc[mt][r], d, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) DU[2][st][r] = d[r];
                }
            }
            const int c = yy - 1;
            const bool act = useful && c >= y0 && c < y1;
            const float actf = act ? 1.f : 0.f;
            const unsigned opix = (unsigned)(img * H * W + c * W + col);
#pragma unroll
            for (int st = 0; st < NSET; ++st) {
                f32x4 dx = {0.f, 0.f, 0.f, 0.f};
                float u[4] = {0.f, 0.f, 0.f, 0.f}, dm[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) dm[r] = actf * DU[1][st][r];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(kw + (ky * 3 + 0) * CI + 16 * st), w1 = *reinterpret_cast<const f32x4*>(kw + (ky * 3 + 1) * CI + 16 * st),
                                w2 = *reinterpret_cast<const f32x4*>(kw + (ky * 3 + 2) * CI + 16 * st);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dv = DU[2 - ky][st][r];
                        dx[r] = fmaf(w2[r], dpp_f<DPP_SHR1>(dv), fmaf(w1[r], dv, fmaf(w0[r], dpp_f<DPP_SHL1>(dv), dx[r])));
                        const float xv = X[ky][st][r], xl = dpp_f<DPP_SHR1>(xv), xr = dpp_f<DPP_SHL1>(xv);
                        u[r] = fmaf(w2[r], xr, fmaf(w1[r], xv, fmaf(w0[r], xl, u[r])));
                        aw[st][r][ky * 3 + 0] = fmaf(dm[r], xl, aw[st][r][ky * 3 + 0]);
                        aw[st][r][ky * 3 + 1] = fmaf(dm[r], xv, aw[st][r][ky * 3 + 1]);
                        aw[st][r][ky * 3 + 2] = fmaf(dm[r], xr, aw[st][r][ky * 3 + 2]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    u[r] *= actf;
                    if constexpr (STATS) {
                        const float gh = X[1][st][r] > 0.f ? actf * dx[r] : 0.f;
                        st1[st][r] += gh;
                        st2[st][r] = fmaf(gh, X[1][st][r], st2[st][r]);
                    }
                    tu[(16 * st + c4 + r) * RS32_TP + n] = u[r];
                }
                const bool fromb = SPLIT && st == 1;
                const unsigned off = fromb ? opix * pb + (unsigned)c4 * 4u : opix * pa + (unsigned)(16 * st + c4) * 4u;
                bst16(fromb ? wb : wa, (act && okI[st]) ? (int)off : -1, dx);
            }
            {
                f32x4 a4[MT], b4[NSET];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a4[mt] = *reinterpret_cast<const f32x4*>(s_tz[wave][c & 1] + (16 * mt + n) * RS32_TP + c4);
#pragma unroll
                for (int st = 0; st < NSET; ++st) b4[st] = *reinterpret_cast<const f32x4*>(tu + (16 * st + n) * RS32_TP + c4);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int st = 0; st < NSET; ++st)
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) dwpw[mt][st] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[mt][tt], b4[st][tt], dwpw[mt][st], 0, 0, 0);
            }
        }
    }
    float* red = s_red[wave];
#pragma unroll
    for (int st = 0; st < NSET; ++st)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) red[(16 * mt + c4 + r) * CI + 16 * st + n] = dwpw[mt][st][r];  // dWpw[o = 16 mt + 4 k + r][ch = 16 st + n]
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float v = quad16_sum(aw[st][r][t]);
                if (n == 0) red[CO * CI + (16 * st + c4 + r) * 9 + t] = v;
            }
            if constexpr (STATS) {
                const float v1 = quad16_sum(st1[st][r]), v2 = quad16_sum(st2[st][r]);
                if (n == 0) {
                    red[CO * CI + 9 * CI + 16 * st + c4 + r] = v1;
                    red[CO * CI + 9 * CI + CI + 16 * st + c4 + r] = v2;
                }
            }
        }
    __syncthreads();
    const int ne = Cout * Cin + 9 * Cin;
    for (int e = tid; e < ne + (STATS ? 2 * Cin : 0); e += 256) {
        int idx;
        if (e < Cout * Cin)
            idx = (e / Cin) * CI + e % Cin;
        else if (e < ne)
            idx = CO * CI + (e - Cout * Cin);
        else
            idx = CO * CI + 9 * CI + ((e - ne) / Cin) * CI + (e - ne) % Cin;
        const float v = (s_red[0][idx] + s_red[1][idx]) + (s_red[2][idx] + s_red[3][idx]);
        if (e < ne)
            A.ws[(long)blockIdx.x * ne + e] = v;
        else if (A.bl.raw)
            bwd_last_add(A.bl, Cin, (e - ne) % Cin, (e - ne) / Cin, v);
    }
    if constexpr (STATS) {
        if (A.bl.raw) bwd_last_finish(A.bl, Cin, A.tra, A.trb, tid, 256, &s_flag);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// The same one-pass backward for a block whose output feeds ONLY MaxPool2d(2) (models.py:54): the gradient g (+ g2) arrives at half resolution and is
// routed to each 2 x 2 window's first maximum in post-ReLU space (row-major order, strict >: ocrs_bn_bwd_reduce's / ocrs_pw_bwd's rule).  A tick is a
// ROW PAIR (2 i, 2 i + 1) = one row of windows; a window's two columns are adjacent lanes (quad_perm [1,0,3,2]), so a strip needs TWO halo columns
// each side (12 output columns of 16 lanes) and a job two halo rows each side.  Rings hold rows 2 i - 2 .. 2 i + 1; the centre rows of a tick are
// 2 i - 1 and 2 i.  Single source, Cin, Cout <= 16.
// ---------------------------------------------------------------------------------------------------------------------------------------------
constexpr int RS32P_COLS = 12;

struct Rs32Acc {
    float aw[4][9];
    f32x4 dwpw;
    float st1[4], st2[4];
};
// centre row from the three ring rows around it: dx~, u (zeroed unless act), dWdw and the producers' sums accumulated
template <bool STATS>
__device__ __forceinline__ void rs32_centre(const float (&Xm)[4], const float (&X0)[4], const float (&Xp)[4], const float (&Dm)[4], const float (&D0)[4],
                                            const float (&Dp)[4], const float* kw, bool act, f32x4& dx, float (&u)[4], Rs32Acc& acc) {
    const float actf = act ? 1.f : 0.f;  // (0 / 1 factor instead of per-lane ternaries: branch-free)
    float dm[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        dm[r] = actf * D0[r];
        dx[r] = 0.f;
        u[r] = 0.f;
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(kw + (ky * 3 + 0) * 16), w1 = *reinterpret_cast<const f32x4*>(kw + (ky * 3 + 1) * 16),
                    w2 = *reinterpret_cast<const f32x4*>(kw + (ky * 3 + 2) * 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dv = ky == 0 ? Dp[r] : (ky == 1 ? D0[r] : Dm[r]);  // dx~(c) = sum w[ky][kx] du(c + 1 - ky, col + 1 - kx)
            dx[r] = fmaf(w2[r], dpp_f<DPP_SHR1>(dv), fmaf(w1[r], dv, fmaf(w0[r], dpp_f<DPP_SHL1>(dv), dx[r])));
            const float xv = ky == 0 ? Xm[r] : (ky == 1 ? X0[r] : Xp[r]);  // u(c) = sum w[ky][kx] x~(c + ky - 1, col + kx - 1)
            const float xl = dpp_f<DPP_SHR1>(xv), xr = dpp_f<DPP_SHL1>(xv);
            u[r] = fmaf(w2[r], xr, fmaf(w1[r], xv, fmaf(w0[r], xl, u[r])));
            acc.aw[r][ky * 3 + 0] = fmaf(dm[r], xl, acc.aw[r][ky * 3 + 0]);
            acc.aw[r][ky * 3 + 1] = fmaf(dm[r], xv, acc.aw[r][ky * 3 + 1]);
            acc.aw[r][ky * 3 + 2] = fmaf(dm[r], xr, acc.aw[r][ky * 3 + 2]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        u[r] *= actf;
        if constexpr (STATS) {
            const float gh = X0[r] > 0.f ? actf * dx[r] : 0.f;
            acc.st1[r] += gh;
            acc.st2[r] = fmaf(gh, X0[r], acc.st2[r]);
        }
    }
}

template <bool G2, bool STATS>
__global__ __launch_bounds__(256, 2) void k_rs32_bwdp(const Rs32B A) {
    __shared__ float s_cf[3 * 16];
    __shared__ __attribute__((aligned(16))) float s_t[4][5][16 * RS32_TP];  // per wave: dz^T of rows (row & 3) | u^T
    __shared__ float s_red[4][16 * 16 + 9 * 16 + 2 * 16];
    __shared__ __attribute__((aligned(16))) float s_w[9 * 16];
    __shared__ __attribute__((aligned(16))) float s_k[8 * 16];
    __shared__ int s_flag;
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = A.Ca, Cout = A.Cout, H = A.H, W = A.W, Hp = H >> 1, Wp = W >> 1;
    bn_fin_coef(A.fin, Cout, s_cf, tid, 256, blockIdx.x == 0);
    for (int i = tid; i < 4 * 5 * 16 * RS32_TP; i += 256) (&s_t[0][0][0])[i] = 0.f;  // (see k_rs32_bwd: the first tick reads a slot it never wrote)
    __syncthreads();
    if (tid < 16) {
        const int c = tid;
        const bool o = c < Cout, i = c < Cin;
        s_k[0 * 16 + c] = o ? s_cf[c] : 0.f;
        s_k[1 * 16 + c] = o ? s_cf[Cout + c] : 0.f;
        s_k[2 * 16 + c] = o ? s_cf[2 * Cout + c] : 0.f;
        s_k[3 * 16 + c] = o ? A.bn[c] : 0.f;
        s_k[4 * 16 + c] = o ? A.bn[Cout + c] : 0.f;
        s_k[5 * 16 + c] = i ? A.tra[c] : 0.f;
        s_k[6 * 16 + c] = i ? A.tra[Cin + c] : 0.f;
        s_k[7 * 16 + c] = i ? A.tra[2 * Cin + c] : 0.f;
        for (int t = 0; t < 9; ++t) s_w[t * 16 + c] = i ? A.wdw[c * 9 + t] : 0.f;
    }
    __syncthreads();
    const int c4 = 4 * q;
    const bool okO = c4 < Cout, okI = c4 < Cin;
    float afd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) afd[r] = (c4 + r < Cout && n < Cin) ? A.wpw[(c4 + r) * A.ldw + n] : 0.f;
    const unsigned npix = (unsigned)A.N * H * W, nppx = (unsigned)A.N * Hp * Wp;
    const rsrc_t ra = mk_rsrc(A.xa, npix * Cin * 4), rz = mk_rsrc(A.z, npix * Cout * 4), wa = mk_rsrc(A.gxa, npix * Cin * 4);
    const rsrc_t rg1 = mk_rsrc(A.g1, nppx * Cout * 4), rg2 = mk_rsrc(G2 ? A.g2 : A.g1, nppx * Cout * 4);
    const unsigned pa = Cin * 4, po = Cout * 4;
    const int e = n & 1;  // column parity inside the window (a strip starts at an even column)

    Rs32Acc acc;
    acc.dwpw = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        acc.st1[r] = acc.st2[r] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc.aw[r][t] = 0.f;
    }
    float* tu = s_t[wave][4];

    const Rs32Sched sched(A.jb.njobs, wave);
    for (int job = sched.first; job < sched.end; job += sched.step) {
        const int cg = job % A.jb.ncg, t = job / A.jb.ncg, rbk = t % A.jb.nrb, img = t / A.jb.nrb;
        const int y0 = rbk * A.jb.rb, y1 = (y0 + A.jb.rb < H) ? y0 + A.jb.rb : H;
        const int col = cg * RS32P_COLS + n - 2;
        const bool colok = (unsigned)col < (unsigned)W;
        const bool useful = n >= 2 && n < 2 + RS32P_COLS && col < W;
        const int i0 = (y0 >> 1) - 1, i1 = y1 >> 1;  // ticks (row pairs) i0 .. i1

        f32x4 pz0, pz1, px0, px1, pg, pg2;
        auto issue = [&](int i) {
            const bool live = i <= i1;
            const int ya = 2 * i, yb = 2 * i + 1;
            const bool oka = live && colok && (unsigned)ya < (unsigned)H, okb = live && colok && (unsigned)yb < (unsigned)H;
            const unsigned pixa = (unsigned)((img * H + ya) * W + col), pixb = pixa + (unsigned)W;
            pz0 = bld16(rz, (oka && okO) ? (int)(pixa * po + (unsigned)c4 * 4u) : -1);
            pz1 = bld16(rz, (okb && okO) ? (int)(pixb * po + (unsigned)c4 * 4u) : -1);
            px0 = bld16(ra, (oka && okI) ? (int)(pixa * pa + (unsigned)c4 * 4u) : -1);
            px1 = bld16(ra, (okb && okI) ? (int)(pixb * pa + (unsigned)c4 * 4u) : -1);
            const bool okp = live && colok && okO && (unsigned)i < (unsigned)Hp && (col >> 1) < Wp;
            const int og = okp ? (int)((unsigned)((img * Hp + i) * Wp + (col >> 1)) * po + (unsigned)c4 * 4u) : -1;
            pg = bld16(rg1, og);
            if constexpr (G2) pg2 = bld16(rg2, og);
        };
        issue(i0);

        float X[4][4], DU[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) X[a][r] = DU[a][r] = 0.f;

        for (int i = i0; i <= i1; ++i) {
            const int ya = 2 * i, yb = 2 * i + 1;
            const bool oka = colok && (unsigned)ya < (unsigned)H, okb = colok && (unsigned)yb < (unsigned)H;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                X[0][r] = X[2][r]; X[1][r] = X[3][r];
                DU[0][r] = DU[2][r]; DU[1][r] = DU[3][r];
            }
            int oz = 0;
            asm volatile("" : "+v"(oz));
            const float* kk = s_k + c4 + oz;
            const float* kw = s_w + c4 + oz;
            float dz0[4], dz1[4];
            const float mOa = (oka && okO) ? 1.f : 0.f, mOb = (okb && okO) ? 1.f : 0.f, mIa = (oka && okI) ? 1.f : 0.f, mIb = (okb && okI) ? 1.f : 0.f;
            {
                const f32x4 z0 = pz0, z1 = pz1, x0 = px0, x1 = px1;
                f32x4 gv = pg;
                if constexpr (G2) gv = gv + pg2;
                const f32x4 cA = *reinterpret_cast<const f32x4*>(kk), cB = *reinterpret_cast<const f32x4*>(kk + 16), cC = *reinterpret_cast<const f32x4*>(kk + 32);
                const f32x4 msc = *reinterpret_cast<const f32x4*>(kk + 48), msh = *reinterpret_cast<const f32x4*>(kk + 64);
                const f32x4 sc = *reinterpret_cast<const f32x4*>(kk + 80), sh = *reinterpret_cast<const f32x4*>(kk + 96), lo = *reinterpret_cast<const f32x4*>(kk + 112);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // the window's first maximum in post-ReLU space, row-major order: (0,0) (0,1) (1,0) (1,1)
                    const float a0 = fmaxf(fmaf(z0[r], msc[r], msh[r]), 0.f), a1 = fmaxf(fmaf(z1[r], msc[r], msh[r]), 0.f);
                    const float b0 = dpp_f<0xB1>(a0), b1 = dpp_f<0xB1>(a1);  // the other column of the window (lane ^ 1)
                    const float L0 = e ? b0 : a0, R0 = e ? a0 : b0, L1 = e ? b1 : a1, R1 = e ? a1 : b1;
                    float best = L0;
                    int k = 0;
                    if (R0 > best) { best = R0; k = 1; }
                    if (L1 > best) { best = L1; k = 2; }
                    if (R1 > best) { best = R1; k = 3; }
                    const float gp = best > 0.f ? gv[r] : 0.f;
                    const float gh0 = (k == e) ? gp : 0.f, gh1 = (k == 2 + e) ? gp : 0.f;
                    dz0[r] = mOa * fmaf(cA[r], gh0, fmaf(cB[r], z0[r], cC[r]));
                    dz1[r] = mOb * fmaf(cA[r], gh1, fmaf(cB[r], z1[r], cC[r]));
                    X[2][r] = mIa * fmaxf(fmaf(x0[r], sc[r], sh[r]), lo[r]);
                    X[3][r] = mIb * fmaxf(fmaf(x1[r], sc[r], sh[r]), lo[r]);
                }
            }
            issue(i + 1);
            {
                float* tza = s_t[wave][ya & 3];
                float* tzb = s_t[wave][yb & 3];
                f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    tza[(c4 + r) * RS32_TP + n] = dz0[r];
                    tzb[(c4 + r) * RS32_TP + n] = dz1[r];
                    d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(afd[r], dz0[r], d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(afd[r], dz1[r], d1, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    DU[2][r] = d0[r];
                    DU[3][r] = d1[r];
                }
            }
            // ---- centre rows 2 i - 1 (rings 0, 1, 2) and 2 i (rings 1, 2, 3)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int c = 2 * i - 1 + half;
                const bool act = useful && c >= y0 && c < y1;
                f32x4 dx;
                float u[4];
                if (half == 0)
                    rs32_centre<STATS>(X[0], X[1], X[2], DU[0], DU[1], DU[2], kw, act, dx, u, acc);
                else
                    rs32_centre<STATS>(X[1], X[2], X[3], DU[1], DU[2], DU[3], kw, act, dx, u, acc);
                bst16(wa, (act && okI) ? (int)((unsigned)((img * H + c) * W + col) * pa + (unsigned)c4 * 4u) : -1, dx);
#pragma unroll
                for (int r = 0; r < 4; ++r) tu[(c4 + r) * RS32_TP + n] = u[r];
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(s_t[wave][c & 3] + n * RS32_TP + c4);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(tu + n * RS32_TP + c4);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) acc.dwpw = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[tt], b4[tt], acc.dwpw, 0, 0, 0);
            }
        }
    }
    float* red = s_red[wave];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[(c4 + r) * 16 + n] = acc.dwpw[r];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float v = quad16_sum(acc.aw[r][t]);
            if (n == 0) red[256 + (c4 + r) * 9 + t] = v;
        }
        if constexpr (STATS) {
            const float v1 = quad16_sum(acc.st1[r]), v2 = quad16_sum(acc.st2[r]);
            if (n == 0) {
                red[256 + 144 + c4 + r] = v1;
                red[256 + 144 + 16 + c4 + r] = v2;
            }
        }
    }
    __syncthreads();
    const int ne = Cout * Cin + 9 * Cin;
    for (int el = tid; el < ne + (STATS ? 2 * Cin : 0); el += 256) {
        int idx;
        if (el < Cout * Cin)
            idx = (el / Cin) * 16 + el % Cin;
        else if (el < ne)
            idx = 256 + (el - Cout * Cin);
        else
            idx = 256 + 144 + ((el - ne) / Cin) * 16 + (el - ne) % Cin;
        const float v = (s_red[0][idx] + s_red[1][idx]) + (s_red[2][idx] + s_red[3][idx]);
        if (el < ne)
            A.ws[(long)blockIdx.x * ne + el] = v;
        else if (A.bl.raw)
            bwd_last_add(A.bl, Cin, (el - ne) % Cin, (el - ne) / Cin, v);
    }
    if constexpr (STATS) {
        if (A.bl.raw) bwd_last_finish(A.bl, Cin, A.tra, A.trb, tid, 256, &s_flag);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// ConvTranspose2d(Cup -> Cout, k 3, s 2) weight + bias gradient in fp32 (models.py:76-78; train_detection.py:96 runs its autograd in fp32):
//   dW[c][o][kh][kw] = sum_{n,i,j} x~[i][j][c] g[2 i + kh][2 j + kw][o],   db[o] = sum g[o]
// a K = pixels GEMM with M = Cup and N = (position, o) = 9 Cout.  Row-streaming over the INPUT grid: a strip is 16 input pixels (no halo: every lane
// loads its own 3 x 3 window of the output gradient -- neighbouring windows overlap in L1 / L2, HBM sees each byte once), x~ and the nine window
// positions go through a wave-private LDS transpose ([row][16 pixels]) and every (16 x 16) tile of dW takes 4 exact-fp32 MFMAs per strip row.
// Replaces k_wgrad_gather<float> + k_channel_sum for the wide levels: that gather kernel (383 registers, scatter-bound) took 3.4 / 2.0 / 0.8 ms at
// levels 0 / 1 / 2 and, on the backward's side stream, kept whole CUs from the main stream's kernels (a 64-thread finalize launch waited 1.1 ms).
// MT = Cup / 16;  CE = output channels per launch (8 or 16: Cout = 32 runs as two launches, o0 = 0 / 16);  NT = ceil(9 CE / 16) N tiles.
// ---------------------------------------------------------------------------------------------------------------------------------------------
struct Rs32CW {
    const float *x, *tr, *g;
    float* ws;  // per-WAVE partials [4 nb][Cup * 9 * CE + CE]: dW[c][o0 + oo][pos] at (c * CE + oo) * 9 + pos, then db[o0 + oo]
    int Cup, Cout, o0, N, h, w, H, W;
    Rs32Jobs jb;
};

template <int MT, int CE>
__global__ __launch_bounds__(256, 2) void k_rs32_ctw(const Rs32CW A) {
    constexpr int NT = (9 * CE + 15) / 16, PPS = 16 / CE;  // N tiles; window positions per register set (2 for CE = 8, 1 for CE = 16)
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];  // per wave: x~^T [16 MT][TP] | g^T [16 NT][TP]
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* tx = s_dyn + wave * (16 * (MT + NT) * RS32_TP);
    float* tg = tx + 16 * MT * RS32_TP;
    const int Cup = A.Cup, Cout = A.Cout, h = A.h, w = A.w, H = A.H, W = A.W;
    const int c4 = 4 * q;
    // the lane's slot of a g set: window position (within the set) and channel quad
    const int ppos = PPS == 2 ? (q >> 1) : 0, oq = PPS == 2 ? 4 * (q & 1) : c4;
    float sc[MT][4], sh[MT][4], lo[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[mt][r] = A.tr[16 * mt + c4 + r];
            sh[mt][r] = A.tr[Cup + 16 * mt + c4 + r];
            lo[mt][r] = A.tr[2 * Cup + 16 * mt + c4 + r];
        }
    const unsigned npi = (unsigned)A.N * h * w, npo = (unsigned)A.N * H * W;
    const rsrc_t rx = mk_rsrc(A.x, npi * Cup * 4), rg = mk_rsrc(A.g, npo * Cout * 4);
    const unsigned px_ = Cup * 4, pg = Cout * 4;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float db[4] = {0.f, 0.f, 0.f, 0.f};

    const Rs32Sched sched(A.jb.njobs, wave);
    for (int job = sched.first; job < sched.end; job += sched.step) {
        const int cg = job % A.jb.ncg, t = job / A.jb.ncg, rbk = t % A.jb.nrb, img = t / A.jb.nrb;
        const int y0 = rbk * A.jb.rb, y1 = (y0 + A.jb.rb < h) ? y0 + A.jb.rb : h;
        const int j = cg * 16 + n;
        const bool jok = j < w;
        // rows are loaded one tick ahead (the first form loaded and consumed in the same tick: every row paid a full memory latency, 1068 us at level 0)
        f32x4 xv[MT], gv[NT], xn[MT], gn[NT];
        auto issue = [&](int i) {
            const bool live = i < y1;
            const unsigned ipix = (unsigned)((img * h + i) * w + j);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xn[mt] = bld16(rx, (live && jok) ? (int)(ipix * px_ + (unsigned)(16 * mt + c4) * 4u) : -1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int pos = nt * PPS + ppos;  // 0 .. 8 (9: the empty half of the last set when CE = 8)
                const int kh = pos / 3, kw = pos - 3 * kh;
                const int oy = 2 * i + kh, ox = 2 * j + kw;
                const bool ok = live && jok && pos < 9 && oy < H && ox < W;
                gn[nt] = bld16(rg, ok ? (int)((unsigned)((img * H + oy) * W + ox) * pg + (unsigned)(A.o0 + oq) * 4u) : -1);
            }
        };
        issue(y0);
        for (int i = y0; i < y1; ++i) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xv[mt] = xn[mt];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) gv[nt] = gn[nt];
            issue(i + 1);
            // ---- x~ and g transposed into the wave's LDS rows (same wave writes and reads: in order, no barrier)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) tx[(16 * mt + c4 + r) * RS32_TP + n] = jok ? fmaxf(fmaf(xv[mt][r], sc[mt][r], sh[mt][r]), lo[mt][r]) : 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int pos = nt * PPS + ppos;
                const int kh = pos / 3, kw = pos - 3 * kh;
                // every output pixel belongs to exactly one (input pixel, position): kh, kw < 2, or the last input row / column
                const bool own = (kh < 2 || i == h - 1) && (kw < 2 || j == w - 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    tg[(16 * nt + c4 + r) * RS32_TP + n] = gv[nt][r];
                    db[r] += own ? gv[nt][r] : 0.f;  // (out-of-range positions loaded 0)
                }
            }
            f32x4 a4[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a4[mt] = *reinterpret_cast<const f32x4*>(tx + (16 * mt + n) * RS32_TP + c4);  // A_t[(m = c, k)] = x~[c][pixel 4 k + t]
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(tg + (16 * nt + n) * RS32_TP + c4);  // B_t[(k, n = column)] = g[column][pixel 4 k + t]
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[mt][tt], b4[tt], acc[mt][nt], 0, 0, 0);
            }
        }
    }
    // ---- the wave's partials: acc[mt][nt] lane (column jj = n, k) reg r = dW[c = 16 mt + 4 k + r][column 16 nt + jj], column = pos * CE + oo
    const int ne = Cup * 9 * CE + CE;
    float* out = A.ws + (long)(blockIdx.x * 4 + wave) * ne;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int colj = 16 * nt + n, pos = colj / CE, oo = colj - pos * CE;
            if (pos < 9) {
#pragma unroll
                for (int r = 0; r < 4; ++r) out[((16 * mt + c4 + r) * CE + oo) * 9 + pos] = acc[mt][nt][r];
            }
        }
    // bias gradient: lanes of one channel quad (all n; for CE = 8 both position halves q, q + 2) -> LDS -> one value per channel
    __syncthreads();  // (every wave is done with its transpose rows: s_dyn is reused as [wave][q][r])
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float v = quad16_sum(db[r]);
        if (n == 0) s_dyn[4096 + (wave * 4 + q) * 4 + r] = v;
    }
    __syncthreads();
    if (lane < CE) {
        const int oo = lane, qq = PPS == 2 ? (oo >> 2) : (oo >> 2), rr = oo & 3;
        float v = s_dyn[4096 + (wave * 4 + qq) * 4 + rr];
        if (PPS == 2) v += s_dyn[4096 + (wave * 4 + qq + 2) * 4 + rr];
        out[Cup * 9 * CE + oo] = v;
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------------------
// ConvTranspose2d(Cup -> Cout, k 3, s 2) + bias + crop, forward, fp32 (models.py:76-78, 87).  Row-streaming over the INPUT grid: input pixel (i, j) owns
// the four output pixels (2 i + py, 2 j + px); out[2i+py][2j+px][o] = b[o] + sum_{di in D(py), dj in D(px)} sum_c x~[i-di][j-dj][c] W[c][o][py+2di][px+2dj],
// D(0) = {0, 1}, D(1) = {0}: a GEMM with M = (parity, o) = 4 Cout, K = (neighbour, c) = 4 Cup whose B operands are the x~ registers of this row and the
// previous one and their left neighbours (DPP row_shr:1): a strip is 16 lanes = 15 input columns + the left halo lane.  The effective weight fragments
// (zero where a parity has no such neighbour) are built once per workgroup in LDS, [tile][neighbour][set][lane][4 K steps]: one ds_read_b128 per four
// MFMAs.  The extra output row / column 2 h, 2 w (kept when the skip tensor is odd-sized) come from the tick i = h / the lane j = w, whose own x~ is 0.
// ---------------------------------------------------------------------------------------------------------------------------------------------
struct Rs32CF {
    const float *x, *tr, *wt, *bias;  // wt: master [Cup][Cout][3][3]
    float* out;
    int Cup, Cout, N, h, w, H, W;
    Rs32Jobs jb;  // over (h + 1) x (w + 1) input positions, 15 columns per strip
};

template <int NSET, int MT>
__global__ __launch_bounds__(256, 2) void k_rs32_ctf(const Rs32CF A) {
    extern __shared__ __attribute__((aligned(16))) float s_af[];  // [MT][4 nb][NSET][64 lanes][4]
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cup = A.Cup, Cout = A.Cout, h = A.h, w = A.w, H = A.H, W = A.W;
    for (int e = tid; e < MT * 4 * NSET * 64 * 4; e += 256) {
        const int r = e & 3, ln = (e >> 2) & 63, rest = e >> 8, s = rest % NSET, nb = (rest / NSET) & 3, mt = rest / (NSET * 4);
        const int m = 16 * mt + (ln & 15), par = m / Cout, o = m - par * Cout, py = par >> 1, px = par & 1;
        const int c = 16 * s + 4 * (ln >> 4) + r, di = nb >> 1, dj = nb & 1;
        const bool ok = par < 4 && c < Cup && !(py == 1 && di == 1) && !(px == 1 && dj == 1);
        s_af[e] = ok ? A.wt[((c * Cout + o) * 3 + (py + 2 * di)) * 3 + (px + 2 * dj)] : 0.f;
    }
    __syncthreads();
    const int c4 = 4 * q;
    float sc[NSET][4], sh[NSET][4], lo[NSET][4];
#pragma unroll
    for (int s = 0; s < NSET; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * s + c4 + r;
            sc[s][r] = c < Cup ? A.tr[c] : 0.f;
            sh[s][r] = c < Cup ? A.tr[Cup + c] : 0.f;
            lo[s][r] = c < Cup ? A.tr[2 * Cup + c] : 0.f;
        }
    // the lane's output slot in every M tile: parity and channel quad
    int parq[MT], oq[MT];
    f32x4 bq[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m0 = 16 * mt + c4;
        parq[mt] = m0 / Cout;
        oq[mt] = m0 - parq[mt] * Cout;
#pragma unroll
        for (int r = 0; r < 4; ++r) bq[mt][r] = parq[mt] < 4 ? A.bias[oq[mt] + r] : 0.f;
    }
    const unsigned npi = (unsigned)A.N * h * w, npo = (unsigned)A.N * H * W;
    const rsrc_t rx = mk_rsrc(A.x, npi * Cup * 4), ro = mk_rsrc(A.out, npo * Cout * 4);
    const unsigned pxb = Cup * 4, pob = Cout * 4;
    const f32x4* af = reinterpret_cast<const f32x4*>(s_af) + lane;

    const Rs32Sched sched(A.jb.njobs, wave);
    for (int job = sched.first; job < sched.end; job += sched.step) {
        const int cg = job % A.jb.ncg, t = job / A.jb.ncg, rbk = t % A.jb.nrb, img = t / A.jb.nrb;
        const int y0 = rbk * A.jb.rb, y1 = (y0 + A.jb.rb < h + 1) ? y0 + A.jb.rb : h + 1;
        const int j = cg * 15 + n - 1;
        const bool jin = (unsigned)j < (unsigned)w;          // a real input column
        const bool useful = n >= 1 && j <= w;                 // owns output columns 2 j, 2 j + 1 (j = w: only 2 w, from its left neighbour)
        auto load = [&](int i, f32x4 (&v)[NSET]) {
            const bool ok = jin && (unsigned)i < (unsigned)h;
            const unsigned pix = (unsigned)((img * h + i) * w + j);
#pragma unroll
            for (int s = 0; s < NSET; ++s) v[s] = bld16(rx, (ok && 16 * s + c4 < Cup) ? (int)(pix * pxb + (unsigned)(16 * s + c4) * 4u) : -1);
        };
        auto xform = [&](int i, const f32x4 (&v)[NSET], float (&X)[NSET][4]) {
            const bool ok = jin && (unsigned)i < (unsigned)h;
#pragma unroll
            for (int s = 0; s < NSET; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r) X[s][r] = (ok && 16 * s + c4 < Cup) ? fmaxf(fmaf(v[s][r], sc[s][r], sh[s][r]), lo[s][r]) : 0.f;
        };
        float Xp[NSET][4], Xc[NSET][4];
        f32x4 raw[NSET];
        load(y0 - 1, raw);
        xform(y0 - 1, raw, Xp);
        load(y0, raw);
        for (int i = y0; i < y1; ++i) {
            xform(i, raw, Xc);
            load(i + 1 < y1 ? i + 1 : i, raw);  // next row (the last tick re-reads its own row: harmless, keeps the loop body uniform)
            int oz = 0;
            asm volatile("" : "+v"(oz));  // (opaque zero: the weight fragments are loop-invariant LDS reads -- hoisted, they cost 32 .. 256 registers)
            const f32x4* afl = af + oz;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 d = bq[mt];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int s = 0; s < NSET; ++s) {
                        const f32x4 a4 = afl[((mt * 4 + nb) * NSET + s) * 64];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float xv = (nb >> 1) ? Xp[s][r] : Xc[s][r];
                            const float bv = (nb & 1) ? dpp_f<DPP_SHR1>(xv) : xv;  // dj = 1: the left neighbour's value
                            d = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[r], bv, d, 0, 0, 0);
                        }
                    }
                const int py = parq[mt] >> 1, px = parq[mt] & 1, oy = 2 * i + py, ox = 2 * j + px;
                const bool st = useful && parq[mt] < 4 && oy < H && ox < W;
                bst16(ro, st ? (int)((unsigned)((img * H + oy) * W + ox) * pob + (unsigned)oq[mt] * 4u) : -1, d);
            }
#pragma unroll
            for (int s = 0; s < NSET; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r) Xp[s][r] = Xc[s][r];
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------------------
// ConvTranspose2d input gradient in fp32 (+ the BatchNorm-backward sums of the block that produced x when the ConvTranspose is its only consumer):
//   dx~[i][j][c] = sum_{kh,kw,o} W[c][o][kh][kw] g[2 i + kh][2 j + kw][o]      -- a stride-2 3 x 3 convolution of the output gradient
// Same walk as k_rs32_ctw (16 input pixels per strip, every lane loads its own 3 x 3 window one row ahead); the window registers ARE the B operands
// (K = (position, o)), the per-(tile, set) weight fragments come from an LDS table built once per workgroup from the master weight.  Replaces
// k_convt_dgrad<float> (635 / 436 us at levels 0 / 1) and, with STATS, the k_bn_bwd_reduce<float> pass behind it (287 / 157 us).
// ---------------------------------------------------------------------------------------------------------------------------------------------
struct Rs32CD {
    const float *g, *wt, *x, *tr;
    float* dx;
    int Cup, Cout, N, h, w, H, W;
    Rs32Jobs jb;
    BwdLast bl;
};

template <int MT, int CE, bool STATS>
__global__ __launch_bounds__(256, 2) void k_rs32_ctd(const Rs32CD A) {
    constexpr int NT = (9 * CE + 15) / 16, PPS = 16 / CE;
    __shared__ __attribute__((aligned(16))) float s_af[MT * NT * 64 * 4];  // [tile][set][lane][K step]
    __shared__ float s_red[4][2][16 * MT];
    __shared__ int s_flag;
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cup = A.Cup, Cout = A.Cout, h = A.h, w = A.w, H = A.H, W = A.W;
    for (int e = tid; e < MT * NT * 64 * 4; e += 256) {
        const int r = e & 3, ln = (e >> 2) & 63, rest = e >> 8, nt = rest % NT, mt = rest / NT;
        const int c = 16 * mt + (ln & 15), col = 16 * nt + 4 * (ln >> 4) + r, pos = col / CE, o = col - pos * CE;  // A_r[(m = c, k)] = W[c][o][pos], column = 16 nt + 4 k + r
        s_af[e] = (c < Cup && pos < 9 && o < Cout) ? A.wt[(c * Cout + o) * 9 + pos] : 0.f;
    }
    __syncthreads();
    const int c4 = 4 * q;
    const int ppos = PPS == 2 ? (q >> 1) : 0, oq = PPS == 2 ? 4 * (q & 1) : c4;
    float sc[MT][4], sh[MT][4], lo[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[mt][r] = STATS ? A.tr[16 * mt + c4 + r] : 0.f;
            sh[mt][r] = STATS ? A.tr[Cup + 16 * mt + c4 + r] : 0.f;
            lo[mt][r] = STATS ? A.tr[2 * Cup + 16 * mt + c4 + r] : 0.f;
        }
    const unsigned npi = (unsigned)A.N * h * w, npo = (unsigned)A.N * H * W;
    const rsrc_t rx = mk_rsrc(STATS ? A.x : A.g, STATS ? npi * Cup * 4 : 16), rg = mk_rsrc(A.g, npo * Cout * 4), wd = mk_rsrc(A.dx, npi * Cup * 4);
    const unsigned px_ = Cup * 4, pg = Cout * 4;
    const f32x4* af = reinterpret_cast<const f32x4*>(s_af) + lane;
    float st1[MT][4], st2[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) st1[mt][r] = st2[mt][r] = 0.f;

    const Rs32Sched sched(A.jb.njobs, wave);
    for (int job = sched.first; job < sched.end; job += sched.step) {
        const int cg = job % A.jb.ncg, t = job / A.jb.ncg, rbk = t % A.jb.nrb, img = t / A.jb.nrb;
        const int y0 = rbk * A.jb.rb, y1 = (y0 + A.jb.rb < h) ? y0 + A.jb.rb : h;
        const int j = cg * 16 + n;
        const bool jok = j < w;
        f32x4 xv[STATS ? MT : 1], gv[NT], xn[STATS ? MT : 1], gn[NT];
        auto issue = [&](int i) {
            const bool live = i < y1;
            const unsigned ipix = (unsigned)((img * h + i) * w + j);
            if constexpr (STATS) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xn[mt] = bld16(rx, (live && jok) ? (int)(ipix * px_ + (unsigned)(16 * mt + c4) * 4u) : -1);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int pos = nt * PPS + ppos;
                const int kh = pos / 3, kw = pos - 3 * kh;
                const int oy = 2 * i + kh, ox = 2 * j + kw;
                const bool ok = live && jok && pos < 9 && oy < H && ox < W && oq < Cout;
                gn[nt] = bld16(rg, ok ? (int)((unsigned)((img * H + oy) * W + ox) * pg + (unsigned)oq * 4u) : -1);
            }
        };
        issue(y0);
        for (int i = y0; i < y1; ++i) {
            if constexpr (STATS) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xv[mt] = xn[mt];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) gv[nt] = gn[nt];
            issue(i + 1);
            int oz = 0;
            asm volatile("" : "+v"(oz));  // (opaque zero: the weight fragments stay LDS reads inside the loop)
            const f32x4* afl = af + oz;
            const unsigned ipix = (unsigned)((img * h + i) * w + j);
            const float actf = jok ? 1.f : 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const f32x4 a4 = afl[(mt * NT + nt) * 64];
#pragma unroll
                    for (int r = 0; r < 4; ++r) d = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[r], gv[nt][r], d, 0, 0, 0);
                }
                bst16(wd, (jok && 16 * mt + c4 < Cup) ? (int)(ipix * px_ + (unsigned)(16 * mt + c4) * 4u) : -1, d);
                if constexpr (STATS) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float xt = fmaxf(fmaf(xv[mt][r], sc[mt][r], sh[mt][r]), lo[mt][r]);
                        const float gh = xt > 0.f ? actf * d[r] : 0.f;
                        st1[mt][r] += gh;
                        st2[mt][r] = fmaf(gh, xt, st2[mt][r]);
                    }
                }
            }
        }
    }
    if constexpr (STATS) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v1 = quad16_sum(st1[mt][r]), v2 = quad16_sum(st2[mt][r]);
                if (n == 0) {
                    s_red[wave][0][16 * mt + c4 + r] = v1;
                    s_red[wave][1][16 * mt + c4 + r] = v2;
                }
            }
        __syncthreads();
        if (A.bl.raw) {
            for (int e = tid; e < 2 * Cup; e += 256) {
                const int which = e / Cup, c = e - which * Cup;
                bwd_last_add(A.bl, Cup, c, which, (s_red[0][which][c] + s_red[1][which][c]) + (s_red[2][which][c] + s_red[3][which][c]));
            }
            bwd_last_finish(A.bl, Cup, A.tr, nullptr, tid, 256, &s_flag);
        }
    }
}

namespace {
static inline int rs32_grid(int njobs, int wg_per_cu) {
    static const int bpc_env = env_int("OCRS_RS32_BPC", 0);
    if (bpc_env > 0) wg_per_cu = bpc_env;
    long g = (long)kNumCU * wg_per_cu;
    const long need = (njobs + 3) / 4;
    if (need < g) g = need;
    if (g >= 8) g &= ~7L;
    return (int)(g < 1 ? 1 : g);
}

// zeroed fp64 scratch of the backward's last-workgroup finalisation (BwdLast): the deferral window's (ocrs_bwd_defer_begin) when one is open, else a
// per-device buffer of this file (allocated and zeroed at first use; every launch leaves its share zeroed)
static double* rs32_last_scratch(int ndoubles, hipStream_t st) {
    if (double* p = bwd_defer_scratch(ndoubles)) return p;
    static double* g_own[16] = {};
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 16) return nullptr;
    constexpr int CAP = BWD_LAST_SLOTS * 2 * 32 + 2;
    if (ndoubles > CAP) return nullptr;
    if (!g_own[d]) {
        double* p = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&p), CAP * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        if (hipMemsetAsync(p, 0, CAP * sizeof(double), st) != hipSuccess) return nullptr;
        g_own[d] = p;
    }
    return g_own[d];
}
}  // namespace

extern "C" {
// ---- ConvTranspose weight / bias gradient launcher for ocrs_convt_bwd_parts (det_bwd.hip): fp32, Cup in {16, 32}, Cout in {8, 16, 32}
long det_rs32_ctw_supported(int Cup, int Cout, int dtype) {
    static const int on = env_int("OCRS_RS32", 1), onc = env_int("OCRS_RS32_CTW", 1);
    return (on && onc && dtype == 0 && (Cup == 16 || Cup == 32) && (Cout == 8 || Cout == 16 || Cout == 32)) ? 1 : 0;
}
}  // extern "C"
static int rs32_ctw_rb() {
    static const int rb_env = env_int("OCRS_RS32_CTW_RB", 32);
    return rb_env > 0 ? rb_env : 32;
}
extern "C" {
long det_rs32_ctw_ws_floats(int Cup, int Cout, int N, int h, int w, int dtype) {
    if (!det_rs32_ctw_supported(Cup, Cout, dtype)) return 0;
    const int CE = Cout >= 16 ? 16 : 8, nl = Cout / CE;
    return (long)rs32_grid(1 << 30, 2) * 4 * (Cup * 9 * CE + CE) * nl;  // (the grid cap: independent of how the rows are cut into jobs)
}
int det_rs32_ctw_launch(const float* x, const float* tr, const float* g, float* dW, float* dbias, float* ws, int Cup, int Cout, int N, int h, int w, int H,
                        int W, hipStream_t st) {
    OCRS_CHECK_ARG(det_rs32_ctw_supported(Cup, Cout, 0) && x && tr && g && dW && dbias && ws);
    OCRS_CHECK_ARG((long)N * H * W * Cout * 4 < (1L << 32) && (long)N * h * w * Cup * 4 < (1L << 32) && H <= 2 * h + 1 && W <= 2 * w + 1);
    const int CE = Cout >= 16 ? 16 : 8, nl = Cout / CE, MT = Cup / 16, NT = (9 * CE + 15) / 16;
    const Rs32Jobs jb = rs32_jobs(N, h, w, 16, rs32_ctw_rb());
    const int grid = rs32_grid(jb.njobs, 2);
    const int ne = Cup * 9 * CE + CE;
    size_t smem = (size_t)4 * 16 * (MT + NT) * RS32_TP * sizeof(float);
    if (smem < (4096 + 64) * sizeof(float)) smem = (4096 + 64) * sizeof(float);
    for (int l = 0; l < nl; ++l) {
        Rs32CW a{x, tr, g, ws + (long)l * grid * 4 * ne, Cup, Cout, l * CE, N, h, w, H, W, jb};
#define CTW32_CASE(MT_, CE_)                                                                                    \
    if (MT == MT_ && CE == CE_) {                                                                              \
        static DevOnce once;                                                                                   \
        if (once.need()) {                                                                                     \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rs32_ctw<MT_, CE_>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess) \
                return OCRS_ERR_HIP;                                                                           \
            once.done();                                                                                       \
        }                                                                                                      \
        OCRS_LAUNCH_T((k_rs32_ctw<MT_, CE_>), dim3(grid), dim3(256), smem, st, a);                             \
    }
        CTW32_CASE(1, 8) CTW32_CASE(1, 16) CTW32_CASE(2, 8) CTW32_CASE(2, 16)
#undef CTW32_CASE
        OCRS_LAUNCH_CHECK();
        // dW[c][o0 + oo][pos] += column (c * CE + oo) * 9 + pos of the partials: rows of 9 CE elements at a pitch of 9 Cout; db[o0 + oo] behind them
        bwd_reduce_or_defer(a.ws, grid * 4, ne, dW + (long)l * CE * 9, Cup * 9 * CE, 9 * CE, 9 * Cout, dbias + l * CE, CE, st);
        OCRS_LAUNCH_CHECK();
    }
    return OCRS_OK;
}

// 1 if ocrs_rs32_fwd runs this block shape: fp32 storage; Cin = Ca + Cb in {8, 16, 32}, a concat split only as 8 | 8 or 16 | 16; Cout in {8, 16, 32}.
long ocrs_rs32_fwd_supported(int Ca, int Cb, int Cout, int dtype) {
    static const int on = env_int("OCRS_RS32", 1);
    if (!on || dtype != 0) return 0;
    const int Cin = Ca + Cb;
    if (!(Cin == 8 || Cin == 16 || Cin == 32) || !(Cout == 8 || Cout == 16 || Cout == 32)) return 0;
    if (Cb && !((Ca == 8 && Cb == 8) || (Ca == 16 && Cb == 16))) return 0;
    return 1;
}

// DepthwiseConv block forward in fp32 (ocrs_models/models.py:11-23; channel concat of models.py:89 folded in as xa | xb), row-streaming form.
//   xa / xb [P][Ca] / [P][Cb] fp32 NHWC; tra / trb their load transforms [3][C]; wdw [Cin][9], wpw [Cout][Cin]: the fp32 masters in the
//   reference layout; z [P][Cout]; gstat [2][Cout] fp64 (sum z | sum z^2) ACCUMULATED (caller zeroes); gamma / pooled (nullable): also write
//   MaxPool2d(2) (models.py:54) of the block output in its pre-BatchNorm form (see ocrs_dwpw_fwd).  counter (nullable): a zeroed device word --
//   the launch's last workgroup finalises the BatchNorm statistics (count .. lo as ocrs_bn_finalize; ocrs_dwpw_fwd_fin's contract).
int ocrs_rs32_fwd(const float* xa, const float* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const float* wpw, float* z,
                  double* gstat, const float* gamma, float* pooled, unsigned* counter, long count, const float* bn_w, const float* bn_b, float eps,
                  float momentum, float* tr, float* saved, float* run_mean, float* run_var, long long* nbt, float lo, int Cout, int N, int H, int W,
                  hipStream_t st) {
    OCRS_CHECK_ARG(ocrs_rs32_fwd_supported(Ca, Cb, Cout, 0) && xa && tra && wdw && wpw && z && gstat && (Cb == 0) == (xb == nullptr) && (Cb == 0 || trb));
    OCRS_CHECK_ARG(N > 0 && H > 0 && W > 0 && (!pooled || gamma) && (!counter || (count > 0 && bn_w && bn_b && tr && saved)));
    const int Cin = Ca + Cb, Cmax = Cin > Cout ? Cin : Cout;
    OCRS_CHECK_ARG((long)N * H * W * Cmax * 4 < (1L << 32));  // 32-bit buffer offsets
    static const int rb_env = env_int("OCRS_RS32_RB", 64);
    const int rb = (rb_env > 1 ? rb_env : 64) & ~1;  // even: a pooling window's two rows belong to one job
    Rs32F a{xa, xb, tra, trb, wdw, wpw, gamma, z, pooled, gstat, Ca, Cb, Cout, N, H, W, rs32_jobs(N, H, W, RS32_COLS, rb),
            FwdFin{counter, count, bn_w, bn_b, eps, momentum, tr, saved, run_mean, run_var, nbt, lo}};
    const int nset = Cin > 16 ? 2 : 1, mt = Cout > 16 ? 2 : 1;
    const int grid = rs32_grid(a.jb.njobs, nset * mt == 1 ? 3 : 2);
#define RS32F_CASE(NS_, MT_, SP_, PL_, P_)                                                                         \
    if (nset == NS_ && mt == MT_ && (Cb != 0) == SP_ && (pooled != nullptr) == PL_) {                             \
        OCRS_LAUNCH_T((k_rs32_fwd<NS_, MT_, SP_, PL_, P_>), dim3(grid), dim3(256), 0, st, a);                     \
        OCRS_LAUNCH_CHECK();                                                                                      \
        return OCRS_OK;                                                                                           \
    }
    RS32F_CASE(1, 1, false, false, 4) RS32F_CASE(1, 1, false, true, 4) RS32F_CASE(1, 1, true, false, 4) RS32F_CASE(1, 1, true, true, 4)
    RS32F_CASE(1, 2, false, false, 4) RS32F_CASE(1, 2, false, true, 4) RS32F_CASE(1, 2, true, false, 4) RS32F_CASE(1, 2, true, true, 4)
    RS32F_CASE(2, 1, false, false, 2) RS32F_CASE(2, 1, false, true, 2) RS32F_CASE(2, 1, true, false, 2) RS32F_CASE(2, 1, true, true, 2)
    RS32F_CASE(2, 2, false, false, 2) RS32F_CASE(2, 2, false, true, 2) RS32F_CASE(2, 2, true, false, 2) RS32F_CASE(2, 2, true, true, 2)
#undef RS32F_CASE
    return OCRS_ERR_ARG;
}

// 1 if ocrs_rs32_bwd runs this block shape: fp32 storage, direct (not max-pooled) gradient source, Cin = Ca + Cb in {8, 16} (concat 8 | 8), Cout in {8, 16}
long ocrs_rs32_bwd_supported(int Ca, int Cb, int Cout, int pooled, int dtype) {
    static const int on = env_int("OCRS_RS32", 1), onb = env_int("OCRS_RS32_BWD", 1), onp = env_int("OCRS_RS32_BWDP", 1);
    if (!on || !onb || dtype != 0 || (pooled && (!onp || Cb))) return 0;
    const int Cin = Ca + Cb;
    static const int onx = env_int("OCRS_RS32_BWDX", 1);
    if (onx && !pooled && ((Ca == 16 && Cb == 16 && Cout == 16) || (Ca == 16 && Cb == 0 && Cout == 32))) return 1;  // level 1: two single-source passes / k_rs32_bwdx
    if (!(Cin == 8 || Cin == 16) || !(Cout == 8 || Cout == 16)) return 0;
    if (Cb && !(Ca == 8 && Cb == 8)) return 0;
    return 1;
}
static int rs32_bwd_rb() {
    static const int rb_env = env_int("OCRS_RS32_RB", 64);
    return (rb_env > 1 ? rb_env : 64) & ~1;
}
long ocrs_rs32_bwd_ws_floats(int Ca, int Cb, int Cout, int N, int H, int W) {
    const int Cin = Ca + Cb;
    (void)N; (void)H; (void)W;
    return (long)rs32_grid(1 << 30, 3) * (Cout * Cin + 9 * Cin) * (Cb == 16 ? 2 : 1);  // (the grid cap; 16 | 16: two passes, each with its own half)
}

// one single-source 16-channel pass of a 16 | 16 block (see ocrs_rs32_bwd): x / tr / gx = that source, wdw / wpw / dwpw / dwdw pre-offset to its channels
static int rs32_bwd_one(const float* x, const float* tr, const float* wdw, const float* wpw, int ldw, const float* g1, const float* g2, const float* z,
                        const float* bn, const double* gsum, const float* gamma, const float* saved, float* dgamma, float* dbeta, float* gx, float* dwpw,
                        float* dwdw, float* ws, const float* saved_p, double* gsum_p, int Cout, int N, int H, int W, hipStream_t st) {
    const int Cin = 16;
    const bool stats = gsum_p != nullptr;
    BwdLast bl{nullptr, nullptr, gsum_p, nullptr, saved_p, nullptr, Cin, 0};
    if (stats) {
        double* p = rs32_last_scratch(BWD_LAST_SLOTS * 2 * Cin + 2, st);
        if (!p) return OCRS_ERR_HIP;
        bl.raw = p;
        bl.counter = reinterpret_cast<unsigned*>(p + BWD_LAST_SLOTS * 2 * Cin);
    }
    Rs32B a{x, nullptr, tr, nullptr, wdw, wpw, g1, g2, z, bn, gx, nullptr, ws, Cin, 0, Cout, N, H, W, rs32_jobs(N, H, W, RS32_COLS, rs32_bwd_rb()),
            BnFin{gsum, gamma, saved, dgamma, dbeta, (long)N * H * W}, bl, ldw, nullptr, nullptr};
    const int grid = rs32_grid(a.jb.njobs, 3);
#define RS32O_CASE(G2_, ST_)                                                                          \
    if ((g2 != nullptr) == G2_ && stats == ST_) {                                                    \
        OCRS_LAUNCH_T((k_rs32_bwd<false, G2_, ST_, 1, false>), dim3(grid), dim3(256), 0, st, a);     \
        OCRS_LAUNCH_CHECK();                                                                         \
    }
    RS32O_CASE(false, false) RS32O_CASE(false, true) RS32O_CASE(true, false) RS32O_CASE(true, true)
#undef RS32O_CASE
    bwd_reduce_or_defer(ws, grid, Cout * Cin + 9 * Cin, dwpw, Cout * Cin, Cin, ldw, dwdw, 9 * Cin, st);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// Backward of one DepthwiseConv block in fp32 as ONE row-streaming pass (replaces ocrs_bn_bwd_finalize + ocrs_pw_bwd + ocrs_dw_bwd; the reference's
// autograd of models.py:11-23): xa | xb with load transforms tra | trb: the block input; wdw [Cin][9], wpw [Cout][Cin]: fp32 masters; g1 (+ g2,
// nullable): dL/d(block output) at full resolution; z, bn: the block's stored pre-BatchNorm output and its load transform [3][Cout]; gsum [2][Cout]
// fp64: the block's COMPLETE BatchNorm-backward sums (sum ghat | sum ghat zhat), gamma, saved [mean | rstd] -> the dz coefficients are derived in
// the prologue, dgamma / dbeta are written; gxa | gxb: dL/dx~; dwpw / dwdw: ACCUMULATED through ws (ocrs_rs32_bwd_ws_floats() floats, alive until the
// second stage ran: ocrs_bwd_defer_flush when a deferral window is open); saved_a / gsum_a, saved_b / gsum_b (nullable): as ocrs_dw_bwd.
int ocrs_rs32_bwd(const float* xa, const float* xb, int Ca, int Cb, const float* tra, const float* trb, const float* wdw, const float* wpw, const float* g1,
                  const float* g2, const float* z, const float* bn, const double* gsum, const float* gamma, const float* saved, float* dgamma, float* dbeta,
                  float* gxa, float* gxb, float* dwpw, float* dwdw, float* ws, const float* saved_a, double* gsum_a, const float* saved_b, double* gsum_b,
                  int pooled, int Cout, int N, int H, int W, hipStream_t st) {
    OCRS_CHECK_ARG(ocrs_rs32_bwd_supported(Ca, Cb, Cout, pooled, 0) && xa && tra && wdw && wpw && g1 && z && bn && gsum && gamma && saved && dgamma && dbeta);
    OCRS_CHECK_ARG(gxa && dwpw && dwdw && ws && (Cb == 0) == (xb == nullptr) && (Cb == 0 || (trb && gxb)) && N > 0 && H > 0 && W > 0);
    OCRS_CHECK_ARG((!gsum_a || saved_a) && (!gsum_b || (saved_b && Cb > 0)));
    const int Cin = Ca + Cb, Cmax = Cin > Cout ? Cin : Cout;
    OCRS_CHECK_ARG((long)N * H * W * Cmax * 4 < (1L << 32));
    if (Ca == 16 && Cb == 16) {
        // 16 | 16 -> Cout (up.1.contract.seq.0): every input-side quantity (dx~, dWdw, the column block of dWpw, the producer's sums) is per input channel,
        // so the block runs as TWO single-source passes of the 16-channel kernel that share only the reads of g and z (a fused two-set form needs 330
        // registers): pass s sees source s, the weight columns [16 s, 16 s + 16) at the concat's row pitch, and its own half of ws
        const long half = ocrs_rs32_bwd_ws_floats(16, 0, Cout, N, H, W);
        int rc = rs32_bwd_one(xa, tra, wdw, wpw, 32, g1, g2, z, bn, gsum, gamma, saved, dgamma, dbeta, gxa, dwpw, dwdw, ws, saved_a, gsum_a, Cout, N, H, W, st);
        if (rc != OCRS_OK) return rc;
        return rs32_bwd_one(xb, trb, wdw + 16 * 9, wpw + 16, 32, g1, g2, z, bn, gsum, gamma, saved, dgamma, dbeta, gxb, dwpw + 16, dwdw + 16 * 9, ws + half, saved_b,
                            gsum_b, Cout, N, H, W, st);
    }
    const bool stats = gsum_a || gsum_b;
    BwdLast bl{nullptr, nullptr, gsum_a, gsum_b, saved_a, saved_b, Ca, 0};
    if (stats) {
        double* p = rs32_last_scratch(BWD_LAST_SLOTS * 2 * Cin + 2, st);
        if (!p) return OCRS_ERR_HIP;
        bl.raw = p;
        bl.counter = reinterpret_cast<unsigned*>(p + BWD_LAST_SLOTS * 2 * Cin);
    }
    static const int dual_on = env_int("OCRS_RS32_DUAL", 1);
    const bool dual = dual_on && !pooled && Cb == 0 && Cin == 8 && Cout == 8;  // two 14-column strips per wave (8-channel tensors fill half the lanes)
    Rs32B a{xa, xb, tra, trb, wdw, wpw, g1, g2, z, bn, gxa, gxb, ws, Ca, Cb, Cout, N, H, W,
            rs32_jobs(N, H, W, pooled ? RS32P_COLS : (dual ? 2 * RS32_COLS : RS32_COLS), rs32_bwd_rb()), BnFin{gsum, gamma, saved, dgamma, dbeta, (long)N * H * W}, bl, Cin, nullptr, nullptr};
    const bool wide = Cin > 16 || Cout > 16;
    const int grid = rs32_grid(a.jb.njobs, (pooled || wide || (dual && g2)) ? 2 : 3);
    if (wide) {
#define RS32X_CASE(NS_, MT_, SP_, G2_, ST_)                                                                                  \
    if (Cin == 16 * NS_ && Cout == 16 * MT_ && (Cb != 0) == SP_ && (g2 != nullptr) == G2_ && stats == ST_) {               \
        OCRS_LAUNCH_T((k_rs32_bwdx<NS_, MT_, SP_, G2_, ST_>), dim3(grid), dim3(256), 0, st, a);                             \
        OCRS_LAUNCH_CHECK();                                                                                                \
    }
        RS32X_CASE(1, 2, false, false, false) RS32X_CASE(1, 2, false, false, true) RS32X_CASE(1, 2, false, true, false) RS32X_CASE(1, 2, false, true, true)
#undef RS32X_CASE
        bwd_reduce_or_defer(ws, grid, Cout * Cin + 9 * Cin, dwpw, Cout * Cin, Cin, Cin, dwdw, 9 * Cin, st);
        OCRS_LAUNCH_CHECK();
        return OCRS_OK;
    }
    if (pooled) {
        OCRS_CHECK_ARG(H >= 2 && W >= 2);
#define RS32P_CASE(G2_, ST_)                                                                    \
    if ((g2 != nullptr) == G2_ && stats == ST_) {                                               \
        OCRS_LAUNCH_T((k_rs32_bwdp<G2_, ST_>), dim3(grid), dim3(256), 0, st, a);                \
        OCRS_LAUNCH_CHECK();                                                                    \
    }
        RS32P_CASE(false, false) RS32P_CASE(false, true) RS32P_CASE(true, false) RS32P_CASE(true, true)
#undef RS32P_CASE
        bwd_reduce_or_defer(ws, grid, Cout * Cin + 9 * Cin, dwpw, Cout * Cin, Cin, Cin, dwdw, 9 * Cin, st);
        OCRS_LAUNCH_CHECK();
        return OCRS_OK;
    }
#define RS32B_CASE(SP_, G2_, ST_)                                                                      \
    if ((Cb != 0) == SP_ && (g2 != nullptr) == G2_ && stats == ST_) {                                 \
        if constexpr (!SP_) {                                                                         \
            if (dual)                                                                                 \
                OCRS_LAUNCH_T((k_rs32_bwd<false, G2_, ST_, 1, true>), dim3(grid), dim3(256), 0, st, a); \
            else                                                                                      \
                OCRS_LAUNCH_T((k_rs32_bwd<false, G2_, ST_, 1, false>), dim3(grid), dim3(256), 0, st, a); \
        } else {                                                                                      \
            OCRS_LAUNCH_T((k_rs32_bwd<SP_, G2_, ST_, 1>), dim3(grid), dim3(256), 0, st, a);           \
        }                                                                                             \
        OCRS_LAUNCH_CHECK();                                                                          \
    }
    RS32B_CASE(false, false, false) RS32B_CASE(false, false, true) RS32B_CASE(false, true, false) RS32B_CASE(false, true, true)
    RS32B_CASE(true, false, false) RS32B_CASE(true, false, true) RS32B_CASE(true, true, false) RS32B_CASE(true, true, true)
#undef RS32B_CASE
    bwd_reduce_or_defer(ws, grid, Cout * Cin + 9 * Cin, dwpw, Cout * Cin, Cin, Cin, dwdw, 9 * Cin, st);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// 1 if ocrs_rs32_convt_fwd runs this shape: fp32, (Cup, Cout) in {(16, 8), (32, 16)} ((32, 32) is built and tested but slower than k_convt_fwd: OCRS_RS32_CTF_3232=1)
long ocrs_rs32_convt_fwd_supported(int Cup, int Cout, int dtype) {
    static const int on = env_int("OCRS_RS32", 1), onf = env_int("OCRS_RS32_CTF", 1);
    static const int on32 = env_int("OCRS_RS32_CTF_3232", 0);  // (32, 32) at level 2: 256 MFMAs per strip row -- measured 296 us vs 253 us on k_convt_fwd: off
    return (on && onf && dtype == 0 && ((Cup == 16 && Cout == 8) || (Cup == 32 && (Cout == 16 || (Cout == 32 && on32))))) ? 1 : 0;
}
// ConvTranspose2d(Cup, Cout, kernel_size=3, stride=2) + bias, cropped to the skip tensor's H x W (models.py:76-78, 87), fp32, row-streaming form.
//   x [N][h][w][Cup] with load transform tr [3][Cup]; wt: the fp32 master weight [Cup][Cout][3][3] (reference layout); out [N][H][W][Cout], H <= 2 h + 1.
int ocrs_rs32_convt_fwd(const float* x, const float* tr, const float* wt, const float* bias, float* out, int Cup, int Cout, int N, int h, int w, int H, int W,
                        hipStream_t st) {
    OCRS_CHECK_ARG(((Cup == 16 && Cout == 8) || (Cup == 32 && (Cout == 16 || Cout == 32))) && x && tr && wt && bias && out && N > 0 && h > 0 && w > 0 && H <= 2 * h + 1 && W <= 2 * w + 1);
    OCRS_CHECK_ARG((long)N * H * W * Cout * 4 < (1L << 32) && (long)N * h * w * Cup * 4 < (1L << 32));
    static const int rb_env = env_int("OCRS_RS32_CTF_RB", 32);
    Rs32CF a{x, tr, wt, bias, out, Cup, Cout, N, h, w, H, W, rs32_jobs(N, h + 1, w + 1, 15, rb_env > 0 ? rb_env : 32)};
    const int nset = Cup / 16, mt = 4 * Cout / 16;
    const int grid = rs32_grid(a.jb.njobs, 2);
    const size_t smem = (size_t)mt * 4 * nset * 64 * 4 * sizeof(float);
#define CTF32_CASE(NS_, MT_)                                                                                   \
    if (nset == NS_ && mt == MT_) {                                                                           \
        static DevOnce once;                                                                                  \
        if (once.need()) {                                                                                    \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rs32_ctf<NS_, MT_>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess) \
                return OCRS_ERR_HIP;                                                                          \
            once.done();                                                                                      \
        }                                                                                                     \
        OCRS_LAUNCH_T((k_rs32_ctf<NS_, MT_>), dim3(grid), dim3(256), smem, st, a);                            \
        OCRS_LAUNCH_CHECK();                                                                                  \
        return OCRS_OK;                                                                                       \
    }
    CTF32_CASE(1, 2) CTF32_CASE(2, 4) CTF32_CASE(2, 8)
#undef CTF32_CASE
    return OCRS_ERR_ARG;
}

// 1 if ocrs_rs32_convt_dgrad runs this shape: fp32, (Cup, Cout) in {(16, 8), (32, 16)}
long ocrs_rs32_convt_dgrad_supported(int Cup, int Cout, int dtype) {
    static const int on = env_int("OCRS_RS32", 1), ond = env_int("OCRS_RS32_CTD", 1);
    return (on && ond && dtype == 0 && ((Cup == 16 && Cout == 8) || (Cup == 32 && Cout == 16))) ? 1 : 0;
}
// ConvTranspose2d input gradient in fp32, row-streaming form (the dx half of ocrs_convt_bwd_parts; autograd of models.py:76-78 as train_detection.py:96
// runs it): g [N][H][W][Cout] the gradient w.r.t. the (cropped) output, wt the fp32 MASTER weight [Cup][Cout][3][3], dx [N][h][w][Cup] = dL/dx~.
// x / tr / saved / gsum (all nullable together): x is the raw output of a block consumed ONLY by this ConvTranspose -- its BatchNorm-backward sums
// [sum ghat | sum ghat zhat] ([2][Cup] fp64, ACCUMULATED) come from this pass instead of ocrs_bn_bwd_reduce.
int ocrs_rs32_convt_dgrad(const float* g, const float* wt, float* dx, const float* x, const float* tr, const float* saved, double* gsum, int Cup, int Cout, int N,
                          int h, int w, int H, int W, hipStream_t st) {
    OCRS_CHECK_ARG(ocrs_rs32_convt_dgrad_supported(Cup, Cout, 0) && g && wt && dx && N > 0 && h > 0 && w > 0 && H <= 2 * h + 1 && W <= 2 * w + 1);
    OCRS_CHECK_ARG((!gsum || (x && tr && saved)) && (long)N * H * W * Cout * 4 < (1L << 32) && (long)N * h * w * Cup * 4 < (1L << 32));
    const bool stats = gsum != nullptr;
    BwdLast bl{nullptr, nullptr, gsum, nullptr, saved, nullptr, Cup, 0};
    if (stats) {
        double* p = rs32_last_scratch(BWD_LAST_SLOTS * 2 * Cup + 2, st);
        if (!p) return OCRS_ERR_HIP;
        bl.raw = p;
        bl.counter = reinterpret_cast<unsigned*>(p + BWD_LAST_SLOTS * 2 * Cup);
    }
    static const int rb_env = env_int("OCRS_RS32_CTD_RB", 32);
    Rs32CD a{g, wt, x, tr, dx, Cup, Cout, N, h, w, H, W, rs32_jobs(N, h, w, 16, rb_env > 0 ? rb_env : 32), bl};
    const int grid = rs32_grid(a.jb.njobs, 2);
#define CTD32_CASE(MT_, CE_, ST_)                                                               \
    if (Cup == 16 * MT_ && Cout == CE_ && stats == ST_) {                                       \
        OCRS_LAUNCH_T((k_rs32_ctd<MT_, CE_, ST_>), dim3(grid), dim3(256), 0, st, a);            \
        OCRS_LAUNCH_CHECK();                                                                    \
        return OCRS_OK;                                                                         \
    }
    CTD32_CASE(1, 8, false) CTD32_CASE(1, 8, true) CTD32_CASE(2, 16, false) CTD32_CASE(2, 16, true)
#undef CTD32_CASE
    return OCRS_ERR_ARG;
}

// 1 if ocrs_rs32_bwd_head runs the block in front of out_conv: fp32, 8 -> 8, single source
long ocrs_rs32_bwd_head_supported(int Ca, int Cb, int Cout, int dtype) {
    static const int onh = env_int("OCRS_RS32_HEAD", 1);
    return (onh && Cb == 0 && Ca == 8 && Cout == 8 && ocrs_rs32_bwd_supported(Ca, Cb, Cout, 0, dtype)) ? 1 : 0;
}
// ocrs_rs32_bwd for the block in front of out_conv (models.py:125-129): its output gradient is formed on the fly, g[p][c] = gl[p] * whead[c], from out_conv's
// dL/dlogit gl [P] fp32 (ocrs_head_bwd_gl / ocrs_head_bwd_loss write 4 instead of 32 bytes per pixel) -- the same fp32 product ocrs_head_bwd stores, so every
// output is bit-identical to ocrs_head_bwd + ocrs_rs32_bwd.  Other arguments as ocrs_rs32_bwd (single source, one gradient).
int ocrs_rs32_bwd_head(const float* xa, int Ca, const float* tra, const float* wdw, const float* wpw, const float* gl, const float* whead, const float* z,
                       const float* bn, const double* gsum, const float* gamma, const float* saved, float* dgamma, float* dbeta, float* gxa, float* dwpw,
                       float* dwdw, float* ws, const float* saved_a, double* gsum_a, int Cout, int N, int H, int W, hipStream_t st) {
    OCRS_CHECK_ARG(ocrs_rs32_bwd_head_supported(Ca, 0, Cout, 0) && xa && tra && wdw && wpw && gl && whead && z && bn && gsum && gamma && saved && dgamma && dbeta);
    OCRS_CHECK_ARG(gxa && dwpw && dwdw && ws && N > 0 && H > 0 && W > 0 && (!gsum_a || saved_a) && (long)N * H * W * 8 * 4 < (1L << 32));
    const int Cin = Ca;
    const bool stats = gsum_a != nullptr;
    BwdLast bl{nullptr, nullptr, gsum_a, nullptr, saved_a, nullptr, Ca, 0};
    if (stats) {
        double* p = rs32_last_scratch(BWD_LAST_SLOTS * 2 * Cin + 2, st);
        if (!p) return OCRS_ERR_HIP;
        bl.raw = p;
        bl.counter = reinterpret_cast<unsigned*>(p + BWD_LAST_SLOTS * 2 * Cin);
    }
    static const int dual_on = env_int("OCRS_RS32_DUAL", 1);
    const bool dual = dual_on != 0;
    Rs32B a{xa, nullptr, tra, nullptr, wdw, wpw, nullptr, nullptr, z, bn, gxa, nullptr, ws, Ca, 0, Cout, N, H, W,
            rs32_jobs(N, H, W, dual ? 2 * RS32_COLS : RS32_COLS, rs32_bwd_rb()), BnFin{gsum, gamma, saved, dgamma, dbeta, (long)N * H * W}, bl, Cin, gl, whead};
    const int grid = rs32_grid(a.jb.njobs, 3);
    if (dual) {
        if (stats) OCRS_LAUNCH_T((k_rs32_bwd<false, false, true, 1, true, true>), dim3(grid), dim3(256), 0, st, a);
        else OCRS_LAUNCH_T((k_rs32_bwd<false, false, false, 1, true, true>), dim3(grid), dim3(256), 0, st, a);
    } else {
        if (stats) OCRS_LAUNCH_T((k_rs32_bwd<false, false, true, 1, false, true>), dim3(grid), dim3(256), 0, st, a);
        else OCRS_LAUNCH_T((k_rs32_bwd<false, false, false, 1, false, true>), dim3(grid), dim3(256), 0, st, a);
    }
    OCRS_LAUNCH_CHECK();
    bwd_reduce_or_defer(ws, grid, Cout * Cin + 9 * Cin, dwpw, Cout * Cin, Cin, Cin, dwdw, 9 * Cin, st);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

}  // extern "C"
