// Pipelined split-bf16 GEMM for the fp32 GRU projections of the CRNN (round 4).  Reference semantics: the `nn.GRU(128, 256, num_layers=2,
// bidirectional=True)` input projections x W_ih^T + b_ih and their input gradients dgi W_ih (ocrs_models/models.py:245, 264-266 -- fp32 under
// autocast), computed as  a b ~ ah bh + ah bl + al bh  exactly like k_gemm_x3 (rec_conv.hip): same products, same accumulation order, the
// results are bit-identical.
//
// Why a second kernel: k_gemm_x3 is latency-bound, not matrix-, LDS- or VALU-bound.  Floor builds of round 4 (tools/experiments/
// r4_gemm_time.py): weights pre-split and out of LDS -> no change; no split arithmetic -> -8 %; a third of the MFMAs AND no stores AND the
// row operand L2-resident -> still 112 of 201 us (K = 512, M = 1536): a 128 x 128 x 32 chunk costs ~3500 cycles whatever is in it, because
// the chunk's loads are issued one chunk (0.3 us of matrix work) ahead and consumed behind a barrier, with two workgroups per CU to hide it.
//
// Here (third form; the first two are in DESIGN.md's round-4 section): one 768-thread workgroup per CU, persistent over (row block, column
// block) tiles of 256 (or 128) rows x 128 columns, with ROLES:
//   * 4 PRODUCER waves issue every LDS-DMA of the workgroup (global_load_lds_dwordx4: no VGPR staging, no ds_write, no VALU) into a ring of
//     THREE stages, two 32-deep K chunks ahead of the matrix work, across tile boundaries.  A wave gets one 1 KB DMA instruction out every
//     ~90 cycles (measured: 24 instructions = 2200 cycles), a chunk needs 48 -- issued by the MFMA waves themselves (first form) every wave
//     lost ~600 cycles per chunk in front of its MFMAs, and every CU-wide phase (issue, LDS reads, MFMA) ran in lock step;
//   * 8 CONSUMER waves (2 along the columns x 4 along the rows, two per SIMD -- one wave per SIMD cannot issue more than ~60 % of the MFMA
//     peak, tools/probes/mfma_issue_probe.hip) only read LDS, split and multiply.  They have NO vector-memory loads, so nothing they do
//     waits for their output stores (in the first form the counted `s_waitcnt vmcnt` in front of every chunk also waited for the previous
//     tile's stores: 20 us of 146).
//   One s_barrier per chunk: producers arrive after "chunk g + 1 has landed" (their own counted vmcnt), consumers after "done reading chunk
//   g": the stage of chunk g is then free for chunk g + 3.
//   row operand X: raw fp32 rows, 8 rows x 128 B per DMA instruction (whole lines), XOR-swizzled by the source address each lane picks so
//     that the fragment reads (two ds_read_b128 per lane = 8 consecutive k) are bank-conflict-free for ds_read_b128's lane groups; the
//     hi / lo split happens in registers after the read (6 VALU per pair of values: v_cvt_pk_bf16_f32 + shift / and + 2 subs + cvt).
//   weight operand: pre-split, pre-packed MFMA A fragments (ocrs_pack_frags mode 2: [kc][mt][hi, lo][lane]), 16 KB per chunk and column
//     block, contiguous -> DMA as is, read back lane-linear.  (Second form: the A fragments straight from L2 into registers instead --
//     half the LDS traffic but 96 KB instead of 48 KB per chunk through the CU's 64 B / clk vector-memory path: slower.)
// XCD-aware tile order: workgroup b runs on XCD b % 8; XCD x owns the row blocks x, x + 8, ... and walks (row block, column block) with the
// column block fastest, so the column blocks that share a row block's X rows meet in one L2.
// Measured (T N = 25856 rows; k_gemm_x3 -> this kernel): K 128 -> M 1536: 71 -> 52 us; 512 -> 1536: 183 -> 137; 1536 -> 512: 195 -> 153;
// 1536 -> 128: 67 -> 40.  What is left: the consumers' LDS traffic (16 KB of fragment reads per 48 MFMAs: a chunk takes ~2400 cycles of
// which 1536 are MFMA issue) and the tile quantisation of M = 512 (404 tiles on 256 CUs).
#include "det_common.h"
#include <type_traits>

#ifndef G_NPROD
#define G_NPROD 4  // producer waves (all LDS-DMA of the workgroup): one wave issues a 1 KB DMA instruction every ~90 cycles, a chunk needs 24 - 48
#endif
#ifndef G_DBG
#define G_DBG 0  // 1: per-phase cycle counters of every wave of workgroup 0 (measurement builds; read with ocrs_gemm_x3p_dbg)
#endif
#if G_DBG
__device__ long long g_gdbg[8][8];
#define G_T() __builtin_readcyclecounter()
#endif
namespace {
// LDS-DMA of 16 bytes per lane from (wave-uniform base + per-lane byte offset): lane i's bytes land at LDS byte lds_dst + 16 i
__device__ __forceinline__ void g_dma16(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void g_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void g_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// swizzle of the 16-byte slots of a 128-byte fp32 row (row = r mod 16 within a 16-row MFMA tile): conflict-free for the four 16-lane groups
// ds_read_b128 is serviced in ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32), checked by brute force (MI355X_MICROARCH.md, LDS table)
__device__ __forceinline__ int g_swz(int r) { return ((r >> 1) & 7) ^ ((((r + 4) >> 3) & 1) << 1); }
}  // namespace

template <int NTW /* 16-row tiles per consumer wave: the workgroup's tile is 64 NTW rows x 128 columns */>
__global__ __launch_bounds__(512 + 64 * G_NPROD) void k_gemm_x3p(const float* __restrict__ X, int ldx, const uint4* __restrict__ Wpk, const float* __restrict__ bias,
                                                  float* __restrict__ out, int ldo, int K, int M, long P, int nrb, int ncb) {
    constexpr int BP = 64 * NTW, XS = BP * 128, STAGE = XS + 16384, NS = 3;
    constexpr int NXI = BP / 8, NPI = (NXI + 16) / G_NPROD;  // DMA instructions per chunk: NXI of X + 16 of W, NPI per producer wave
    extern __shared__ __attribute__((aligned(1024))) char smem[];  // [NS][STAGE] + bias [M]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int MT = M >> 4, nkc = K >> 5;
    // ---- this workgroup's tiles: XCD x = b % 8 owns row blocks x, x + 8, ...; tile q of the XCD = (row block 8 (q / ncb) + x, column block q % ncb)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = (gridDim.x + 7 - xcd) >> 3;
    const int nq = ((nrb - xcd + 7) >> 3) * ncb;  // tiles of this XCD
    if (slot >= nq) return;
    const int ntile = (nq - slot + nslot - 1) / nslot;
    const int G = ntile * nkc;  // chunks of this workgroup; chunk g lives in stage g % 3
    float* sbias = reinterpret_cast<float*>(smem + NS * STAGE);
    if (bias)
        for (int i = tid; i < M; i += 512 + 64 * G_NPROD) sbias[i] = bias[i];
    auto tile_of = [&](int ti, int& rb, int& cb) {
        const int q = slot + ti * nslot;
        rb = 8 * (q / ncb) + xcd;
        cb = q % ncb;
    };
    if (wave >= 8) {
        // ================= producer waves (G_NPROD): all LDS-DMA of the workgroup, two chunks ahead of the MFMAs =================
        // X: instruction u covers rows 8 u .. 8 u + 7 of the tile: lane -> (row 8 u + lane / 8, slot lane % 8) fetches k-slot (lane % 8) ^ swz(row % 16);
        // W: instruction f = fragment (m-tile f / 2, plane f % 2) of the column block, lane-linear.  Producer pw issues the instructions of its parity.
        const int pw = wave - 8;
        const unsigned lds0 = (unsigned)(uintptr_t)smem;
        const int r8 = lane >> 3;
        const unsigned ks0 = (unsigned)(((lane & 7) ^ g_swz(r8)) * 16), ks1 = (unsigned)(((lane & 7) ^ g_swz(8 + r8)) * 16);  // rows 16 t + r8 / 16 t + 8 + r8
        const unsigned ld4 = (unsigned)ldx * 4u;
        int pti = 0, pkc = 0, pst = 0, prb = 0, pcb = 0;
        tile_of(0, prb, pcb);
        auto issue = [&]() {  // chunk (pti, pkc) -> stage pst; then advance the cursor
            const unsigned sb = lds0 + (unsigned)pst * STAGE;
            const char* xb = reinterpret_cast<const char*>(X) + (long)pkc * 128;
            const long p0 = (long)prb * BP + r8;
#pragma unroll
            for (int i = 0; i < NXI / G_NPROD; ++i) {
                const int u = G_NPROD * i + pw;  // (u & 1 == pw & 1: the row parity within a 16-row tile is this producer's constant)
                long p = p0 + u * 8;
                p = p < P ? p : P - 1;  // (rows past the end: any valid row -- their outputs are not stored)
                g_dma16(xb, (unsigned)p * ld4 + ((pw & 1) ? ks1 : ks0), __builtin_amdgcn_readfirstlane(sb + u * 1024));
            }
            const char* wb = reinterpret_cast<const char*>(Wpk) + ((long)pkc * MT + pcb * 8) * 2048;
#pragma unroll
            for (int i = 0; i < 16 / G_NPROD; ++i) {
                const int f = G_NPROD * i + pw;
                g_dma16(wb, (unsigned)(f * 1024 + lane * 16), __builtin_amdgcn_readfirstlane(sb + XS + f * 1024));
            }
            pst = pst == NS - 1 ? 0 : pst + 1;
            if (++pkc == nkc) {
                pkc = 0;
                ++pti;
                tile_of(pti, prb, pcb);
            }
        };
        issue();
        if (G > 1) {
            issue();
            g_wait_vm<NPI>();
        } else {
            g_wait_vm<0>();
        }
#if G_DBG
        long long pb = 0, pi = 0, pwt = 0;
#endif
        for (int g = 0; g < G; ++g) {
#if G_DBG
            const long long q0 = G_T();
            g_barrier();
            const long long q1 = G_T();
            if (g + 2 < G) issue();
            const long long q2 = G_T();
            g_wait_vm<0>();
            pb += q1 - q0; pi += q2 - q1; pwt += G_T() - q2;
            if (g == G - 1 && blockIdx.x == 0 && lane == 0 && wave == 8) { g_gdbg[0][6] = pb; g_gdbg[1][6] = pi; g_gdbg[2][6] = pwt; }
            continue;
#endif
            g_barrier();  // chunk g has landed (waited for below / above); the consumers are done with chunk g - 1: its stage takes chunk g + 2
            if (g + 2 < G) {
                issue();
                g_wait_vm<NPI>();  // chunk g + 1 has landed (only the NPI pieces of chunk g + 2 may be in flight)
            } else {
                g_wait_vm<0>();
            }
        }
        return;
    }
    // ================= consumer waves (8 = 2 along the columns x 4 along the rows): LDS -> split -> MFMA, no vector-memory loads =================
    const int wm = wave & 1, wn = wave >> 1;
    const int n16 = lane & 15, kq = lane >> 4;
    // fragment read offsets: row 16 (wn NTW + j) + n16 of the tile, 16-byte slots (2 kq, 2 kq + 1) ^ swz
    const int sw = g_swz(n16);
    const unsigned xo0 = (unsigned)((wn * NTW * 16 + n16) * 128 + (((2 * kq) ^ sw) << 4)), xo1 = (unsigned)((wn * NTW * 16 + n16) * 128 + (((2 * kq + 1) ^ sw) << 4));
    const unsigned wo = (unsigned)(XS + wm * 8 * 1024 + lane * 16);
    f32x4 acc[4][NTW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int cst = 0;
#if G_DBG
    long long d_bar = 0, d_epi = 0;
    const long long d_t0 = G_T();
#endif
    for (int ti = 0; ti < ntile; ++ti) {
        for (int kc = 0; kc < nkc; ++kc) {
#if G_DBG
            const long long t0 = G_T();
#endif
            g_barrier();
#if G_DBG
            d_bar += G_T() - t0;
#endif
            const char* sb = smem + cst * STAGE;
            cst = cst == NS - 1 ? 0 : cst + 1;
            bf16x8 wh[4], wl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wh[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sb + wo + i * 2048));
                wl[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sb + wo + i * 2048 + 1024));
            }
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const float4 a = *reinterpret_cast<const float4*>(sb + xo0 + j * 2048), b = *reinterpret_cast<const float4*>(sb + xo1 + j * 2048);
                const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                unsigned hp[4], lp[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hp[e] = pack2bf(x[2 * e], x[2 * e + 1]);
                    const float l0 = x[2 * e] - __uint_as_float(hp[e] << 16), l1 = x[2 * e + 1] - __uint_as_float(hp[e] & 0xffff0000u);
                    lp[e] = pack2bf(l0, l1);
                }
                const bf16x8 xh = __builtin_bit_cast(bf16x8, make_uint4(hp[0], hp[1], hp[2], hp[3]));
                const bf16x8 xl = __builtin_bit_cast(bf16x8, make_uint4(lp[0], lp[1], lp[2], lp[3]));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], xh, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], xl, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[i], xh, acc[i][j], 0, 0, 0);
                }
            }
        }
        // ---- tile done: + bias, 16 bytes per lane (4 consecutive output columns of one row); nothing here waits for the stores
#if G_DBG
        const long long e0 = G_T();
#endif
        int rb, cb;
        tile_of(ti, rb, cb);
        const long p0 = (long)rb * BP + wn * NTW * 16 + n16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = cb * 128 + (wm * 4 + i) * 16 + kq * 4;
            float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias) bs = *reinterpret_cast<const float4*>(sbias + m);
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const long p = p0 + j * 16;
                const f32x4 v = acc[i][j];
                if (p < P) *reinterpret_cast<float4*>(out + p * ldo + m) = make_float4(v[0] + bs.x, v[1] + bs.y, v[2] + bs.z, v[3] + bs.w);
                acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
#if G_DBG
        d_epi += G_T() - e0;
#endif
    }
#if G_DBG
    if (blockIdx.x == 0 && lane == 0) {
        g_gdbg[wave][0] = G_T() - d_t0;
        g_gdbg[wave][1] = 0;
        g_gdbg[wave][2] = d_bar;
        g_gdbg[wave][3] = 0;
        g_gdbg[wave][4] = d_epi;
        g_gdbg[wave][5] = G;
    }
#endif
}

// true when ocrs_gemm_x3p can take the shape (the caller falls back to ocrs_gemm_x3 otherwise)
static bool gemm_x3p_ok(int ldx, int K, int ldo, int M, long P) {
    return K % 32 == 0 && M % 128 == 0 && M <= 2048 && ldx % 4 == 0 && ldo % 4 == 0 && P > 0 && (long)P * ldx * 4 < (1L << 31);
}

extern "C" {

// out [P][ldo] (first M columns) = X [P][ldx] (first K columns) * W^T (+ bias [M]) as split-bf16 products, weights pre-split / pre-packed:
//   wpk = ocrs_pack_frags(mode 2, dtype 1) of W as A[m][k] (2 * ocrs_pack_frags_bytes(K, M, 1) bytes).  K % 32 == 0, M % 128 == 0, M <= 2048,
//   P * ldx * 4 < 2^31 (returns 1 otherwise: ocrs_gemm_x3p_supported() tells beforehand).  Bit-identical to ocrs_gemm_x3.
long ocrs_gemm_x3p_supported(int ldx, int K, int ldo, int M, long P) { return gemm_x3p_ok(ldx, K, ldo, M, P) ? 1 : 0; }
int ocrs_gemm_x3p_tiles(const float* X, int ldx, int K, const void* wpk, const float* bias, float* out, int ldo, int M, long P, int ntw, hipStream_t st);
int ocrs_gemm_x3p(const float* X, int ldx, int K, const void* wpk, const float* bias, float* out, int ldo, int M, long P, hipStream_t st) {
    return ocrs_gemm_x3p_tiles(X, ldx, K, wpk, bias, out, ldo, M, P, 0, st);
}
// ntw: rows per workgroup tile / 64 -- 4 (256 rows), 2 (128 rows) or 0 (automatic: 256-row tiles unless that leaves CUs without a tile)
int ocrs_gemm_x3p_tiles(const float* X, int ldx, int K, const void* wpk, const float* bias, float* out, int ldo, int M, long P, int ntw_arg, hipStream_t st) {
    OCRS_CHECK_ARG(X && wpk && out && ldx >= K && ldo >= M && gemm_x3p_ok(ldx, K, ldo, M, P));
    OCRS_CHECK_ARG(((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(wpk) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0);
    const int ncb = M / 128;
    // 256-row tiles unless that leaves CUs without a tile (M = 128: one column block)
    OCRS_CHECK_ARG(ntw_arg == 0 || ntw_arg == 2 || ntw_arg == 4);
    static const int force = env_int("OCRS_GEMM_X3P_NTW", 0);
    const long t4 = ((P + 255) / 256) * ncb;
    const int ntw = ntw_arg ? ntw_arg : (force == 2 || force == 4) ? force : (t4 >= 2 * kNumCU ? 4 : 2);
    if (ntw == 4) {
        const int nrb = (int)((P + 255) / 256);
        const size_t lds = 3 * (256 * 128 + 16384) + (size_t)M * 4;
        static DevOnce attr4;  // (per device and thread-safe, like r3_launch / r4_launch: ADVICE r04)
        if (attr4.need()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_x3p<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return OCRS_ERR_HIP;
            attr4.done();
        }
        hipLaunchKernelGGL(k_gemm_x3p<4>, dim3(kNumCU), dim3(512 + 64 * G_NPROD), lds, st, X, ldx, reinterpret_cast<const uint4*>(wpk), bias, out, ldo, K, M, P, nrb, ncb);
    } else {
        const int nrb = (int)((P + 127) / 128);
        const size_t lds = 3 * (128 * 128 + 16384) + (size_t)M * 4;
        static DevOnce attr2;
        if (attr2.need()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_x3p<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return OCRS_ERR_HIP;
            attr2.done();
        }
        hipLaunchKernelGGL(k_gemm_x3p<2>, dim3(kNumCU), dim3(512 + 64 * G_NPROD), lds, st, X, ldx, reinterpret_cast<const uint4*>(wpk), bias, out, ldo, K, M, P, nrb, ncb);
    }
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}


#if G_DBG
int ocrs_gemm_x3p_dbg(long long* host64) { return hipMemcpyFromSymbol(host64, HIP_SYMBOL(g_gdbg), sizeof(long long) * 64) == hipSuccess ? 0 : 2; }
#endif

}  // extern "C"
