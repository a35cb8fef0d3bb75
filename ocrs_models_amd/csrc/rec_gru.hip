// Bidirectional GRU recurrence (gfx950), fp32 as the reference forces it (ocrs_models/models.py:245, 264-266).
// PyTorch gate order (r, z, n):  r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = (1 - z) * n + z * h,
// gh = W_hh h + b_hh; the reverse direction scans t = T-1..0 and stores at its own t; h0 = 0.
//
// The input projections gi = W_ih x + b_ih for all time steps and both directions are one big GEMM (ocrs_conv_igemm, fp32 MFMA);
// only the T sequential steps run here, one launch per step (both directions in the same launch, grid.z = 2):
//   block = 16 hidden units x 3 gates (3 MFMA M-tiles) x 16 batch columns, 8 waves, K = 256 on v_mfma_f32_16x16x4_f32 (exact fp32).
// Backward (BPTT) mirrors it: one launch per step computes dh_carry = z*dh + W_hh^T dgh (K = 768) and, in its epilogue, the
// gate derivatives of the NEXT step to process; the weight/input gradients are big GEMMs over all steps afterwards.
#include "common.h"

static constexpr int GH = 256;       // hidden size
static constexpr int G3 = 3 * GH;    // 768

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// gi   [T][N][2*768]  (direction-major: d*768 + gate*256 + j), b_ih already added
// whh  packed W_hh fragments per direction (K=256, M=768): [2][8 chunks][48 mtiles][64][8] fp32
// bhh  [2][768];  out [T][N][512] (d*256 + j);  saved [T][N][2][4][256] = r | z | n | gh_n
// Block = 16 hidden units x 16 batch columns x 3 gates; the K = 256 reduction is SPLIT over the block's 8 waves (32 each, so the
// dependent MFMA chain is 8x shorter; a step is pure latency: 6.9 us with 4 waves, 6.6 with 8) and combined through LDS;
// grid = 16 x ceil(N/16) x 2 directions (512 blocks of 512 threads at N = 256).
static constexpr int FNW = 8, FKW = GH / FNW;  // forward: K = 256 split over 8 waves (32 each), 512 threads; threads 0..255 own the epilogue pairs
static constexpr int SPITCH = FKW + 4;         // LDS pitch of a slab row
__global__ __launch_bounds__(512) void k_gru_step_fwd(const float* __restrict__ gi, const float* __restrict__ whh, const float* __restrict__ bhh,
                                                      float* __restrict__ out, float* __restrict__ saved, int T, int N, int s) {
    __shared__ __attribute__((aligned(16))) float hs[FNW][16 * SPITCH];  // per wave: h_prev[16 batch][FKW k]
    __shared__ float red[FNW][3][16][17];                                // per wave partial gh[gate][hidden][batch]
    const int d = blockIdx.z, jt = blockIdx.x, b0 = blockIdx.y * 16;
    const int t = d == 0 ? s : T - 1 - s;
    const int tp = d == 0 ? t - 1 : t + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool first = s == 0;
    // epilogue operands of this thread's (hidden, batch) pair: independent of the recurrent GEMM, so their loads are issued first
    const int jl = tid & 15, bl = (tid >> 4) & 15;
    const int b = b0 + bl, j = jt * 16 + jl;
    const bool bv = b < N && tid < 256;
    float gi_r = 0.f, gi_z = 0.f, gi_n = 0.f, bh_r = 0.f, bh_z = 0.f, bh_n = 0.f, hp = 0.f;
    if (bv) {
        const float* gir = gi + ((long)t * N + b) * (2 * G3) + d * G3;
        const float* bh = bhh + d * G3;
        gi_r = gir[j];
        gi_z = gir[GH + j];
        gi_n = gir[2 * GH + j];
        bh_r = bh[j];
        bh_z = bh[GH + j];
        bh_n = bh[2 * GH + j];
        if (!first) hp = out[((long)tp * N + b) * 512 + d * GH + j];
    }
    if (!first) {
        // this wave's K slab: h_prev[b0 .. b0+16][FKW*wave .. +FKW]
        for (int it = lane; it < 16 * (FKW / 4); it += 64) {
            const int row = it / (FKW / 4), c4 = (it % (FKW / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b0 + row < N) v = *reinterpret_cast<const float4*>(out + ((long)tp * N + b0 + row) * 512 + d * GH + wave * FKW + c4);
            *reinterpret_cast<float4*>(&hs[wave][row * SPITCH + c4]) = v;
        }
        __syncthreads();
        f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* wd = whh + (long)d * 8 * 48 * 64 * 8;
#pragma unroll
        for (int kc = 0; kc < FKW / 32; ++kc) {
            const Mma<float>::Frag pf = Mma<float>::load_p(&hs[wave][kc * 32], SPITCH, 0, lane, 32);
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const Mma<float>::Frag wf = Mma<float>::load_w(wd, (long)(wave * (FKW / 32) + kc) * 48 + g * 16 + jt, lane);
                acc[g] = Mma<float>::mma<8>(wf, pf, acc[g]);
            }
        }
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][g][(lane >> 4) * 4 + r][lane & 15] = acc[g][r];
        __syncthreads();
    }
    // epilogue: one (hidden, batch) pair per thread
    if (!bv) return;
    float gh[3] = {0.f, 0.f, 0.f};
    if (!first) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            float sum = 0.f;
#pragma unroll
            for (int wv = 0; wv < FNW; ++wv) sum += red[wv][g][jl][bl];
            gh[g] = sum;
        }
    }
    const float rv = sigmoidf_(gi_r + gh[0] + bh_r);
    const float zv = sigmoidf_(gi_z + gh[1] + bh_z);
    const float hn = gh[2] + bh_n;
    const float nv = tanhf(gi_n + rv * hn);
    const float hv = (1.f - zv) * nv + zv * hp;
    out[((long)t * N + b) * 512 + d * GH + j] = hv;
    if (saved) {
        float* sv = saved + (((long)t * N + b) * 2 + d) * 4 * GH + j;
        sv[0] = rv;
        sv[GH] = zv;
        sv[2 * GH] = nv;
        sv[3 * GH] = hn;
    }
}

// gate derivatives of one (t, b, d, hidden j) given the total dh: writes dgi / dgh entries, returns dh*z
__device__ __forceinline__ float gru_gate_bwd1(const float* __restrict__ saved, const float* __restrict__ out, float* __restrict__ dgi,
                                               float* __restrict__ dgh, int T, int N, int t, int b, int d, int j, float dh) {
    const float* sv = saved + (((long)t * N + b) * 2 + d) * 4 * GH + j;
    const float rv = sv[0], zv = sv[GH], nv = sv[2 * GH], hn = sv[3 * GH];
    const int tp = d == 0 ? t - 1 : t + 1;
    const float hp = (tp >= 0 && tp < T) ? out[((long)tp * N + b) * 512 + d * GH + j] : 0.f;
    const float dn_pre = dh * (1.f - zv) * (1.f - nv * nv);
    const float dz = dh * (hp - nv) * zv * (1.f - zv);
    const float dr = dn_pre * hn * rv * (1.f - rv);
    float* gi_ = dgi + ((long)t * N + b) * (2 * G3) + d * G3 + j;
    float* gh_ = dgh + ((long)t * N + b) * (2 * G3) + d * G3 + j;
    gi_[0] = dr;
    gi_[GH] = dz;
    gi_[2 * GH] = dn_pre;
    gh_[0] = dr;
    gh_[GH] = dz;
    gh_[2 * GH] = dn_pre * rv;
    return dh * zv;
}

// backward step s (s = 0 .. T-1).  Direction 0 processes t = T-1-s, direction 1 processes t = s.
//   s == 0 : dh_total = dout[t]                                 (no GEMM)
//   s >= 1 : dh_total = dout[t] + dhz_prev + W_hh^T dgh[t_prev]  (t_prev = the time processed at step s-1)
// wT: packed W_hh^T fragments per direction (K=768, M=256): [2][24 chunks][16 mtiles][64][8].  dhz: ping-pong [2][2][N][256].
// Block = 16 hidden x 16 batch, 512 threads: K = 768 split over the block's EIGHT waves (96 each: the dependent MFMA chain and the slab
// load per wave are half of the 4-wave version's); threads 0..255 own the (hidden, batch) pairs of the epilogue.
__global__ __launch_bounds__(512) void k_gru_step_bwd(const float* __restrict__ dout, const float* __restrict__ saved, const float* __restrict__ out,
                                                      const float* __restrict__ wT, float* __restrict__ dgi, float* __restrict__ dgh,
                                                      float* __restrict__ dhz, int T, int N, int s) {
    constexpr int NW = 8, KW_ = 768 / NW, GP = KW_ + 4;  // LDS pitch of a 96-wide slab row
    __shared__ __attribute__((aligned(16))) float gs[NW][16 * GP];
    __shared__ float red[NW][16][17];
    const int d = blockIdx.z, jt = blockIdx.x, b0 = blockIdx.y * 16;
    const int t = d == 0 ? T - 1 - s : s;
    const int tq = d == 0 ? t + 1 : t - 1;  // time processed at the previous step
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // this thread's epilogue operands (independent of the GEMM): issue their loads first
    const int jl = tid & 15, bl = (tid >> 4) & 15;
    const int b = b0 + bl, j = jt * 16 + jl;
    const bool bv = b < N && tid < 256;
    float e_dout = 0.f, e_dhz = 0.f, e_r = 0.f, e_z = 0.f, e_n = 0.f, e_hn = 0.f, e_hp = 0.f;
    if (bv) {
        e_dout = dout[((long)t * N + b) * 512 + d * GH + j];
        if (s > 0) e_dhz = dhz[(((long)((s - 1) & 1) * 2 + d) * N + b) * GH + j];
        const float* sv = saved + (((long)t * N + b) * 2 + d) * 4 * GH + j;
        e_r = sv[0];
        e_z = sv[GH];
        e_n = sv[2 * GH];
        e_hn = sv[3 * GH];
        const int tp = d == 0 ? t - 1 : t + 1;
        if (tp >= 0 && tp < T) e_hp = out[((long)tp * N + b) * 512 + d * GH + j];
    }
    if (s > 0) {
        for (int it = lane; it < 16 * (KW_ / 4); it += 64) {
            const int row = it / (KW_ / 4), c4 = (it % (KW_ / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b0 + row < N) v = *reinterpret_cast<const float4*>(dgh + ((long)tq * N + b0 + row) * (2 * G3) + d * G3 + wave * KW_ + c4);
            *reinterpret_cast<float4*>(&gs[wave][row * GP + c4]) = v;
        }
        __syncthreads();
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* wd = wT + (long)d * 24 * 16 * 64 * 8;
#pragma unroll
        for (int kc = 0; kc < KW_ / 32; ++kc) {
            const Mma<float>::Frag pf = Mma<float>::load_p(&gs[wave][kc * 32], GP, 0, lane, 32);
            const Mma<float>::Frag wf = Mma<float>::load_w(wd, (long)(wave * (KW_ / 32) + kc) * 16 + jt, lane);
            acc = Mma<float>::mma<8>(wf, pf, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][(lane >> 4) * 4 + r][lane & 15] = acc[r];
        __syncthreads();
    }
    if (!bv) return;
    float dh = e_dout;
    if (s > 0) dh += e_dhz + ((red[0][jl][bl] + red[1][jl][bl]) + (red[2][jl][bl] + red[3][jl][bl])) + ((red[4][jl][bl] + red[5][jl][bl]) + (red[6][jl][bl] + red[7][jl][bl]));
    const float dn_pre = dh * (1.f - e_z) * (1.f - e_n * e_n);
    const float dz = dh * (e_hp - e_n) * e_z * (1.f - e_z);
    const float dr = dn_pre * e_hn * e_r * (1.f - e_r);
    float* gi_ = dgi + ((long)t * N + b) * (2 * G3) + d * G3 + j;
    float* gh_ = dgh + ((long)t * N + b) * (2 * G3) + d * G3 + j;
    gi_[0] = dr;
    gi_[GH] = dz;
    gi_[2 * GH] = dn_pre;
    gh_[0] = dr;
    gh_[GH] = dz;
    gh_[2 * GH] = dn_pre * e_r;
    dhz[(((long)(s & 1) * 2 + d) * N + b) * GH + j] = dh * e_z;
}

extern "C" {

// One bidirectional GRU layer, recurrent part (nn.GRU at models.py:245).  See kernel comment for layouts.  saved may be null (eval).
int ocrs_gru_layer_fwd(const float* gi, const float* whh_pk, const float* bhh, float* out, float* saved, int T, int N, hipStream_t st) {
    OCRS_CHECK_ARG(gi && whh_pk && bhh && out && T > 0 && N > 0);
    const dim3 grid(GH / 16, (N + 15) / 16, 2);
    for (int s = 0; s < T; ++s) hipLaunchKernelGGL(k_gru_step_fwd, grid, dim3(512), 0, st, gi, whh_pk, bhh, out, saved, T, N, s);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// BPTT of the recurrent part: dout [T][N][512] -> dgi, dgh [T][N][1536] (gradients w.r.t. gi and gh = W_hh h + b_hh).
// whhT_pk: ocrs_pack_frags(K=768, M=256) of W_hh^T per direction; dhz: workspace [2][2][N][256] fp32.
int ocrs_gru_layer_bwd(const float* dout, const float* saved, const float* out, const float* whhT_pk, float* dgi, float* dgh, float* dhz, int T,
                       int N, hipStream_t st) {
    OCRS_CHECK_ARG(dout && saved && out && whhT_pk && dgi && dgh && dhz && T > 0 && N > 0);
    const dim3 grid(GH / 16, (N + 15) / 16, 2);
    for (int s = 0; s < T; ++s) hipLaunchKernelGGL(k_gru_step_bwd, grid, dim3(512), 0, st, dout, saved, out, whhT_pk, dgi, dgh, dhz, T, N, s);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

}  // extern "C"
