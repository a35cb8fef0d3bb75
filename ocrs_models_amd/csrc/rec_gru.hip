// Bidirectional GRU recurrence (gfx950), fp32 as the reference forces it (ocrs_models/models.py:245, 264-266).
// PyTorch gate order (r, z, n):  r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = (1 - z) * n + z * h,
// gh = W_hh h + b_hh; the reverse direction scans t = T-1..0 and stores at its own t; h0 = 0.
//
// The input projections gi = W_ih x + b_ih for all time steps and both directions are one big GEMM (ocrs_conv_igemm, fp32 MFMA);
// only the T sequential steps run here, one launch per step (both directions in the same launch, grid.z = 2):
//   block = 16 hidden units x 3 gates (3 MFMA M-tiles) x 64 batch columns, K = 256 on v_mfma_f32_16x16x4_f32 (exact fp32).
// Backward (BPTT) mirrors it: one launch per step computes dh_carry = z*dh + W_hh^T dgh (K = 768) and, in its epilogue, the
// gate derivatives of the NEXT step to process; the weight/input gradients are big GEMMs over all steps afterwards.
#include "common.h"

static constexpr int GH = 256;       // hidden size
static constexpr int G3 = 3 * GH;    // 768
static constexpr int HPITCH = 260;   // LDS pitch (floats) of a 256-wide fp32 row

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// gi   [T][N][2*768]  (direction-major: d*768 + gate*256 + j), b_ih already added
// whh  packed W_hh fragments per direction (K=256, M=768): [2][8 chunks][48 mtiles][64][8] fp32
// bhh  [2][768];  out [T][N][512] (d*256 + j);  saved [T][N][2][4][256] = r | z | n | gh_n
__global__ __launch_bounds__(256) void k_gru_step_fwd(const float* __restrict__ gi, const float* __restrict__ whh, const float* __restrict__ bhh,
                                                      float* __restrict__ out, float* __restrict__ saved, int T, int N, int s) {
    extern __shared__ __attribute__((aligned(16))) float hs[];  // [64][HPITCH]
    const int d = blockIdx.z, jt = blockIdx.x, bt = blockIdx.y;
    const int t = d == 0 ? s : T - 1 - s;
    const int tp = d == 0 ? t - 1 : t + 1;  // time index of h_prev
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool first = s == 0;
    f32x4 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!first) {
        for (int it = tid; it < 64 * 64; it += 256) {
            const int row = it >> 6, c4 = (it & 63) * 4;
            const int b = bt * 64 + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b < N) v = *reinterpret_cast<const float4*>(out + ((long)tp * N + b) * 512 + d * GH + c4);
            *reinterpret_cast<float4*>(hs + row * HPITCH + c4) = v;
        }
        __syncthreads();
        const float* wd = whh + (long)d * 8 * 48 * 64 * 8;
#pragma unroll 2
        for (int kc = 0; kc < 8; ++kc) {
            const Mma<float>::Frag pf = Mma<float>::load_p(hs + kc * 32, HPITCH, wave * 16, lane, 32);
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const Mma<float>::Frag wf = Mma<float>::load_w(wd, (long)kc * 48 + g * 16 + jt, lane);
                acc[g] = Mma<float>::mma<8>(wf, pf, acc[g]);
            }
        }
    }
    const int b = bt * 64 + wave * 16 + (lane & 15);
    const int j0 = jt * 16 + (lane >> 4) * 4;
    if (b >= N) return;
    const float* gir = gi + ((long)t * N + b) * (2 * G3) + d * G3;
    const float4 ir = *reinterpret_cast<const float4*>(gir + j0);
    const float4 iz = *reinterpret_cast<const float4*>(gir + GH + j0);
    const float4 in_ = *reinterpret_cast<const float4*>(gir + 2 * GH + j0);
    const float4 br = *reinterpret_cast<const float4*>(bhh + d * G3 + j0);
    const float4 bz = *reinterpret_cast<const float4*>(bhh + d * G3 + GH + j0);
    const float4 bn = *reinterpret_cast<const float4*>(bhh + d * G3 + 2 * GH + j0);
    float4 hp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!first) hp = *reinterpret_cast<const float4*>(hs + (wave * 16 + (lane & 15)) * HPITCH + j0);
    const float irv[4] = {ir.x, ir.y, ir.z, ir.w}, izv[4] = {iz.x, iz.y, iz.z, iz.w}, inv[4] = {in_.x, in_.y, in_.z, in_.w};
    const float brv[4] = {br.x, br.y, br.z, br.w}, bzv[4] = {bz.x, bz.y, bz.z, bz.w}, bnv[4] = {bn.x, bn.y, bn.z, bn.w};
    const float hpv[4] = {hp.x, hp.y, hp.z, hp.w};
    float rv[4], zv[4], nv[4], hn[4], hv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        rv[q] = sigmoidf_(irv[q] + acc[0][q] + brv[q]);
        zv[q] = sigmoidf_(izv[q] + acc[1][q] + bzv[q]);
        hn[q] = acc[2][q] + bnv[q];
        nv[q] = tanhf(inv[q] + rv[q] * hn[q]);
        hv[q] = (1.f - zv[q]) * nv[q] + zv[q] * hpv[q];
    }
    *reinterpret_cast<float4*>(out + ((long)t * N + b) * 512 + d * GH + j0) = make_float4(hv[0], hv[1], hv[2], hv[3]);
    if (saved) {
        float* sv = saved + (((long)t * N + b) * 2 + d) * 4 * GH + j0;
        *reinterpret_cast<float4*>(sv) = make_float4(rv[0], rv[1], rv[2], rv[3]);
        *reinterpret_cast<float4*>(sv + GH) = make_float4(zv[0], zv[1], zv[2], zv[3]);
        *reinterpret_cast<float4*>(sv + 2 * GH) = make_float4(nv[0], nv[1], nv[2], nv[3]);
        *reinterpret_cast<float4*>(sv + 3 * GH) = make_float4(hn[0], hn[1], hn[2], hn[3]);
    }
}

// gate derivatives of one (t, b, d, 4 hidden units) given the total dh:  writes dgi, dgh rows and returns dh*z
__device__ __forceinline__ void gru_gate_bwd(const float* __restrict__ saved, const float* __restrict__ out, float* __restrict__ dgi,
                                             float* __restrict__ dgh, int T, int N, int t, int b, int d, int j0, const float (&dh)[4],
                                             float (&dhz)[4]) {
    const float* sv = saved + (((long)t * N + b) * 2 + d) * 4 * GH + j0;
    const float4 r4 = *reinterpret_cast<const float4*>(sv), z4 = *reinterpret_cast<const float4*>(sv + GH);
    const float4 n4 = *reinterpret_cast<const float4*>(sv + 2 * GH), q4 = *reinterpret_cast<const float4*>(sv + 3 * GH);
    const int tp = d == 0 ? t - 1 : t + 1;
    float4 hp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tp >= 0 && tp < T) hp = *reinterpret_cast<const float4*>(out + ((long)tp * N + b) * 512 + d * GH + j0);
    const float rv[4] = {r4.x, r4.y, r4.z, r4.w}, zv[4] = {z4.x, z4.y, z4.z, z4.w}, nv[4] = {n4.x, n4.y, n4.z, n4.w};
    const float hn[4] = {q4.x, q4.y, q4.z, q4.w}, hpv[4] = {hp.x, hp.y, hp.z, hp.w};
    float dr[4], dz[4], dn[4], dnr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float dn_pre = dh[q] * (1.f - zv[q]) * (1.f - nv[q] * nv[q]);
        dz[q] = dh[q] * (hpv[q] - nv[q]) * zv[q] * (1.f - zv[q]);
        dr[q] = dn_pre * hn[q] * rv[q] * (1.f - rv[q]);
        dn[q] = dn_pre;
        dnr[q] = dn_pre * rv[q];
        dhz[q] = dh[q] * zv[q];
    }
    float* gi_ = dgi + ((long)t * N + b) * (2 * G3) + d * G3 + j0;
    float* gh_ = dgh + ((long)t * N + b) * (2 * G3) + d * G3 + j0;
    *reinterpret_cast<float4*>(gi_) = make_float4(dr[0], dr[1], dr[2], dr[3]);
    *reinterpret_cast<float4*>(gi_ + GH) = make_float4(dz[0], dz[1], dz[2], dz[3]);
    *reinterpret_cast<float4*>(gi_ + 2 * GH) = make_float4(dn[0], dn[1], dn[2], dn[3]);
    *reinterpret_cast<float4*>(gh_) = make_float4(dr[0], dr[1], dr[2], dr[3]);
    *reinterpret_cast<float4*>(gh_ + GH) = make_float4(dz[0], dz[1], dz[2], dz[3]);
    *reinterpret_cast<float4*>(gh_ + 2 * GH) = make_float4(dnr[0], dnr[1], dnr[2], dnr[3]);
}

// backward step s (s = 0 .. T-1).  Direction 0 processes t = T-1-s, direction 1 processes t = s.
//   s == 0 : dh_total = dout[t]                                 (no GEMM)
//   s >= 1 : dh_total = dout[t] + dhz_prev + W_hh^T dgh[t_prev]  (t_prev = the time processed at step s-1)
// wT: packed W_hh^T fragments per direction (K=768, M=256): [2][24 chunks][16 mtiles][64][8].  dhz: ping-pong [2][2][N][256].
__global__ __launch_bounds__(256) void k_gru_step_bwd(const float* __restrict__ dout, const float* __restrict__ saved, const float* __restrict__ out,
                                                      const float* __restrict__ wT, float* __restrict__ dgi, float* __restrict__ dgh,
                                                      float* __restrict__ dhz, int T, int N, int s) {
    extern __shared__ __attribute__((aligned(16))) float gs[];  // [64][HPITCH]
    const int d = blockIdx.z, jt = blockIdx.x, bt = blockIdx.y;
    const int t = d == 0 ? T - 1 - s : s;
    const int tq = d == 0 ? t + 1 : t - 1;  // time processed at the previous step
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
        const float* wd = wT + (long)d * 24 * 16 * 64 * 8;
        for (int part = 0; part < 3; ++part) {
            __syncthreads();
            for (int it = tid; it < 64 * 64; it += 256) {
                const int row = it >> 6, c4 = (it & 63) * 4;
                const int b = bt * 64 + row;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (b < N) v = *reinterpret_cast<const float4*>(dgh + ((long)tq * N + b) * (2 * G3) + d * G3 + part * GH + c4);
                *reinterpret_cast<float4*>(gs + row * HPITCH + c4) = v;
            }
            __syncthreads();
#pragma unroll 2
            for (int kc = 0; kc < 8; ++kc) {
                const Mma<float>::Frag pf = Mma<float>::load_p(gs + kc * 32, HPITCH, wave * 16, lane, 32);
                const Mma<float>::Frag wf = Mma<float>::load_w(wd, (long)(part * 8 + kc) * 16 + jt, lane);
                acc = Mma<float>::mma<8>(wf, pf, acc);
            }
        }
    }
    const int b = bt * 64 + wave * 16 + (lane & 15);
    const int j0 = jt * 16 + (lane >> 4) * 4;
    if (b >= N) return;
    const float4 go = *reinterpret_cast<const float4*>(dout + ((long)t * N + b) * 512 + d * GH + j0);
    float dh[4] = {go.x, go.y, go.z, go.w};
    if (s > 0) {
        const float4 pz = *reinterpret_cast<const float4*>(dhz + (((long)((s - 1) & 1) * 2 + d) * N + b) * GH + j0);
        dh[0] += pz.x + acc[0];
        dh[1] += pz.y + acc[1];
        dh[2] += pz.z + acc[2];
        dh[3] += pz.w + acc[3];
    }
    float hz[4];
    gru_gate_bwd(saved, out, dgi, dgh, T, N, t, b, d, j0, dh, hz);
    *reinterpret_cast<float4*>(dhz + (((long)(s & 1) * 2 + d) * N + b) * GH + j0) = make_float4(hz[0], hz[1], hz[2], hz[3]);
}

extern "C" {

// One bidirectional GRU layer, recurrent part (nn.GRU at models.py:245).  See kernel comment for layouts.  saved may be null (eval).
int ocrs_gru_layer_fwd(const float* gi, const float* whh_pk, const float* bhh, float* out, float* saved, int T, int N, hipStream_t st) {
    OCRS_CHECK_ARG(gi && whh_pk && bhh && out && T > 0 && N > 0);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_step_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)
            return OCRS_ERR_HIP;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_step_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)
            return OCRS_ERR_HIP;
        attr_set = true;
    }
    const dim3 grid(GH / 16, (N + 63) / 64, 2);
    for (int s = 0; s < T; ++s)
        hipLaunchKernelGGL(k_gru_step_fwd, grid, dim3(256), 64 * HPITCH * sizeof(float), st, gi, whh_pk, bhh, out, saved, T, N, s);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// BPTT of the recurrent part: dout [T][N][512] -> dgi, dgh [T][N][1536] (gradients w.r.t. gi and gh = W_hh h + b_hh).
// whhT_pk: ocrs_pack_frags(K=768, M=256) of W_hh^T per direction; dhz: workspace [2][2][N][256] fp32.
int ocrs_gru_layer_bwd(const float* dout, const float* saved, const float* out, const float* whhT_pk, float* dgi, float* dgh, float* dhz, int T,
                       int N, hipStream_t st) {
    OCRS_CHECK_ARG(dout && saved && out && whhT_pk && dgi && dgh && dhz && T > 0 && N > 0);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gru_step_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)
            return OCRS_ERR_HIP;
        attr_set = true;
    }
    const dim3 grid(GH / 16, (N + 63) / 64, 2);
    for (int s = 0; s < T; ++s)
        hipLaunchKernelGGL(k_gru_step_bwd, grid, dim3(256), 64 * HPITCH * sizeof(float), st, dout, saved, out, whhT_pk, dgi, dgh, dhz, T, N, s);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

}  // extern "C"
