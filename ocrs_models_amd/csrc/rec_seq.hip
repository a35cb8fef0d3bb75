// Sequence-side kernels of the CRNN recogniser (gfx950): log-softmax, CTC loss (alpha / beta+gradient), greedy decode.
//
// Reference call sites: nn.LogSoftmax(dim=2) (ocrs_models/models.py:250), torch.nn.CTCLoss() defaults -- blank 0, reduction 'mean',
// zero_infinity False -- (ocrs_models/train_rec.py:104,121), argmax + ctc_greedy_decode_text (train_rec.py:52, datasets/util.py:147-177).
// CTC follows the published alpha-beta recursion (Graves et al. 2006) in fp32 log space; the gradient uses ATen's convention
// grad = exp(lp) - exp(alpha+beta - lp + nll), i.e. it is already the gradient w.r.t. the logits.
#include "common.h"

static constexpr float NEG_INF = -__builtin_inff();

__device__ __forceinline__ float lse2(float a, float b) {
    const float m = fmaxf(a, b);
    if (m == NEG_INF) return NEG_INF;
    return m + log1pf(__expf(-fabsf(a - b)));
}
#ifndef OCRS_CTC_FAST_LSE
#define OCRS_CTC_FAST_LSE 1  // 0: libm expf / logf in the lattice recursion (rounds 1-3)
#endif
// The CTC recursion is one dependent lse3 per time step: with libm's expf / logf (~15-20 instructions each) it is ~70 dependent VALU
// instructions per state and step, which IS the step time of both kernel forms.  The hardware transcendentals (v_exp_f32 = 2^x, v_log_f32 =
// log2 x, 1 ulp each) make it 13: the arguments are in (-inf, 0] with the largest term exactly 2^0, so the rounding of (x - m) * log2(e)
// matters only for terms that are negligible in the sum, and the error per step stays ~2-3e-7 absolute on alpha -- the same order as libm's
// (tests: loss within 1e-5 relative of torch up to T = 4200, gradient 1e-5 absolute on the reference's golden case).
__device__ __forceinline__ float lse3(float a, float b, float c) {
    const float m = fmaxf(fmaxf(a, b), c);
    if (m == NEG_INF) return NEG_INF;
#if OCRS_CTC_FAST_LSE
    constexpr float L2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    const float s = __builtin_amdgcn_exp2f((a - m) * L2E) + __builtin_amdgcn_exp2f((b - m) * L2E) + __builtin_amdgcn_exp2f((c - m) * L2E);
    return fmaf(__builtin_amdgcn_logf(s), LN2, m);
#else
    return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// log_softmax over the last dim (C classes); one 16-lane DPP row per (t, n) row, 16 rows per 256-thread block.
template <class TIN>
__global__ __launch_bounds__(256) void k_log_softmax(const TIN* __restrict__ logits, float* __restrict__ out, long rows, int C, int ld) {
    const int sub = threadIdx.x & 15;
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= rows) return;
    const TIN* src = logits + row * ld;
    float m = NEG_INF;
    for (int c = sub; c < C; c += 16) m = fmaxf(m, Elem<TIN>::ld(src + c));
    m = fmaxf(m, dpp_f<0xB1>(m));
    m = fmaxf(m, dpp_f<0x4E>(m));
    m = fmaxf(m, dpp_f<0x141>(m));
    m = fmaxf(m, dpp_f<0x140>(m));
    float s = 0.f;
    for (int c = sub; c < C; c += 16) s += expf(Elem<TIN>::ld(src + c) - m);
    s = quad16_sum(s);
    const float lz = m + logf(s);
    for (int c = sub; c < C; c += 16) out[row * C + c] = Elem<TIN>::ld(src + c) - lz;
}

// backward: dlogits = g - exp(lp) * sum_c g
template <class TOUT>
__global__ __launch_bounds__(256) void k_log_softmax_bwd(const float* __restrict__ lp, const float* __restrict__ g, TOUT* __restrict__ dlogits,
                                                         long rows, int C, int ld) {
    const int sub = threadIdx.x & 15;
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = sub; c < C; c += 16) s += g[row * C + c];
    s = quad16_sum(s);
    for (int c = sub; c < ld; c += 16) Elem<TOUT>::st(dlogits + row * ld + c, c < C ? g[row * C + c] - expf(lp[row * C + c]) * s : 0.f);
}

// ---------------------------------------------------------------------------------------------------------------------
// CTC forward: one block per sample, one thread per extended-label state s (S = 2L+1 <= blockDim.x * SPT).
// alpha rows are written to the global workspace (needed by the backward), the rolling row lives in LDS.
//   lp      [T][N][C] fp32 log-probs;  targets [N][Lpad] int32 (padded with anything);  in_len/tg_len [N] int64
//   alpha   [N][T][Smax] fp32 workspace;  nll [N] fp32
// states per thread SPT (template parameter): 3 -> S <= 768 (L <= 383: every lattice the CRNN's own widths can need), 8 -> 2048, 16 -> 4096 (L <= 2047)
static constexpr int CTC_SPT_MAX = 16;

// H16 (the separately-toleranced "fp16 alpha/beta" variant of BASELINE configs[4]): the alpha lattice kept for the backward is stored as
// fp16 of (alpha[t][s] - max_s alpha[t][s]) plus one fp32 row maximum per time step -- half the lattice bytes; the states that carry
// probability mass sit within a few units of the row maximum, where fp16 resolves <= 2^-8.  The recursion itself (LDS rolling rows) and the
// loss stay fp32, so the loss is IDENTICAL to the fp32 variant; only the gradient sees the rounding (measured ~1e-3 relative).
__device__ __forceinline__ float block_max256(float v, float* s_red /*[4]*/) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    __syncthreads();
    return m;
}
template <bool H16, int SPT>
__global__ __launch_bounds__(256) void k_ctc_alpha(const float* __restrict__ lp, const int* __restrict__ targets, const long long* __restrict__ in_len,
                                                   const long long* __restrict__ tg_len, void* __restrict__ alpha_v, float* __restrict__ rowmax,
                                                   float* __restrict__ nll, int T, int N, int C, int Lpad, int Smax) {
    __shared__ float row[2][256 * SPT + 2];
    __shared__ float s_red[4];
    float* alpha = reinterpret_cast<float*>(alpha_v);
    _Float16* alpha16 = reinterpret_cast<_Float16*>(alpha_v);
    const int n = blockIdx.x;
    // device-side lengths are clamped to the tensor extents (torch raises for input_lengths > T / target_lengths > Lpad on host lengths --
    // losses.py does the same check there; lengths that only exist on the device cannot raise without a sync, so they must not go out of bounds)
    int Ti = (int)in_len[n], L = (int)tg_len[n];
    Ti = Ti > T ? T : Ti;
    L = L < 0 ? 0 : (L > Lpad ? Lpad : L);
    L = 2 * L + 1 > Smax ? (Smax - 1) / 2 : L;
    const int S = 2 * L + 1;
    const int* tg = targets + (long)n * Lpad;
    float* al = alpha + (long)n * T * Smax;
    _Float16* al16 = alpha16 + (long)n * T * Smax;
    int ext[SPT];
    bool skip[SPT];
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
        const int s = threadIdx.x + j * 256;
        ext[j] = (s < S && (s & 1)) ? tg[s >> 1] : 0;
        skip[j] = s < S && (s & 1) && s >= 2 && tg[s >> 1] != tg[(s >> 1) - 1];
    }
    // store one lattice row: fp32 as is, or fp16 relative to the row maximum (a block reduction per time step)
    auto store_row = [&](int t, const float (&a)[SPT]) {
        if constexpr (!H16) {
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                const int s = threadIdx.x + j * 256;
                if (s < S) al[(long)t * Smax + s] = a[j];
            }
        } else {
            float m = NEG_INF;
#pragma unroll
            for (int j = 0; j < SPT; ++j) m = fmaxf(m, a[j]);
            m = block_max256(m, s_red);
            if (threadIdx.x == 0) rowmax[(long)n * T + t] = m;
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                const int s = threadIdx.x + j * 256;
                if (s < S) al16[(long)t * Smax + s] = (_Float16)((a[j] == NEG_INF || m == NEG_INF) ? -65504.f : fmaxf(a[j] - m, -65504.f));
            }
        }
    };
    if (Ti <= 0) {
        if (threadIdx.x == 0) nll[n] = (L == 0) ? 0.f : -NEG_INF;
        return;
    }
    // t = 0
    // rolling rows: state s lives at index s + 2; indices 0,1 are -inf pad slots (written below)
    {
        float a0v[SPT];
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const int s = threadIdx.x + j * 256;
            float a = NEG_INF;
            if (s < S && s < 2) a = lp[(long)n * C + ext[j]];
            row[0][s + 2] = a;
            a0v[j] = s < S ? a : NEG_INF;
        }
        store_row(0, a0v);
    }
    if (threadIdx.x < 2) row[0][threadIdx.x] = row[1][threadIdx.x] = NEG_INF;
    __syncthreads();
    int cur = 0;
    for (int t = 1; t < Ti; ++t) {
        const float* lpt = lp + ((long)t * N + n) * C;
        float av[SPT];
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const int s = threadIdx.x + j * 256;
            av[j] = NEG_INF;
            if (s < S) {
                const float a0 = row[cur][s + 2], a1 = row[cur][s + 1], a2 = skip[j] ? row[cur][s] : NEG_INF;
                const float a = lse3(a0, a1, a2) + lpt[ext[j]];
                row[cur ^ 1][s + 2] = a;
                av[j] = a;
            }
        }
        store_row(t, av);
        cur ^= 1;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float l1 = row[cur][S - 1 + 2];
        const float l2 = S > 1 ? row[cur][S - 2 + 2] : NEG_INF;
        nll[n] = -lse2(l1, l2);
    }
}

// 'mean' reduction: loss = mean_n( nll_n / max(L_n, 1) )   (one block)
__global__ __launch_bounds__(256) void k_ctc_reduce(const float* __restrict__ nll, const long long* __restrict__ tg_len, float* __restrict__ loss,
                                                    int N) {
    __shared__ float s_sum[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) {
        const float L = (float)(tg_len[i] > 1 ? tg_len[i] : 1);
        s += nll[i] / L;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *loss = (s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]) / (float)N;
}

// CTC backward: beta recursion + gradient, one block per sample.  grad [T][N][C] is fully written (zeros for t >= T_n).
template <bool H16, int SPT>
__global__ __launch_bounds__(256) void k_ctc_beta_grad(const float* __restrict__ lp, const int* __restrict__ targets, const long long* __restrict__ in_len,
                                                       const long long* __restrict__ tg_len, const void* __restrict__ alpha_v,
                                                       const float* __restrict__ rowmax, const float* __restrict__ nll,
                                                       const float* __restrict__ gout, float* __restrict__ grad, int T, int N, int C, int Lpad, int Smax) {
    __shared__ float row[2][256 * SPT + 2];
    extern __shared__ unsigned s_occ[];  // [C] state-occupancy sums per class in 2^-30 fixed point (see below)
    const int n = blockIdx.x;
    // device-side lengths are clamped to the tensor extents (torch raises for input_lengths > T / target_lengths > Lpad on host lengths --
    // losses.py does the same check there; lengths that only exist on the device cannot raise without a sync, so they must not go out of bounds)
    int Ti = (int)in_len[n], L = (int)tg_len[n];
    Ti = Ti > T ? T : Ti;
    L = L < 0 ? 0 : (L > Lpad ? Lpad : L);
    L = 2 * L + 1 > Smax ? (Smax - 1) / 2 : L;
    const int S = 2 * L + 1;
    const int* tg = targets + (long)n * Lpad;
    const float* al = reinterpret_cast<const float*>(alpha_v) + (long)n * T * Smax;
    const _Float16* al16 = reinterpret_cast<const _Float16*>(alpha_v) + (long)n * T * Smax;
    const float nl = nll[n];
    const float scale = gout[0] / ((float)N * (float)(L > 1 ? L : 1));
    int ext[SPT];
    bool skip[SPT];  // transition s -> s+2 allowed
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
        const int s = threadIdx.x + j * 256;
        ext[j] = (s < S && (s & 1)) ? tg[s >> 1] : 0;
        skip[j] = (s & 1) && s + 2 < S && tg[s >> 1] != tg[(s >> 1) + 1];
    }
    // rows t >= Ti get zero gradient
    for (int t = (Ti > 0 ? Ti : 0); t < T; ++t)
        for (int c = threadIdx.x; c < C; c += 256) grad[((long)t * N + n) * C + c] = 0.f;
    if (Ti <= 0) return;
    // row layout: state s at index s, two NEG_INF pad slots behind S
    for (int i = threadIdx.x; i < 256 * SPT + 2; i += 256) row[0][i] = row[1][i] = NEG_INF;
    __syncthreads();
    int cur = 0;
    for (int t = Ti - 1; t >= 0; --t) {
        const float* lpt = lp + ((long)t * N + n) * C;
        for (int c = threadIdx.x; c < C; c += 256) s_occ[c] = 0u;
        float rm = 0.f;
        if constexpr (H16) rm = rowmax[(long)n * T + t];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const int s = threadIdx.x + j * 256;
            if (s < S) {
                float b;
                if (t == Ti - 1) {
                    b = (s >= S - 2) ? lpt[ext[j]] : NEG_INF;
                } else {
                    const float b0 = row[cur][s], b1 = row[cur][s + 1], b2 = skip[j] ? row[cur][s + 2] : NEG_INF;
                    b = lse3(b0, b1, b2) + lpt[ext[j]];
                }
                row[cur ^ 1][s] = b;
                float a;
                if constexpr (H16) {
                    const float d = (float)al16[(long)t * Smax + s];
                    a = d <= -65504.f ? NEG_INF : d + rm;
                } else
                    a = al[(long)t * Smax + s];
                const float ab = a + b;
                // occupancy gamma_t(s) = exp(alpha + beta - lp + nll) in [0, 1], summed per class.  Accumulated as 2^-30 fixed point with INTEGER
                // LDS atomics: integer addition is associative, so the sum does not depend on the order the waves arrive in (float LDS atomics
                // made this gradient differ in the last bits from run to run); resolution 9e-10 absolute, the class total is <= 1 (+ rounding)
                if (ab != NEG_INF) {
                    const float g = fminf(expf(ab + nl - lpt[ext[j]]), 2.f);
                    atomicAdd(&s_occ[ext[j]], (unsigned)(g * 1073741824.f + 0.5f));
                }
            }
        }
        cur ^= 1;
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) grad[((long)t * N + n) * C + c] = (expf(lpt[c]) - (float)s_occ[c] * (1.f / 1073741824.f)) * scale;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// CTC with the two recursions side by side (round 4).  k_ctc_beta_grad runs T dependent steps of [three barriers, the
// alpha row from global memory, an LDS-atomic occupancy sum and a gradient-row store]: 121 us against the alpha recursion's 50.  But beta does
// not depend on alpha, and the gradient of a time step depends on nothing else once both lattices exist:
//   k_ctc_ab    grid (N, 2): blockIdx.y = 0 runs the alpha recursion (exactly k_ctc_alpha), 1 the beta recursion of the same sample, storing
//               its lattice -- both at once, two workgroups per CU;
//   k_ctc_grad  one WAVE per (sample, time step): occupancy sums with the same 2^-30 fixed-point integer atomics, then the gradient row.
// Same lse3, same association, integer sums: loss and gradient are bit-identical to k_ctc_alpha + k_ctc_beta_grad.
template <int SPT>
__global__ __launch_bounds__(256) void k_ctc_ab(const float* __restrict__ lp, const int* __restrict__ targets, const long long* __restrict__ in_len,
                                                const long long* __restrict__ tg_len, float* __restrict__ alpha, float* __restrict__ beta,
                                                float* __restrict__ nll, int T, int N, int C, int Lpad, int Smax) {
    __shared__ float row[2][256 * SPT + 2];
    const int n = blockIdx.x;
    int Ti = (int)in_len[n], L = (int)tg_len[n];
    Ti = Ti > T ? T : Ti;
    L = L < 0 ? 0 : (L > Lpad ? Lpad : L);
    L = 2 * L + 1 > Smax ? (Smax - 1) / 2 : L;
    const int S = 2 * L + 1;
    const int* tg = targets + (long)n * Lpad;
    if (blockIdx.y == 0) {  // ---- alpha (the code of k_ctc_alpha<false, SPT>)
        float* al = alpha + (long)n * T * Smax;
        int ext[SPT];
        bool skip[SPT];
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const int s = threadIdx.x + j * 256;
            ext[j] = (s < S && (s & 1)) ? tg[s >> 1] : 0;
            skip[j] = s < S && (s & 1) && s >= 2 && tg[s >> 1] != tg[(s >> 1) - 1];
        }
        if (Ti <= 0) {
            if (threadIdx.x == 0) nll[n] = (L == 0) ? 0.f : -NEG_INF;
            return;
        }
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const int s = threadIdx.x + j * 256;
            float a = NEG_INF;
            if (s < S && s < 2) a = lp[(long)n * C + ext[j]];
            row[0][s + 2] = a;
            if (s < S) al[s] = a;
        }
        if (threadIdx.x < 2) row[0][threadIdx.x] = row[1][threadIdx.x] = NEG_INF;
        __syncthreads();
        int cur = 0;
        for (int t = 1; t < Ti; ++t) {
            const float* lpt = lp + ((long)t * N + n) * C;
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                const int s = threadIdx.x + j * 256;
                if (s < S) {
                    const float a0 = row[cur][s + 2], a1 = row[cur][s + 1], a2 = skip[j] ? row[cur][s] : NEG_INF;
                    const float a = lse3(a0, a1, a2) + lpt[ext[j]];
                    row[cur ^ 1][s + 2] = a;
                    al[(long)t * Smax + s] = a;
                }
            }
            cur ^= 1;
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const float l1 = row[cur][S - 1 + 2];
            const float l2 = S > 1 ? row[cur][S - 2 + 2] : NEG_INF;
            nll[n] = -lse2(l1, l2);
        }
    } else {  // ---- beta (the recursion of k_ctc_beta_grad<false, SPT>, its rows stored)
        float* be = beta + (long)n * T * Smax;
        int ext[SPT];
        bool skip[SPT];  // transition s -> s+2 allowed
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const int s = threadIdx.x + j * 256;
            ext[j] = (s < S && (s & 1)) ? tg[s >> 1] : 0;
            skip[j] = (s & 1) && s + 2 < S && tg[s >> 1] != tg[(s >> 1) + 1];
        }
        if (Ti <= 0) return;
        for (int i = threadIdx.x; i < 256 * SPT + 2; i += 256) row[0][i] = row[1][i] = NEG_INF;
        __syncthreads();
        int cur = 0;
        for (int t = Ti - 1; t >= 0; --t) {
            const float* lpt = lp + ((long)t * N + n) * C;
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                const int s = threadIdx.x + j * 256;
                if (s < S) {
                    float b;
                    if (t == Ti - 1) {
                        b = (s >= S - 2) ? lpt[ext[j]] : NEG_INF;
                    } else {
                        const float b0 = row[cur][s], b1 = row[cur][s + 1], b2 = skip[j] ? row[cur][s + 2] : NEG_INF;
                        b = lse3(b0, b1, b2) + lpt[ext[j]];
                    }
                    row[cur ^ 1][s] = b;
                    be[(long)t * Smax + s] = b;
                }
            }
            cur ^= 1;
            __syncthreads();
        }
    }
}
// gradient rows from the two lattices: workgroup = 4 waves = 4 time steps of one sample; grad [T][N][C] fully written (zeros for t >= T_n)
__global__ __launch_bounds__(256) void k_ctc_grad(const float* __restrict__ lp, const int* __restrict__ targets, const long long* __restrict__ in_len,
                                                  const long long* __restrict__ tg_len, const float* __restrict__ alpha, const float* __restrict__ beta,
                                                  const float* __restrict__ nll, const float* __restrict__ gout, float* __restrict__ grad, int T, int N,
                                                  int C, int Lpad, int Smax) {
    extern __shared__ unsigned s_occ[];  // [4 waves][C]
    const int n = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6, t = blockIdx.x * 4 + w;
    int Ti = (int)in_len[n], L = (int)tg_len[n];
    Ti = Ti > T ? T : Ti;
    L = L < 0 ? 0 : (L > Lpad ? Lpad : L);
    L = 2 * L + 1 > Smax ? (Smax - 1) / 2 : L;
    const int S = 2 * L + 1;
    const int* tg = targets + (long)n * Lpad;
    unsigned* occ = s_occ + w * C;
    for (int c = lane; c < C; c += 64) occ[c] = 0u;
    __syncthreads();
    const bool live = t < T && t < Ti;
    const float nl = nll[n];
    const float scale = gout[0] / ((float)N * (float)(L > 1 ? L : 1));
    const float* lpt = lp + ((long)(t < T ? t : 0) * N + n) * C;
    if (live) {
        const float* al = alpha + ((long)n * T + t) * Smax;
        const float* be = beta + ((long)n * T + t) * Smax;
        for (int s = lane; s < S; s += 64) {
            const int e = (s & 1) ? tg[s >> 1] : 0;
            const float ab = al[s] + be[s];
            if (ab != NEG_INF) {
                const float g = fminf(expf(ab + nl - lpt[e]), 2.f);
                atomicAdd(&occ[e], (unsigned)(g * 1073741824.f + 0.5f));
            }
        }
    }
    __syncthreads();
    if (t < T) {
        float* gr = grad + ((long)t * N + n) * C;
        for (int c = lane; c < C; c += 64) gr[c] = live ? (expf(lpt[c]) - (float)occ[c] * (1.f / 1073741824.f)) * scale : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-level CTC (round 4; BASELINE north_star "wavefront shuffles for the CTC alpha / beta reductions").  The block kernels above give a
// sample a whole 256-thread block: one LDS round trip + one __syncthreads() per time step and a dependent global gather lp[t][ext[s]] inside
// the serial loop, with at most S = 2 L + 1 (81 at BASELINE configs[2]) of the 256 threads holding a state: 59 + 136 us per step.  Here a
// sample is ONE wave (four samples per block, no block-level synchronisation at all): lane l holds states SPL l .. SPL l + SPL - 1 in
// registers (SPL = 2: S <= 128, SPL = 4: S <= 256), the only values that cross lanes are the neighbour lane's two boundary states -- whole-
// wave DPP shifts (wave_shr:1 / wave_shl:1) -- and everything that does not depend on the recursion is fetched a chunk of CTC_TC time steps
// ahead: the lane's label log-probabilities (one gather per time step and label, CTC_TC of them in flight at once), the blank column, and
// in the backward the alpha row and the log-prob row of the gradient.  Same lse3 / same association as the block kernels: alpha, nll and
// (through the integer occupancy sums) the gradient are bit-identical to them.
static constexpr int CTC_TC = 8;
__device__ __forceinline__ float wave_shr1(float v, float fill) {  // result[l] = v[l - 1]; lane 0 gets `fill`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float wave_shl1(float v, float fill) {  // result[l] = v[l + 1]; lane 63 gets `fill`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, false));
}
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (unsigned)__shfl_xor((int)v, o, 64);
    return v;
}
template <int SPL>
__device__ __forceinline__ float pick_state(const float (&a)[SPL], int s) {  // state s of the wave's lattice row, broadcast to every lane
    float v = a[0];
#pragma unroll
    for (int j = 1; j < SPL; ++j) v = (s % SPL == j) ? a[j] : v;
    return __shfl(v, s / SPL, 64);
}

template <bool H16, int SPL>
__global__ __launch_bounds__(256) void k_ctc_alpha_w(const float* __restrict__ lp, const int* __restrict__ targets, const long long* __restrict__ in_len,
                                                     const long long* __restrict__ tg_len, void* __restrict__ alpha_v, float* __restrict__ rowmax,
                                                     float* __restrict__ nll, int T, int N, int C, int Lpad, int Smax) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    int Ti = (int)in_len[n], L = (int)tg_len[n];
    Ti = Ti > T ? T : Ti;
    L = L < 0 ? 0 : (L > Lpad ? Lpad : L);
    L = 2 * L + 1 > Smax ? (Smax - 1) / 2 : L;
    const int S = 2 * L + 1;
    const int* tg = targets + (long)n * Lpad;
    float* al = reinterpret_cast<float*>(alpha_v) + (long)n * T * Smax;
    _Float16* al16 = reinterpret_cast<_Float16*>(alpha_v) + (long)n * T * Smax;
    int ext[SPL];
    bool skip[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const int s = lane * SPL + j;
        ext[j] = (s < S && (s & 1)) ? tg[s >> 1] : 0;
        skip[j] = s < S && (s & 1) && s >= 2 && tg[s >> 1] != tg[(s >> 1) - 1];
    }
    if (Ti <= 0) {
        if (lane == 0) nll[n] = (L == 0) ? 0.f : -NEG_INF;
        return;
    }
    auto store_row = [&](int t, const float (&a)[SPL]) {
        if constexpr (!H16) {
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
                const int s = lane * SPL + j;
                if (s < S) al[(long)t * Smax + s] = a[j];
            }
        } else {
            float m = NEG_INF;
#pragma unroll
            for (int j = 0; j < SPL; ++j) m = fmaxf(m, a[j]);
            m = wave_max(m);
            if (lane == 0) rowmax[(long)n * T + t] = m;
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
                const int s = lane * SPL + j;
                if (s < S) al16[(long)t * Smax + s] = (_Float16)((a[j] == NEG_INF || m == NEG_INF) ? -65504.f : fmaxf(a[j] - m, -65504.f));
            }
        }
    };
    // label / blank log-probabilities of time steps t0 .. t0 + CTC_TC - 1 (clamped to the sample's last step): SPL / 2 gathers + 1 broadcast each
    auto fetch = [&](int t0, float (&lv)[CTC_TC][SPL]) {
#pragma unroll
        for (int i = 0; i < CTC_TC; ++i) {
            const int t = t0 + i < Ti ? t0 + i : Ti - 1;
            const float* lpt = lp + ((long)t * N + n) * C;
            const float blank = lpt[0];
#pragma unroll
            for (int j = 0; j < SPL; ++j) lv[i][j] = (j & 1) ? lpt[ext[j]] : blank;
        }
    };
    float a[SPL], lvA[CTC_TC][SPL], lvB[CTC_TC][SPL];
    fetch(0, lvA);
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const int s = lane * SPL + j;
        a[j] = (s < S && s < 2) ? lvA[0][j] : NEG_INF;
    }
    store_row(0, a);
    // chunk c covers time steps c * CTC_TC + i; the first chunk starts at i = 1
    auto chunk = [&](int t0, const float (&lv)[CTC_TC][SPL], int i0) {
#pragma unroll
        for (int i = 0; i < CTC_TC; ++i) {
            const int t = t0 + i;
            if (i < i0 || t >= Ti) continue;  // (wave-uniform)
            const float pm1 = wave_shr1(a[SPL - 1], NEG_INF), pm2 = wave_shr1(a[SPL - 2], NEG_INF);
            float an[SPL];
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
                const int s = lane * SPL + j;
                const float a1 = j >= 1 ? a[j - 1] : pm1;
                const float a2 = skip[j] ? (j >= 2 ? a[j - 2] : (j == 1 ? pm1 : pm2)) : NEG_INF;
                an[j] = s < S ? lse3(a[j], a1, a2) + lv[i][j] : NEG_INF;
            }
#pragma unroll
            for (int j = 0; j < SPL; ++j) a[j] = an[j];
            store_row(t, a);
        }
    };
    for (int t0 = 0; t0 < Ti; t0 += 2 * CTC_TC) {
        if (t0 + CTC_TC < Ti) fetch(t0 + CTC_TC, lvB);
        chunk(t0, lvA, t0 == 0 ? 1 : 0);
        if (t0 + 2 * CTC_TC < Ti) fetch(t0 + 2 * CTC_TC, lvA);
        chunk(t0 + CTC_TC, lvB, 0);
    }
    const float l1 = pick_state<SPL>(a, S - 1), l2 = S > 1 ? pick_state<SPL>(a, S - 2) : NEG_INF;
    if (lane == 0) nll[n] = -lse2(l1, l2);
}

template <bool H16, int SPL>
__global__ __launch_bounds__(256) void k_ctc_beta_grad_w(const float* __restrict__ lp, const int* __restrict__ targets, const long long* __restrict__ in_len,
                                                         const long long* __restrict__ tg_len, const void* __restrict__ alpha_v,
                                                         const float* __restrict__ rowmax, const float* __restrict__ nll, const float* __restrict__ gout,
                                                         float* __restrict__ grad, int T, int N, int C, int Lpad, int Smax) {
    extern __shared__ unsigned s_occ_all[];  // [4 waves][C] state-occupancy sums per class in 2^-30 fixed point
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    unsigned* s_occ = s_occ_all + wave * C;
    int Ti = (int)in_len[n], L = (int)tg_len[n];
    Ti = Ti > T ? T : Ti;
    L = L < 0 ? 0 : (L > Lpad ? Lpad : L);
    L = 2 * L + 1 > Smax ? (Smax - 1) / 2 : L;
    const int S = 2 * L + 1;
    const int* tg = targets + (long)n * Lpad;
    const float* al = reinterpret_cast<const float*>(alpha_v) + (long)n * T * Smax;
    const _Float16* al16 = reinterpret_cast<const _Float16*>(alpha_v) + (long)n * T * Smax;
    const float nl = nll[n];
    const float scale = gout[0] / ((float)N * (float)(L > 1 ? L : 1));
    int ext[SPL];
    bool skip[SPL];  // transition s -> s + 2 allowed
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const int s = lane * SPL + j;
        ext[j] = (s < S && (s & 1)) ? tg[s >> 1] : 0;
        skip[j] = (s & 1) && s + 2 < S && tg[s >> 1] != tg[(s >> 1) + 1];
    }
    for (int t = (Ti > 0 ? Ti : 0); t < T; ++t)
        for (int c = lane; c < C; c += 64) grad[((long)t * N + n) * C + c] = 0.f;
    if (Ti <= 0) return;
    for (int c = lane; c < C; c += 64) s_occ[c] = 0u;
    asm volatile("" ::: "memory");
    constexpr int CR = 2;  // classes per lane of a gradient row (C <= 128; the launcher checks)
    struct Pre {
        float lv[SPL], av[SPL], row[CR], rm;
    };
    // everything of time step t that does not depend on the recursion
    auto fetch = [&](int t, Pre& p) {
        t = t < 0 ? 0 : t;
        const float* lpt = lp + ((long)t * N + n) * C;
        const float blank = lpt[0];
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const int s = lane * SPL + j;
            p.lv[j] = (j & 1) ? lpt[ext[j]] : blank;
            const int sc = s < S ? s : S - 1;
            if constexpr (H16) p.av[j] = (float)al16[(long)t * Smax + sc];
            else p.av[j] = al[(long)t * Smax + sc];
        }
        if constexpr (H16) p.rm = rowmax[(long)n * T + t];
        else p.rm = 0.f;
#pragma unroll
        for (int k = 0; k < CR; ++k) p.row[k] = lpt[lane + 64 * k < C ? lane + 64 * k : 0];
    };
    float b[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) b[j] = NEG_INF;
    Pre pA[CTC_TC], pB[CTC_TC];
    auto chunk = [&](int thi, const Pre (&pp)[CTC_TC]) {  // time steps thi, thi - 1, ... (entry i = step thi - i)
#pragma unroll
        for (int i = 0; i < CTC_TC; ++i) {
            const int t = thi - i;
            if (t < 0) continue;  // (wave-uniform)
            const Pre& p = pp[i];
            const float np1 = wave_shl1(b[0], NEG_INF), np2 = wave_shl1(b[1], NEG_INF);
            float bn[SPL];
            unsigned blank_u = 0u;
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
                const int s = lane * SPL + j;
                float v = NEG_INF;
                if (s < S) {
                    if (t == Ti - 1) {
                        v = (s >= S - 2) ? p.lv[j] : NEG_INF;
                    } else {
                        const float b1 = j + 1 < SPL ? b[j + 1] : np1;
                        const float b2 = skip[j] ? (j + 2 < SPL ? b[j + 2] : (j + 2 == SPL ? np1 : np2)) : NEG_INF;
                        v = lse3(b[j], b1, b2) + p.lv[j];
                    }
                    float a = p.av[j];
                    if constexpr (H16) a = a <= -65504.f ? NEG_INF : a + p.rm;
                    const float ab = a + v;
                    // occupancy gamma_t(s) = exp(alpha + beta - lp + nll) in [0, 1], summed per class as 2^-30 fixed point: integer addition
                    // is associative, so neither the atomics' arrival order nor the shuffle tree of the blank column changes the sum
                    if (ab != NEG_INF) {
                        const unsigned u = (unsigned)(fminf(expf(ab + nl - p.lv[j]), 2.f) * 1073741824.f + 0.5f);
                        if (j & 1) atomicAdd(&s_occ[ext[j]], u);
                        else blank_u += u;
                    }
                }
                bn[j] = v;
            }
#pragma unroll
            for (int j = 0; j < SPL; ++j) b[j] = bn[j];
            blank_u = wave_sum_u32(blank_u);
            if (lane == 0) atomicAdd(&s_occ[0], blank_u);
            // the gradient row of this time step; the counters are cleared for the next one.  LDS operations of a wave execute in issue order, so
            // no hardware synchronisation is needed between the lanes' atomics and the reads -- only the compiler must keep that order (a
            // lane's read of counter c has no data dependence on ITS OWN atomic to another counter)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = 0; k < CR; ++k) {
                const int c = lane + 64 * k;
                if (c < C) {
                    const unsigned o = s_occ[c];
                    s_occ[c] = 0u;
                    grad[((long)t * N + n) * C + c] = (expf(p.row[k]) - (float)o * (1.f / 1073741824.f)) * scale;
                }
            }
            asm volatile("" ::: "memory");
        }
    };
    auto fetch_chunk = [&](int thi, Pre (&pp)[CTC_TC]) {
#pragma unroll
        for (int i = 0; i < CTC_TC; ++i) fetch(thi - i, pp[i]);
    };
    fetch_chunk(Ti - 1, pA);
    for (int thi = Ti - 1; thi >= 0; thi -= 2 * CTC_TC) {
        if (thi - CTC_TC >= 0) fetch_chunk(thi - CTC_TC, pB);
        chunk(thi, pA);
        if (thi - 2 * CTC_TC >= 0) fetch_chunk(thi - 2 * CTC_TC, pA);
        chunk(thi - CTC_TC, pB);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// greedy decode: argmax over classes (first maximum on ties), then collapse repeats (compare with the previous class BEFORE
// the blank test) and drop blanks -- datasets/util.py:163-175.  labels [N][T] int32 (collapsed, left-aligned), lens [N] int32,
// argmax_out [N][T] int32 (raw arg-max, optional).
__global__ __launch_bounds__(256) void k_argmax(const float* __restrict__ lp, int* __restrict__ amax, int T, int N, int C) {
    const int sub = threadIdx.x & 15;
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);  // row = t*N + n
    if (row >= (long)T * N) return;
    float best = NEG_INF;
    int bi = C;
    for (int c = sub; c < C; c += 16) {
        const float v = lp[row * C + c];
        if (v > best || (v == best && c < bi)) {
            best = v;
            bi = c;
        }
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    if (sub == 0) {
        const int t = (int)(row / N), n = (int)(row - (long)t * N);
        amax[(long)n * T + t] = bi;
    }
}

__global__ void k_ctc_collapse(const int* __restrict__ amax, const long long* __restrict__ in_len, int* __restrict__ labels, int* __restrict__ lens,
                               int T, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    int Ti = (int)in_len[n];
    if (Ti > T) Ti = T;
    int prev = -1, k = 0;
    for (int t = 0; t < Ti; ++t) {
        const int c = amax[(long)n * T + t];
        if (c == prev) continue;
        prev = c;
        if (c != 0) labels[(long)n * T + k++] = c;
    }
    lens[n] = k;
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused wave-level CTC: forward AND the gradient in one launch, everything the recursions touch in LDS.  Measured on the two-kernel wave form
// above: 36 + 86 us at BASELINE configs[2] -- no better than the block kernels -- because every step stores a lattice / gradient row to global
// memory, and on gfx9 stores and loads share one in-order counter: with stores pending hipcc must wait vmcnt(0) for the prefetched
// log-probabilities, i.e. for the NEXT chunk's loads as well (a full memory latency every chunk).  Here a wave first copies its sample's
// log-probability rows (T x C) into LDS, runs the alpha recursion with the lattice rows kept in LDS (T x Smax), then the beta recursion +
// occupancy sums + gradient rows: no global load inside either recursion, and the gradient stores are fire-and-forget.
// grad_pre = (exp(lp) - occupancy) / (N max(L, 1)): the upstream gradient multiplies it in the backward call (ocrs_scale_by_dev) -- bit-identical
// to ocrs_ctc_bwd for gout = 1.  grad_pre may be null (loss only: validation).  LDS per wave: (T (C + Smax) + C) floats.
template <int SPL>
__global__ __launch_bounds__(256) void k_ctc_fused_w(const float* __restrict__ lp, const int* __restrict__ targets, const long long* __restrict__ in_len,
                                                     const long long* __restrict__ tg_len, float* __restrict__ nll, float* __restrict__ grad_pre, int T, int N,
                                                     int C, int Lpad, int Smax, int wpb) {
    extern __shared__ float s_f[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = blockIdx.x * wpb + wave;
    if (n >= N) return;
    const long per = (long)T * (C + Smax) + C;
    float* lpS = s_f + wave * per;
    float* alS = lpS + (long)T * C;
    unsigned* s_occ = reinterpret_cast<unsigned*>(alS + (long)T * Smax);
    int Ti = (int)in_len[n], L = (int)tg_len[n];
    Ti = Ti > T ? T : Ti;
    L = L < 0 ? 0 : (L > Lpad ? Lpad : L);
    L = 2 * L + 1 > Smax ? (Smax - 1) / 2 : L;
    const int S = 2 * L + 1;
    const int* tg = targets + (long)n * Lpad;
    if (grad_pre)
        for (int t = (Ti > 0 ? Ti : 0); t < T; ++t)
            for (int c = lane; c < C; c += 64) grad_pre[((long)t * N + n) * C + c] = 0.f;
    if (Ti <= 0) {
        if (lane == 0) nll[n] = (L == 0) ? 0.f : -NEG_INF;
        return;
    }
    int ext[SPL];
    bool skipa[SPL], skipb[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const int s = lane * SPL + j;
        ext[j] = (s < S && (s & 1)) ? tg[s >> 1] : 0;
        skipa[j] = s < S && (s & 1) && s >= 2 && tg[s >> 1] != tg[(s >> 1) - 1];
        skipb[j] = (s & 1) && s + 2 < S && tg[s >> 1] != tg[(s >> 1) + 1];
    }
    // ---- the sample's log-probability rows -> LDS (coalesced rows, every load in flight at once)
    for (int t = 0; t < Ti; ++t)
        for (int c = lane; c < C; c += 64) lpS[t * C + c] = lp[((long)t * N + n) * C + c];
    for (int c = lane; c < C; c += 64) s_occ[c] = 0u;
    asm volatile("" ::: "memory");  // (LDS operations of one wave execute in issue order: only the compiler must not reorder across lanes' dependences)
    // ---- alpha
    float a[SPL], lv[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const int s = lane * SPL + j;
        a[j] = (s < S && s < 2) ? lpS[ext[j]] : NEG_INF;
        if (s < S) alS[s] = a[j];
    }
#pragma unroll
    for (int j = 0; j < SPL; ++j) lv[j] = lpS[(Ti > 1 ? C : 0) + ext[j]];
    for (int t = 1; t < Ti; ++t) {
        float ln[SPL];
        const int tn = t + 1 < Ti ? t + 1 : t;
#pragma unroll
        for (int j = 0; j < SPL; ++j) ln[j] = lpS[tn * C + ext[j]];  // next step's values: off the dependent chain
        const float pm1 = wave_shr1(a[SPL - 1], NEG_INF), pm2 = wave_shr1(a[SPL - 2], NEG_INF);
        float an[SPL];
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const int s = lane * SPL + j;
            const float a1 = j >= 1 ? a[j - 1] : pm1;
            const float a2 = skipa[j] ? (j >= 2 ? a[j - 2] : (j == 1 ? pm1 : pm2)) : NEG_INF;
            an[j] = s < S ? lse3(a[j], a1, a2) + lv[j] : NEG_INF;
        }
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            a[j] = an[j];
            lv[j] = ln[j];
            if (lane * SPL + j < S) alS[t * Smax + lane * SPL + j] = an[j];
        }
    }
    const float l1 = pick_state<SPL>(a, S - 1), l2 = S > 1 ? pick_state<SPL>(a, S - 2) : NEG_INF;
    const float nl = -lse2(l1, l2);
    if (lane == 0) nll[n] = nl;
    if (!grad_pre) return;
    asm volatile("" ::: "memory");
    // ---- beta + occupancy + gradient rows
    const float scale = 1.f / ((float)N * (float)(L > 1 ? L : 1));
    float b[SPL], av[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const int s = lane * SPL + j, sc = s < S ? s : S - 1;
        b[j] = NEG_INF;
        lv[j] = lpS[(Ti - 1) * C + ext[j]];
        av[j] = alS[(Ti - 1) * Smax + sc];
    }
    for (int t = Ti - 1; t >= 0; --t) {
        float ln[SPL], avn[SPL], rowv[2];
        const int tp = t > 0 ? t - 1 : 0;
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const int s = lane * SPL + j, sc = s < S ? s : S - 1;
            ln[j] = lpS[tp * C + ext[j]];
            avn[j] = alS[tp * Smax + sc];
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) rowv[k] = lpS[t * C + (lane + 64 * k < C ? lane + 64 * k : 0)];
        const float np1 = wave_shl1(b[0], NEG_INF), np2 = wave_shl1(b[1], NEG_INF);
        float bn[SPL];
        unsigned blank_u = 0u;
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const int s = lane * SPL + j;
            float v = NEG_INF;
            if (s < S) {
                if (t == Ti - 1) {
                    v = (s >= S - 2) ? lv[j] : NEG_INF;
                } else {
                    const float b1 = j + 1 < SPL ? b[j + 1] : np1;
                    const float b2 = skipb[j] ? (j + 2 < SPL ? b[j + 2] : (j + 2 == SPL ? np1 : np2)) : NEG_INF;
                    v = lse3(b[j], b1, b2) + lv[j];
                }
                const float ab = av[j] + v;
                if (ab != NEG_INF) {
                    const unsigned u = (unsigned)(fminf(expf(ab + nl - lv[j]), 2.f) * 1073741824.f + 0.5f);
                    if (j & 1) atomicAdd(&s_occ[ext[j]], u);
                    else blank_u += u;
                }
            }
            bn[j] = v;
        }
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            b[j] = bn[j];
            lv[j] = ln[j];
            av[j] = avn[j];
        }
        blank_u = wave_sum_u32(blank_u);
        if (lane == 0) atomicAdd(&s_occ[0], blank_u);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int c = lane + 64 * k;
            if (c < C) {
                const unsigned o = s_occ[c];
                s_occ[c] = 0u;
                grad_pre[((long)t * N + n) * C + c] = (expf(rowv[k]) - (float)o * (1.f / 1073741824.f)) * scale;
            }
        }
        asm volatile("" ::: "memory");
    }
}
__global__ __launch_bounds__(256) void k_scale_by_dev(const float* __restrict__ in, const float* __restrict__ g, float* __restrict__ out, long n) {
    const float s = g[0];
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(in + i);
            *reinterpret_cast<float4*>(out + i) = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
        } else
            for (long k = i; k < n; ++k) out[k] = in[k] * s;
    }
}

// one wave per sample (k_ctc_*_w): lattices of up to 64 SPL states, gradient rows of up to 128 classes.  Measured at BASELINE configs[2]
// (T = 101, N = 256, S <= 81; rocprofv3 kernel times, all forms with the hardware-transcendental lse3): block kernels 37.8 + 88.0 us, two-kernel
// wave form 36.2 + 85.6 us, fused wave form 132.9 us -- a single wave per SIMD issues its ~55 dependent VALU / transcendental instructions per
// time step at ~10 cycles each (0.35 us per step), the same time the block form spends on its LDS round trip + barrier with the lse3 of the
// states spread over four waves.  The wave forms are bit-identical to the block form (tests) and selectable (OCRS_CTC_WAVE=1 /
// OCRS_CTC_FUSED=1); what DID pay is the lse3 itself (59.5 + 136 -> 37.8 + 88 us).  A faster recursion needs fewer instructions per state
// (scaled linear-domain alpha / beta instead of log space), not a different thread mapping.
static bool ctc_wave_ok(int Smax, int C, int spl) {
    const char* e = getenv("OCRS_CTC_WAVE");  // measurement / test knob, default off (no faster than the block kernels, see above)
    return (e && e[0] == '1') && Smax <= 64 * spl && C <= 128;
}

extern "C" {

// nn.LogSoftmax(dim=2) (models.py:250).  logits [rows][ld] (ld >= C row pitch; dtype 0 fp32 / 1 bf16) -> log-probs fp32 [rows][C].
int ocrs_log_softmax_fwd(const void* logits, float* out, long rows, int C, int ld, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(logits && out && rows > 0 && C > 0 && ld >= C);
    const int grid = (int)((rows + 15) / 16);
    if (dtype == 1)
        hipLaunchKernelGGL(k_log_softmax<bf16>, dim3(grid), dim3(256), 0, st, (const bf16*)logits, out, rows, C, ld);
    else
        hipLaunchKernelGGL(k_log_softmax<float>, dim3(grid), dim3(256), 0, st, (const float*)logits, out, rows, C, ld);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// dlogits [rows][ld] (pad columns written as 0)
int ocrs_log_softmax_bwd(const float* lp, const float* g, void* dlogits, long rows, int C, int ld, int dtype, hipStream_t st) {
    OCRS_CHECK_ARG(lp && g && dlogits && rows > 0 && C > 0 && ld >= C);
    const int grid = (int)((rows + 15) / 16);
    if (dtype == 1)
        hipLaunchKernelGGL(k_log_softmax_bwd<bf16>, dim3(grid), dim3(256), 0, st, lp, g, (bf16*)dlogits, rows, C, ld);
    else
        hipLaunchKernelGGL(k_log_softmax_bwd<float>, dim3(grid), dim3(256), 0, st, lp, g, (float*)dlogits, rows, C, ld);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// torch.nn.CTCLoss() forward (train_rec.py:104,121).  alpha: workspace [N][T][Smax] with Smax >= 2*max(tg_len)+1; nll [N]; loss [1].
int ocrs_ctc_fwd(const float* lp, const int* targets, const long long* in_len, const long long* tg_len, float* alpha, float* nll, float* loss,
                 int T, int N, int C, int Lpad, int Smax, hipStream_t st) {
    OCRS_CHECK_ARG(lp && targets && in_len && tg_len && alpha && nll && loss && T > 0 && N > 0 && C > 0);
    OCRS_CHECK_ARG(Smax <= 256 * CTC_SPT_MAX && Smax >= 1);
    if (ctc_wave_ok(Smax, C, 2)) {
        hipLaunchKernelGGL((k_ctc_alpha_w<false, 2>), dim3((N + 3) / 4), dim3(256), 0, st, lp, targets, in_len, tg_len, (void*)alpha, (float*)nullptr, nll, T, N, C, Lpad, Smax);
    } else if (ctc_wave_ok(Smax, C, 4)) {
        hipLaunchKernelGGL((k_ctc_alpha_w<false, 4>), dim3((N + 3) / 4), dim3(256), 0, st, lp, targets, in_len, tg_len, (void*)alpha, (float*)nullptr, nll, T, N, C, Lpad, Smax);
    } else if (Smax <= 256 * 3) {
        hipLaunchKernelGGL((k_ctc_alpha<false, 3>), dim3(N), dim3(256), 0, st, lp, targets, in_len, tg_len, (void*)alpha, (float*)nullptr, nll, T, N, C, Lpad, Smax);
    } else if (Smax <= 256 * 8) {
        hipLaunchKernelGGL((k_ctc_alpha<false, 8>), dim3(N), dim3(256), 0, st, lp, targets, in_len, tg_len, (void*)alpha, (float*)nullptr, nll, T, N, C, Lpad, Smax);
    } else {
        hipLaunchKernelGGL((k_ctc_alpha<false, 16>), dim3(N), dim3(256), 0, st, lp, targets, in_len, tg_len, (void*)alpha, (float*)nullptr, nll, T, N, C, Lpad, Smax);
    }
    hipLaunchKernelGGL(k_ctc_reduce, dim3(1), dim3(256), 0, st, nll, tg_len, loss, N);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}
// ocrs_ctc_fwd with the beta lattice computed by the same launch (for a backward by ocrs_ctc_grad_ab): alpha, beta [N][T][Smax] fp32.
// Loss and (with ocrs_ctc_grad_ab) gradient bit-identical to ocrs_ctc_fwd + ocrs_ctc_bwd.
int ocrs_ctc_fwd_ab(const float* lp, const int* targets, const long long* in_len, const long long* tg_len, float* alpha, float* beta, float* nll,
                    float* loss, int T, int N, int C, int Lpad, int Smax, hipStream_t st) {
    OCRS_CHECK_ARG(lp && targets && in_len && tg_len && alpha && beta && nll && loss && T > 0 && N > 0 && C > 0);
    OCRS_CHECK_ARG(Smax <= 256 * CTC_SPT_MAX && Smax >= 1);
    const dim3 grid(N, 2);
    if (Smax <= 256 * 3)
        hipLaunchKernelGGL((k_ctc_ab<3>), grid, dim3(256), 0, st, lp, targets, in_len, tg_len, alpha, beta, nll, T, N, C, Lpad, Smax);
    else if (Smax <= 256 * 8)
        hipLaunchKernelGGL((k_ctc_ab<8>), grid, dim3(256), 0, st, lp, targets, in_len, tg_len, alpha, beta, nll, T, N, C, Lpad, Smax);
    else
        hipLaunchKernelGGL((k_ctc_ab<16>), grid, dim3(256), 0, st, lp, targets, in_len, tg_len, alpha, beta, nll, T, N, C, Lpad, Smax);
    hipLaunchKernelGGL(k_ctc_reduce, dim3(1), dim3(256), 0, st, nll, tg_len, loss, N);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}
// backward of ocrs_ctc_fwd_ab: grad [T][N][C] written; gout = upstream scalar gradient (device fp32 [1])
int ocrs_ctc_grad_ab(const float* lp, const int* targets, const long long* in_len, const long long* tg_len, const float* alpha, const float* beta,
                     const float* nll, const float* gout, float* grad, int T, int N, int C, int Lpad, int Smax, hipStream_t st) {
    OCRS_CHECK_ARG(lp && targets && in_len && tg_len && alpha && beta && nll && gout && grad && T > 0 && N > 0 && C > 0 && C <= 4096);
    hipLaunchKernelGGL(k_ctc_grad, dim3((T + 3) / 4, N), dim3(256), 4 * C * sizeof(unsigned), st, lp, targets, in_len, tg_len, alpha, beta, nll, gout, grad,
                       T, N, C, Lpad, Smax);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}
// The "fp16 alpha/beta" variant (BASELINE configs[4]; SURVEY D5: offered as a separately-toleranced variant): the lattice kept for the
// backward is alpha16 [N][T][Smax] fp16 = alpha - rowmax with rowmax [N][T] fp32.  Same loss bits as ocrs_ctc_fwd.
int ocrs_ctc_fwd_h16(const float* lp, const int* targets, const long long* in_len, const long long* tg_len, void* alpha16, float* rowmax, float* nll,
                     float* loss, int T, int N, int C, int Lpad, int Smax, hipStream_t st) {
    OCRS_CHECK_ARG(lp && targets && in_len && tg_len && alpha16 && rowmax && nll && loss && T > 0 && N > 0 && C > 0);
    OCRS_CHECK_ARG(Smax <= 256 * CTC_SPT_MAX && Smax >= 1);
    if (ctc_wave_ok(Smax, C, 2)) {
        hipLaunchKernelGGL((k_ctc_alpha_w<true, 2>), dim3((N + 3) / 4), dim3(256), 0, st, lp, targets, in_len, tg_len, alpha16, rowmax, nll, T, N, C, Lpad, Smax);
    } else if (ctc_wave_ok(Smax, C, 4)) {
        hipLaunchKernelGGL((k_ctc_alpha_w<true, 4>), dim3((N + 3) / 4), dim3(256), 0, st, lp, targets, in_len, tg_len, alpha16, rowmax, nll, T, N, C, Lpad, Smax);
    } else if (Smax <= 256 * 3) {
        hipLaunchKernelGGL((k_ctc_alpha<true, 3>), dim3(N), dim3(256), 0, st, lp, targets, in_len, tg_len, alpha16, rowmax, nll, T, N, C, Lpad, Smax);
    } else if (Smax <= 256 * 8) {
        hipLaunchKernelGGL((k_ctc_alpha<true, 8>), dim3(N), dim3(256), 0, st, lp, targets, in_len, tg_len, alpha16, rowmax, nll, T, N, C, Lpad, Smax);
    } else {
        hipLaunchKernelGGL((k_ctc_alpha<true, 16>), dim3(N), dim3(256), 0, st, lp, targets, in_len, tg_len, alpha16, rowmax, nll, T, N, C, Lpad, Smax);
    }
    hipLaunchKernelGGL(k_ctc_reduce, dim3(1), dim3(256), 0, st, nll, tg_len, loss, N);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// Fused form (csrc: k_ctc_fused_w): nll, loss and grad_pre = dloss/dlog_probs for an upstream gradient of 1 in ONE launch, the lattice and the
// sample's log-probabilities in LDS.  ocrs_ctc_fused_lds_bytes: a sample's LDS working set ((T (C + Smax) + C) floats), or 0 when the shape is
// not covered (more than 150 KB, Smax > 256, C > 128, OCRS_CTC_FUSED=0): the caller then uses ocrs_ctc_fwd / ocrs_ctc_bwd.  grad_pre nullable.
long ocrs_ctc_fused_lds_bytes(int T, int C, int Smax) {
    const long per = ((long)T * (C + Smax) + C) * 4;
    const char* e = getenv("OCRS_CTC_FUSED");  // measurement / test knob, default off: see the note at k_ctc_fused_w
    return ((e && e[0] == '1') && T > 0 && per <= 150 * 1024 && Smax >= 1 && Smax <= 256 && C > 0 && C <= 128) ? per : 0;
}
int ocrs_ctc_fused(const float* lp, const int* targets, const long long* in_len, const long long* tg_len, float* nll, float* loss, float* grad_pre, int T,
                   int N, int C, int Lpad, int Smax, hipStream_t st) {
    OCRS_CHECK_ARG(lp && targets && in_len && tg_len && nll && loss && T > 0 && N > 0 && C > 0 && Smax >= 1);
    const long per = ocrs_ctc_fused_lds_bytes(T, C, Smax);
    OCRS_CHECK_ARG(per > 0);
    int wpb = (int)(150 * 1024 / per);
    wpb = wpb > 4 ? 4 : wpb;
    const size_t smem = (size_t)per * wpb;
    static DevOnce attr_set;
    if (attr_set.need()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ctc_fused_w<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ctc_fused_w<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return OCRS_ERR_HIP;
        attr_set.done();
    }
    if (Smax <= 128)
        hipLaunchKernelGGL(k_ctc_fused_w<2>, dim3((N + wpb - 1) / wpb), dim3(64 * wpb), smem, st, lp, targets, in_len, tg_len, nll, grad_pre, T, N, C, Lpad, Smax, wpb);
    else
        hipLaunchKernelGGL(k_ctc_fused_w<4>, dim3((N + wpb - 1) / wpb), dim3(64 * wpb), smem, st, lp, targets, in_len, tg_len, nll, grad_pre, T, N, C, Lpad, Smax, wpb);
    hipLaunchKernelGGL(k_ctc_reduce, dim3(1), dim3(256), 0, st, nll, tg_len, loss, N);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}
// out[i] = in[i] * g[0] (the upstream gradient applied to ocrs_ctc_fused's grad_pre; in == out allowed)
int ocrs_scale_by_dev(const float* in, const float* g, float* out, long n, hipStream_t st) {
    OCRS_CHECK_ARG(in && g && out && n > 0);
    long blocks = (n / 4 + 255) / 256;
    blocks = blocks > 2048 ? 2048 : (blocks < 1 ? 1 : blocks);
    hipLaunchKernelGGL(k_scale_by_dev, dim3((int)blocks), dim3(256), 0, st, in, g, out, n);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// backward: grad [T][N][C] written (gradient w.r.t. log-probs in ATen's convention); gout = upstream scalar gradient (device fp32 [1]).
int ocrs_ctc_bwd(const float* lp, const int* targets, const long long* in_len, const long long* tg_len, const float* alpha, const float* nll,
                 const float* gout, float* grad, int T, int N, int C, int Lpad, int Smax, hipStream_t st) {
    OCRS_CHECK_ARG(lp && targets && in_len && tg_len && alpha && nll && gout && grad);
    OCRS_CHECK_ARG(Smax <= 256 * CTC_SPT_MAX && Smax >= 1);
    if (ctc_wave_ok(Smax, C, 2)) {
        hipLaunchKernelGGL((k_ctc_beta_grad_w<false, 2>), dim3((N + 3) / 4), dim3(256), 4 * C * sizeof(unsigned), st, lp, targets, in_len, tg_len, (const void*)alpha, (const float*)nullptr, nll, gout, grad, T, N, C, Lpad, Smax);
    } else if (ctc_wave_ok(Smax, C, 4)) {
        hipLaunchKernelGGL((k_ctc_beta_grad_w<false, 4>), dim3((N + 3) / 4), dim3(256), 4 * C * sizeof(unsigned), st, lp, targets, in_len, tg_len, (const void*)alpha, (const float*)nullptr, nll, gout, grad, T, N, C, Lpad, Smax);
    } else if (Smax <= 256 * 3) {
        hipLaunchKernelGGL((k_ctc_beta_grad<false, 3>), dim3(N), dim3(256), C * sizeof(unsigned), st, lp, targets, in_len, tg_len, (const void*)alpha, (const float*)nullptr, nll, gout, grad, T, N, C, Lpad, Smax);
    } else if (Smax <= 256 * 8) {
        hipLaunchKernelGGL((k_ctc_beta_grad<false, 8>), dim3(N), dim3(256), C * sizeof(unsigned), st, lp, targets, in_len, tg_len, (const void*)alpha, (const float*)nullptr, nll, gout, grad, T, N, C, Lpad, Smax);
    } else {
        hipLaunchKernelGGL((k_ctc_beta_grad<false, 16>), dim3(N), dim3(256), C * sizeof(unsigned), st, lp, targets, in_len, tg_len, (const void*)alpha, (const float*)nullptr, nll, gout, grad, T, N, C, Lpad, Smax);
    }
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}
int ocrs_ctc_bwd_h16(const float* lp, const int* targets, const long long* in_len, const long long* tg_len, const void* alpha16, const float* rowmax,
                     const float* nll, const float* gout, float* grad, int T, int N, int C, int Lpad, int Smax, hipStream_t st) {
    OCRS_CHECK_ARG(lp && targets && in_len && tg_len && alpha16 && rowmax && nll && gout && grad);
    OCRS_CHECK_ARG(Smax <= 256 * CTC_SPT_MAX && Smax >= 1);
    if (ctc_wave_ok(Smax, C, 2)) {
        hipLaunchKernelGGL((k_ctc_beta_grad_w<true, 2>), dim3((N + 3) / 4), dim3(256), 4 * C * sizeof(unsigned), st, lp, targets, in_len, tg_len, alpha16, rowmax, nll, gout, grad, T, N, C, Lpad, Smax);
    } else if (ctc_wave_ok(Smax, C, 4)) {
        hipLaunchKernelGGL((k_ctc_beta_grad_w<true, 4>), dim3((N + 3) / 4), dim3(256), 4 * C * sizeof(unsigned), st, lp, targets, in_len, tg_len, alpha16, rowmax, nll, gout, grad, T, N, C, Lpad, Smax);
    } else if (Smax <= 256 * 3) {
        hipLaunchKernelGGL((k_ctc_beta_grad<true, 3>), dim3(N), dim3(256), C * sizeof(unsigned), st, lp, targets, in_len, tg_len, alpha16, rowmax, nll, gout, grad, T, N, C, Lpad, Smax);
    } else if (Smax <= 256 * 8) {
        hipLaunchKernelGGL((k_ctc_beta_grad<true, 8>), dim3(N), dim3(256), C * sizeof(unsigned), st, lp, targets, in_len, tg_len, alpha16, rowmax, nll, gout, grad, T, N, C, Lpad, Smax);
    } else {
        hipLaunchKernelGGL((k_ctc_beta_grad<true, 16>), dim3(N), dim3(256), C * sizeof(unsigned), st, lp, targets, in_len, tg_len, alpha16, rowmax, nll, gout, grad, T, N, C, Lpad, Smax);
    }
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// argmax(-1) + greedy CTC collapse (train_rec.py:52; datasets/util.py:147-177).  amax/labels [N][T] int32, lens [N] int32.
int ocrs_ctc_greedy_decode(const float* lp, const long long* in_len, int* amax, int* labels, int* lens, int T, int N, int C, hipStream_t st) {
    OCRS_CHECK_ARG(lp && in_len && amax && labels && lens && T > 0 && N > 0 && C > 0);
    const long rows = (long)T * N;
    hipLaunchKernelGGL(k_argmax, dim3((int)((rows + 15) / 16)), dim3(256), 0, st, lp, amax, T, N, C);
    hipLaunchKernelGGL(k_ctc_collapse, dim3((N + 63) / 64), dim3(64), 0, st, amax, in_len, labels, lens, T, N);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

}  // extern "C"
