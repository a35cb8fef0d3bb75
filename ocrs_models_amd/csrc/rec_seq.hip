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
__device__ __forceinline__ float lse3(float a, float b, float c) {
    const float m = fmaxf(fmaxf(a, b), c);
    if (m == NEG_INF) return NEG_INF;
    return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
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
    if (Smax <= 256 * 3) {
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
// The "fp16 alpha/beta" variant (BASELINE configs[4]; SURVEY D5: offered as a separately-toleranced variant): the lattice kept for the
// backward is alpha16 [N][T][Smax] fp16 = alpha - rowmax with rowmax [N][T] fp32.  Same loss bits as ocrs_ctc_fwd.
int ocrs_ctc_fwd_h16(const float* lp, const int* targets, const long long* in_len, const long long* tg_len, void* alpha16, float* rowmax, float* nll,
                     float* loss, int T, int N, int C, int Lpad, int Smax, hipStream_t st) {
    OCRS_CHECK_ARG(lp && targets && in_len && tg_len && alpha16 && rowmax && nll && loss && T > 0 && N > 0 && C > 0);
    OCRS_CHECK_ARG(Smax <= 256 * CTC_SPT_MAX && Smax >= 1);
    if (Smax <= 256 * 3) {
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

// backward: grad [T][N][C] written (gradient w.r.t. log-probs in ATen's convention); gout = upstream scalar gradient (device fp32 [1]).
int ocrs_ctc_bwd(const float* lp, const int* targets, const long long* in_len, const long long* tg_len, const float* alpha, const float* nll,
                 const float* gout, float* grad, int T, int N, int C, int Lpad, int Smax, hipStream_t st) {
    OCRS_CHECK_ARG(lp && targets && in_len && tg_len && alpha && nll && gout && grad);
    OCRS_CHECK_ARG(Smax <= 256 * CTC_SPT_MAX && Smax >= 1);
    if (Smax <= 256 * 3) {
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
    if (Smax <= 256 * 3) {
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
