// Detection loss (class-balanced hard-example BCE) and optimiser kernels (gfx950).
//
// Reference: balanced_cross_entropy_loss, ocrs_models/train_detection.py:225-263
//   pos = t > 0.5, neg = t < 0.5, t = clamp(t,0,1); l = BCE(p, t) elementwise (log clamp -100);
//   k = min(#pos, #neg); loss = mean(topk(pos*l, k) ++ topk(neg*l, k)).
// Here everything stays on the device (the reference does two .item() syncs): exact k-th-largest threshold by a
// 3-pass MSB radix select (11/10/10 bits of the non-negative fp32 bit pattern), then a masked sum.
// Ties at the threshold: the reference's topk picks an implementation-defined subset; we weight every tie
// equally with need/ties (same loss value, a valid sub-gradient).
//
// Optimiser: torch.optim.Adam defaults (train_detection.py:378, train_rec.py:381-382) as one multi-tensor launch,
// clip_grad_norm_ (train_rec.py:148) as multi-tensor sum-of-squares + scale.
#include "common.h"

#include "loss_state.h"

// Workspace behind the two histograms (ocrs_loss_hist_bytes): per-block partials, reduced in a fixed order by the one-block scan kernels -- the
// element counts of k_bce_fwd and the sums of k_select_hist's last pass (1024 blocks x 2 same-address fp64 atomics were a 30 us serial tail
// on each of those kernels, and made the loss depend on their arrival order in its last bits).
constexpr int LOSS_MAXB = 1024;                              // blocks of the streaming kernels
constexpr int LOSS_HIST_WORDS = 2 * 2048;
struct LossParts {
    unsigned long long cnt[LOSS_MAXB][2];                    // k_bce_fwd: #pos, #neg of the block
    double sum[LOSS_MAXB][2];                                // last k_select_hist pass: sum of the block's losses above the 21-bit prefix, per class
};
__device__ __forceinline__ LossParts* loss_parts(unsigned* hist) { return reinterpret_cast<LossParts*>(hist + LOSS_HIST_WORDS); }

// log(1 - p) for p in [0, 1] with log1p accuracy from ONE logf: log1p(x) = log(u) - ((u - 1) - x) / u, u = fl(1 + x)  (x = -p).
// (OCML's log1pf costs ~3x a logf; the loss forward was VALU-bound on it.)  p = 1 -> -inf (clamped by the caller).
__device__ __forceinline__ float log1p_neg(float p) {
    const float u = 1.f - p;
    if (u <= 0.f) return -__builtin_inff();
    return logf(u) - ((u - 1.f) + p) / u;
}

// The elementwise loss kernels walk the pixels in quads (16-byte loads / stores of pred, target, lpx, gpred; 4-byte of cls) when the
// buffers are 16-byte aligned (VEC = 4), else one by one (VEC = 1).  The < 4 tail elements are done by the last quad-loop thread.
template <int VEC>
struct Quad {
    float v[VEC];
};
template <int VEC>
__device__ __forceinline__ Quad<VEC> ldq(const float* p, long q) {
    Quad<VEC> r;
    if constexpr (VEC == 4) {
        const float4 a = reinterpret_cast<const float4*>(p)[q];
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    } else
        r.v[0] = p[q];
    return r;
}
template <int VEC>
__device__ __forceinline__ void stq(float* p, long q, const Quad<VEC>& r) {
    if constexpr (VEC == 4)
        reinterpret_cast<float4*>(p)[q] = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
    else
        p[q] = r.v[0];
}
template <int VEC>
struct QuadB {
    unsigned char v[VEC];
};
template <int VEC>
__device__ __forceinline__ QuadB<VEC> ldqb(const unsigned char* p, long q) {
    QuadB<VEC> r;
    if constexpr (VEC == 4) {
        const unsigned a = reinterpret_cast<const unsigned*>(p)[q];
        r.v[0] = a & 255; r.v[1] = (a >> 8) & 255; r.v[2] = (a >> 16) & 255; r.v[3] = a >> 24;
    } else
        r.v[0] = p[q];
    return r;
}
template <int VEC>
__device__ __forceinline__ void stqb(unsigned char* p, long q, const QuadB<VEC>& r) {
    if constexpr (VEC == 4)
        reinterpret_cast<unsigned*>(p)[q] = r.v[0] | (r.v[1] << 8) | (r.v[2] << 16) | ((unsigned)r.v[3] << 24);
    else
        p[q] = r.v[0];
}

template <int VEC>
__global__ __launch_bounds__(256) void k_bce_fwd(const float* __restrict__ pred, const float* __restrict__ target, float* __restrict__ lpx,
                                                 unsigned char* __restrict__ cls, unsigned* __restrict__ hist, long P) {
    // the first radix-select pass (top 11 bits of every classified element: it needs neither k nor a prefix) rides on this kernel
    __shared__ unsigned s_h[LOSS_HIST_WORDS];
    for (int i = threadIdx.x; i < LOSS_HIST_WORDS; i += 256) s_h[i] = 0;
    __syncthreads();
    unsigned long long np = 0, nn = 0;
    const long nq = P / VEC;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long)gridDim.x * 256) {
        const Quad<VEC> pq = ldq<VEC>(pred, q), tq = ldq<VEC>(target, q);
        Quad<VEC> lq;
        QuadB<VEC> cq;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float p = pq.v[e], t0 = tq.v[e];
            const unsigned char c = t0 > 0.5f ? 1 : (t0 < 0.5f ? 2 : 0);
            const float t = fminf(fmaxf(t0, 0.f), 1.f);
            lq.v[e] = -(t * fmaxf(logf(p), -100.f) + (1.f - t) * fmaxf(log1p_neg(p), -100.f));
            cq.v[e] = c;
            np += c == 1;
            nn += c == 2;
        }
        stq<VEC>(lpx, q, lq);
        stqb<VEC>(cls, q, cq);
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            if (cq.v[e]) atomicAdd(&s_h[(cq.v[e] - 1) * 2048 + (loss_key(lq.v[e]) >> 20)], 1u);
    }
    if (VEC > 1 && blockIdx.x == 0 && threadIdx.x < P - nq * VEC) {  // tail
        const long i = nq * VEC + threadIdx.x;
        const float p = pred[i], t0 = target[i];
        const unsigned char c = t0 > 0.5f ? 1 : (t0 < 0.5f ? 2 : 0);
        const float t = fminf(fmaxf(t0, 0.f), 1.f);
        const float l = -(t * fmaxf(logf(p), -100.f) + (1.f - t) * fmaxf(log1p_neg(p), -100.f));
        lpx[i] = l;
        cls[i] = c;
        np += c == 1;
        nn += c == 2;
        if (c) atomicAdd(&s_h[(c - 1) * 2048 + (loss_key(l) >> 20)], 1u);
    }
    // the block's counts: one partial per block (same-address global atomics serialise at ~15 ns each)
    __shared__ unsigned s_cnt[2];
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) {
        np += __shfl_xor(np, o, 64);
        nn += __shfl_xor(nn, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&s_cnt[0], (unsigned)np);
        atomicAdd(&s_cnt[1], (unsigned)nn);
    }
    __syncthreads();
    if (threadIdx.x < 2) loss_parts(hist)->cnt[blockIdx.x][threadIdx.x] = s_cnt[threadIdx.x];
    for (int i = threadIdx.x; i < LOSS_HIST_WORDS; i += 256)
        if (s_h[i]) atomicAdd(&hist[i], s_h[i]);
}

// pass: 0 -> bits 30..20 (2048 bins), 1 -> bits 19..10 (1024), 2 -> bits 9..0 (1024)
__device__ __forceinline__ void pass_bits(int pass, int& shift, int& nb) {
    shift = pass == 0 ? 20 : (pass == 1 ? 10 : 0);
    nb = pass == 0 ? 2048 : 1024;
}

template <int VEC>
__global__ __launch_bounds__(256) void k_select_hist(const float* __restrict__ lpx, const unsigned char* __restrict__ cls,
                                                     const LossState* __restrict__ stt, unsigned* __restrict__ hist /*[2][2048]*/, int pass,
                                                     long P) {
    __shared__ unsigned s_h[2 * 2048];
    for (int i = threadIdx.x; i < 4096; i += 256) s_h[i] = 0;
    __syncthreads();
    int shift, nb;
    pass_bits(pass, shift, nb);
    const unsigned pf0 = stt->prefix[0], pf1 = stt->prefix[1];
    const int hs = shift + (pass == 0 ? 11 : 10);  // bits above the current digit
    // last pass: an element whose upper 21 bits exceed the prefix lies above the threshold whatever the last digit turns out to be -- summed here
    // (the elements inside the prefix bucket are summed from the histogram by k_select_scan: a key IS its value), so no further pass over lpx
    double s0 = 0.0, s1 = 0.0;
    auto visit = [&](unsigned char c, float v) {
        if (!c) return;
        const unsigned key = loss_key(v);
        const unsigned pf = c == 1 ? pf0 : pf1;
        const unsigned up = key >> hs;
        if (pass == 0 || up == pf) atomicAdd(&s_h[(c - 1) * 2048 + ((key >> shift) & (nb - 1))], 1u);
        if (pass == 2 && up > pf) {
            if (c == 1) s0 += (double)v;
            else s1 += (double)v;
        }
    };
    const long nq = P / VEC;
    // four quads' loads in flight per thread (one per iteration left these passes latency-bound: 2-3 TB/s)
    const long stride = (long)gridDim.x * 256;
    long q = (long)blockIdx.x * 256 + threadIdx.x;
    for (; q + 3 * stride < nq; q += 4 * stride) {
        QuadB<VEC> cq[4];
        Quad<VEC> lq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            cq[u] = ldqb<VEC>(cls, q + u * stride);
            lq[u] = ldq<VEC>(lpx, q + u * stride);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < VEC; ++e) visit(cq[u].v[e], lq[u].v[e]);
    }
    for (; q < nq; q += stride) {
        const QuadB<VEC> cq = ldqb<VEC>(cls, q);
        const Quad<VEC> lq = ldq<VEC>(lpx, q);
#pragma unroll
        for (int e = 0; e < VEC; ++e) visit(cq.v[e], lq.v[e]);
    }
    if (VEC > 1 && blockIdx.x == 0 && threadIdx.x < P - nq * VEC) visit(cls[nq * VEC + threadIdx.x], lpx[nq * VEC + threadIdx.x]);
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 256)
        if (s_h[i]) atomicAdd(&hist[i], s_h[i]);
    if (pass == 2) {  // (fixed order inside the block; the blocks' partials are added in block order by k_select_scan: a reproducible loss)
        __shared__ double s_sum[2][4];
        for (int o = 32; o > 0; o >>= 1) {
            s0 += __shfl_xor(s0, o, 64);
            s1 += __shfl_xor(s1, o, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            s_sum[0][threadIdx.x >> 6] = s0;
            s_sum[1][threadIdx.x >> 6] = s1;
        }
        __syncthreads();
        if (threadIdx.x < 2) loss_parts(hist)->sum[blockIdx.x][threadIdx.x] = (s_sum[threadIdx.x][0] + s_sum[threadIdx.x][1]) + (s_sum[threadIdx.x][2] + s_sum[threadIdx.x][3]);
    }
}

// ONE block of 256 threads after each histogram pass: per class, the bin that holds the need-th largest element.  Pass 0 first reduces the
// blocks' element counts (k = min(#pos, #neg)) and initialises the selection state; pass 2 finishes the loss: sum above the threshold =
// the blocks' partial sums of k_select_hist (added in block order) + the histogram bins above the chosen one (a key is its value), then
// loss = mean of the 2k selected elements (ties at the threshold weighted need / ties).
__global__ __launch_bounds__(256) void k_select_scan(LossState* __restrict__ stt, unsigned* __restrict__ hist, int pass, int nblk, float* __restrict__ loss_out) {
    __shared__ unsigned long long s_sum[256];
    __shared__ double s_d[256];
    __shared__ unsigned long long s_need[2], s_ties[2], s_k;
    __shared__ int s_bin[2];
    int shift, nb;
    pass_bits(pass, shift, nb);
    const int per = nb / 256;
    const int tid = threadIdx.x;
    LossParts* lp = loss_parts(hist);
    auto block_sum_u64 = [&](unsigned long long v) -> unsigned long long {
        __syncthreads();
        s_sum[tid] = v;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) s_sum[tid] += s_sum[tid + o];
            __syncthreads();
        }
        return s_sum[0];
    };
    auto block_sum_f64 = [&](double v) -> double {  // fixed tree: reproducible
        __syncthreads();
        s_d[tid] = v;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) s_d[tid] += s_d[tid + o];
            __syncthreads();
        }
        return s_d[0];
    };
    if (pass == 0) {
        unsigned long long c0 = 0, c1 = 0;
        for (int b = tid; b < nblk; b += 256) {
            c0 += lp->cnt[b][0];
            c1 += lp->cnt[b][1];
        }
        const unsigned long long n0 = block_sum_u64(c0), n1 = block_sum_u64(c1);
        if (tid == 0) {
            const unsigned long long k = n0 < n1 ? n0 : n1;
            stt->cnt[0] = n0;
            stt->cnt[1] = n1;
            stt->k = k;
            s_k = k;
            for (int c = 0; c < 2; ++c) {
                stt->prefix[c] = 0;
                stt->ties[c] = 0;
                stt->sum_gt[c] = 0.0;
                s_need[c] = k;
            }
        }
    } else if (tid == 0) {
        s_need[0] = stt->need[0];
        s_need[1] = stt->need[1];
        s_k = stt->k;
    }
    if (tid < 2) s_bin[tid] = -1;
    __syncthreads();
    double tot = 0.0;
    for (int c = 0; c < 2; ++c) {
        unsigned* h = hist + c * 2048;
        // thread t owns bins [nb - (t+1)*per, nb - t*per) : descending order over t
        unsigned long long loc = 0;
        for (int j = 0; j < per; ++j) loc += h[nb - 1 - (tid * per + j)];
        const unsigned pfx = pass == 0 ? 0u : stt->prefix[c];  // (read by every thread in front of the barriers: the finder below overwrites it)
        __syncthreads();
        s_sum[tid] = loc;
        __syncthreads();
        if (tid == 0) {
            unsigned long long run = 0;
            for (int t = 0; t < 256; ++t) {
                const unsigned long long v = s_sum[t];
                s_sum[t] = run;  // exclusive prefix (count of strictly larger digits before this thread's bins)
                run += v;
            }
        }
        __syncthreads();
        const unsigned long long need = s_need[c];
        const unsigned long long before = s_sum[tid];
        if (need > 0 && before < need && before + loc >= need) {
            unsigned long long run = before;
            for (int j = 0; j < per; ++j) {
                const int bin = nb - 1 - (tid * per + j);
                const unsigned long long v = h[bin];
                if (run + v >= need) {
                    stt->prefix[c] = (pfx << (pass == 0 ? 11 : 10)) | (unsigned)bin;
                    stt->need[c] = need - run;
                    stt->ties[c] = v;
                    s_need[c] = need - run;
                    s_ties[c] = v;
                    s_bin[c] = bin;
                    break;
                }
                run += v;
            }
        } else if (need == 0 && tid == 0) {
            stt->need[c] = 0;
        }
        __syncthreads();
        if (pass == 2) {
            const int bin = s_bin[c];
            const unsigned long long ties = bin >= 0 ? s_ties[c] : 0ull, needc = bin >= 0 ? s_need[c] : 0ull;
            const unsigned thr = bin >= 0 ? ((pfx << 10) | (unsigned)bin) : pfx;
            // the prefix bucket's elements above the chosen bin, from the histogram
            double in_b = 0.0;
            if (bin >= 0)
                for (int j = 0; j < per; ++j) {
                    const int bj = nb - 1 - (tid * per + j);
                    if (bj > bin && h[bj]) in_b += (double)h[bj] * (double)__uint_as_float((pfx << 10) | (unsigned)bj);
                }
            double above = 0.0;
            for (int b = tid; b < nblk; b += 256) above += lp->sum[b][c];
            const double sum_gt = block_sum_f64(above) + block_sum_f64(in_b);
            if (tid == 0) {
                stt->sum_gt[c] = sum_gt;
                stt->frac[c] = ties ? (float)((double)needc / (double)ties) : 0.f;
            }
            tot += sum_gt + (double)needc * (double)__uint_as_float(thr);
        }
        __syncthreads();
    }
    for (int i = tid; i < LOSS_HIST_WORDS; i += 256) hist[i] = 0;  // ready for the next pass / the next call
    if (pass == 2 && tid == 0) {
        const unsigned long long k = s_k;
        const float loss = k ? (float)(tot / (2.0 * (double)k)) : __uint_as_float(0x7fc00000u);  // mean of an empty tensor = NaN
        stt->loss = loss;
        stt->inv2k = k ? (float)(1.0 / (2.0 * (double)k)) : 0.f;
        *loss_out = loss;
    }
}

// d loss / d pred: weight * (p - t) / max(p (1 - p), 1e-12)   (ATen binary_cross_entropy_backward)
template <int VEC>
__global__ __launch_bounds__(256) void k_bce_bwd(const float* __restrict__ pred, const float* __restrict__ target, const float* __restrict__ lpx,
                                                 const unsigned char* __restrict__ cls, const LossState* __restrict__ stt,
                                                 const float* __restrict__ gout, float* __restrict__ gpred, long P) {
    const unsigned t0 = stt->prefix[0], t1 = stt->prefix[1];
    const float f0 = stt->frac[0], f1 = stt->frac[1];
    const float s = gout[0] * stt->inv2k;
    auto grad = [&](unsigned char c, float l, float p, float tg) -> float {
        if (!c) return 0.f;
        const unsigned key = loss_key(l);
        const unsigned thr = c == 1 ? t0 : t1;
        const float wgt = key > thr ? 1.f : (key == thr ? (c == 1 ? f0 : f1) : 0.f);
        if (wgt == 0.f) return 0.f;
        const float t = fminf(fmaxf(tg, 0.f), 1.f);
        return s * wgt * (p - t) / fmaxf((1.f - p) * p, 1e-12f);
    };
    const long nq = P / VEC;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long)gridDim.x * 256) {
        const QuadB<VEC> cq = ldqb<VEC>(cls, q);
        const Quad<VEC> lq = ldq<VEC>(lpx, q), pq = ldq<VEC>(pred, q), tq = ldq<VEC>(target, q);
        Quad<VEC> gq;
#pragma unroll
        for (int e = 0; e < VEC; ++e) gq.v[e] = grad(cq.v[e], lq.v[e], pq.v[e], tq.v[e]);
        stq<VEC>(gpred, q, gq);
    }
    if (VEC > 1 && blockIdx.x == 0 && threadIdx.x < P - nq * VEC) {
        const long i = nq * VEC + threadIdx.x;
        gpred[i] = grad(cls[i], lpx[i], pred[i], target[i]);
    }
}

// ----------------------------------------------------------------------------------------------
// multi-tensor optimiser kernels.  table: [nt][5] int64 = {param, grad, exp_avg, exp_avg_sq, numel};
// chunks: [nchunks][2] int32 = {tensor index, chunk index}; one block handles CHUNK = 2048 elements.
// ----------------------------------------------------------------------------------------------
static constexpr int OPT_CHUNK = 2048;

__global__ __launch_bounds__(256) void k_multi_sumsq(const long long* __restrict__ table, const int* __restrict__ chunks,
                                                     double* __restrict__ out) {
    const int t = chunks[2 * blockIdx.x], ch = chunks[2 * blockIdx.x + 1];
    const float* g = reinterpret_cast<const float*>(table[5 * t + 1]);
    const long n = table[5 * t + 4];
    float s = 0.f;
    for (long i = (long)ch * OPT_CHUNK + threadIdx.x; i < n && i < (long)(ch + 1) * OPT_CHUNK; i += 256) s = fmaf(g[i], g[i], s);
    // one fp64 atomic per block (they all hit ONE address and serialise at ~11 ns each: per-wave atomics made this launch 62 us for 12 MB)
    __shared__ float s_w[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (double)((s_w[0] + s_w[1]) + (s_w[2] + s_w[3])));
}

// norm_out = sqrt(sumsq); coef = min(1, max_norm / (norm + 1e-6))
__global__ void k_clip_coef(const double* sumsq, float max_norm, float* norm_out, float* coef_out) {
    const float norm = (float)sqrt(*sumsq);
    *norm_out = norm;
    *coef_out = fminf(1.f, max_norm / (norm + 1e-6f));
}

__global__ __launch_bounds__(256) void k_multi_scale(const long long* __restrict__ table, const int* __restrict__ chunks,
                                                     const float* __restrict__ coef) {
    const int t = chunks[2 * blockIdx.x], ch = chunks[2 * blockIdx.x + 1];
    float* g = reinterpret_cast<float*>(table[5 * t + 1]);
    const long n = table[5 * t + 4];
    const float c = *coef;
    for (long i = (long)ch * OPT_CHUNK + threadIdx.x; i < n && i < (long)(ch + 1) * OPT_CHUNK; i += 256) g[i] *= c;
}

__global__ __launch_bounds__(256) void k_multi_adam(const long long* __restrict__ table, const int* __restrict__ chunks, float b1, float b2,
                                                    float eps, float step_size, float bc2_sqrt, const float* __restrict__ gscale) {
    const int t = chunks[2 * blockIdx.x], ch = chunks[2 * blockIdx.x + 1];
    float* p = reinterpret_cast<float*>(table[5 * t + 0]);
    const float* g = reinterpret_cast<const float*>(table[5 * t + 1]);
    float* m = reinterpret_cast<float*>(table[5 * t + 2]);
    float* v = reinterpret_cast<float*>(table[5 * t + 3]);
    const long n = table[5 * t + 4];
    const float gs = gscale ? *gscale : 1.f;
    for (long i = (long)ch * OPT_CHUNK + threadIdx.x; i < n && i < (long)(ch + 1) * OPT_CHUNK; i += 256) {
        const float gi = g[i] * gs;
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);        // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] -= step_size * (mi / denom);
    }
}

// Capturable form (hipGraph replay): the step count lives on the device; every block derives the bias corrections from it (double powers of
// the betas, as torch.optim.Adam's capturable path does), a one-thread launch in front increments it.
__global__ void k_step_inc(float* step) { *step += 1.f; }
__global__ __launch_bounds__(256) void k_multi_adam_dev(const long long* __restrict__ table, const int* __restrict__ chunks, double b1d, double b2d,
                                                        float eps, double lr, const float* __restrict__ step, const float* __restrict__ gscale) {
    const float b1 = (float)b1d, b2 = (float)b2d;  // (the moment updates use fp32 betas like k_multi_adam; the bias corrections the exact doubles)
    const int t = chunks[2 * blockIdx.x], ch = chunks[2 * blockIdx.x + 1];
    float* p = reinterpret_cast<float*>(table[5 * t + 0]);
    const float* g = reinterpret_cast<const float*>(table[5 * t + 1]);
    float* m = reinterpret_cast<float*>(table[5 * t + 2]);
    float* v = reinterpret_cast<float*>(table[5 * t + 3]);
    const long n = table[5 * t + 4];
    const double st = (double)*step;
    const float step_size = (float)(lr / (1.0 - pow(b1d, st)));
    const float bc2_sqrt = (float)sqrt(1.0 - pow(b2d, st));
    const float gs = gscale ? *gscale : 1.f;
    for (long i = (long)ch * OPT_CHUNK + threadIdx.x; i < n && i < (long)(ch + 1) * OPT_CHUNK; i += 256) {
        const float gi = g[i] * gs;
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] -= step_size * (mi / denom);
    }
}

__global__ void k_fill_f32(float* p, float v, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = v;
}

__global__ void k_fold64_multi(const int* __restrict__ table, float* __restrict__ dst, const double* __restrict__ src) {
    const int d = table[blockIdx.x * 3], s0 = table[blockIdx.x * 3 + 1], n = table[blockIdx.x * 3 + 2];
    for (int i = threadIdx.x; i < n; i += 64) dst[d + i] = (float)((double)dst[d + i] + src[s0 + i]);
}

static inline int ew_grid(long items) {
    long g = (items + 255) / 256;
    const long cap = (long)kNumCU * 8;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

extern "C" {

long ocrs_loss_state_bytes() { return (long)sizeof(LossState); }
long ocrs_loss_hist_bytes() { return LOSS_HIST_WORDS * (long)sizeof(unsigned) + (long)sizeof(LossParts); }  // the two histograms + the per-block partials

// Class-balanced BCE forward (train_detection.py:225-263).  pred/target fp32 [P]; lpx fp32 [P] and cls u8 [P] are saved
// for backward; state (ocrs_loss_state_bytes) and hist (ocrs_loss_hist_bytes) are device workspaces; loss_out fp32 [1].
// 7 launches: the elementwise loss + first histogram, then (scan | histogram) x 2, the last histogram pass also summing what lies above the
// prefix, and a final scan that finishes the loss (round 5; before: 12 launches with two further passes over the loss map).
int ocrs_balanced_bce_fwd(const float* pred, const float* target, float* lpx, unsigned char* cls, void* state, void* hist, float* loss_out,
                          long P, hipStream_t st) {
    OCRS_CHECK_ARG(pred && target && lpx && cls && state && hist && loss_out && P > 0);
    LossState* stt = (LossState*)state;
    if (hipMemsetAsync(hist, 0, LOSS_HIST_WORDS * sizeof(unsigned), st) != hipSuccess) return OCRS_ERR_HIP;  // (the state is written before it is read)
    int grid = ew_grid((P + 3) / 4);
    if (grid > LOSS_MAXB) grid = LOSS_MAXB;  // 4 blocks per CU are plenty; LossParts holds one partial per block
    const bool vec = ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(target) | reinterpret_cast<uintptr_t>(lpx)) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(cls) & 3) == 0;
    if (vec)
        hipLaunchKernelGGL(k_bce_fwd<4>, dim3(grid), dim3(256), 0, st, pred, target, lpx, cls, (unsigned*)hist, P);
    else
        hipLaunchKernelGGL(k_bce_fwd<1>, dim3(grid), dim3(256), 0, st, pred, target, lpx, cls, (unsigned*)hist, P);
    for (int pass = 0; pass < 3; ++pass) {
        if (pass > 0) {
            if (vec)
                hipLaunchKernelGGL(k_select_hist<4>, dim3(grid), dim3(256), 0, st, lpx, cls, stt, (unsigned*)hist, pass, P);
            else
                hipLaunchKernelGGL(k_select_hist<1>, dim3(grid), dim3(256), 0, st, lpx, cls, stt, (unsigned*)hist, pass, P);
        }
        hipLaunchKernelGGL(k_select_scan, dim3(1), dim3(256), 0, st, stt, (unsigned*)hist, pass, grid, loss_out);
    }
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// Backward of the above: gpred fp32 [P] written; gout = upstream gradient of the scalar loss (device fp32 [1]).
int ocrs_balanced_bce_bwd(const float* pred, const float* target, const float* lpx, const unsigned char* cls, const void* state,
                          const float* gout, float* gpred, long P, hipStream_t st) {
    OCRS_CHECK_ARG(pred && target && lpx && cls && state && gout && gpred && P > 0);
    const bool vec = ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(target) | reinterpret_cast<uintptr_t>(lpx) |
                       reinterpret_cast<uintptr_t>(gpred)) & 15) == 0 && (reinterpret_cast<uintptr_t>(cls) & 3) == 0;
    if (vec)
        hipLaunchKernelGGL(k_bce_bwd<4>, dim3(ew_grid((P + 3) / 4)), dim3(256), 0, st, pred, target, lpx, cls, (const LossState*)state, gout, gpred, P);
    else
        hipLaunchKernelGGL(k_bce_bwd<1>, dim3(ew_grid(P)), dim3(256), 0, st, pred, target, lpx, cls, (const LossState*)state, gout, gpred, P);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

int ocrs_opt_chunk() { return OPT_CHUNK; }

// clip_grad_norm_(params, max_norm) (train_rec.py:148): norm_out/coef_out fp32 [1] device; sumsq double [1] device workspace.
// scale_in_place != 0 multiplies the gradients by coef (reference behaviour); 0 leaves that to ocrs_adam_step(gscale = coef_out).
int ocrs_clip_grad_norm(const long long* table, const int* chunks, int nchunks, float max_norm, double* sumsq, float* norm_out,
                        float* coef_out, int scale_in_place, hipStream_t st) {
    OCRS_CHECK_ARG(table && chunks && nchunks > 0 && sumsq && norm_out && coef_out);
    if (hipMemsetAsync(sumsq, 0, sizeof(double), st) != hipSuccess) return OCRS_ERR_HIP;
    hipLaunchKernelGGL(k_multi_sumsq, dim3(nchunks), dim3(256), 0, st, table, chunks, sumsq);
    hipLaunchKernelGGL(k_clip_coef, dim3(1), dim3(1), 0, st, sumsq, max_norm, norm_out, coef_out);
    if (scale_in_place) hipLaunchKernelGGL(k_multi_scale, dim3(nchunks), dim3(256), 0, st, table, chunks, coef_out);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// torch.optim.Adam single step over every tensor in the table.  step_size = lr / (1 - b1^t), bc2_sqrt = sqrt(1 - b2^t).
int ocrs_adam_step(const long long* table, const int* chunks, int nchunks, float b1, float b2, float eps, float step_size, float bc2_sqrt,
                   const float* gscale, hipStream_t st) {
    OCRS_CHECK_ARG(table && chunks && nchunks > 0);
    hipLaunchKernelGGL(k_multi_adam, dim3(nchunks), dim3(256), 0, st, table, chunks, b1, b2, eps, step_size, bc2_sqrt, gscale);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// The same step with the step count on the device (fp32 [1], incremented here before use) and the learning rate as the only host scalar:
// nothing in the launch arguments changes from step to step, so a captured train step (hipGraph) replays correctly.
int ocrs_adam_step_dev(const long long* table, const int* chunks, int nchunks, double b1, double b2, float eps, double lr, float* step,
                       const float* gscale, hipStream_t st) {
    OCRS_CHECK_ARG(table && chunks && nchunks > 0 && step);
    hipLaunchKernelGGL(k_step_inc, dim3(1), dim3(1), 0, st, step);
    hipLaunchKernelGGL(k_multi_adam_dev, dim3(nchunks), dim3(256), 0, st, table, chunks, b1, b2, eps, lr, step, gscale);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

// dst[table[r][0] + i] += src[table[r][1] + i] for i < table[r][2], r < nrows: the fp64 accumulators of a detection backward (head, first block,
// ConvTranspose biases: ~10 tensors of <= 17 elements) folded into the flat fp32 gradient buffer by one launch (models.py: fold64).
int ocrs_fold64_multi(const int* table, int nrows, float* dst, const double* src, hipStream_t st) {
    OCRS_CHECK_ARG(table && nrows > 0 && dst && src);
    hipLaunchKernelGGL(k_fold64_multi, dim3(nrows), dim3(64), 0, st, table, dst, src);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

int ocrs_fill_f32(float* p, float v, long n, hipStream_t st) {
    OCRS_CHECK_ARG(p && n >= 0);
    if (n == 0) return OCRS_OK;
    hipLaunchKernelGGL(k_fill_f32, dim3(ew_grid(n)), dim3(256), 0, st, p, v, n);
    OCRS_LAUNCH_CHECK();
    return OCRS_OK;
}

}  // extern "C"
